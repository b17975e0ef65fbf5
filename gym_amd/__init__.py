"""gym_amd — MI355X-native batched environment engine behind the ava6969/gym VecEnv interface.

Only the hot path of SURVEY.md §8 lives here: HIP kernels + C ABI (csrc/, lib/libgymhip.so), the ctypes binding
(_lib), the engine owner (engine) and the host-side mirror of the reference's VecEnv API (vec_env, spaces).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
