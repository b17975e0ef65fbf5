"""ctypes binding of libgymhip.so (the C ABI in include/gymhip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GYMHIP_LIB") or os.path.join(_HERE, "lib", "libgymhip.so")   # GYMHIP_LIB: tuning builds only

OK = 0
ENV_CARTPOLE, ENV_MOUNTAINCAR = 0, 1
PRECISION_LOCKSTEP_F64, PRECISION_F32 = 0, 1
RESET_TAPE, RESET_COUNTER_RNG, RESET_MT19937 = 0, 1, 2
EXCHANGE_MOMENTS, EXCHANGE_ROWS = 0, 1
WRAP_TIME_LIMIT, WRAP_CLIP_REWARD, WRAP_MONITOR = 1, 2, 4
WRAP_NORM_OBS, WRAP_NORM_REWARD, WRAP_NORM_TRAINING = 8, 16, 32
WRAP_NO_AUTO_RESET = 64
ACT_I32, ACT_I64, ACT_U8 = 0, 1, 2
STACK_INTENDED, STACK_AS_WRITTEN = 0, 1
OUT_F16, OUT_BF16 = 0, 1
NOALLOC_MAX_ROWS = 1 << 26


class GymhipError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("env_kind", C.c_int32),
        ("precision", C.c_int32),
        ("reset_mode", C.c_int32),
        ("wrappers", C.c_uint32),
        ("max_episode_steps", C.c_int32),
        ("n_envs", C.c_int64),
        ("env_offset", C.c_int64),
        ("seed", C.c_uint64),
        ("clip_obs", C.c_float),
        ("clip_reward", C.c_float),
        ("gamma", C.c_float),
        ("epsilon", C.c_float),
    ]


class Buffers(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32),
        ("moments_len", C.c_int32),
        ("state", C.c_void_p),
        ("obs", C.c_void_p),
        ("obs_norm", C.c_void_p),
        ("reward", C.c_void_p),
        ("done", C.c_void_p),
        ("counter", C.c_void_p),
        ("returns", C.c_void_p),
        ("norm_stats", C.c_void_p),
        ("moments", C.c_void_p),
        ("episode_totals", C.c_void_p),
        ("mt_state", C.c_void_p),
    ]


class AtariConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("n_envs", C.c_int64),
        ("src_h", C.c_int32),
        ("src_w", C.c_int32),
        ("dst_h", C.c_int32),
        ("dst_w", C.c_int32),
        ("n_stack", C.c_int32),
        ("gray_bits", C.c_int32),
        ("clear_on_done", C.c_int32),
    ]


# Every symbol include/gymhip.h declares: (restype, argtypes)
_vp, _i64, _i32, _u64 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64
SIGNATURES = {
    "gymhip_last_error": (C.c_char_p, []),
    "gymhip_version": (C.c_int, []),
    "gymhip_workspace_bytes": (C.c_size_t, [C.POINTER(Config)]),
    "gymhip_create": (C.c_int, [C.POINTER(Config), C.c_int, _vp, C.c_size_t, C.POINTER(_vp)]),
    "gymhip_destroy": (C.c_int, [_vp]),
    "gymhip_get_buffers": (C.c_int, [_vp, C.POINTER(Buffers)]),
    "gymhip_seed": (C.c_int, [_vp, _u64]),
    "gymhip_upload_reset_tape": (C.c_int, [_vp, _vp, _i64]),
    "gymhip_seed_reset_tape": (C.c_int, [_vp, _u64, _i64]),
    "gymhip_unseeded_reset_tape": (C.c_int, [_vp, _i64]),
    "gymhip_tape_cursor_max": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_extend_reset_tape": (C.c_int, [_vp, _i64]),
    "gymhip_generate_reset_tape": (C.c_int, [_vp, _i64, _vp]),
    "gymhip_download_reset_tape": (C.c_int, [_vp, _vp, _i64]),
    "gymhip_reset": (C.c_int, [_vp, _vp]),
    "gymhip_step": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "gymhip_step_begin": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "gymhip_step_finish": (C.c_int, [_vp, _vp]),
    "gymhip_reset_begin": (C.c_int, [_vp, _vp]),
    "gymhip_reset_finish": (C.c_int, [_vp, _vp]),
    "gymhip_rollout": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gymhip_decode_counters": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gymhip_read_state_f64": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_poll_error": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_reset_index": (C.c_int, [_vp, _i64, _vp]),
    "gymhip_update_norm_stats": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "gymhip_update_norm_stats_noalloc": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "gymhip_diag_pole_sincos": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "gymhip_get_norm_stats": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_set_norm_stats": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_host_env_post_step": (C.c_int, [_i64] + [_vp] * 7 + [C.c_int] + [_vp] * 6),
    "gymhip_atari_create": (C.c_int, [C.POINTER(AtariConfig), C.c_int, C.POINTER(_vp)]),
    "gymhip_atari_destroy": (C.c_int, [_vp]),
    "gymhip_atari_stack_buffer": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_i32)]),
    "gymhip_atari_reset": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_atari_step": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gymhip_atari_preprocess": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gymhip_atari_set_palette": (C.c_int, [_vp, _vp]),
    "gymhip_atari_preprocess_indexed": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gymhip_atari_reset_indexed": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_atari_step_indexed": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gymhip_launch_info": (C.c_int, [_vp, _vp]),
    "gymhip_set_exchange_form": (C.c_int, [_vp, C.c_int]),
    "gymhip_atari_push_reset": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_atari_push": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gymhip_atari_push_index": (C.c_int, [_vp, _i64, _vp, C.c_int, _vp]),
    "gymhip_atari_materialize_nhwc": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_atari_set_stack_view": (C.c_int, [_vp, C.c_int]),
    "gymhip_atari_materialize_nchw_f32": (C.c_int, [_vp, _vp, _vp]),
    "gymhip_atari_materialize_nchw_16": (C.c_int, [_vp, _vp, C.c_int, _vp]),
}

_lib = None


def load():
    """Load libgymhip.so; raises GymhipError if it was not built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GymhipError(f"{LIB_PATH} not found: the HIP engine is not built and there is no CPU fallback")
    # torch first: its bundled HIP runtime (SONAME libamdhip64.so.7) must already be loaded so that libgymhip's NEEDED entry
    # resolves to that same copy — one HIP runtime per process (gym_amd/csrc/Makefile, INTEGRATION.md "link order").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != OK:
        msg = load().gymhip_last_error().decode("utf-8", "replace")
        raise GymhipError(f"{what or 'gymhip call'} failed (code {rc}): {msg}")
