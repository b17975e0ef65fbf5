"""BASELINE configs[2] (MountainCar + TimeLimit(200) + ClipReward + Monitor + obs/return normaliser) at a given env count:
prints wall-clock us per VecEnv::step; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
    python tools/profile_config3.py <envs> <steps>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
torch.cuda.set_device(0)
w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                          wrappers=w, max_episode_steps=200), 0)
eng.reset()
acts = [torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda") for _ in range(4)]
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for k in range(16):
        eng.step(acts[k % 4])
    torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    eng.step(acts[k % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"CONFIG3 envs={n} steps={steps} us_per_step={dt * 1e6:.2f} env_steps_per_s={n / dt:.4e} GBs={84 * n / dt / 1e9:.1f} frac={84 * n / dt / 8e12:.4f}")
eng.close()
