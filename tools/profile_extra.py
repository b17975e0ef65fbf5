"""Workload for the secondary-kernel trace (tools/profile_r01.sh): BASELINE config 3 (MountainCar + VecNormalize, the
step / reduce / normalise launch triple) at 16,777,216 envs and config 4 (Atari preprocess + stack materialisation) at
4096 envs, 40 steps each.  Run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd import _lib as L
from gym_amd.atari import AtariPreprocessor
from gym_amd.engine import Engine, EngineConfig

torch.cuda.set_device(0)
n = 1 << 24
w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                          wrappers=w, max_episode_steps=200), 0)
eng.reset()
acts = torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda")
for _ in range(40):
    eng.step(acts)
torch.cuda.synchronize()
eng.close()
del eng, acts
torch.cuda.empty_cache()

m = 4096
pp = AtariPreprocessor(m)
frames = torch.randint(0, 256, (m, 2, 210, 160, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((m, 4, 84, 84), dtype=torch.float32, device="cuda")
out16 = torch.empty((m, 4, 84, 84), dtype=torch.bfloat16, device="cuda")
pp.reset(frames)
for _ in range(40):
    pp.step(frames)
    pp.stacked_nchw_f32(out)
    pp.stacked_nchw_16(torch.bfloat16, out16)
torch.cuda.synchronize()
pp.close()
print("PROFILE_EXTRA_OK")
