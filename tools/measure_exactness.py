#!/usr/bin/env python3
"""How exact is the lockstep MountainCar path?  Its only non-reference arithmetic is the device library's double cos(3*pos)
(mountain_car.cpp:19 calls glibc's); unlike CartPole the double result is used without a float narrowing, so a 1-ulp
difference is visible in the fp64 state.  Prints, over n envs x T steps: f32 observation values that differ from the
oracle's, fp64 state values that differ at the end, and the largest fp64 difference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_amd import _lib as L  # noqa: E402
from gym_amd.engine import Engine, EngineConfig  # noqa: E402
from tests.oracle_lib import MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, Oracle, splitmix_actions  # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
o = Oracle()
ov = o.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, 200)
eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, wrappers=L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR, max_episode_steps=200), 0)
eng.reset(); ov.reset()
acts = splitmix_actions(77, T, n, 3)
dev = torch.from_numpy(acts).cuda()
bad32 = 0
worst = 0.0
for t in range(T):
    eng.step(dev[t])
    r = ov.step(acts[t], want64=True)
    bad32 += int((eng.obs.cpu().numpy() != r["obs"]).sum())
    assert np.array_equal(eng.done.cpu().numpy(), r["done"])
    if t % 50 == 49:
        g64 = eng.state_f64().cpu().numpy()
        worst = max(worst, float(np.abs(g64 - r["obs64"]).max()))
g64 = eng.state_f64().cpu().numpy()
print(f"mountaincar lockstep {n} envs x {T} steps: f32 obs values differing {bad32} of {n * T * 2}; final fp64 state values differing "
      f"{int((g64 != r['obs64']).sum())} of {g64.size}; largest |fp64 difference| seen {worst:.3e}")
