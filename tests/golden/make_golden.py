#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

Provenance: the reference ships no golden vectors for this path and cannot be built in this image without stand-in
headers, so these vectors are produced by oracle/gym_oracle.cpp — whose pins are listed in its header and exercised by
tests/test_oracle_known_answers.py / test_oracle_atari.py.  They freeze the oracle's outputs so that (a) an accidental
change of the oracle is caught on CPU and (b) the HIP path is checked against committed data on the GPU box.
Inputs are regenerated from seeds (splitmix action tapes, counter-hash frames); only outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, Oracle, splitmix_actions  # noqa: E402
from tests.test_gpu_atari import counter_frames  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CHECK_STEPS = [0, 1, 2, 10, 50, 150, 299]


def main():
    o = Oracle()
    # --- CartPole, seed 0, 64 envs x 300 steps, TimeLimit(500)+Monitor (config 2 in miniature) --------------------
    n, T = 64, 300
    v = o.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    v.seed(0)
    obs0 = v.reset()
    r = v.rollout(splitmix_actions(1234, T, n, 2), rec_obs=True)
    np.savez_compressed(os.path.join(OUT, "cartpole_seed0_64x300.npz"), seed=0, action_seed=1234, reset_obs=obs0,
                        done=np.packbits(r["done"], axis=1), truncated=np.packbits(r["truncated"], axis=1),
                        obs_at=np.array(CHECK_STEPS), obs=r["obs"][CHECK_STEPS], final_obs64=r["final_obs64"],
                        reset_tape=o.cartpole_reset_tape(0, n, 40))
    # --- BASELINE configs[0]: ONE CartPole env, seed 0, TimeLimit(500), action tape a_t = mt19937(1)() & 1; first 8192 steps --
    K = 8192
    tape = (o.mt19937_raw(1, K) & 1).astype(np.uint8).reshape(K, 1)
    v = o.vec(CARTPOLE, 1, F_TIME_LIMIT | F_MONITOR, 500, threads=1)
    v.seed(0)
    obs0 = v.reset()
    r = v.rollout(tape, rec_obs=True)
    at = np.array([0, 1, 7, 100, 1000, K - 1])
    np.savez_compressed(os.path.join(OUT, "cartpole_config0_prefix.npz"), seed=0, action_mt19937_seed=1, steps=K, reset_obs=obs0,
                        done=np.packbits(r["done"][:, 0]), obs_at=at, obs=r["obs"][at, 0], final_obs64=r["final_obs64"],
                        episodes=int(r["done"].sum()))
    # --- MountainCar, 64 envs x 450 steps, TimeLimit(200)+ClipReward+Monitor+VecNormalize (config 3 in miniature) --
    n, T = 64, 450
    v = o.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, 200)
    v.attach_norm()
    nobs0, _ = v.norm_reset()
    acts = splitmix_actions(77, T, n, 3)
    stats, nobs, nrew, dones = [v.norm_stats()], [], [], []
    for t in range(T):
        s = v.norm_step(acts[t])
        dones.append(s["done"] != 0)
        if t % 50 == 49:
            stats.append(v.norm_stats()); nobs.append(s["obs"]); nrew.append(s["reward"])
    np.savez_compressed(os.path.join(OUT, "mountaincar_norm_64x450.npz"), action_seed=77, reset_obs_norm=nobs0, stats=np.stack(stats),
                        obs_norm=np.stack(nobs), reward_norm=np.stack(nrew), done=np.packbits(np.stack(dones), axis=1))
    # --- Atari preprocessing of two counter-hash frame pairs ----------------------------------------------------
    fr = counter_frames(7, 2)
    np.savez_compressed(os.path.join(OUT, "atari_preprocess_seed7.npz"), frame_seed=7, gray15=o.atari_preprocess(fr, gray15=1),
                        gray14=o.atari_preprocess(fr, gray15=0))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
