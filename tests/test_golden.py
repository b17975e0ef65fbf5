"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the CPU oracle).

CPU leg: the oracle still reproduces them bit for bit.  GPU leg: the HIP path reproduces them through the C ABI
without consulting the oracle at run time."""
import os

import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, splitmix_actions
from tests.test_gpu_atari import counter_frames

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ---------------------------------------------------------------- CPU: oracle vs golden
def test_oracle_reproduces_cartpole_golden(oracle):
    g = load("cartpole_seed0_64x300.npz")
    n, T = 64, 300
    v = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    v.seed(int(g["seed"]))
    assert np.array_equal(v.reset(), g["reset_obs"])
    r = v.rollout(splitmix_actions(int(g["action_seed"]), T, n, 2), rec_obs=True)
    assert np.array_equal(np.packbits(r["done"], axis=1), g["done"])
    assert np.array_equal(np.packbits(r["truncated"], axis=1), g["truncated"])
    assert np.array_equal(r["obs"][g["obs_at"]], g["obs"])
    assert np.array_equal(r["final_obs64"], g["final_obs64"])
    assert np.array_equal(oracle.cartpole_reset_tape(0, n, 40), g["reset_tape"])
    assert np.array_equal(g["reset_tape"][0], g["reset_obs"])


def _config0_tape(oracle, K):
    """BASELINE configs[0]'s action tape: a_t = mt19937(1)() & 1."""
    return (oracle.mt19937_raw(1, K) & 1).astype(np.uint8).reshape(K, 1)


def test_oracle_reproduces_config0_prefix_golden(oracle):
    """BASELINE configs[0] (single env, seed 0, TimeLimit(500), mt19937(1) action tape): first 8192 steps."""
    g = load("cartpole_config0_prefix.npz")
    K = int(g["steps"])
    v = oracle.vec(CARTPOLE, 1, F_TIME_LIMIT | F_MONITOR, 500, threads=1)
    v.seed(0)
    assert np.array_equal(v.reset(), g["reset_obs"])
    r = v.rollout(_config0_tape(oracle, K), rec_obs=True)
    assert np.array_equal(np.packbits(r["done"][:, 0]), g["done"]) and int(r["done"].sum()) == int(g["episodes"])
    assert np.array_equal(r["obs"][g["obs_at"], 0], g["obs"]) and np.array_equal(r["final_obs64"], g["final_obs64"])
    # the survey's probe of the compiled reference: seed 0 resets to (0.0048813522, 0.0092844591, 0.0215189345, 0.0344265737)
    assert np.allclose(g["reset_obs"][0], [0.0048813522, 0.0092844591, 0.0215189345, 0.0344265737], rtol=0, atol=5e-11)


def test_oracle_reproduces_mountaincar_norm_golden(oracle):
    g = load("mountaincar_norm_64x450.npz")
    n, T = 64, 450
    v = oracle.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, 200)
    v.attach_norm()
    assert np.array_equal(v.norm_reset()[0], g["reset_obs_norm"])
    acts = splitmix_actions(int(g["action_seed"]), T, n, 3)
    stats, k = [v.norm_stats()], 0
    done = np.unpackbits(g["done"], axis=1)[:, :n].astype(bool)
    for t in range(T):
        s = v.norm_step(acts[t])
        assert np.array_equal(s["done"] != 0, done[t])
        if t % 50 == 49:
            assert np.array_equal(s["obs"], g["obs_norm"][k]) and np.array_equal(s["reward"], g["reward_norm"][k])
            stats.append(v.norm_stats()); k += 1
    assert np.array_equal(np.stack(stats), g["stats"])


def test_oracle_reproduces_atari_golden(oracle):
    g = load("atari_preprocess_seed7.npz")
    fr = counter_frames(int(g["frame_seed"]), 2)
    assert np.array_equal(oracle.atari_preprocess(fr, gray15=1), g["gray15"])
    assert np.array_equal(oracle.atari_preprocess(fr, gray15=0), g["gray14"])


# ---------------------------------------------------------------- GPU: HIP path vs golden (no oracle call)
@pytest.mark.gpu
def test_hip_reproduces_cartpole_golden():
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    g = load("cartpole_seed0_64x300.npz")
    n, T = 64, 300
    for via_library_seed in (False, True):
        eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                                  wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500), 0)
        if via_library_seed:
            eng.seed_reset_tape(0, 40)            # the library's own std::mt19937 draw (gymhip_seed_reset_tape)
            assert np.array_equal(eng.download_reset_tape(40), g["reset_tape"])
        else:
            eng.upload_reset_tape(g["reset_tape"])
        eng.reset()
        assert np.array_equal(eng.obs.cpu().numpy(), g["reset_obs"])
        acts = torch.from_numpy(splitmix_actions(int(g["action_seed"]), T, n, 2)).cuda()
        done, trunc, obs = [], [], {}
        for t in range(T):
            eng.step(acts[t])
            done.append(eng.done.cpu().numpy() != 0)
            trunc.append(eng.episode_info()[1].cpu().numpy() != 0)
            if t in g["obs_at"]:
                obs[t] = eng.obs.cpu().numpy()
        assert np.array_equal(np.packbits(np.stack(done), axis=1), g["done"])
        assert np.array_equal(np.packbits(np.stack(trunc), axis=1), g["truncated"])
        for k, t in enumerate(g["obs_at"]):
            assert np.array_equal(obs[int(t)], g["obs"][k])
        assert np.array_equal(eng.state_f64().cpu().numpy(), g["final_obs64"])


@pytest.mark.gpu
def test_hip_reproduces_config0_prefix_golden(oracle):
    """configs[0] on the HIP path: ONE env (a 1-thread grid), library-drawn std::mt19937(0) resets, the 8192-step prefix as
    one rollout launch and again as 8192 single-step launches (the oracle is used only to draw the mt19937(1) action tape)."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    g = load("cartpole_config0_prefix.npz")
    K = int(g["steps"])
    tape = torch.from_numpy(_config0_tape(oracle, K)).cuda()
    for single_steps in (False, True):
        eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=1, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                                  wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500), 0)
        eng.seed_reset_tape(0, K // 8 + 2)
        eng.reset()
        assert np.array_equal(eng.obs.cpu().numpy(), g["reset_obs"])
        if single_steps:
            done = torch.empty((K, 1), dtype=torch.uint8, device="cuda")
            obs = {}
            for t in range(K):
                eng.step(tape[t])
                done[t] = eng.done.to(torch.uint8)
                if t in g["obs_at"]:
                    obs[t] = eng.obs.cpu().numpy()[0]
            got_obs = np.stack([obs[int(t)] for t in g["obs_at"]])
        else:
            done = torch.empty((K, 1), dtype=torch.uint8, device="cuda")
            rec = torch.empty((K, 1, 4), dtype=torch.float32, device="cuda")
            eng.rollout(tape, rec_done=done, rec_obs=rec)
            got_obs = rec.cpu().numpy()[g["obs_at"], 0]
        assert np.array_equal(np.packbits(done.cpu().numpy()[:, 0]), g["done"])
        assert np.array_equal(got_obs, g["obs"])
        assert np.array_equal(eng.state_f64().cpu().numpy(), g["final_obs64"])
        assert int(eng.episode_stats()[0]) == int(g["episodes"]) and eng.poll_error() == 0


@pytest.mark.gpu
def test_hip_reproduces_mountaincar_norm_golden():
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    g = load("mountaincar_norm_64x450.npz")
    n, T = 64, 450
    w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, wrappers=w, max_episode_steps=200), 0)
    eng.reset()
    assert np.allclose(eng.obs_norm.cpu().numpy(), g["reset_obs_norm"], rtol=1e-5, atol=1e-6)
    acts = torch.from_numpy(splitmix_actions(int(g["action_seed"]), T, n, 3)).cuda()
    done = np.unpackbits(g["done"], axis=1)[:, :n].astype(bool)
    k = 0
    for t in range(T):
        eng.step(acts[t])
        assert np.array_equal(eng.done.cpu().numpy() != 0, done[t])
        if t % 50 == 49:
            k += 1
            st = g["stats"][k]
            assert np.allclose(eng.get_norm_stats(), st, rtol=1e-5, atol=1e-9)
            cond = np.maximum(1.0, np.abs(st[:2]) / np.sqrt(st[2:4] + 1e-8))
            assert (np.abs(eng.obs_norm.cpu().numpy() - g["obs_norm"][k - 1]) <= 1e-5 * cond * np.maximum(1, np.abs(g["obs_norm"][k - 1]))).all()
            assert np.allclose(eng.reward.cpu().numpy(), g["reward_norm"][k - 1], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_reproduces_atari_golden():
    from gym_amd.atari import AtariPreprocessor
    g = load("atari_preprocess_seed7.npz")
    fr = torch.from_numpy(counter_frames(int(g["frame_seed"]), 2)).cuda()
    for bits, key in ((15, "gray15"), (14, "gray14")):
        got = AtariPreprocessor(2, gray_bits=bits).preprocess(fr).cpu().numpy().astype(int)
        assert np.abs(got - g[key].astype(int)).max() <= 1          # +-1 LSB contract (bit-exact in practice)
