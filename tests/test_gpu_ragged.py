"""GPU: env counts that are not multiples of the block, wave or vector widths (1, 3, 255, 257, 1001, 4099): every tail path —
the last partial block, the pipelined loop's clamped prefetch, the 2- and 4-env vectors of the normalise passes, the
single-block partial reduce — against the CPU oracle, same bar as the aligned sizes."""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, splitmix_actions

pytestmark = pytest.mark.gpu

SIZES = [1, 3, 255, 257, 1001, 4099]


@pytest.mark.parametrize("n", SIZES)
def test_cartpole_lockstep_ragged(oracle, n):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    T, seed = 80, 23
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(seed)
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                              wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500), 0)
    eng.upload_reset_tape(oracle.cartpole_reset_tape(seed, n, T // 8 + 4))
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    acts = splitmix_actions(3, T, n, 2)
    dev = torch.from_numpy(acts.astype(np.int64)).cuda()
    episodes = 0
    for t in range(T):
        eng.step(dev[t])
        r = ov.step(acts[t], want64=True)
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        assert np.array_equal(eng.reward.cpu().numpy(), r["reward"]), t
        assert np.array_equal(eng.obs.cpu().numpy(), r["obs"]), t
        episodes += int(r["done"].sum())
    assert np.array_equal(eng.state_f64().cpu().numpy(), r["obs64"])
    assert eng.episode_stats()[0] == episodes and eng.poll_error() == 0
    # the rollout form from the same point
    tape = splitmix_actions(4, 16, n, 2)
    eng.rollout(torch.from_numpy(tape).cuda())
    rr = ov.rollout(tape)
    assert np.allclose(eng.obs.cpu().numpy(), rr["final_obs"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n", SIZES)
def test_mountaincar_vecnormalize_ragged(oracle, n):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    T = 40
    w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    ov = oracle.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, 25)      # short limit: resets happen inside the run
    ov.attach_norm()
    eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                              wrappers=w, max_episode_steps=25), 0)
    eng.reset()
    o_obs, o_raw = ov.norm_reset()
    assert np.array_equal(eng.obs.cpu().numpy(), o_raw)
    acts = splitmix_actions(8, T, n, 3)
    dev = torch.from_numpy(acts).cuda()
    for t in range(T):
        eng.step(dev[t])
        r = ov.norm_step(acts[t])
        assert np.array_equal(eng.obs.cpu().numpy(), r["raw_obs"]), t
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        st = ov.norm_stats()
        g = eng.get_norm_stats()
        assert (np.abs(g - st) <= 1e-5 * np.maximum(np.abs(st), 1e-6)).all(), (t, g, st)
        cond = np.maximum(1.0, np.abs(st[:2]) / np.sqrt(st[2:4] + 1e-8))
        assert (np.abs(eng.obs_norm.cpu().numpy() - r["obs"]) <= 1e-5 * cond * np.maximum(1.0, np.abs(r["obs"]))).all(), t
        assert (np.abs(eng.reward.cpu().numpy() - r["reward"]) <= 1e-5 * np.maximum(1.0, np.abs(r["reward"]))).all(), t
    assert int(r["done"].sum()) >= 0 and eng.episode_stats()[0] == n * (T // 25)
