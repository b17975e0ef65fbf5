"""GPU: seeded random configurations of the fused step (env kind, wrapper set, TimeLimit length, env count, seed, action
dtype, step vs split begin/finish calls) against the CPU oracle — the combinations the hand-written tests do not
enumerate.  Same bar as everywhere: done / truncated / episode records bit-exact, observations 1e-5 relative."""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, splitmix_actions

pytestmark = pytest.mark.gpu


def _case(rng):
    cart = bool(rng.integers(0, 2))
    return dict(cart=cart, n=int(rng.choice([1, 2, 63, 64, 65, 257, 1000, 4097, 20001])), T=int(rng.integers(40, 260)),
                time_limit=int(rng.choice([0, 1, 2, 7, 33, 200, 500])), clip=bool(rng.integers(0, 2)), monitor=bool(rng.integers(0, 2)),
                seed=int(rng.integers(0, 1 << 30)), act=str(rng.choice(["i32", "i64", "u8", "f32", "host_i64"])), split=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("case_id", range(24))
def test_random_configuration_matches_oracle(oracle, case_id):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    c = _case(np.random.default_rng(1000 + case_id))
    cart, n, T = c["cart"], c["n"], c["T"]
    w = (L.WRAP_TIME_LIMIT if c["time_limit"] else 0) | (L.WRAP_CLIP_REWARD if c["clip"] else 0) | (L.WRAP_MONITOR if c["monitor"] else 0)
    f = (F_TIME_LIMIT if c["time_limit"] else 0) | (F_CLIP_REWARD if c["clip"] else 0) | (F_MONITOR if c["monitor"] else 0)
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64,
                              reset_mode=L.RESET_TAPE, wrappers=w, max_episode_steps=c["time_limit"]), 0)
    ov = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, n, f, c["time_limit"])
    if cart:
        ov.seed(c["seed"])
        eng.seed_reset_tape(c["seed"], T + 2)          # TimeLimit(1) ends an episode every step
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset()), c
    n_act = 2 if cart else 3
    acts = splitmix_actions(c["seed"] & 0xFFFF, T, n, n_act)
    finished = 0
    for t in range(T):
        a = acts[t]
        if c["act"] == "i32":
            dev = torch.from_numpy(a.astype(np.int32)).cuda()
        elif c["act"] == "i64":
            dev = torch.from_numpy(a.astype(np.int64)).cuda()
        elif c["act"] == "u8":
            dev = torch.from_numpy(a).cuda()
        elif c["act"] == "f32":
            dev = torch.from_numpy(a.astype(np.float32) + 0.25).cuda()     # TensorAdapter::decode truncates to int32 (tensor_adapter.cpp:22-23)
        else:
            dev = torch.from_numpy(a.astype(np.int64)).view(n, 1)          # host tensor [N, 1], as test/atari_env.cpp:60 passes it
        if c["split"]:
            eng.step_begin(dev); eng.step_finish()
        else:
            eng.step(dev)
        r = ov.step(a)
        d = r["done"] != 0
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), (c, t)
        assert np.array_equal(eng.reward.cpu().numpy(), r["reward"]), (c, t)
        g = eng.obs.cpu().numpy()
        assert (np.abs(g - r["obs"]) <= 1e-5 * np.maximum(np.abs(r["obs"]), 1e-2)).all(), (c, t)
        ln, tr, ret = (x.cpu().numpy() for x in eng.episode_info())
        if c["time_limit"]:
            assert np.array_equal(tr[d], r["truncated"][d]), (c, t)
        if c["monitor"]:
            assert np.array_equal(ln[d].astype(np.float32), r["ep_l"][d]) and np.array_equal(ret[d], r["ep_r"][d]), (c, t)
        finished += int(d.sum())
    assert eng.poll_error() == 0, c
    if c["monitor"]:
        assert eng.episode_stats()[0] == finished
    eng.close()


def _norm_case(rng):
    cart = bool(rng.integers(0, 2))
    return dict(cart=cart, n=int(rng.choice([1, 3, 64, 511, 513, 2049, 9000, 70001])), T=int(rng.integers(30, 140)),
                time_limit=int(rng.choice([0, 3, 17, 60, 200])), clip=bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1 << 30)),
                training=bool(rng.integers(0, 4)), norm_reward=bool(rng.integers(0, 4)), norm_obs=bool(rng.integers(0, 4)),
                clip_obs=float(rng.choice([10.0, 5.0, 1.5])), clip_rew=float(rng.choice([10.0, 2.0])), gamma=float(rng.choice([0.99, 0.9, 0.5])),
                eps=float(rng.choice([1e-8, 1e-4])), split=bool(rng.integers(0, 2)), load=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("case_id", range(16))
def test_random_vecnormalize_configuration_matches_oracle(oracle, case_id):
    """The fused VecNormalize arms (vec_normalize.h:20,145-177) under random template flags, clips, gamma, epsilon, env counts
    around the block size, statistics fresh or loaded, and the two call forms (gymhip_step = 2 launches, begin + finish = 3)."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    c = _norm_case(np.random.default_rng(5000 + case_id))
    if not (c["norm_reward"] or c["norm_obs"]):
        c["norm_obs"] = True
    cart, n, T, F = c["cart"], c["n"], c["T"], 4 if c["cart"] else 2
    w = (L.WRAP_TIME_LIMIT if c["time_limit"] else 0) | (L.WRAP_CLIP_REWARD if c["clip"] else 0) | L.WRAP_MONITOR
    w |= (L.WRAP_NORM_TRAINING if c["training"] else 0) | (L.WRAP_NORM_REWARD if c["norm_reward"] else 0) | (L.WRAP_NORM_OBS if c["norm_obs"] else 0)
    f = (F_TIME_LIMIT if c["time_limit"] else 0) | (F_CLIP_REWARD if c["clip"] else 0) | F_MONITOR
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                              wrappers=w, max_episode_steps=c["time_limit"], clip_obs=c["clip_obs"], clip_reward=c["clip_rew"], gamma=c["gamma"],
                              epsilon=c["eps"]), 0)
    ov = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, n, f, c["time_limit"])
    if cart:
        ov.seed(c["seed"])
        eng.seed_reset_tape(c["seed"], T + 2)
    ov.attach_norm(c["clip_obs"], c["clip_rew"], c["gamma"], c["eps"])
    ov.norm_flags(c["training"], c["norm_reward"], c["norm_obs"])
    if c["load"]:
        loaded = np.concatenate([np.linspace(-0.2, 0.1, F), np.linspace(0.05, 0.4, F), [300.0], [0.3, 1.7, 300.0]]).astype(np.float32)
        eng.set_norm_stats(loaded)
        ov.norm_set_stats(loaded)
    if c["split"]:
        eng.reset_begin(); eng.reset_finish()
    else:
        eng.reset()
    o_obs, o_raw = ov.norm_reset()
    assert np.array_equal(eng.obs.cpu().numpy(), o_raw), c
    acts = splitmix_actions(c["seed"] & 0xFFFF, T, n, 2 if cart else 3)
    dev = torch.from_numpy(acts).cuda()
    for t in range(T):
        if c["split"]:
            eng.step_begin(dev[t]); eng.step_finish()
        else:
            eng.step(dev[t])
        r = ov.norm_step(acts[t])
        st = ov.norm_stats()
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), (c, t)
        g_raw = eng.obs.cpu().numpy()
        if cart:
            assert np.array_equal(g_raw, r["raw_obs"]), (c, t)
        else:
            assert (np.abs(g_raw - r["raw_obs"]) <= 1e-5 * np.maximum(np.abs(r["raw_obs"]), 1e-2)).all(), (c, t)
        err = np.abs(eng.get_norm_stats() - st) / np.maximum(np.abs(st), 1e-6)
        assert err.max() <= 1e-5, (c, t, err)
        mean, var = st[:F], st[F:2 * F]
        cond = np.maximum(1.0, np.abs(mean) / np.sqrt(var + c["eps"]))
        g_obs = (eng.obs_norm if c["norm_obs"] else eng.obs).cpu().numpy()
        assert (np.abs(g_obs - r["obs"]) <= 1e-5 * cond * np.maximum(1.0, np.abs(r["obs"]))).all(), (c, t)
        g_rew = eng.reward.cpu().numpy()
        assert (np.abs(g_rew - r["reward"]) <= 1e-5 * np.maximum(1.0, np.abs(r["reward"]))).all(), (c, t)
    assert eng.poll_error() == 0, c
    eng.close()


@pytest.mark.parametrize("case_id", range(8))
def test_random_auto_reset_off_schedules_match_oracle(oracle, case_id):
    """SyncVecEnv(envs, auto_reset = false) (sync_env.h:28,92,128-139) at random env kinds, env counts, TimeLimit lengths and
    seeds: terminal observations are returned, each finished env is reset alone by reset(index) before the next step."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    rng = np.random.default_rng(9000 + case_id)
    cart = bool(rng.integers(0, 2))
    n, T = int(rng.choice([1, 5, 64, 130, 700])), int(rng.integers(60, 200))
    tl = int(rng.choice([5, 23, 80]))
    seed = int(rng.integers(0, 1 << 20))
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | L.WRAP_NO_AUTO_RESET
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                              wrappers=w, max_episode_steps=tl), 0)
    ov = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, n, F_TIME_LIMIT | F_MONITOR, tl)
    ov.set_auto_reset(False)
    if cart:
        ov.seed(seed)
        eng.seed_reset_tape(seed, T + 2)
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    acts = splitmix_actions(seed & 0xFFFF, T, n, 2 if cart else 3)
    episodes = 0
    for t in range(T):
        eng.step(torch.from_numpy(acts[t].astype(np.int32)).cuda())
        r = ov.step(acts[t])
        g_done, g_rew, g_obs = eng.done.cpu().numpy(), eng.reward.cpu().numpy(), eng.obs.cpu().numpy()
        assert np.array_equal(g_done, r["done"]) and np.array_equal(g_rew, r["reward"]), (case_id, t)
        if cart:
            assert np.array_equal(g_obs, r["obs"]), (case_id, t)
        else:
            assert np.abs(g_obs - r["obs"]).max() <= 1e-7, (case_id, t)
        for i in np.nonzero(r["done"])[0]:
            episodes += 1
            eng.reset_index(int(i))
            orow, _ = ov.reset_index(int(i))
            assert np.array_equal(eng.obs[int(i)].cpu().numpy(), orow), (case_id, t, i)
    assert eng.poll_error() == 0 and eng.episode_stats()[0] == episodes and episodes > 0
    eng.close()
