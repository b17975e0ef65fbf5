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
