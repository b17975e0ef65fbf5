"""GPU: gymhip_step under HIP graph capture.  A step neither allocates nor synchronises; with host-drawn reset tapes and no
normaliser all of its per-step state lives on the device, so a recorded step can be replayed — checked against an engine
stepped eagerly and against the CPU oracle.  Where a replay would silently reuse host-side state (COUNTER_RNG step counter,
normaliser slot) the call refuses to be captured."""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_MONITOR, F_TIME_LIMIT, splitmix_actions

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["cartpole", "mountaincar"])
def test_recorded_step_replays_like_eager_steps(oracle, kind):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    cart = kind == "cartpole"
    n, T, seed, tl = 3000, 120, 4, 40
    mk = lambda: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                                     wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=tl), 0)
    graphed, eager = mk(), mk()
    ov = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, n, F_TIME_LIMIT | F_MONITOR, tl)
    if cart:
        ov.seed(seed)
        for e in (graphed, eager):
            e.seed_reset_tape(seed, T + 4)
    graphed.reset(); eager.reset()
    assert np.array_equal(graphed.obs.cpu().numpy(), ov.reset())
    acts = splitmix_actions(seed, T, n, 2 if cart else 3)
    static_actions = torch.zeros(n, dtype=torch.int32, device="cuda")
    # one eager step on a side stream first (torch's capture recipe), mirrored on the other engine and the oracle
    side = torch.cuda.Stream()
    a0 = torch.from_numpy(acts[0].astype(np.int32)).cuda()
    static_actions.copy_(a0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        graphed.step(static_actions)
    side.synchronize()
    eager.step(a0); ov.step(acts[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        graphed.step(static_actions)          # recorded, not executed
    for t in range(1, T):
        a = torch.from_numpy(acts[t].astype(np.int32)).cuda()
        static_actions.copy_(a)
        g.replay()
        eager.step(a)
        r = ov.step(acts[t])
        torch.cuda.synchronize()
        for name in ("obs", "reward", "done", "counter"):
            assert torch.equal(getattr(graphed, name), getattr(eager, name)), (t, name)
        assert np.array_equal(graphed.done.cpu().numpy(), r["done"]), t
        if cart:
            assert np.array_equal(graphed.obs.cpu().numpy(), r["obs"]), t
    assert torch.equal(graphed.state_f64(), eager.state_f64())
    assert graphed.poll_error() == 0 and graphed.episode_stats()[0] == eager.episode_stats()[0] > n


@pytest.mark.parametrize("what", ["counter_rng", "normaliser"])
def test_capture_is_refused_where_a_replay_would_reuse_host_state(what):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n = 512
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR
    if what == "normaliser":
        cfg = EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                           wrappers=w | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING, max_episode_steps=50)
    else:
        cfg = EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG, wrappers=w,
                           max_episode_steps=50, seed=1)
    eng = Engine(cfg, 0)
    eng.reset()
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        eng.step(a)                             # fine on a non-capturing stream
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception, match="captur"):
        with torch.cuda.graph(g, stream=side):
            eng.step(a)
    torch.cuda.synchronize()
    eng.step(a)                                 # and the engine is still usable
    torch.cuda.synchronize()
    assert eng.poll_error() == 0
