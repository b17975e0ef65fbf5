"""GPU: gymhip_step under HIP graph capture.  A step neither allocates nor synchronises; with host-drawn reset tapes and no
normaliser all of its per-step state lives on the device, so a recorded step can be replayed — checked against an engine
stepped eagerly and against the CPU oracle — and so do the counter-RNG episode index and the normaliser's statistics slot."""
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


@pytest.mark.parametrize("what", ["counter_rng", "normaliser", "normaliser_split", "cartpole_normaliser_rng"])
def test_recorded_step_replays_with_device_side_step_state(what):
    """The two values a step used to take from the host — the step counter that keyed COUNTER_RNG resets and the normaliser's
    current statistics slot — live on the device (each env's episode index in its counter word; the tick pair), so a step
    recorded once replays like eager stepping: every output, the running statistics and the fp64 state bit for bit."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n, T = 5000, 150
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR
    norm = L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    if what == "counter_rng":
        cfg = dict(env_kind=L.ENV_CARTPOLE, reset_mode=L.RESET_COUNTER_RNG, wrappers=w, max_episode_steps=30, seed=5)
    elif what == "cartpole_normaliser_rng":
        cfg = dict(env_kind=L.ENV_CARTPOLE, reset_mode=L.RESET_COUNTER_RNG, wrappers=w | norm, max_episode_steps=30, seed=6)
    else:
        cfg = dict(env_kind=L.ENV_MOUNTAINCAR, reset_mode=L.RESET_TAPE, wrappers=w | L.WRAP_CLIP_REWARD | norm, max_episode_steps=40)
    mk = lambda: Engine(EngineConfig(n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, **cfg), 0)
    graphed, eager = mk(), mk()
    split = what == "normaliser_split"

    def step(e, a):
        if split:
            e.step_begin(a); e.step_finish()
        else:
            e.step(a)

    graphed.reset(); eager.reset()
    n_act = 2 if cfg["env_kind"] == L.ENV_CARTPOLE else 3
    g_ = torch.Generator(device="cuda").manual_seed(2)
    acts = torch.randint(0, n_act, (T, n), dtype=torch.int32, device="cuda", generator=g_)
    static_actions = acts[0].clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        step(graphed, static_actions)
    side.synchronize()
    step(eager, acts[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step(graphed, static_actions)           # recorded, not executed
    names = ["obs", "reward", "done", "counter"] + (["obs_norm", "returns", "norm_stats"] if what != "counter_rng" else [])
    for t in range(1, T):
        static_actions.copy_(acts[t])
        g.replay()
        step(eager, acts[t])
        torch.cuda.synchronize()
        for name in names:
            assert torch.equal(getattr(graphed, name), getattr(eager, name)), (t, name)
    assert torch.equal(graphed.state_f64(), eager.state_f64())
    if what != "counter_rng":
        assert np.array_equal(graphed.get_norm_stats(), eager.get_norm_stats())
    assert graphed.poll_error() == 0 and graphed.episode_stats()[0] == eager.episode_stats()[0] > n
    # and eager stepping continues from the replayed state
    step(graphed, acts[0]); step(eager, acts[0])
    torch.cuda.synchronize()
    assert torch.equal(graphed.obs, eager.obs) and torch.equal(graphed.counter, eager.counter)


def test_counter_rng_resets_equal_a_tape_filled_by_the_same_generator():
    """With the counter-RNG draw keyed by (seed, global env index, the env's episode index) the in-kernel generator and a device
    tape filled by the same generator (gymhip_generate_reset_tape) are two routes to the same reset states: an engine in
    COUNTER_RNG mode and one in TAPE mode over that tape produce identical streams — a cross-check of the tape gather, the cursor
    bookkeeping and the hash, including a sharded engine (env_offset) and a second reset()."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n, T, seed, off = 6000, 260, 12, 1000
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR
    mk = lambda mode: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=mode, wrappers=w,
                                          max_episode_steps=25, env_offset=off, seed=seed), 0)
    rng, tape = mk(L.RESET_COUNTER_RNG), mk(L.RESET_TAPE)
    tape.generate_reset_tape(T + 8)
    rng.reset(); tape.reset()
    assert torch.equal(rng.obs, tape.obs)
    g = torch.Generator(device="cuda").manual_seed(9)
    for t in range(T):
        a = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda", generator=g)
        rng.step(a); tape.step(a)
        assert torch.equal(rng.obs, tape.obs) and torch.equal(rng.done, tape.done) and torch.equal(rng.counter, tape.counter), t
        if t == T // 2:
            rng.reset(); tape.reset()                 # a second VecEnv::reset(): both continue from each env's episode index
            assert torch.equal(rng.obs, tape.obs) and torch.equal(rng.counter, tape.counter)
    assert torch.equal(rng.state_f64(), tape.state_f64())
    assert rng.poll_error() == 0 and tape.poll_error() == 0 and rng.episode_stats()[0] == tape.episode_stats()[0] > 5 * n


def test_recorded_step_follows_a_replaced_reset_tape_and_a_new_seed(oracle):
    """The reset source is device state (a block the kernels read through a pointer), not launch arguments: a step recorded BEFORE
    gymhip_extend_reset_tape — which allocates a larger tape and frees the old one — gathers from the new tape when replayed (it used
    to read freed memory with the old row count), and a recorded COUNTER_RNG step draws with the seed gymhip_seed() set afterwards."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n, T, seed, tl = 2000, 160, 8, 6
    mk = lambda mode: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=mode,
                                          wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=tl, seed=seed), 0)
    for mode in (L.RESET_TAPE, L.RESET_COUNTER_RNG):
        graphed, eager = mk(mode), mk(mode)
        if mode == L.RESET_TAPE:
            graphed.seed_reset_tape(seed, 12)          # 12 rows: exhausted after ~70 steps of TimeLimit(6)
            eager.seed_reset_tape(seed, T)             # the eager twin has the whole tape from the start
        graphed.reset(); eager.reset()
        g_ = torch.Generator(device="cuda").manual_seed(4)
        acts = torch.randint(0, 2, (T, n), dtype=torch.int32, device="cuda", generator=g_)
        static_actions = acts[0].clone()
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            graphed.step(static_actions)
        side.synchronize()
        eager.step(acts[0])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            graphed.step(static_actions)
        for t in range(1, T):
            if t == 40:
                torch.cuda.synchronize()
                if mode == L.RESET_TAPE:
                    graphed.extend_reset_tape(T)       # new allocation, old tape freed; the graph is NOT re-recorded
                else:
                    graphed.seed(seed + 5); eager.seed(seed + 5)
            static_actions.copy_(acts[t])
            g.replay()
            eager.step(acts[t])
            torch.cuda.synchronize()
            for name in ("obs", "reward", "done", "counter"):
                assert torch.equal(getattr(graphed, name), getattr(eager, name)), (mode, t, name)
        assert graphed.poll_error() == 0 and eager.poll_error() == 0
        assert graphed.episode_stats()[0] > n * (T // tl - 1)
