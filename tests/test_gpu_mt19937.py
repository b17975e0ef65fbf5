"""GPU: GYMHIP_RESET_MT19937 — every env's std::mt19937 lives on the device and the step kernel draws CartPole's reset states from
it (cartpole.cpp:19-31, env.h:60-68,83): libstdc++'s seeding, in-place twist (one per 156 resets of an env, done by the whole wave
through LDS), tempering and uniform_real_distribution<float> arithmetic.  Bar: bit-identical to the CPU oracle (whose envs own real
std::mt19937 objects) and to an engine fed with a HOST-drawn tape of the same streams, across several regenerations of every
env's generator block, for ragged env counts, both precisions, reset() / reset(index) / seed() mid-run, the rollout kernel, the
normaliser variant and HIP-graph replays.  All calls go through the C ABI."""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, F_MONITOR, F_TIME_LIMIT, splitmix_actions

pytestmark = pytest.mark.gpu


def _engine(n, reset_mode, max_steps, precision=None, wrappers=None, env_offset=0, seed=0):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    return Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64 if precision is None else precision,
                               reset_mode=reset_mode, wrappers=(L.WRAP_TIME_LIMIT | L.WRAP_MONITOR) if wrappers is None else wrappers,
                               max_episode_steps=max_steps, env_offset=env_offset, seed=seed), 0)


def _temper(y):
    y = y.astype(np.uint32)
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
    y ^= y >> np.uint32(18)
    return y


def test_device_generators_equal_libstdcxx_mt19937(oracle):
    """Seeding + first twist + tempering against libstdc++ itself: after the first reset env i's state array, tempered, IS outputs
    0..623 of std::mt19937(seed + env_offset + i); after enough resets to regenerate the block, outputs 624..1247.  Also the
    never-seeded engine: every env is the default-constructed generator (5489)."""
    from gym_amd import _lib as L
    n, seed, off = 200, 123456789012, 77            # the seed is reduced mod 2^32 as mersenne_twister_engine::seed does
    eng = _engine(n, L.RESET_MT19937, 1, env_offset=off)
    eng.seed(seed)
    eng.reset()
    st = eng.mt_state.cpu().numpy().view(np.uint32)
    for i in (0, 1, 63, 64, 199):
        want = oracle.mt19937_raw(seed + off + i, 1248)
        assert np.array_equal(_temper(st[i]), want[:624]), i
    # TimeLimit(1): every step resets every env; 156 rows per block -> after 156 more resets the block has been regenerated once
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(156):
        eng.step(a)
    assert int(eng.counters()["cursor"].min()) == int(eng.counters()["cursor"].max()) == 157
    st = eng.mt_state.cpu().numpy().view(np.uint32)
    for i in (0, 63, 64, 199):
        want = oracle.mt19937_raw(seed + off + i, 1248)
        assert np.array_equal(_temper(st[i]), want[624:]), i
    assert eng.poll_error() == 0
    fresh = _engine(130, L.RESET_MT19937, 1)
    fresh.reset()
    st = fresh.mt_state.cpu().numpy().view(np.uint32)
    want = oracle.mt19937_raw(5489, 624)
    assert np.array_equal(_temper(st[0]), want) and np.array_equal(_temper(st[129]), want)


def test_mt19937_config2_size_rollout_vs_oracle_across_regenerations(oracle):
    """BASELINE configs[1]'s shape with the generators on the device: N = 65,536, 12,000-step trajectories (every env's generator
    block is regenerated at least 3 times after the first), rollout kernel, against the oracle's per-env std::mt19937 objects: done /
    truncated / reward streams for every step, fp64 state and observations at every chunk boundary, all observations of the last chunk."""
    from gym_amd import _lib as L
    n, T, chunk, seed = 65536, 12000, 500, 3
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(seed)
    eng = _engine(n, L.RESET_MT19937, 500)
    eng.seed(seed)
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    rec_done = torch.empty((chunk, n), dtype=torch.uint8, device="cuda")
    rec_trunc = torch.empty((chunk, n), dtype=torch.uint8, device="cuda")
    rec_rew = torch.empty((chunk, n), dtype=torch.float32, device="cuda")
    rec_obs = torch.empty((chunk, n, 4), dtype=torch.float32, device="cuda")
    episodes = 0
    for c in range(T // chunk):
        last = c == T // chunk - 1
        acts = splitmix_actions(4321 + c * chunk * n, chunk, n, 2)
        eng.rollout(torch.from_numpy(acts).cuda(), rec_done, rec_rew, rec_trunc, rec_obs if last else None)
        r = ov.rollout(acts, rec_obs=last)
        assert np.array_equal(rec_done.cpu().numpy(), r["done"]), f"done stream diverged in chunk {c}"
        assert np.array_equal(rec_trunc.cpu().numpy(), r["truncated"]), f"truncated stream diverged in chunk {c}"
        assert np.array_equal(rec_rew.cpu().numpy(), r["reward"])
        assert np.array_equal(eng.obs.cpu().numpy(), r["final_obs"]), f"obs diverged in chunk {c}"
        assert np.array_equal(eng.state_f64().cpu().numpy(), r["final_obs64"]), f"fp64 state diverged in chunk {c}"
        if last:
            assert np.array_equal(rec_obs.cpu().numpy(), r["obs"])
        episodes += int(r["done"].sum())
    cur = eng.counters()["cursor"]
    assert int(cur.min()) > 3 * 156, int(cur.min())        # >= 3 regenerations after the first twist, for EVERY env
    assert eng.poll_error() == 0 and eng.episode_stats()[0] == episodes


@pytest.mark.parametrize("precision", ["lockstep", "f32"])
def test_mt19937_step_api_equals_host_drawn_tape(oracle, precision):
    """gymhip_step with device generators against the same engine fed a tape the HOST drew with std::mt19937
    (gymhip_seed_reset_tape): every buffer equal after every step, over > 2 regenerations per env; ragged env count (partial
    waves), a sharded engine (env_offset), a second reset(), reset(index), and a mid-run re-seed."""
    from gym_amd import _lib as L
    n, T, seed, off, tl = 5003, 3000, 9, 4096, 9
    prec = L.PRECISION_LOCKSTEP_F64 if precision == "lockstep" else L.PRECISION_F32
    mt = _engine(n, L.RESET_MT19937, tl, prec, env_offset=off)
    tp = _engine(n, L.RESET_TAPE, tl, prec, env_offset=off)
    mt.seed(seed); tp.seed_reset_tape(seed, T // 8 + 16)
    mt.reset(); tp.reset()
    g = torch.Generator(device="cuda").manual_seed(1)

    def same(t):
        for name in ("obs", "reward", "done", "counter"):
            assert torch.equal(getattr(mt, name), getattr(tp, name)), (t, name)

    same(-1)
    for t in range(T):
        a = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda", generator=g)
        mt.step(a); tp.step(a)
        same(t)
        if t == 700:
            mt.reset(); tp.reset(); same("reset")                        # a second VecEnv::reset() draws each env's NEXT row
        if t in (900, 901, 1500):
            for idx in (0, 63, 64, n - 1, (t * 7) % n):
                mt.reset_index(idx); tp.reset_index(idx)
            same("reset_index")
        if t == 2000:                                                      # VecEnv::seed mid-run: every stream restarts (env.h:60-63)
            mt.seed(seed + 100); tp.seed_reset_tape(seed + 100, T // 8 + 16)
            same("seed")
    assert torch.equal(mt.state_f64(), tp.state_f64())
    assert int(mt.counters()["cursor"].max()) > 100
    assert mt.poll_error() == 0 and tp.poll_error() == 0 and mt.episode_stats()[0] == tp.episode_stats()[0] > n * 250


def test_mt19937_unseeded_and_normaliser_variants_equal_the_tape_engine():
    """Never-seeded envs (all draw the default-constructed stream, env.h:83) and the normaliser instance of the step kernel."""
    from gym_amd import _lib as L
    n, T = 3000, 1700
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    mt = _engine(n, L.RESET_MT19937, 10, wrappers=w)
    tp = _engine(n, L.RESET_TAPE, 10, wrappers=w)
    tp.seed_reset_tape_unseeded(T // 8 + 8)
    mt.reset(); tp.reset()
    g = torch.Generator(device="cuda").manual_seed(2)
    for t in range(T):
        a = torch.randint(0, 2, (n,), dtype=torch.int64, device="cuda", generator=g)
        mt.step(a); tp.step(a)
        if t % 50 == 0 or t == T - 1:
            for name in ("obs", "obs_norm", "reward", "done", "counter", "returns", "norm_stats"):
                assert torch.equal(getattr(mt, name), getattr(tp, name)), (t, name)
    assert int(mt.counters()["cursor"].min()) > 156
    assert mt.poll_error() == 0 and tp.poll_error() == 0


def test_mt19937_recorded_step_replays_across_regenerations_and_a_reseed():
    """The generator state, its read position and the seed are all device state: a step recorded once into a HIP graph replays
    like eager stepping across regenerations of the generator blocks, and keeps following the engine after gymhip_seed()."""
    from gym_amd import _lib as L
    n, T = 4000, 900
    graphed, eager = _engine(n, L.RESET_MT19937, 4), _engine(n, L.RESET_MT19937, 4)
    for e in (graphed, eager):
        e.seed(21); e.reset()
    g_ = torch.Generator(device="cuda").manual_seed(3)
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
        if t == 500:
            graphed.seed(99); eager.seed(99)                               # outside the graph; the replays follow
        static_actions.copy_(acts[t])
        g.replay()
        eager.step(acts[t])
        if t % 25 == 0 or t == T - 1:
            torch.cuda.synchronize()
            for name in ("obs", "reward", "done", "counter"):
                assert torch.equal(getattr(graphed, name), getattr(eager, name)), (t, name)
    assert torch.equal(graphed.state_f64(), eager.state_f64()) and torch.equal(graphed.mt_state, eager.mt_state)
    assert int(graphed.counters()["cursor"].min()) >= (T - 500) // 4 - 1
    assert graphed.poll_error() == 0


def test_mt19937_engine_refuses_tapes_and_cursor_period_is_a_whole_number_of_blocks():
    from gym_amd import _lib as L
    eng = _engine(64, L.RESET_MT19937, 5)
    with pytest.raises(L.GymhipError):
        eng.seed_reset_tape(1, 8)
    with pytest.raises(L.GymhipError):
        eng.generate_reset_tape(8)
    assert 1_048_320 % 156 == 0 and 1_048_320 < (1 << 20)


def test_mt19937_cursor_wraps_on_a_block_boundary_and_the_stream_goes_on(oracle):
    """The 20-bit cursor field counts an env's reset rows modulo 1,048,320 = 6,720 blocks of 156 rows.  TimeLimit(1) makes every
    step a reset; 130 envs (a full and a partial wave) are driven to three rows before the wrap by the rollout kernel, then across
    it through gymhip_step: the oracle's std::mt19937 objects know nothing of any period — the device stream must simply go on
    (row 0 after the wrap = a regeneration, like row 0 of any block), and no error bit is raised."""
    from gym_amd import _lib as L
    n, seed, period, chunk = 130, 21, 1_048_320, 1 << 16
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 1)
    ov.seed(seed)
    eng = _engine(n, L.RESET_MT19937, 1)
    eng.seed(seed)
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    done_steps, target = 0, period - 4                          # the reset consumed row 0: cursor = 1 + steps
    zeros = torch.zeros((chunk, n), dtype=torch.uint8, device="cuda")
    while done_steps < target:
        k = min(chunk, target - done_steps)
        eng.rollout(zeros[:k].contiguous())
        r = ov.rollout(np.zeros((k, n), np.uint8), rec_reward=False, rec_done=False, rec_trunc=False)
        assert np.array_equal(eng.obs.cpu().numpy(), r["final_obs"]), done_steps
        done_steps += k
    cur = eng.counters()["cursor"]
    assert int(cur.min()) == int(cur.max()) == period - 3
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    seen = []
    for t in range(160):                                        # across the wrap and through the whole first block after it
        eng.step(a)
        r = ov.step(np.zeros(n, np.int32), want64=True)
        assert np.array_equal(eng.obs.cpu().numpy(), r["obs"]) and np.array_equal(eng.state_f64().cpu().numpy(), r["obs64"]), t
        seen.append(int(eng.counters()["cursor"][0]))
    assert seen[:5] == [period - 2, period - 1, 0, 1, 2] and seen[-1] == 157
    assert eng.poll_error() == 0
