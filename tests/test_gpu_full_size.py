"""GPU: BASELINE configs[4] at its full size — 16,777,216 envs per GPU — through properties that do not need a CPU replay
of 16.7 M envs: (1) a shard of the env range, run as its own small engine, reproduces the big engine's rows bit for bit
(results do not depend on how many envs / GPUs share the launch) and that small engine is replayed by the oracle;
(2) conservation: every env-step is accounted for either in a running episode or in a finished one, and the cumulative
episode statistics equal the sums of the per-step records; (3) determinism: two engines, same seed, same bits."""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT

pytestmark = pytest.mark.gpu

N_FULL = 16 * 1024 * 1024


def _actions(n, t, n_act, offset=0):
    """Per-step device actions as a pure function of the GLOBAL env index, so a shard can regenerate its own."""
    i = torch.arange(offset, offset + n, dtype=torch.int64, device="cuda")
    return (((i * 2654435761 + t * 40503) >> 7) % n_act).to(torch.int32)


def test_cartpole_lockstep_16m_envs_shard_equality_conservation_oracle(oracle):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    T, seed = 48, 17
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR
    mk = lambda n, off: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                                            wrappers=w, max_episode_steps=500, env_offset=off, seed=seed), 0)
    big = mk(N_FULL, 0)
    big.generate_reset_tape(T // 8 + 2)
    big.reset()
    shards = [(0, 4096), (N_FULL // 2 - 1000, 8192), (N_FULL - 4096, 4096)]       # first, middle (unaligned), last
    small = []
    for off, n in shards:
        e = mk(n, off)
        e.generate_reset_tape(T // 8 + 2)
        e.reset()
        assert torch.equal(e.obs, big.obs[off:off + n])
        small.append(e)
    # the oracle replays the first shard from that shard's own tape
    ov = oracle.vec(CARTPOLE, shards[0][1], F_TIME_LIMIT | F_MONITOR, 500)
    ov.set_reset_tape(small[0].download_reset_tape(T // 8 + 2))
    assert np.array_equal(ov.reset(), small[0].obs.cpu().numpy())

    done_total = 0.0
    len_total = 0.0
    for t in range(T):
        big.step(_actions(N_FULL, t, 2))
        length, _, ret = big.episode_info()
        d = big.done != 0
        done_total += float(d.sum())
        len_total += float(length[d].sum(dtype=torch.float64))
        assert torch.equal(ret[d], length[d].to(torch.float32))                    # CartPole: return == length (reward 1 per step)
        for (off, n), e in zip(shards, small):
            e.step(_actions(n, t, 2, off))
            assert torch.equal(e.obs, big.obs[off:off + n]) and torch.equal(e.done, big.done[off:off + n]), (t, off)
            assert torch.equal(e.state_f64(), big.state_f64()[off:off + n]) if t % 16 == 15 else True
        r = ov.step(_actions(shards[0][1], t, 2).cpu().numpy())
        assert np.array_equal(small[0].done.cpu().numpy(), r["done"]), t
        assert np.allclose(small[0].obs.cpu().numpy(), r["obs"], rtol=1e-5, atol=1e-7), t
    assert big.poll_error() == 0
    tot = big.episode_stats()
    assert tot[0] == done_total and tot[2] == len_total and tot[1] == len_total     # episodes, sum of returns, sum of lengths
    running = float(big.counters()["elapsed"].sum(dtype=torch.float64))
    assert running + len_total == float(N_FULL) * T                                 # every env-step is in exactly one episode
    assert done_total > N_FULL                                                      # on average every env finished an episode


def test_cartpole_mt19937_16m_envs_shard_equality_and_oracle_across_a_regeneration(oracle):
    """The reference-exact reset source at the full size: 16,777,216 device-resident std::mt19937 states (39 GiB), TimeLimit(9) so
    that every env finishes an episode at least every 9 steps and all of them regenerate their 624-word block inside the run (a
    regeneration per 156 episodes: the wave-cooperative in-place twist at scale).  Shards of the env range run as their own small
    engines reproduce the big engine's rows bit for bit, and the oracle — per-env std::mt19937(seed + global index) objects on the
    CPU — replays each shard: observations, done flags and the fp64 state."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    T, seed, tl = 1500, 11, 9
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR

    def mk(n, off):
        e = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_MT19937,
                                wrappers=w, max_episode_steps=tl, env_offset=off), 0)
        e.seed(seed)
        e.reset()
        return e

    big = mk(N_FULL, 0)
    shards = [(0, 2048), (N_FULL // 2 - 1000, 3000), (N_FULL - 2048, 2048)]          # first, middle (unaligned: partial waves), last
    small, ovs = [], []
    for off, n in shards:
        e = mk(n, off)
        assert torch.equal(e.obs, big.obs[off:off + n])
        ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, tl)
        ov.seed(seed + off)                                                          # env i of the shard <- mt19937(seed + off + i)
        assert np.array_equal(ov.reset(), e.obs.cpu().numpy())
        small.append(e); ovs.append(ov)
    for t in range(T):
        big.step(_actions(N_FULL, t, 2))
        check = t % 100 == 99 or t == T - 1
        for (off, n), e, ov in zip(shards, small, ovs):
            a = _actions(n, t, 2, off)
            e.step(a)
            r = ov.step(a.cpu().numpy(), want64=check)
            if check:
                assert torch.equal(e.obs, big.obs[off:off + n]) and torch.equal(e.done, big.done[off:off + n]) and \
                    torch.equal(e.counter, big.counter[off:off + n]), (t, off)
                assert np.array_equal(e.obs.cpu().numpy(), r["obs"]) and np.array_equal(e.done.cpu().numpy(), r["done"]), (t, off)
                assert np.array_equal(e.state_f64().cpu().numpy(), r["obs64"]), (t, off)
    cur = big.counters()["cursor"]
    assert int(cur.min()) > 156, int(cur.min())                                      # every one of the 16.7 M envs regenerated after its first twist
    for (off, n), e in zip(shards, small):
        assert torch.equal(e.state_f64(), big.state_f64()[off:off + n]), off
        assert torch.equal(e.mt_state, big.mt_state[off:off + n]), off               # the generator blocks themselves
    assert big.poll_error() == 0


def test_mountaincar_normalised_16m_envs_determinism_and_float32_count():
    """config 3's stack at config 5's size: two engines agree bit for bit, and `m_Count` follows the reference's float32
    accumulation (the 1e-4 initial count is absorbed by the first 2^24-row batch)."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    mk = lambda: Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=N_FULL, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                                     wrappers=w, max_episode_steps=200), 0)
    a, b = mk(), mk()
    a.reset(); b.reset()
    count = np.float32(1e-4) + np.float32(N_FULL)
    assert a.get_norm_stats()[4] == count == np.float32(2 ** 24)
    for t in range(6):
        act = _actions(N_FULL, t, 3)
        a.step(act); b.step(act)
        count = np.float32(count + np.float32(N_FULL))
        assert a.get_norm_stats()[4] == count
    assert torch.equal(a.obs_norm, b.obs_norm) and torch.equal(a.reward, b.reward) and np.array_equal(a.get_norm_stats(), b.get_norm_stats())
    st = a.get_norm_stats()
    assert st[4] == count                                                           # obs count: float32 accumulation, as in the reference
    assert -0.6 < st[0] < -0.4 and abs(st[1]) < 0.01 and (st[2:4] > 0).all()        # mean position near the start, tiny mean velocity
    assert torch.isfinite(a.obs_norm).all() and float(a.obs_norm.abs().max()) <= 10.0   # clip_obs


def test_cartpole_lockstep_1m_envs_2000_steps_done_stream_exact(oracle):
    """A longer soak than config 2: 1,048,576 envs x 2,000 steps = 2.1e9 env-steps through the rollout kernel, every
    done / truncated flag compared with the oracle (bit-exact), observations at every chunk end and the final fp64 state (bit-identical)."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n, T, chunk, seed = 1 << 20, 2000, 500, 3
    gen = torch.Generator(device="cuda").manual_seed(99)
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(seed)
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                              wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500), 0)
    eng.upload_reset_tape(oracle.cartpole_reset_tape(seed, n, T // 8 + 8))
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    rec_done = torch.empty((chunk, n), dtype=torch.uint8, device="cuda")
    rec_trunc = torch.empty((chunk, n), dtype=torch.uint8, device="cuda")
    episodes = 0
    for c in range(T // chunk):
        acts_dev = torch.randint(0, 2, (chunk, n), dtype=torch.uint8, device="cuda", generator=gen)
        eng.rollout(acts_dev, rec_done, None, rec_trunc, None)
        r = ov.rollout(acts_dev.cpu().numpy(), rec_reward=False)
        assert np.array_equal(rec_done.cpu().numpy(), r["done"]), f"done stream diverged in chunk {c}"
        assert np.array_equal(rec_trunc.cpu().numpy(), r["truncated"]), f"truncated stream diverged in chunk {c}"
        assert np.array_equal(eng.obs.cpu().numpy(), r["final_obs"]), c
        episodes += int(r["done"].sum())
    st = eng.state_f64().cpu().numpy()
    assert np.array_equal(st, r["final_obs64"]), f"{int((st != r['final_obs64']).sum())} of {st.size} fp64 state values differ"
    assert eng.poll_error() == 0 and eng.episode_stats()[0] == episodes


def test_cartpole_268m_envs_index_arithmetic_beyond_4_gib():
    """Maximum sizes: 2^28 + 4099 envs on one GPU — the observation and residual arrays are > 4 GiB each, so env rows sit
    beyond byte offsets 2^31 and 2^32 — with shard equality at the first rows, across both byte boundaries and at the ragged
    tail, plus the conservation identities.  Counter-RNG resets under TimeLimit(13), so every env resets several times."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n_big, T, seed, tl = (1 << 28) + 4099, 30, 5, 13
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR
    mk = lambda n, off: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                                            wrappers=w, max_episode_steps=tl, env_offset=off, seed=seed), 0)
    big = mk(n_big, 0)
    big.reset()
    assert big.obs.numel() * 4 > (1 << 32)
    shards = [(0, 4096), ((1 << 27) - 2048, 4096), ((1 << 28) - 2048, 4096), (n_big - 4099, 4099)]
    small = []
    for off, n in shards:
        e = mk(n, off)
        e.reset()
        assert torch.equal(e.obs, big.obs[off:off + n]), off
        small.append(e)
    done_total = len_total = 0.0
    for t in range(T):
        a = _actions(n_big, t, 2)
        big.step(a)
        d = big.done != 0
        length, _, _ = big.episode_info()
        done_total += float(d.sum())
        len_total += float(length[d].sum(dtype=torch.float64))
        del d, length
        for (off, n), e in zip(shards, small):
            e.step(a[off:off + n].contiguous())
            assert torch.equal(e.obs, big.obs[off:off + n]) and torch.equal(e.done, big.done[off:off + n]) and \
                torch.equal(e.reward, big.reward[off:off + n]), (t, off)
        del a
    for (off, n), e in zip(shards, small):
        assert torch.equal(e.state_f64(), big.state_f64()[off:off + n]), off
    assert big.poll_error() == 0
    tot = big.episode_stats()
    assert tot[0] == done_total and tot[2] == len_total
    running = float(big.counters()["elapsed"].sum(dtype=torch.float64))
    assert running + len_total == float(n_big) * T
    assert done_total >= 2.0 * n_big                       # TimeLimit(13) over 30 steps: at least two finished episodes per env


@pytest.mark.parametrize("kind,precision,norm", [("cartpole", "lockstep", False), ("cartpole", "f32", False), ("mountaincar", "lockstep", True)])
def test_every_residency_plan_computes_the_same_bits(monkeypatch, kind, precision, norm):
    """The cache policy of an engine that streams non-temporally is a template argument of its step kernel (plan_residency: all
    4-byte lanes kept in the Infinity Cache / all but done / counter / none), i.e. four instantiations per kernel; the plan
    changes speed only.  12.6 M envs (past the non-temporal threshold for every variant here), 40 steps, each plan forced through
    GYMHIP_KEEP_SMALL: observations, rewards, dones, counter words, the exact state and the running statistics are bit-identical."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n, T = 12 * 1024 * 1024, 40
    cart = kind == "cartpole"
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | ((L.WRAP_CLIP_REWARD | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING) if norm else 0)
    acts = [_actions(n, t, 2 if cart else 3) for t in range(4)]
    want = None
    for plan in ("0", "1", "11", "15", None):
        if plan is None:
            monkeypatch.delenv("GYMHIP_KEEP_SMALL", raising=False)     # the planner's own choice
        else:
            monkeypatch.setenv("GYMHIP_KEEP_SMALL", plan)
        eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n,
                                  precision=L.PRECISION_LOCKSTEP_F64 if precision == "lockstep" else L.PRECISION_F32, reset_mode=L.RESET_COUNTER_RNG,
                                  wrappers=w, max_episode_steps=9 if cart else 13, seed=3, clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8), 0)
        eng.reset()
        for t in range(T):
            eng.step(acts[t % 4])
        got = [eng.obs.clone(), eng.reward.clone(), eng.done.clone(), eng.counter.clone()]
        if precision == "lockstep":
            got.append(eng.state_f64().clone())
        if norm:
            got += [eng.obs_norm.clone(), torch.from_numpy(eng.get_norm_stats())]
        assert eng.poll_error() == 0
        got.append(torch.from_numpy(np.asarray(eng.episode_stats(), np.float64)))
        del eng
        if want is None:
            want = got
            assert float(want[-1][0]) >= 3 * n         # every env finished at least 3 episodes (TimeLimit 9 / 13): the done path is in play
        else:
            for a, b in zip(got, want):
                assert torch.equal(a.cpu().view(torch.uint8) if a.dtype.is_floating_point else a.cpu(), b.cpu().view(torch.uint8) if b.dtype.is_floating_point else b.cpu()), plan
        torch.cuda.empty_cache()


def test_launch_shape_and_cache_policy_follow_the_engine_size():
    """gymhip_launch_info: what gymhip_create chose, against DESIGN.md §4 — small engines spread over 256 workgroups, plain accesses
    while the step's footprint fits the Infinity Cache, beyond it a residency plan (all 4-byte lanes while they fit 240 MiB, tape
    engines all but done, then fewer), and the headline kernel's 2-envs-per-thread shape once it streams."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig

    def info(n, kind=L.ENV_CARTPOLE, precision=L.PRECISION_LOCKSTEP_F64, reset=L.RESET_COUNTER_RNG, norm=False):
        w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | ((L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING) if norm else 0)
        e = Engine(EngineConfig(env_kind=kind, n_envs=n, precision=precision, reset_mode=reset, wrappers=w, max_episode_steps=200, seed=0,
                                clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8), 0)
        i = e.launch_info()
        e.close()
        torch.cuda.empty_cache()
        return i

    assert info(65536) == {"grid": 256, "lanes": 256, "streams_non_temporally": False, "kept_lanes": 0}                       # configs[1]
    assert info(1 << 17, L.ENV_MOUNTAINCAR, norm=True) == {"grid": 256, "lanes": 512, "streams_non_temporally": False, "kept_lanes": 0}   # configs[2]'s shard
    assert info(4 << 20)["streams_non_temporally"] is False
    assert info(6 << 20) == {"grid": 6144, "lanes": 512, "streams_non_temporally": True, "kept_lanes": 15}
    assert info(N_FULL) == {"grid": 16384, "lanes": 512, "streams_non_temporally": True, "kept_lanes": 15}                     # the headline engine
    assert info(N_FULL, reset=L.RESET_TAPE) == {"grid": 2048, "lanes": 512, "streams_non_temporally": True, "kept_lanes": 11}
    assert info(N_FULL, precision=L.PRECISION_F32) == {"grid": 512, "lanes": 512, "streams_non_temporally": True, "kept_lanes": 15}
    assert info(N_FULL, L.ENV_MOUNTAINCAR, norm=True) == {"grid": 512, "lanes": 512, "streams_non_temporally": True, "kept_lanes": 11}   # four lanes = 256 MiB: done streams
    assert info(32 << 20)["kept_lanes"] == 1 and info(96 << 20, L.ENV_MOUNTAINCAR, precision=L.PRECISION_F32)["kept_lanes"] == 0
