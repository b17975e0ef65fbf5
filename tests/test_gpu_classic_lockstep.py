"""GPU parity: the HIP lockstep path vs the CPU oracle on the same host-drawn reset tape and action tape.

Bar (BASELINE.json north_star): done / truncated / step-count / episode streams bit-exact; obs and reward within
1e-5 relative (they are in fact compared exactly where noted).  All calls go through the C ABI (gym_amd._lib).
"""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, splitmix_actions

pytestmark = pytest.mark.gpu

REL = 1e-5


def assert_mountaincar_obs(got, want, where):
    """MountainCar's only non-reference arithmetic is the device library's double cos(3*pos) (mountain_car.cpp:19 calls
    glibc's), used WITHOUT a float narrowing, so the fp64 state may differ in its last bit (measured: <= 8e-16, 0 of 131,072
    final values after 1000 steps) and a float32 observation lands on the other side of a rounding tie once in ~1e8 values
    (measured: 1 of 131,072,000; tools/measure_exactness.py).  Bar: bit-identical except at most max(1, 1e-6 x size) values,
    and those one float ulp apart."""
    bad = got != want
    if bad.any():
        assert bad.sum() <= max(1, int(1e-6 * got.size)), f"{int(bad.sum())} observation values differ at {where}"
        assert (np.abs(got[bad].view(np.int32) - want[bad].view(np.int32)) <= 1).all(), where


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-2)   # abs floor 1e-2 as in SURVEY.md §7


def check_episode_info(eng, r, d):
    """info["episode"] / info["TimeLimit.truncated"] of the envs that just finished (decoded from the record words)."""
    ln, tr, ret = (x.cpu().numpy() for x in eng.episode_info())
    assert np.array_equal(tr[d], r["truncated"][d])
    assert np.array_equal(ln[d].astype(np.float32), r["ep_l"][d])
    assert np.array_equal(ret[d], r["ep_r"][d])
    assert not ln[~d].any() and not tr[~d].any()
    el = eng.counters()["elapsed"].cpu().numpy()
    assert (el[~d] > 0).all() and (el[d] == 0).all()
    w = eng.counter.cpu().numpy()
    assert ((w < 0) == d).all()          # bit 31 of the packed word <=> the last step finished an episode


def make_engine(kind, n, precision, wrappers, max_steps, reset_mode=None, seed=0, env_offset=0):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    cfg = EngineConfig(env_kind=kind, n_envs=n, precision=precision, reset_mode=L.RESET_TAPE if reset_mode is None else reset_mode,
                       wrappers=wrappers, max_episode_steps=max_steps, seed=seed, env_offset=env_offset)
    return Engine(cfg, 0)


def test_cartpole_step_api_lockstep_every_step(oracle):
    """N=2048 x T=1500 through gymhip_step: every output of every step compared."""
    from gym_amd import _lib as L
    n, T, seed = 2048, 1500, 0
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(seed)
    tape = oracle.cartpole_reset_tape(seed, n, T // 8 + 4)
    eng = make_engine(L.ENV_CARTPOLE, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, 500)
    eng.upload_reset_tape(tape)
    eng.reset()
    obs0, obs0_64 = ov.reset(want64=True)
    assert np.array_equal(eng.obs.cpu().numpy(), obs0)
    acts = splitmix_actions(1234, T, n, 2)
    acts_dev = torch.from_numpy(acts.astype(np.int32)).cuda()
    n_done = 0
    for t in range(T):
        eng.step(acts_dev[t])
        r = ov.step(acts[t], want64=True)
        g_obs = eng.obs.cpu().numpy()
        g_done = eng.done.cpu().numpy()
        assert np.array_equal(g_done, r["done"]), f"done stream diverged at step {t}"
        assert np.array_equal(eng.reward.cpu().numpy(), r["reward"])
        assert np.array_equal(g_obs, r["obs"]), f"obs diverged at step {t}: {int((g_obs != r['obs']).sum())} values"
        d = r["done"] != 0
        if d.any():
            n_done += int(d.sum())
            check_episode_info(eng, r, d)
        if t % 250 == 0 or t == T - 1:
            # the fp64 state itself is BIT-IDENTICAL to the oracle's: the only arithmetic that is not the reference's own
            # op sequence is sin/cos, whose float-narrowed values equal glibc's except within ~1 double ulp of a float tie
            # (tests/test_sincos_parity.py: 0 mismatches in 3.2e10 evaluations), and the split-double store is lossless
            g64 = eng.state_f64().cpu().numpy()                 # [n][4], split-double halves joined on the device
            assert np.array_equal(g64, r["obs64"]), f"{int((g64 != r['obs64']).sum())} state values differ at step {t}"
            assert np.array_equal(g64.astype(np.float32), g_obs)   # obs IS the high half of the state
    assert n_done > n * T / 40            # episodes did end (mean length ~22)
    assert eng.poll_error() == 0
    tot = eng.episode_stats()
    assert tot[0] == n_done


def test_cartpole_config2_rollout_65536x10000(oracle):
    """BASELINE config 2: N=65,536, 10^4-step trajectories, fixed action tape, lockstep vs the CPU oracle.
    Streams (done, truncated, reward) are compared for EVERY step; obs for every step of the last chunk and at every
    chunk boundary."""
    from gym_amd import _lib as L
    n, T, chunk, seed = 65536, 10000, 500, 0
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(seed)
    eng = make_engine(L.ENV_CARTPOLE, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, 500)
    eng.upload_reset_tape(oracle.cartpole_reset_tape(seed, n, T // 8 + 8))
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    rec_done = torch.empty((chunk, n), dtype=torch.uint8, device="cuda")
    rec_trunc = torch.empty((chunk, n), dtype=torch.uint8, device="cuda")
    rec_rew = torch.empty((chunk, n), dtype=torch.float32, device="cuda")
    rec_obs = torch.empty((chunk, n, 4), dtype=torch.float32, device="cuda")
    episodes = 0
    for c in range(T // chunk):
        last = c == T // chunk - 1
        acts = splitmix_actions(1234 + c * chunk * n, chunk, n, 2)
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
    assert eng.poll_error() == 0
    assert eng.episode_stats()[0] == episodes


def test_mountaincar_lockstep(oracle):
    """MountainCar + TimeLimit(200) + ClipReward + Monitor, N=8192 x T=600 via gymhip_step (i64 and u8 actions)."""
    from gym_amd import _lib as L
    n, T = 8192, 600
    ov = oracle.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, 200)
    eng = make_engine(L.ENV_MOUNTAINCAR, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR, 200)
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    acts = splitmix_actions(77, T, n, 3)
    a64 = torch.from_numpy(acts.astype(np.int64)).cuda()
    a8 = torch.from_numpy(acts).cuda()
    for t in range(T):
        eng.step(a64[t] if t % 2 else a8[t])
        r = ov.step(acts[t])
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        assert np.array_equal(eng.reward.cpu().numpy(), r["reward"])
        assert_mountaincar_obs(eng.obs.cpu().numpy(), r["obs"], t)
        d = r["done"] != 0
        if d.any():
            check_episode_info(eng, r, d)
    assert eng.episode_stats()[0] == 3 * n   # every episode truncates at 200 under random actions


def test_mountaincar_goal_reached_directed_policy(oracle):
    """Force the rare branch: an energy-pumping policy reaches the goal (done without truncation) in both paths."""
    from gym_amd import _lib as L
    n, T = 256, 400
    ov = oracle.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_MONITOR, 200)
    eng = make_engine(L.ENV_MOUNTAINCAR, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, 200)
    eng.reset()
    obs = ov.reset()
    terminated = 0
    for t in range(T):
        a = np.where(obs[:, 1] >= 0, 2, 0).astype(np.int32)   # push along the velocity
        a[: n // 4] = 1                                     # a quarter of the envs idle (they truncate)
        eng.step(torch.from_numpy(a).cuda())
        r = ov.step(a)
        obs = r["obs"]
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        d = r["done"] != 0
        if d.any():
            check_episode_info(eng, r, d)
            terminated += int((d & (r["truncated"] == 0)).sum())
        assert_mountaincar_obs(eng.obs.cpu().numpy(), obs, t)
    assert terminated > 0


def test_cartpole_no_time_limit_uses_internal_500(oracle):
    """Without TimeLimit the env's own `timeStep == 500` test (cartpole.cpp:64, pre-increment) ends the episode on the
    501st step.  A balancing bang-bang policy keeps the pole up long enough to reach it."""
    from gym_amd import _lib as L
    n, T = 128, 1100
    ov = oracle.vec(CARTPOLE, n, F_MONITOR, 0)
    ov.seed(3)
    eng = make_engine(L.ENV_CARTPOLE, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_MONITOR, 0)
    eng.upload_reset_tape(oracle.cartpole_reset_tape(3, n, 64))
    eng.reset()
    obs = ov.reset()
    lens = []
    for t in range(T):
        a = (3.0 * obs[:, 2] + obs[:, 3] + 0.05 * obs[:, 0] + 0.1 * obs[:, 1] > 0).astype(np.int32)
        eng.step(torch.from_numpy(a).cuda())
        r = ov.step(a)
        obs = r["obs"]
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        d = r["done"] != 0
        if d.any():
            lens += list(r["ep_l"][d])
            check_episode_info(eng, r, d)
        assert np.array_equal(eng.obs.cpu().numpy(), obs), t
    assert 501.0 in lens


def test_engine_errors():
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    with pytest.raises(L.GymhipError):
        Engine(EngineConfig(n_envs=0), 0)
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=64), 0)
    with pytest.raises(L.GymhipError, match="no reset tape"):
        eng.reset()
    eng2 = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=64), 0)
    with pytest.raises(L.GymhipError, match="before reset"):
        eng2.step(torch.zeros(64, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        eng2.reset()
        eng2.step(torch.zeros(63, dtype=torch.int32, device="cuda"))


def test_reset_tape_exhaustion_sets_error_flag(oracle):
    from gym_amd import _lib as L
    n = 256
    eng = make_engine(L.ENV_CARTPOLE, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_TIME_LIMIT, 500)
    eng.upload_reset_tape(oracle.cartpole_reset_tape(0, n, 2))
    eng.reset()
    a = torch.ones(n, dtype=torch.int32, device="cuda")
    for _ in range(60):
        eng.step(a)
    assert eng.poll_error() & 1


def test_cartpole_lockstep_counter_rng_resets_replayed_by_oracle(oracle):
    """COUNTER_RNG resets (bench.py's headline reset source): the reset states come from an
    in-kernel counter hash, so the oracle cannot draw them — it REPLAYS them.  Every reset observation the engine
    produced (exact: a reset state is four floats widened to double) is collected, per env and in order of use, into
    a reset tape, and the oracle re-runs the whole trajectory from it.  done / truncated / episode records bit-exact,
    observations and the fp64 state at the end bit-identical."""
    from gym_amd import _lib as L
    n, T = 4096, 700
    eng = make_engine(L.ENV_CARTPOLE, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, 500, reset_mode=L.RESET_COUNTER_RNG, seed=5)
    eng.reset()
    obs0 = eng.obs.cpu().numpy().copy()
    assert (obs0 >= -0.05).all() and (obs0 < 0.05).all() and len(np.unique(obs0)) > 0.99 * obs0.size
    tape = np.zeros((T // 8 + 3, n, 4), np.float32)
    used = np.ones(n, np.int64)
    tape[0] = obs0
    acts = splitmix_actions(77, T, n, 2)
    acts_dev = torch.from_numpy(acts.astype(np.int32)).cuda()
    g_obs, g_done, g_info = [], [], []
    for t in range(T):
        eng.step(acts_dev[t])
        o = eng.obs.cpu().numpy().copy()
        d = eng.done.cpu().numpy() != 0
        idx = np.nonzero(d)[0]
        assert (o[idx] >= -0.05).all() and (o[idx] < 0.05).all()         # the rows of done envs ARE their reset states
        tape[used[idx], idx] = o[idx]
        used[idx] += 1
        g_obs.append(o); g_done.append(d)
        g_info.append(tuple(x.cpu().numpy().copy() for x in eng.episode_info()))
    g64 = eng.state_f64().cpu().numpy()
    assert eng.poll_error() == 0 and used.max() > 10
    # no two resets share a state: the key is (seed, global env index, global step)
    first_rows = tape[:2].reshape(-1, 4)
    assert len(np.unique(first_rows, axis=0)) == len(first_rows)

    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.set_reset_tape(tape[:int(used.max())])
    assert np.array_equal(ov.reset(), obs0)
    for t in range(T):
        r = ov.step(acts[t], want64=(t == T - 1))
        d = r["done"] != 0
        assert np.array_equal(g_done[t], d), f"done stream diverged at step {t}"
        assert np.array_equal(g_obs[t], r["obs"]), t
        ln, tr, ret = g_info[t]
        assert np.array_equal(tr[d], r["truncated"][d]) and np.array_equal(ln[d].astype(np.float32), r["ep_l"][d]) and np.array_equal(ret[d], r["ep_r"][d])
    assert np.array_equal(g64, r["obs64"])


@pytest.mark.parametrize("kind,precision,reset_mode", [("cartpole", "lockstep", "tape"), ("cartpole", "lockstep", "rng"), ("cartpole", "f32", "tape"),
                                                        ("cartpole", "f32", "rng"), ("mountaincar", "lockstep", "rng"), ("mountaincar", "f32", "rng")])
def test_rollout_of_k_steps_equals_k_step_calls(kind, precision, reset_mode):
    """gymhip_rollout(K) is K x gymhip_step with the state held in registers: same obs / state / counter words / episode
    totals, bit for bit, for both env kinds, both precisions and both reset sources, and its recorded streams equal the
    per-step outputs.  (The counter-RNG key of step k inside a rollout is the global step it would have had.)"""
    from gym_amd import _lib as L
    cart = kind == "cartpole"
    n, K = 5000, 130
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | (0 if cart else L.WRAP_CLIP_REWARD)
    mk = lambda: make_engine(L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n, L.PRECISION_LOCKSTEP_F64 if precision == "lockstep" else L.PRECISION_F32,
                             w, 40, reset_mode=L.RESET_TAPE if (cart and reset_mode == "tape") else L.RESET_COUNTER_RNG, seed=4)
    a, b = mk(), mk()
    for e in (a, b):
        if cart and reset_mode == "tape":
            e.generate_reset_tape(K // 8 + 4)
        e.reset()
    acts = torch.from_numpy(splitmix_actions(3, 2 * K, n, 2 if cart else 3)).cuda()
    F = 4 if cart else 2
    for half in range(2):                                  # two rollouts in a row: the second starts mid-episode
        tape = acts[half * K:(half + 1) * K].contiguous()
        rec_done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
        rec_trunc = torch.empty((K, n), dtype=torch.uint8, device="cuda")
        rec_rew = torch.empty((K, n), dtype=torch.float32, device="cuda")
        rec_obs = torch.empty((K, n, F), dtype=torch.float32, device="cuda")
        a.rollout(tape, rec_done, rec_rew, rec_trunc, rec_obs)
        for k in range(K):
            b.step(tape[k])
            assert torch.equal(rec_obs[k], b.obs) and torch.equal(rec_done[k].float(), b.done) and torch.equal(rec_rew[k], b.reward), (half, k)
            _, tr, _ = b.episode_info()
            assert torch.equal(rec_trunc[k], tr), (half, k)
        assert torch.equal(a.obs, b.obs) and torch.equal(a.counter, b.counter) and torch.equal(a.state_f64(), b.state_f64())
        assert torch.equal(a.done, b.done) and torch.equal(a.reward, b.reward)
    assert np.array_equal(a.episode_stats(), b.episode_stats()) and a.episode_stats()[0] > n
    assert a.poll_error() == 0 and b.poll_error() == 0


def test_counter_rng_episode_index_wraps_after_2_pow_20_episodes_and_says_so():
    """COUNTER_RNG keys an env's reset draw by a 20-bit episode index kept in its counter word (gymhip.h): after 2^20 episodes of one
    env the reset states repeat with that period.  Documented, and reported: the wrap sets bit 4 of the error word (information; no
    other bit).  TimeLimit(1) makes every step an episode; the rollout kernel takes the 2^20 steps in 16 launches."""
    from gym_amd import _lib as L
    n, K = 64, 1 << 16
    eng = make_engine(L.ENV_CARTPOLE, n, L.PRECISION_LOCKSTEP_F64, L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, 1, reset_mode=L.RESET_COUNTER_RNG, seed=3)
    eng.reset()
    head = torch.zeros((8, n), dtype=torch.uint8, device="cuda")
    rec_a = torch.empty((8, n, 4), dtype=torch.float32, device="cuda")
    eng.rollout(head, rec_obs=rec_a)                          # episode indices 1..8: with TimeLimit(1) every observation is a reset state
    assert eng.poll_error() == 0
    tape = torch.zeros((K, n), dtype=torch.uint8, device="cuda")
    for _ in range((1 << 20) // K - 1):
        eng.rollout(tape)
    eng.rollout(tape[:K - 8])
    assert eng.poll_error() == 16                             # the index wrapped to 0 on the way
    assert int(eng.counters()["cursor"].min()) == int(eng.counters()["cursor"].max()) == 1
    rec_b = torch.empty_like(rec_a)
    eng.rollout(head, rec_obs=rec_b)
    assert torch.equal(rec_a, rec_b)                          # the same reset states again: period 2^20 episodes per env
    assert eng.episode_stats()[0] == n * ((1 << 20) + 8)
