"""GPU parity of the fused VecNormalize path (BASELINE config 3) against the CPU oracle.

What is compared, per step (SURVEY.md §7 "Normaliser ordering"):
  (i)   raw obs / reward / done streams: exact (same bar as the lockstep tests);
  (ii)  running (mean, var, count) of the obs and return statistics: <= 1e-5 relative;
  (iii) normalised obs / reward: <= 1e-5 scaled by the conditioning of (x - mean)/sqrt(var + eps), i.e. absolute
        tolerance 1e-5 * max(1, |mean| / sqrt(var + eps)) — while every MountainCar env still sits at the identical
        start state the batch variance is ~0 and the division amplifies the last-bit difference between two float32
        batch-mean reduction orders by up to 1e4 (measured in the survey against the compiled reference).
"""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, splitmix_actions

pytestmark = pytest.mark.gpu


def _engine(kind, n, wrappers, max_steps, env_offset=0, precision=None):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    return Engine(EngineConfig(env_kind=kind, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64 if precision is None else precision,
                               reset_mode=L.RESET_TAPE, wrappers=wrappers, max_episode_steps=max_steps, env_offset=env_offset), 0)


def _norm_flags():
    from gym_amd import _lib as L
    return L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING


def _check_stats(g, o, F, where):
    # g, o: [mean F, var F, count, ret mean, ret var, ret count]
    err = np.abs(g - o) / np.maximum(np.abs(o), 1e-6)
    assert err.max() <= 1e-5, f"running stats diverged at {where}: {err}\n{g}\n{o}"


def _check_normalised(g_obs, o_obs, g_rew, o_rew, stats, F, eps, where):
    mean, var = stats[:F], stats[F:2 * F]
    cond = np.maximum(1.0, np.abs(mean) / np.sqrt(var + eps))
    assert (np.abs(g_obs - o_obs) <= 1e-5 * cond * np.maximum(1.0, np.abs(o_obs))).all(), f"normalised obs diverged at {where}"
    assert (np.abs(g_rew - o_rew) <= 1e-5 * np.maximum(1.0, np.abs(o_rew))).all(), f"normalised reward diverged at {where}"


@pytest.mark.parametrize("n,T", [(4096, 450), (1 << 20, 60)])
def test_mountaincar_vecnormalize_config3(oracle, n, T):
    """MountainCar + TimeLimit(200) + ClipReward + Monitor + TrainingVecNormRO (config 3; N = 1,048,576 is its full size)."""
    from gym_amd import _lib as L
    ov = oracle.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, 200)
    ov.attach_norm()
    eng = _engine(L.ENV_MOUNTAINCAR, n, L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | _norm_flags(), 200)
    eng.reset()
    o_obs, o_raw = ov.norm_reset()
    assert np.array_equal(eng.obs.cpu().numpy(), o_raw)
    _check_stats(eng.get_norm_stats(), ov.norm_stats(), 2, "reset")
    _check_normalised(eng.obs_norm.cpu().numpy(), o_obs, np.zeros(n, np.float32), np.zeros(n, np.float32), ov.norm_stats(), 2, 1e-8, "reset")
    acts = splitmix_actions(5, T, n, 3)
    dev = torch.from_numpy(acts).cuda()
    for t in range(T):
        eng.step(dev[t])
        r = ov.norm_step(acts[t])
        assert np.array_equal(eng.obs.cpu().numpy(), r["raw_obs"]), t
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        st = ov.norm_stats()
        _check_stats(eng.get_norm_stats(), st, 2, f"step {t}")
        _check_normalised(eng.obs_norm.cpu().numpy(), r["obs"], eng.reward.cpu().numpy(), r["reward"], st, 2, 1e-8, f"step {t}")
        ret_err = np.abs(eng.returns.cpu().numpy() - ov.norm_returns())
        assert ret_err.max() <= 1e-5 * max(1.0, np.abs(ov.norm_returns()).max())
    assert np.array_equal(eng.norm_stats.cpu().numpy(), eng.get_norm_stats())   # public copy == current slot


def test_cartpole_vecnormalize(oracle):
    """CartPole + Monitor + TrainingVecNormRO (the stack of test/classic_control.cpp:105-107), N=8192 x 400 steps."""
    from gym_amd import _lib as L
    n, T = 8192, 400
    ov = oracle.vec(CARTPOLE, n, F_MONITOR, 0)
    ov.seed(11)
    ov.attach_norm()
    eng = _engine(L.ENV_CARTPOLE, n, L.WRAP_MONITOR | _norm_flags(), 0)
    eng.upload_reset_tape(oracle.cartpole_reset_tape(11, n, T // 8 + 4))
    eng.reset()
    o_obs, o_raw = ov.norm_reset()
    assert np.array_equal(eng.obs.cpu().numpy(), o_raw)
    _check_stats(eng.get_norm_stats(), ov.norm_stats(), 4, "reset")
    acts = splitmix_actions(6, T, n, 2)
    dev = torch.from_numpy(acts).cuda()
    for t in range(T):
        eng.step(dev[t])
        r = ov.norm_step(acts[t])
        assert np.array_equal(eng.obs.cpu().numpy(), r["raw_obs"]), t
        assert np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        st = ov.norm_stats()
        _check_stats(eng.get_norm_stats(), st, 4, f"step {t}")
        _check_normalised(eng.obs_norm.cpu().numpy(), r["obs"], eng.reward.cpu().numpy(), r["reward"], st, 4, 1e-8, f"step {t}")
    assert eng.poll_error() == 0


def test_first_normalised_rewards_match_reference_probe(oracle):
    """SURVEY.md §8a A10 [probed on the compiled reference]: first CartPole step => normalised reward 10 (clipped),
    second => 2.02."""
    from gym_amd import _lib as L
    n = 4
    eng = _engine(L.ENV_CARTPOLE, n, L.WRAP_MONITOR | _norm_flags(), 0)
    eng.upload_reset_tape(oracle.cartpole_reset_tape(0, n, 8))
    eng.reset()
    a = torch.ones(n, dtype=torch.int32, device="cuda")
    eng.step(a)
    assert np.array_equal(eng.reward.cpu().numpy(), np.full(n, 10.0, np.float32))
    eng.step(a)
    assert np.allclose(eng.reward.cpu().numpy(), 2.02, atol=5e-3)


@pytest.mark.parametrize("form,n", [("moments", 16384), ("rows", 16384), ("rows", 600_000), ("rows", 3000)])
def test_sharded_moments_allreduce_equals_single_engine(oracle, form, n):
    """Two engines holding the two halves of the env range, with the `moments` payloads summed between
    step_begin and step_finish (what the RCCL all-reduce does across ranks), must track the single-engine run:
    raw streams exactly, running statistics to 1e-6, normalised outputs to 1e-5 (conditioning-scaled).
    "rows": the exchange form without the reduce launch (gymhip_set_exchange_form) — the payload is the partial rows themselves
    (row 0 carries each rank's env count), padded to 256 or 512 rows whatever the engine's grid is; an Updateable::update pass in
    between (which borrows the same rows) must not leak into the next step's payload."""
    from gym_amd import _lib as L
    T = 260 if n <= 16384 else 40
    w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | _norm_flags()
    one = _engine(L.ENV_MOUNTAINCAR, n, w, 200)
    halves = [_engine(L.ENV_MOUNTAINCAR, n // 2, w, 200, env_offset=k * n // 2) for k in range(2)]
    if form == "rows":
        for h in halves:
            h.set_exchange_form(L.EXCHANGE_ROWS)
            assert h.moments.numel() in (256 * 16, 512 * 16)
        assert halves[0].moments.numel() == halves[1].moments.numel()

    def allreduce():
        tot = halves[0].moments + halves[1].moments
        for h in halves:
            h.moments.copy_(tot)

    one.reset()
    for h in halves:
        h.reset_begin()
    allreduce()
    for h in halves:
        h.reset_finish()
    acts = torch.from_numpy(splitmix_actions(8, T, n, 3)).cuda()
    for t in range(T):
        one.step(acts[t])
        for k, h in enumerate(halves):
            h.step_begin(acts[t, k * n // 2:(k + 1) * n // 2])
        allreduce()
        assert float(halves[0].moments[0 if form == "moments" else 15]) == n          # global batch count
        for h in halves:
            h.step_finish()
        if form == "rows" and t == 20:
            # Updateable::update borrows the partial rows (more of them than the step kernel owns at small n)
            g_ = torch.Generator(device="cuda").manual_seed(0)
            rows = torch.randn((300_001, 2), device="cuda", generator=g_) * 0.1 - 0.4
            for e in [one] + halves:
                e.update_norm_stats(rows, None)
        s1 = one.get_norm_stats()
        for k, h in enumerate(halves):
            sl = slice(k * n // 2, (k + 1) * n // 2)
            assert torch.equal(h.obs, one.obs[sl]) and torch.equal(h.done, one.done[sl])
            s2 = h.get_norm_stats()
            assert (np.abs(s1 - s2) <= 1e-6 * np.maximum(np.abs(s1), 1e-6)).all(), (t, s1, s2)
            cond = np.maximum(1.0, np.abs(s1[:2]) / np.sqrt(s1[2:4] + 1e-8))
            d = (h.obs_norm - one.obs_norm[sl]).abs().cpu().numpy()
            assert (d <= 1e-5 * cond * np.maximum(1.0, one.obs_norm[sl].abs().cpu().numpy())).all(), t
            assert torch.allclose(h.reward, one.reward[sl], rtol=1e-5, atol=1e-6)
    assert np.array_equal(halves[0].get_norm_stats(), halves[1].get_norm_stats())   # replicated statistics stay identical


def test_norm_stats_save_load_roundtrip():
    from gym_amd import _lib as L
    n = 1024
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | _norm_flags()
    a = _engine(L.ENV_MOUNTAINCAR, n, w, 200)
    a.reset()
    acts = torch.randint(0, 3, (50, n), dtype=torch.int32, device="cuda")
    for t in range(25):
        a.step(acts[t])
    saved = a.get_norm_stats()
    b = _engine(L.ENV_MOUNTAINCAR, n, w & ~L.WRAP_NORM_TRAINING, 200)   # evaluation copy: isTraining=false
    b.set_norm_stats(saved)
    b.reset()
    for t in range(25):
        b.step(acts[t])
    assert np.array_equal(b.get_norm_stats(), saved)                    # frozen statistics
    mean, var = saved[:2], saved[2:4]
    want = np.clip((b.obs.cpu().numpy() - mean) / np.sqrt(var + np.float32(1e-8)), -10, 10)
    assert np.allclose(b.obs_norm.cpu().numpy(), want, rtol=1e-6, atol=1e-6)


def test_reset_index_resets_one_env_without_touching_the_statistics(oracle):
    """VecEnv::reset(int index) under VecNormalize (sync_env.h:128-139, vec_normalize.h:179-189): env `index` takes its next
    reset-tape row, its return lane is zeroed, obs_norm[index] uses the CURRENT statistics, nothing else changes."""
    from gym_amd import _lib as L
    n, seed = 512, 3
    eng = _engine(L.ENV_CARTPOLE, n, L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | _norm_flags(), 500)
    tape = oracle.cartpole_reset_tape(seed, n, 16)
    eng.upload_reset_tape(tape)
    eng.reset()
    acts = torch.from_numpy(splitmix_actions(9, 30, n, 2)).cuda()
    for t in range(30):
        eng.step(acts[t])
    before = {k: getattr(eng, k).clone() for k in ("obs", "obs_norm", "returns", "counter", "reward", "done")}
    state_before = eng.state_f64().clone()
    stats = eng.get_norm_stats()
    cur = eng.counters()
    for index in (0, 77, n - 1):
        cursor = int(cur["cursor"][index])
        eng.reset_index(index)
        row = tape[cursor, index]
        assert np.array_equal(eng.obs[index].cpu().numpy(), row)
        assert np.array_equal(eng.state_f64()[index].cpu().numpy(), row.astype(np.float64))       # floats widened (cartpole.cpp:23-26)
        want = np.clip((row - stats[:4]) / np.sqrt(stats[4:8] + np.float32(1e-8)), -10, 10).astype(np.float32)
        assert np.allclose(eng.obs_norm[index].cpu().numpy(), want, rtol=1e-6, atol=1e-7)
        assert float(eng.returns[index]) == 0.0
        after = eng.counters()
        assert int(after["elapsed"][index]) == 0 and int(after["cursor"][index]) == cursor + 1
    assert np.array_equal(eng.get_norm_stats(), stats)                                              # no RMS update
    keep = np.ones(n, bool); keep[[0, 77, n - 1]] = False
    for k in ("obs", "obs_norm", "returns", "counter"):
        assert torch.equal(getattr(eng, k)[torch.from_numpy(keep).cuda()], before[k][torch.from_numpy(keep).cuda()]), k
    assert torch.equal(eng.reward, before["reward"]) and torch.equal(eng.done, before["done"])
    assert torch.equal(eng.state_f64()[torch.from_numpy(keep).cuda()], state_before[torch.from_numpy(keep).cuda()])
    with pytest.raises(Exception):
        eng.reset_index(n)
    eng.step(acts[0])                                                                               # the engine keeps stepping
    assert eng.poll_error() == 0


def test_update_pushes_rollout_rows_into_the_statistics(oracle):
    """Updateable::update (vec_normalize.h:191-206) == RunningMeanStd::update on the flattened rows
    (running_mean_std.h:42-66), checked against the oracle's RunningMeanStd (itself pinned to the reference's)."""
    from gym_amd import _lib as L
    n, F = 256, 2
    eng = _engine(L.ENV_MOUNTAINCAR, n, L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | _norm_flags(), 200)
    eng.reset()
    st = eng.get_norm_stats()
    obs_st = np.concatenate([st[:F], st[F:2 * F], st[2 * F:2 * F + 1]]).astype(np.float32)
    ret_st = st[2 * F + 1:2 * F + 4].astype(np.float32).copy()
    rng = np.random.default_rng(4)
    for k in range(5):
        T = int(rng.integers(1, 40))
        state = (rng.standard_normal((T, n, F)) * [0.3, 0.02] + [-0.5, 0.0]).astype(np.float32)
        reward = rng.standard_normal((T, n)).astype(np.float32) * 3
        eng.update_norm_stats(torch.from_numpy(state).cuda(), torch.from_numpy(reward).cuda() if k % 2 == 0 else None)
        bm, bv = oracle.rms_batch_moments(state.reshape(-1, F))
        oracle.rms_update_from_moments(obs_st, bm, bv, T * n)
        if k % 2 == 0:
            bm, bv = oracle.rms_batch_moments(reward.reshape(-1, 1))
            oracle.rms_update_from_moments(ret_st, bm, bv, T * n)
        got = eng.get_norm_stats()
        _check_stats(got, np.concatenate([obs_st, ret_st]), F, f"update {k}")
        assert got[2 * F] == obs_st[2 * F] and got[2 * F + 3] == ret_st[2]                          # counts are exact
    eng.step(torch.zeros(n, dtype=torch.int32, device="cuda"))                                      # and the engine steps on
    assert np.array_equal(eng.norm_stats.cpu().numpy(), eng.get_norm_stats())


def test_learner_updater_reaches_vecnormalize_update(oracle):
    """A learner's Updater (updateable.h:16-31) subscribed to VecNormalize pushes [T, N, F] rollout tensors from the HOST
    (vec_normalize.h:197 takes .cpu() tensors) into the fused statistics: same numbers as RunningMeanStd::update."""
    from gym_amd import vec_env as V
    n, F, T = 128, 4, 9
    env = V.TrainingVecNormRO(V.HipVecEnv("CartPole-v1", n, max_episode_steps=500))
    env.seed(3)
    env.reset()
    st = env.venv.engine.get_norm_stats()
    obs_st = np.concatenate([st[:F], st[F:2 * F], st[2 * F:2 * F + 1]]).astype(np.float32)
    ret_st = st[2 * F + 1:2 * F + 4].astype(np.float32).copy()
    rng = np.random.default_rng(8)
    state = rng.standard_normal((T, n, F)).astype(np.float32)
    reward = rng.standard_normal((T, n)).astype(np.float32)
    learner = V.Updater()
    learner.subscribe(env)
    learner.notifyOne(torch.from_numpy(state), torch.from_numpy(reward))
    bm, bv = oracle.rms_batch_moments(state.reshape(-1, F)); oracle.rms_update_from_moments(obs_st, bm, bv, T * n)
    bm, bv = oracle.rms_batch_moments(reward.reshape(-1, 1)); oracle.rms_update_from_moments(ret_st, bm, bv, T * n)
    _check_stats(env.venv.engine.get_norm_stats(), np.concatenate([obs_st, ret_st]), F, "updater")


@pytest.mark.parametrize("training,norm_reward,norm_obs", [(True, True, False), (True, False, True), (False, True, True),
                                                            (False, True, False), (False, False, True)])
@pytest.mark.parametrize("kind", ["cartpole", "mountaincar"])
def test_vecnormalize_template_flag_combinations(oracle, kind, training, norm_reward, norm_obs):
    """VecNormalize<dict, isTraining, normReward, normObs> (vec_normalize.h:20, 145-177): each `if constexpr` arm on its
    own — statistics only move for the arms that are both trained and normalised, the other outputs pass through — with
    non-default clip / gamma / epsilon and statistics loaded from a checkpoint for the frozen cases."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    cart = kind == "cartpole"
    n, T, F = 3000, 150, 4 if cart else 2
    clip_obs, clip_rew, gamma, eps = 2.5, 1.5, 0.9, 1e-4
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | (0 if cart else L.WRAP_CLIP_REWARD)
    w |= (L.WRAP_NORM_TRAINING if training else 0) | (L.WRAP_NORM_REWARD if norm_reward else 0) | (L.WRAP_NORM_OBS if norm_obs else 0)
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64,
                              reset_mode=L.RESET_TAPE, wrappers=w, max_episode_steps=60, clip_obs=clip_obs, clip_reward=clip_rew,
                              gamma=gamma, epsilon=eps), 0)
    ov = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, n, F_TIME_LIMIT | F_MONITOR | (0 if cart else F_CLIP_REWARD), 60)
    if cart:
        ov.seed(21)
        eng.upload_reset_tape(oracle.cartpole_reset_tape(21, n, T // 8 + 4))
    ov.attach_norm(clip_obs, clip_rew, gamma, eps)
    ov.norm_flags(training, norm_reward, norm_obs)
    loaded = np.concatenate([np.linspace(-0.3, 0.2, F), np.linspace(0.02, 0.3, F), [5000.0], [0.7, 2.5, 5000.0]]).astype(np.float32)
    eng.set_norm_stats(loaded)
    ov.norm_set_stats(loaded)
    eng.reset()
    o_obs, o_raw = ov.norm_reset()
    assert np.array_equal(eng.obs.cpu().numpy(), o_raw)
    acts = splitmix_actions(31, T, n, 2 if cart else 3)
    dev = torch.from_numpy(acts).cuda()
    for t in range(T):
        eng.step(dev[t])
        r = ov.norm_step(acts[t])
        st = ov.norm_stats()
        assert np.array_equal(eng.obs.cpu().numpy(), r["raw_obs"]) and np.array_equal(eng.done.cpu().numpy(), r["done"]), t
        _check_stats(eng.get_norm_stats(), st, F, f"step {t}")
        g_obs = (eng.obs_norm if norm_obs else eng.obs).cpu().numpy()
        _check_normalised(g_obs, r["obs"], eng.reward.cpu().numpy(), r["reward"], st, F, eps, f"step {t}")
        if not norm_reward:
            assert np.array_equal(eng.reward.cpu().numpy(), r["reward"])            # raw rewards pass through
    st = eng.get_norm_stats()
    if not (training and norm_obs):
        assert np.array_equal(st[:2 * F + 1], loaded[:2 * F + 1])                   # obs statistics frozen
    if not (training and norm_reward):
        assert np.array_equal(st[2 * F + 1:], loaded[2 * F + 1:])                   # return statistics frozen
    assert eng.poll_error() == 0


@pytest.mark.parametrize("kind,n", [("mountaincar", 1), ("mountaincar", 4097), ("cartpole", 70001), ("mountaincar", 1 << 20)])
def test_unsplit_step_equals_begin_plus_finish_bit_for_bit(kind, n):
    """gymhip_step / gymhip_reset launch two kernels (the normalise kernel folds the step kernel's partial rows in its
    prologue); gymhip_*_begin + gymhip_*_finish launch three (the separate reduce feeds the all-reduce).  Same summation
    function, same order: every output, the statistics and the moments buffer must agree bit for bit."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    cart = kind == "cartpole"
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | (0 if cart else L.WRAP_CLIP_REWARD) | _norm_flags()
    mk = lambda: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n, reset_mode=L.RESET_COUNTER_RNG, wrappers=w,
                                     max_episode_steps=37, seed=3), 0)
    a, b = mk(), mk()
    a.reset()
    b.reset_begin(); b.reset_finish()
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(60):
        act = torch.randint(0, 2 if cart else 3, (n,), dtype=torch.int32, device="cuda", generator=g)
        a.step(act)
        b.step_begin(act); b.step_finish()
        if t % 7 == 0 or t == 59:
            for k in ("obs", "obs_norm", "reward", "done", "returns", "counter", "norm_stats", "moments"):
                assert torch.equal(getattr(a, k), getattr(b, k)), (t, k)
    assert np.array_equal(a.episode_stats(), b.episode_stats()) and a.episode_stats()[0] > 0
    assert a.poll_error() == 0 and b.poll_error() == 0
