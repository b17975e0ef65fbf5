"""GPU: the host-side mirror of gym::VecEnv<false> / VecNormalize (gym_amd.vec_env) drives the engine the way a training
loop drives the reference (test/classic_control.cpp:26-41): same call sequence, same shapes and dtypes, same values."""
import numpy as np
import pytest
import torch

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, splitmix_actions

pytestmark = pytest.mark.gpu


def test_training_loop_shape_cartpole_monitor_vecnormalize(oracle, tmp_path):
    """test/classic_control.cpp:105-107: Monitor<CartPoleEnv> x N -> SyncVecEnv -> TrainingVecNormRO, driven by step()."""
    from gym_amd.vec_env import HipVecEnv, TrainingVecNormRO
    n, T, seed = 1024, 200, 5
    env = TrainingVecNormRO(HipVecEnv.make("CartPole-v1", n, monitor=True, tape_episodes=T // 8 + 4))
    assert env.seed(seed)[:2] == [seed, seed + 1]
    ov = oracle.vec(CARTPOLE, n, F_MONITOR, 0)
    ov.seed(seed)
    ov.attach_norm()
    n_act = env.actionSpace().as_(type(env.actionSpace())).n
    assert n_act == 2
    obs = env.reset()
    o_obs, o_raw = ov.norm_reset()
    assert obs.shape == (n, 4) and obs.dtype == torch.float32                      # test/classic_control.cpp:85-86
    assert np.array_equal(env.oldObs().cpu().numpy(), o_raw)
    acts = splitmix_actions(21, T, n, n_act)
    seen_episode = False
    for t in range(T):
        a = torch.from_numpy(acts[t].astype(np.int64))                             # torch::randint(n, N_env) is int64 on the host
        s = env.step(a)
        r = ov.norm_step(acts[t])
        assert s.reward.shape == (n, 1) and s.done.shape == (n, 1) and s.done.dtype == torch.float32   # sync_env.h:107-112
        assert np.array_equal(s.done.cpu().numpy()[:, 0], r["done"])
        assert np.array_equal(env.oldObs().cpu().numpy(), r["raw_obs"])
        assert np.allclose(s.reward.cpu().numpy()[:, 0], r["reward"], rtol=1e-5, atol=1e-6)
        assert len(s.info) == n
        d = np.nonzero(r["done"])[0]
        if len(d):
            i = int(d[0])
            assert list(s.info.done_indices()) == list(d)
            ep = s.info[i]["episode"]                                              # monitor.h:86-97
            assert ep["r"] == r["ep_r"][i] and ep["l"] == r["ep_l"][i] and ep["t"] > 0
            assert "TimeLimit.truncated" not in s.info[i]                          # no TimeLimit in this chain
            nd = int(np.nonzero(r["done"] == 0)[0][0])
            assert s.info[nd] == {}
            seen_episode = True
    assert seen_episode
    obs_rms, ret_rms = env.get()
    st = ov.norm_stats()
    assert np.allclose(obs_rms["mean"], st[:4], rtol=1e-5, atol=1e-8) and np.allclose(obs_rms["m_Var"], st[4:8], rtol=1e-5)
    assert obs_rms["m_Count"] == st[8] and ret_rms["m_Count"] == st[11]
    # save / load (vec_normalize.h:79-108): returns.pt + observation.pt with keys mean / m_Var / m_Count
    env.save(tmp_path)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["observation.pt", "returns.pt"]
    # the reference's archive format: a TorchScript module file with three non-buffer tensors (tests/test_ref_pin.py pins it)
    assert sorted(k for k, _ in torch.jit.load(str(tmp_path / "observation.pt")).named_parameters()) == ["m_Count", "m_Var", "mean"]
    from gym_amd.vec_env import VecNormalize
    ev = VecNormalize(HipVecEnv("CartPole-v1", n, tape_episodes=8), training=False)
    ev.seed(seed)
    ev.load(tmp_path)
    o1, o2 = ev.get()
    assert np.array_equal(o1["mean"], obs_rms["mean"]) and o2["m_Var"] == ret_rms["m_Var"]
    ev.reset()
    assert np.array_equal(ev.get()[0]["mean"], obs_rms["mean"])                    # frozen when not training


def test_time_limit_info_and_unseeded_default_stream(oracle):
    from gym_amd.vec_env import HipVecEnv
    n = 256
    env = HipVecEnv("CartPole-v1", n, max_episode_steps=12, tape_episodes=64)      # never seeded
    obs = env.reset().cpu().numpy()
    assert (obs == obs[0]).all()                                                   # every env replays the default mt19937
    assert np.array_equal(obs[0], oracle.cartpole_reset_tape(5489, 1, 1)[0, 0])
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 12)
    ov.reset()
    acts = splitmix_actions(2, 40, n, 2)
    trunc_seen = term_seen = 0
    for t in range(40):
        s = env.step(torch.from_numpy(acts[t].astype(np.int32)).cuda())
        r = ov.step(acts[t])
        assert np.array_equal(s.done.cpu().numpy()[:, 0], r["done"])
        for i in s.info.done_indices()[:8]:
            assert s.info[int(i)]["TimeLimit.truncated"] == bool(r["truncated"][i])
            assert s.info[int(i)]["episode"]["l"] == r["ep_l"][i]
        trunc_seen += int(r["truncated"].sum()); term_seen += int(((r["done"] != 0) & (r["truncated"] == 0)).sum())
    assert trunc_seen > 0 and term_seen > 0
    tot = env.episode_statistics()
    assert tot[0] == trunc_seen + term_seen


def test_mountaincar_f32_mode_single_step_parity(oracle):
    """Throughput (float-state) mode: from identical states one step agrees with the reference arithmetic to 1e-5;
    over long horizons float32 state drifts — that drift is measured, not asserted away."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n, T = 4096, 200
    eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_F32, reset_mode=L.RESET_COUNTER_RNG,
                              wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=200), 0)
    ov = oracle.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_MONITOR, 200)
    eng.reset()
    ov.reset()
    acts = splitmix_actions(12, T, n, 3)
    worst = 0.0
    for t in range(T):
        eng.step(torch.from_numpy(acts[t]).cuda())
        r = ov.step(acts[t])
        assert np.array_equal(eng.done.cpu().numpy(), r["done"])                   # all episodes truncate at 200
        err = np.abs(eng.obs.cpu().numpy() - r["obs"]) / np.maximum(np.abs(r["obs"]), 1e-2)
        worst = max(worst, err.max())
    assert worst <= 2e-4, worst        # survey: max 5.9e-5 for float state + cosf over 200-step episodes


def test_cartpole_f32_mode_short_horizon_and_counter_rng(oracle):
    """Float-state CartPole with the stateless COUNTER_RNG resets: (a) reset draws lie in [-0.05, 0.05) and are
    reproducible; (b) replaying a device-generated tape, the first steps of every episode agree with the oracle to 1e-5
    (episode age <= 10: SURVEY.md §7 measured 0.08 % of samples above 1e-5 there, all below 1e-4)."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    n, T = 8192, 10
    mk = lambda mode: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_F32, reset_mode=mode,
                                          wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500, seed=99), 0)
    a, b = mk(L.RESET_COUNTER_RNG), mk(L.RESET_COUNTER_RNG)
    a.reset(); b.reset()
    oa = a.obs.cpu().numpy()
    assert np.array_equal(oa, b.obs.cpu().numpy()) and (oa >= -0.05).all() and (oa < 0.05).all()
    assert np.unique(oa[:, 0]).size > n * 0.99 and abs(oa.mean()) < 2e-3
    acts = splitmix_actions(31, 60, n, 2)
    for t in range(60):
        a.step(torch.from_numpy(acts[t]).cuda()); b.step(torch.from_numpy(acts[t]).cuda())
    assert torch.equal(a.obs, b.obs) and a.episode_stats()[0] == b.episode_stats()[0] > n

    eng = mk(L.RESET_TAPE)
    eng.generate_reset_tape(8)
    tape = eng.download_reset_tape(8)
    assert (tape >= -0.05).all() and (tape < 0.05).all()
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.set_reset_tape(tape)
    eng.reset()
    assert np.array_equal(eng.obs.cpu().numpy(), ov.reset())
    bad = 0
    valid = np.ones(n, bool)            # an env whose done stream flipped once is on a different episode from then on
    for t in range(T):
        eng.step(torch.from_numpy(acts[t]).cuda())
        r = ov.step(acts[t])
        valid &= eng.done.cpu().numpy() == r["done"]
        err = np.abs(eng.obs.cpu().numpy() - r["obs"]) / np.maximum(np.abs(r["obs"]), 1e-2)
        assert err[valid].max() <= 1e-4
        bad += int((err[valid] > 1e-5).sum())
    assert valid.mean() >= 0.9999 and bad <= 0.002 * n * T * 4


def test_monitor_csv_logs_the_finished_episodes_of_chosen_envs(oracle, tmp_path):
    """Monitor(env, ..., id, fileName) per env (monitor.h:41-50,93-94): one CSV per logged env, rows (t, r, l) of its episodes."""
    from gym_amd.monitor_csv import MonitorCsv
    from gym_amd.vec_env import HipVecEnv
    n, T, seed = 256, 150, 2
    logged = [0, 5, 255]
    env = MonitorCsv(HipVecEnv.make("CartPole-v1", n, max_episode_steps=500, monitor=True, tape_episodes=T // 8 + 4), str(tmp_path), logged)
    env.seed(seed)
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(seed)
    env.reset(); ov.reset()
    acts = splitmix_actions(31, T, n, 2)
    want = {i: [] for i in logged}
    for t in range(T):
        env.step(torch.from_numpy(acts[t]).cuda())
        r = ov.step(acts[t])
        for i in logged:
            if r["done"][i]:
                want[i].append((float(r["ep_r"][i]), float(r["ep_l"][i])))
    env.close()
    for i in logged:
        text = open(tmp_path / f"env{i}.csv").read()
        head, body = text.split("t,r,l", 1)
        assert head.splitlines()[1] == f"id: env{i}"
        rows = [tuple(float(x) for x in line.split(",")) for line in body.strip().split("\n")] if body.strip() else []
        assert [(r_, l_) for _, r_, l_ in rows] == want[i] and len(rows) >= 3
        assert all(b[0] >= a[0] for a, b in zip(rows, rows[1:]))                   # t is monotone


def test_second_reset_and_midrun_reseed_follow_the_envs_generators(oracle):
    """A CartPoleEnv's mt19937 keeps running across reset() calls (cartpole.cpp:21-26) and restarts on seed()
    (env.h:60-63): a second VecEnv::reset() draws each env's NEXT four values, a mid-run seed() starts new streams."""
    from gym_amd.vec_env import HipVecEnv
    n, seed = 777, 9
    env = HipVecEnv.make("CartPole-v1", n, max_episode_steps=500, monitor=True, tape_episodes=64)
    env.seed(seed)
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(seed)
    acts = splitmix_actions(5, 120, n, 2)

    def run(t0, t1):
        for t in range(t0, t1):
            s = env.step(torch.from_numpy(acts[t].astype(np.int32)).cuda())
            r = ov.step(acts[t])
            assert np.array_equal(s.done.cpu().numpy()[:, 0], r["done"]), t
            assert np.allclose(s.observation.cpu().numpy(), r["obs"], rtol=1e-5, atol=1e-7), t

    first = env.reset().clone()
    assert np.array_equal(first.cpu().numpy(), ov.reset())
    run(0, 40)                                      # episodes end and auto-reset along the way: cursors differ per env
    second = env.reset()
    assert np.array_equal(second.cpu().numpy(), ov.reset()) and not torch.equal(first, second)
    run(40, 80)
    env.seed(seed + 1000)                           # mid-run: the NEXT reset of every env (auto or explicit) uses the new stream
    ov.seed(seed + 1000)
    run(80, 100)
    assert np.array_equal(env.reset().cpu().numpy(), ov.reset())
    run(100, 120)
    assert env.engine.poll_error() == 0


@pytest.mark.parametrize("max_steps", [None, 3])
def test_reset_tape_is_extended_before_any_env_runs_out(oracle, max_steps):
    """The drop-in's default must never replay reset states: starting from FOUR tape rows per env, the mirror's tape guard
    (cursor look-ups budgeted by the shortest possible episode, then the same std::mt19937 streams redrawn with more rows)
    keeps every stream identical to the reference generator's across 600 steps (~27 resets per env, 200 with TimeLimit(3))."""
    from gym_amd.vec_env import HipVecEnv
    n, T, seed = 300, 600, 11
    env = HipVecEnv("CartPole-v1", n, max_episode_steps=max_steps, tape_episodes=4)
    env.seed(seed)
    ov = oracle.vec(CARTPOLE, n, (F_TIME_LIMIT if max_steps else 0) | F_MONITOR, max_steps or 0)
    ov.seed(seed)
    assert np.array_equal(env.reset().cpu().numpy(), ov.reset())
    acts = splitmix_actions(5, T, n, 2)
    dev = torch.from_numpy(acts.astype(np.int32)).cuda()
    for t in range(T):
        s = env.step(dev[t])
        r = ov.step(acts[t])
        assert np.array_equal(s.done.cpu().numpy()[:, 0], r["done"]), t
        assert np.array_equal(s.observation.cpu().numpy(), r["obs"]), t
        if t == 300:   # per-index and whole-batch resets consume rows too
            assert np.array_equal(env.reset().cpu().numpy(), ov.reset())
    assert env._tape_episodes > 4 and env.engine.tape_cursor_max() > 20
    assert env.engine.tape_cursor_max() <= env._tape_episodes and env.engine.poll_error() == 0


def test_raw_engine_still_reports_a_tape_that_ran_out():
    """Below the mirrors, the C ABI keeps its documented behaviour: rows wrap and the sticky error word says so."""
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=64, reset_mode=L.RESET_TAPE, wrappers=L.WRAP_TIME_LIMIT, max_episode_steps=2), 0)
    eng.seed_reset_tape(0, 3)
    eng.reset()
    a = torch.zeros(64, dtype=torch.int32, device="cuda")
    for _ in range(3):
        eng.step(a)
    assert eng.poll_error() == 0 and eng.tape_cursor_max() == 2
    for _ in range(4):
        eng.step(a)
    assert eng.poll_error() & 1
    with pytest.raises(L.GymhipError):
        eng.extend_reset_tape(2)                       # not more rows than it has
    up = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=8, reset_mode=L.RESET_TAPE), 0)
    up.upload_reset_tape(np.zeros((2, 8, 4), np.float32))
    with pytest.raises(L.GymhipError):
        up.extend_reset_tape(16)                       # only host-drawn streams can be extended


def test_step_info_must_be_read_before_the_next_step():
    """The reference returns concrete per-step info maps; LazyInfos reads the engine's buffers on first use and therefore
    refuses to be read after a later step instead of silently returning that step's records."""
    from gym_amd.vec_env import HipVecEnv
    env = HipVecEnv("CartPole-v1", 64, max_episode_steps=5)
    env.seed(0)
    env.reset()
    a = torch.zeros(64, dtype=torch.int32, device="cuda")
    kept = None
    for t in range(5):
        s = env.step(a)
        if t == 3:
            stale, kept = s.info, env.step(a).info.materialize()
            break
    with pytest.raises(RuntimeError, match="before the next step"):
        stale[0]
    assert len(kept.done_indices()) == 64 and kept[0]["episode"]["l"] == 5.0 and kept[0]["TimeLimit.truncated"] in (True, False)
    env.step(a)
    assert kept[0]["episode"]["l"] == 5.0             # materialised records survive later steps


@pytest.mark.parametrize("kind", ["cartpole", "mountaincar"])
def test_auto_reset_off_returns_terminal_observations_and_waits_for_reset_index(oracle, kind):
    """SyncVecEnv(envs, auto_reset = false) (sync_env.h:28,44,92): a finished env is not reset — the step returns its TERMINAL
    observation — and it is reset alone by reset(index) (sync_env.h:128-139).  Against the oracle's SyncVecEnv with the flag off."""
    from gym_amd.vec_env import HipVecEnv
    cart = kind == "cartpole"
    n, T, seed = 48, 260 if cart else 420, 9
    env = HipVecEnv("CartPole-v1" if cart else "MountainCar-v0", n, max_episode_steps=60 if cart else 200, auto_reset=False)
    ov = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, n, F_TIME_LIMIT | F_MONITOR, 60 if cart else 200)
    ov.set_auto_reset(False)
    env.seed(seed); ov.seed(seed)
    assert np.array_equal(env.reset().cpu().numpy(), ov.reset())
    acts = splitmix_actions(6, T, n, 2 if cart else 3)
    n_done = 0
    for t in range(T):
        s = env.step(torch.from_numpy(acts[t].astype(np.int32)).cuda())
        r = ov.step(acts[t], want64=True)
        assert np.array_equal(s.done.cpu().numpy()[:, 0], r["done"]) and np.array_equal(s.reward.cpu().numpy()[:, 0], r["reward"]), t
        got = s.observation.cpu().numpy()
        if cart:
            assert np.array_equal(got, r["obs"]), t                         # terminal rows included
            assert np.array_equal(env.engine.state_f64().cpu().numpy(), r["obs64"]), t
        else:
            assert np.abs(got - r["obs"]).max() <= 1e-7, t
        for i in np.nonzero(r["done"])[0]:
            n_done += 1
            info = s.info[int(i)]
            assert info["episode"]["l"] == r["ep_l"][i] and info["TimeLimit.truncated"] == bool(r["truncated"][i])
            row = env.reset(int(i)).cpu().numpy()
            orow, _ = ov.reset_index(int(i))
            assert np.array_equal(row, orow), (t, i)
    assert n_done > n and env.engine.poll_error() == 0 and env.engine.episode_stats()[0] == n_done


def test_auto_reset_off_stepping_a_finished_env_is_reported_not_simulated():
    """The misuse the reference's Monitor reports on stderr (monitor.h:75-77): the env stays at its terminal state, reward 0,
    done 1, and bit 3 of the error word is set."""
    from gym_amd.vec_env import HipVecEnv
    env = HipVecEnv("CartPole-v1", 8, max_episode_steps=3, auto_reset=False)
    env.seed(0)
    env.reset()
    a = torch.zeros(8, dtype=torch.int32, device="cuda")
    for _ in range(3):
        s = env.step(a)
    assert s.done.sum().item() == 8 and env.engine.poll_error() == 0
    terminal = s.observation.clone()
    s = env.step(a)
    assert torch.equal(s.observation, terminal) and s.reward.abs().sum().item() == 0 and s.done.sum().item() == 8
    assert env.engine.poll_error() & 8 and env.engine.episode_stats()[0] == 8      # no second episode record
    assert env.reset(2).shape == (4,)
    s = env.step(a)
    assert s.done[2].item() == 0 and s.reward[2].item() == 1.0 and s.done.sum().item() == 7


def test_the_references_own_smoke_scenarios_at_their_sizes(oracle):
    """test/classic_control.cpp at the sizes it uses: (a) :105-107 — 10 CartPole envs under Monitor + TrainingVecNormRO for 20,000
    env-steps (2,000 vec steps: a long horizon for the float32 running statistics); (b) :109-112 — 1028 envs for 1e6 env-steps
    (973 vec steps), where the reference only asserts obs.size(0) == N_env and obs.size(1) == 4 (:85-86) and prints a wall time.
    Here every step is also compared with the oracle."""
    from gym_amd.vec_env import HipVecEnv, TrainingVecNormRO
    # (a)
    n, T, seed = 10, 2000, 1
    env = TrainingVecNormRO(HipVecEnv.make("CartPole-v1", n, monitor=True))
    env.seed(seed)
    ov = oracle.vec(CARTPOLE, n, F_MONITOR, 0)
    ov.seed(seed); ov.attach_norm()
    obs = env.reset()
    o_obs, _ = ov.norm_reset()
    assert obs.shape == (n, 4) and np.allclose(obs.cpu().numpy(), o_obs, atol=1e-5)
    acts = splitmix_actions(3, T, n, 2)
    episodes = 0
    for t in range(T):
        s = env.step(torch.from_numpy(acts[t].astype(np.int64)))
        r = ov.norm_step(acts[t])
        assert np.array_equal(s.done.cpu().numpy()[:, 0], r["done"]) and np.array_equal(env.oldObs().cpu().numpy(), r["raw_obs"]), t
        st = ov.norm_stats()
        cond = np.maximum(1.0, np.abs(st[:4]) / np.sqrt(st[4:8] + 1e-8))
        assert (np.abs(s.observation.cpu().numpy() - r["obs"]) <= 1e-5 * cond * np.maximum(1.0, np.abs(r["obs"]))).all(), t
        assert np.allclose(s.reward.cpu().numpy()[:, 0], r["reward"], rtol=1e-5, atol=1e-6), t
        episodes += int(r["done"].sum())
    got, want = env.venv.engine.get_norm_stats(), ov.norm_stats()
    assert (np.abs(got - want) <= 1e-5 * np.maximum(np.abs(want), 1e-6)).all() and got[8] == want[8] == np.float32(1e-4 + n * (T + 1))
    assert episodes > 500 and env.venv.engine.episode_stats()[0] == episodes
    # (b)
    n, T = 1028, 973
    env = HipVecEnv.make("CartPole-v1", n, max_episode_steps=500, monitor=True)
    env.seed(0)
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 500)
    ov.seed(0)
    obs = env.reset()
    assert obs.size(0) == n and obs.size(1) == 4 and np.array_equal(obs.cpu().numpy(), ov.reset())
    acts = splitmix_actions(9, T, n, 2)
    for t in range(T):
        s = env.step(torch.from_numpy(acts[t].astype(np.int64)))
        r = ov.step(acts[t])
        assert np.array_equal(s.observation.cpu().numpy(), r["obs"]) and np.array_equal(s.done.cpu().numpy()[:, 0], r["done"]), t
    assert env.engine.poll_error() == 0


@pytest.mark.parametrize("seed", [None, 17])
def test_default_reset_source_is_the_device_resident_mt19937_and_matches_the_oracle(oracle, seed):
    """HipVecEnv without a reset_mode / tape_episodes: every env's std::mt19937 lives on the device (GYMHIP_RESET_MT19937) — no tape,
    no tape guard — and the streams are the reference's: seeded `seed + i` (sync_env.h:74-83) or never seeded (env.h:83), a mid-run
    re-seed included, against the oracle's per-env generators over ~3 regenerations of every generator block."""
    from gym_amd import _lib as L
    from gym_amd.vec_env import HipVecEnv
    n, T = 700, 1500
    env = HipVecEnv("CartPole-v1", n, max_episode_steps=3)
    ov = oracle.vec(CARTPOLE, n, F_TIME_LIMIT | F_MONITOR, 3)
    if seed is not None:
        assert list(env.seed(seed)) == [seed + i for i in range(n)]
        ov.seed(seed)
    assert env._reset_mode == L.RESET_MT19937
    assert np.array_equal(env.reset().cpu().numpy(), ov.reset())
    acts = splitmix_actions(6, T, n, 2)
    for t in range(T):
        s = env.step(torch.from_numpy(acts[t]).cuda())
        r = ov.step(acts[t])
        if t % 10 == 0 or t == T - 1:
            assert np.array_equal(s.observation.cpu().numpy(), r["obs"]), t
            assert np.array_equal(s.done.cpu().numpy()[:, 0], r["done"]), t
        if t == 900 and seed is not None:
            env.seed(seed + 1); ov.seed(seed + 1)       # VecEnv::seed mid-run: every generator restarts (env.h:60-63)
    assert env.engine.poll_error() == 0 and int(env.engine.counters()["cursor"].min()) >= (190 if seed is not None else 480)
