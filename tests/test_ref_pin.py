"""CPU: pin the oracle against the REFERENCE'S OWN CODE for rows A4 / A5 / A8 / A9.

tests/golden/ref_*.npz hold outputs of oracle/_ref/ref_wrappers_driver — the reference's TimeLimit, ClipRewardEnv,
Monitor and RunningMeanStd headers compiled in place against libtorch (oracle/Makefile target _ref; generator
tests/golden/make_golden_ref.py).  The frozen vectors are checked always; when the driver binary is present (build
container) it is re-run live as well."""
import os

import numpy as np
import pytest

from tests.golden import make_golden_ref as R
from tests.oracle_lib import SCRIPTED, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT

G = os.path.dirname(os.path.abspath(R.__file__))
HAVE_DRIVER = os.path.exists(R.DRIVER)


def check_wrappers(oracle, rec, rew, done, max_steps):
    v = oracle.vec(SCRIPTED, 1, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, max_steps, script=(rew, done))
    v.reset()
    a = np.zeros(1, np.int32)
    for s in range(len(rec)):
        o = v.step(a)
        r = rec[s] if isinstance(rec, np.ndarray) else {k: rec[k][s] for k in R.REC.names}
        assert o["reward"][0] == r["reward"] and np.signbit(o["reward"][0]) == np.signbit(r["reward"]), s   # keeps -0.0 (clip_reward.h:23)
        assert bool(o["done"][0]) == bool(r["done"]), s
        assert int(o["truncated"][0]) == int(r["has_trunc"] and r["trunc"]), s
        assert (o["ep_l"][0] > 0) == bool(r["has_ep"]), s
        assert o["ep_r"][0] == r["ep_r"] and o["ep_l"][0] == r["ep_l"], s


def check_rms(oracle, rows, batches, F):
    st = oracle.rms_init(F)
    st_exact = st.copy()
    for k, x in enumerate(batches):
        ref_bm, ref_bv, ref_mean, ref_var, ref_cnt = rows[k, :F], rows[k, F:2 * F], rows[k, 2 * F:3 * F], rows[k, 3 * F:4 * F], rows[k, 4 * F]
        bm, bv = oracle.rms_batch_moments(x)
        assert np.allclose(bm, ref_bm, rtol=3e-7, atol=1e-7), k         # ATen's float32 cascade mean vs the exact mean: few ulp
        if x.shape[0] > 1:
            assert np.allclose(bv, ref_bv, rtol=3e-7, atol=1e-12), k
        # the merge (running_mean_std.h:50-57) on the reference's own batch moments: bit-exact
        oracle.rms_update_from_moments(st_exact, ref_bm, np.nan_to_num(ref_bv), x.shape[0]) if x.shape[0] > 1 else None
        if x.shape[0] > 1:
            assert np.array_equal(st_exact[:F], ref_mean) and np.array_equal(st_exact[F:2 * F], ref_var) and st_exact[2 * F] == ref_cnt, k
        # end to end with the oracle's own batch moments: running statistics within 1e-6
        oracle.rms_update_from_moments(st, bm, bv, x.shape[0])
        if x.shape[0] > 1:
            assert np.allclose(st[:F], ref_mean, rtol=2e-6, atol=1e-7) and np.allclose(st[F:2 * F], ref_var, rtol=2e-6, atol=1e-10), k
        else:
            st[:] = np.concatenate([ref_mean, ref_var, [ref_cnt]])      # n=1: torch var of one row is 0; resync and go on
            st_exact[:] = st


def test_oracle_wrappers_vs_frozen_reference_output(oracle):
    g = np.load(os.path.join(G, "ref_wrappers.npz"))
    rew, done = R.wrapper_script(int(g["script_seed"]), int(g["script_len"]))
    assert int(g["done"].sum()) > 300 and int(g["trunc"].sum()) > 50 and int(g["has_ep"].sum()) == int(g["done"].sum())
    check_wrappers(oracle, g, rew, done, int(g["max_steps"]))


def test_oracle_running_mean_std_vs_frozen_reference_output(oracle):
    g = np.load(os.path.join(G, "ref_rms.npz"))
    F = int(g["F"])
    check_rms(oracle, g["rows"], R.rms_batches(int(g["batch_seed"]), int(g["n_batches"]), F), F)


@pytest.mark.skipif(not HAVE_DRIVER, reason="oracle/_ref not built (needs /root/reference; build container only)")
def test_oracle_vs_live_reference_driver(oracle):
    for seed, L, max_steps in ((1, 101, 5), (2, 64, 1000), (3, 333, 2)):
        rew, done = R.wrapper_script(seed, L)
        check_wrappers(oracle, R.run_wrappers(rew, done, max_steps, 1500), rew, done, max_steps)
    for seed in (8, 9):
        b = R.rms_batches(seed, 12, 2)
        check_rms(oracle, R.run_rms(b, 2), b, 2)


def test_checkpoint_files_written_by_the_reference_load_here():
    """(f2) VecNormalize::save format: files produced by the reference's torch::save(RunningMeanStd) (frozen under
    tests/golden/) read back bit-exactly through gym_amd.checkpoint."""
    from gym_amd.checkpoint import load_running_mean_std
    for name in ("observation", "returns"):
        g = np.load(os.path.join(G, f"ref_rms_checkpoint_{name}.npz"))
        mean, var, cnt = load_running_mean_std(os.path.join(G, f"ref_rms_checkpoint_{name}.pt"))
        assert mean.shape == ((int(g["F"]),) if int(g["F"]) else ())
        assert np.array_equal(mean.reshape(-1), g["mean"]) and np.array_equal(var.reshape(-1), g["var"]) and np.float32(cnt) == g["count"]


@pytest.mark.skipif(not HAVE_DRIVER, reason="oracle/_ref not built (needs /root/reference; build container only)")
def test_checkpoint_round_trip_with_live_reference_code(tmp_path):
    """Both directions against the reference's own save()/load() (running_mean_std.h:28-40)."""
    from gym_amd.checkpoint import load_running_mean_std, save_running_mean_std
    for F, seed in ((4, 1), (2, 2), (0, 3)):
        pt = str(tmp_path / f"ref_{F}.pt")
        ref = R.run_ckpt_save(R.ckpt_batch(seed, F), F, pt)
        got = load_running_mean_std(pt)
        assert np.array_equal(got[0].reshape(-1), ref[0]) and np.array_equal(got[1].reshape(-1), ref[1]) and np.float32(got[2]) == np.float32(ref[2])
        ours = str(tmp_path / f"ours_{F}.pt")
        save_running_mean_std(ours, got[0], got[1], got[2])
        back = R.run_ckpt_load(ours)
        assert np.array_equal(back[0], ref[0]) and np.array_equal(back[1], ref[1]) and np.float32(back[2]) == np.float32(ref[2])


def test_monitor_csv_bytes_match_the_reference_writer(oracle, tmp_path):
    """(f3) Monitor's CSV log: tests/golden/ref_monitor.csv was written by the reference's ResultWriter (monitor.h:20-35).
    Re-writing its rows through gym_amd.monitor_csv.ResultWriter reproduces the file byte for byte (header without a
    trailing newline, %g floats), and the (r, l) columns equal the oracle's Monitor over the same script."""
    from gym_amd.monitor_csv import ResultWriter
    raw = open(os.path.join(G, "ref_monitor.csv"), "rb").read().decode()
    head, body = raw.split("t,r,l", 1)
    t_start = int(head.splitlines()[0].split(": ")[1]); env_id = head.splitlines()[1].split(": ")[1]
    rows = [tuple(float(x) for x in line.split(",")) for line in body.strip().split("\n")]
    w = ResultWriter(str(tmp_path / "again.whatever"), t_start, env_id)
    for t, r, l in rows:
        w.write(r, l, t)
    w.close()
    assert w.path.endswith("again.csv")
    assert open(w.path, "rb").read().decode() == raw
    # the episodes themselves: TimeLimit(13) + Monitor over the scripted env, 1500 steps
    rew, done = R.wrapper_script(11, 997)
    v = oracle.vec(SCRIPTED, 1, F_TIME_LIMIT | F_MONITOR, 13, script=(rew, done))
    v.reset()
    eps = []
    for _ in range(1500):
        o = v.step(np.zeros(1, np.int32))
        if o["done"][0]:
            eps.append((o["ep_r"][0], o["ep_l"][0]))
    assert len(eps) == len(rows)
    for (t, r, l), (er, el) in zip(rows, eps):
        assert float("%g" % er) == r and float(el) == l


MIRROR_WRAPPERS = os.path.join(os.path.dirname(G), "cpp", "bin", "wrappers_mirror_demo")


def test_cpp_wrapper_mirror_equals_the_references_templates(tmp_path):
    """gym/wrappers.h (stand-alone Wrapper / RewardWrapper / TimeLimit / ClipRewardEnv / Monitor / ResultWriter) under the very
    probe the reference's own templates were put through (tests/cpp/wrappers_probe.h): per-step reward (sign of zero included),
    done, info["TimeLimit.truncated"], info["episode"] equal the frozen reference output record for record, and the CSV file is
    the reference's byte for byte apart from the wall-clock column and the t_start stamp."""
    assert os.path.exists(MIRROR_WRAPPERS), "build tests/cpp first"
    g = np.load(os.path.join(G, "ref_wrappers.npz"))
    rew, done = R.wrapper_script(int(g["script_seed"]), int(g["script_len"]))
    rec = R.run_wrappers(rew, done, int(g["max_steps"]), len(g["done"]), driver=MIRROR_WRAPPERS)
    for name in R.REC.names:
        assert np.ascontiguousarray(rec[name]).tobytes() == np.ascontiguousarray(g[name]).astype(rec[name].dtype).tobytes(), name      # bit patterns: keeps -0.0
    # the CSV: same script and wrapper stack as tests/golden/ref_monitor.csv
    rew, done = R.wrapper_script(11, 997)
    out = tmp_path / "mirror.anything"
    R.run_csv(rew, done, 13, 1500, str(out), driver=MIRROR_WRAPPERS)
    mine = open(tmp_path / "mirror.csv", "rb").read().decode()
    ref = open(os.path.join(G, "ref_monitor.csv"), "rb").read().decode()
    mh, mb = mine.split("t,r,l", 1); rh, rb = ref.split("t,r,l", 1)
    assert mh.splitlines()[0].startswith("t_start: ") and mh.splitlines()[1] == rh.splitlines()[1] == "id: env7" and mine.count("\n") == ref.count("\n")
    for a, b in zip(mb.strip().split("\n"), rb.strip().split("\n")):
        assert a.split(",")[1:] == b.split(",")[1:]                      # r, l as the reference printed them; t is wall time


@pytest.mark.skipif(not HAVE_DRIVER, reason="oracle/_ref not built (needs /root/reference; build container only)")
def test_cpp_wrapper_mirror_vs_live_reference_driver():
    assert os.path.exists(MIRROR_WRAPPERS), "build tests/cpp first"
    for seed, L, max_steps in ((1, 101, 5), (2, 64, 1000), (3, 333, 2), (4, 50, 1)):
        rew, done = R.wrapper_script(seed, L)
        a = R.run_wrappers(rew, done, max_steps, 1500)
        b = R.run_wrappers(rew, done, max_steps, 1500, driver=MIRROR_WRAPPERS)
        assert a.tobytes() == b.tobytes(), (seed, L, max_steps)


def unroll_expected(rew, done, steps):
    """What the probe's two stacks must produce, from the definitions: ScriptEnv's observation is its step count, the chain
    clips the reward to its sign (zeros pass, clip_reward.h:23) and maps obs -> 2 obs + 1; the adapter returns float(obs / 2 - 3)."""
    rows = np.zeros(steps + 1, R.UNROLL)
    rows["obs"][0], rows["obs2"][0] = 1.0, np.float32(-3.0)
    for s in range(steps):
        i = s % rew.size
        r = rew[i]
        rows["obs"][s + 1] = 2.0 * (s + 1) + 1
        rows["reward"][s + 1] = r if r == 0 else np.float32(1.0 if r > 0 else -1.0)
        rows["obs2"][s + 1] = np.float32(0.5 * (s + 1) - 3.0)
        rows["reward2"][s + 1] = r
        rows["done"][s + 1] = rows["done2"][s + 1] = done[i]
    return rows


def test_cpp_wrapper_unroll_and_observation_wrapper2_mirror_equal_the_references(tmp_path):
    """common/wrapper.h's remaining templates (WrapperUnroll, ObservationWrapper2, lastType / NthType) in gym/wrappers.h, under
    the probe the reference's own were put through: every row equals the frozen reference output bit for bit, and both equal
    what the definitions give."""
    assert os.path.exists(MIRROR_WRAPPERS), "build tests/cpp first"
    g = np.load(os.path.join(G, "ref_unroll.npz"))
    rew, done = R.wrapper_script(int(g["script_seed"]), int(g["script_len"]))
    rows, t, flags, updater = R.run_unroll(rew, done, int(g["steps"]), driver=MIRROR_WRAPPERS)
    want = unroll_expected(rew, done, int(g["steps"]))
    assert int(g["done"].sum()) > 30
    for name in R.UNROLL.names:
        assert rows[name].tobytes() == np.ascontiguousarray(g[name]).tobytes() == want[name].tobytes(), name
    assert (t, flags) == (int(g["t"]), int(g["flags"])) == (int(g["steps"]), 4)     # no spaces on a bare scripted env; info() forwarded
    # Updater::notifyOne (updateable.h:26-29): only the FIRST subscriber is told; nothing is queued before a subscription
    assert updater == list(g["updater"]) == [23, 3, 0, 0]


@pytest.mark.skipif(not HAVE_DRIVER, reason="oracle/_ref not built (needs /root/reference; build container only)")
def test_cpp_wrapper_unroll_mirror_vs_live_reference_driver():
    for seed, L, steps in ((5, 17, 300), (6, 400, 900)):
        rew, done = R.wrapper_script(seed, L)
        a, b = R.run_unroll(rew, done, steps), R.run_unroll(rew, done, steps, driver=MIRROR_WRAPPERS)
        assert a[0].tobytes() == b[0].tobytes() == unroll_expected(rew, done, steps).tobytes() and a[1:] == b[1:] and a[3] == [23, 3, 0, 0]


# ---------------------------------------------------------------------------------------------------------------------
# Spaces, Env::seed and the in-tree drop-in: everything of the reference that compiles here without a stand-in header
# ---------------------------------------------------------------------------------------------------------------------
import json
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPACES_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_spaces_driver")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "ref_dropin_tu")
MIRROR_DEMO = os.path.join(ROOT, "tests", "cpp", "bin", "spaces_mirror_demo")


def _frozen_spaces():
    return json.load(open(os.path.join(G, "ref_spaces.json")))


@pytest.mark.skipif(not os.path.exists(SPACES_DRIVER), reason="reference not mounted: oracle/_ref not built (frozen file is checked below)")
def test_live_reference_spaces_probe_equals_frozen():
    r = subprocess.run([SPACES_DRIVER], capture_output=True, text=True, check=True)
    assert json.loads(r.stdout) == _frozen_spaces()


def test_cpp_mirror_spaces_equal_the_references():
    """gym/vec_env.h under the same probe (tests/cpp/spaces_probe.h) the reference's own spaces/*.cpp + env.h were run under."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "bin/spaces_mirror_demo"], capture_output=True, check=True)
    r = subprocess.run([MIRROR_DEMO], capture_output=True, text=True)
    assert r.returncode == 0 and "factories ok" in r.stderr, r.stderr[-500:]
    assert json.loads(r.stdout) == _frozen_spaces()


def test_python_spaces_match_the_references():
    from gym_amd import spaces as S
    g = _frozen_spaces()
    KCHAR, KINT, KFLOAT = 1, 3, 6                                                  # c10::ScalarType values
    code = {np.dtype(np.int8): KCHAR, np.dtype(np.int32): KINT, np.dtype(np.float32): KFLOAT}
    d = S.Discrete(5)
    assert d.size() == g["discrete5"]["size"] and code[d.type()] == g["discrete5"]["type"] and d.n == g["discrete5"]["n"]
    assert d.keys() == [g["discrete5"]["keys"]] and all(0 <= d.sample() < 5 for _ in range(200))
    assert g["discrete0_throws"] == g["discrete_neg_throws"] == "invalid_argument"
    for bad in (0, -3):
        with pytest.raises(ValueError):                                            # Python's counterpart of std::invalid_argument
            S.Discrete(bad)
    b = S.makeBoxSpace(2)
    assert b.size() == g["box_default_float"]["size"] and code[b.type()] == g["box_default_float"]["type"]
    assert [[np.float32(r.low), np.float32(r.high)] for r in b.getRange()] == [[np.float32(x) for x in rr] for rr in g["box_default_float"]["ranges"]]
    rr = g["box_vector_float"]["ranges"]
    v = S.makeBoxSpace([x[0] for x in rr], [x[1] for x in rr], 4)
    assert v.size() == g["box_vector_float"]["size"] and [[r.low, r.high] for r in v.getRange()] == rr
    c = v.clone()
    assert c.as_(S.Box) is c and c.as_(S.Discrete) is None and [[r.low, r.high] for r in c.getRange()] == g["box_clone"]["ranges"]
    assert g["box_vector_zero_dim_throws"] == g["box_vector_two_dim_throws"] == "runtime_error" and g["box_vector_empty_range_throws"] is None
    with pytest.raises(RuntimeError):
        S.Box([0.0], [1.0], shape=(2, 2))
    with pytest.raises(RuntimeError):
        S.Box([0.0], [1.0], shape=())
    img = S.makeBoxSpace(0, 255, [84, 84, 1], dtype=np.uint8)
    assert img.size() == g["box_image_u8"]["size"] and code[img.type()] == g["box_image_u8"]["type"] == KCHAR     # the uint8 -> kChar quirk
    assert [[r.low, r.high] for r in img.getRange()] == g["box_image_u8"]["ranges"]
    o, a = S.cartpole_spaces()
    assert [[np.float32(r.low), np.float32(r.high)] for r in o.getRange()][:3] == [[np.float32(x) for x in q] for q in rr[:3]]   # the probe used cartpole.cpp:10-16's values


def test_env_seed_streams_match_the_references(oracle):
    """Env::seed -> std::mt19937 (env.h:60-68,83) as the reference's own env.h runs it: raw outputs and the reset draws
    (uniform_real_distribution<float>(-0.05f, 0.05f), cartpole.cpp:21-26) for several seeds, against the oracle's generator,
    the oracle's reset tape and the library's host-drawn tape arithmetic."""
    g = _frozen_spaces()["env_seed"]
    assert list(oracle.mt19937_raw(5489, 2)) == g["unseeded_raw"]                  # never seeded == default-constructed mt19937
    for seed in (0, 1, 7, 12345, 4294967301):
        assert list(oracle.mt19937_raw(seed, 4)) == g[f"raw_{seed}"]
        tape = oracle.cartpole_reset_tape(seed, 1, 2)                              # env 0 of a vec env seeded `seed`: episodes 0 and 1
        assert np.array_equal(tape.reshape(-1), np.array(g[f"draws_{seed}"], np.float32))
    # sync_env.h:74-83: env i of a vec env seeded s is seeded s + i
    assert np.array_equal(oracle.cartpole_reset_tape(0, 8, 1)[0, 7], np.array(g["draws_7"][:4], np.float32))
    # the survey's probe of the compiled reference
    assert np.allclose(g["draws_0"][:4], [0.0048813522, 0.0092844591, 0.0215189345, 0.0344265737], rtol=0, atol=5e-11)


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="reference not mounted: oracle/_ref not built")
def test_dropin_compiles_against_the_reference_headers_and_describes_itself():
    """oracle/_ref/ref_dropin_tu: gym/hip_vec_env.h with GYMHIP_USE_REFERENCE_HEADERS over the reference's base.h,
    vec_env_wrapper.h, monitor.h and spaces, linked against libgymhip.so — that it exists means it compiled and linked."""
    r = subprocess.run([DROPIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = r.stdout.strip().splitlines()
    assert lines[1].split() == ["single", "obs_size=4", "act_n=2"]     # Monitor<TimeLimit<HipCartPoleEnv>>: the reference's own wrapper templates over the single-env face
    f = dict(kv.split("=") for kv in lines[0].split()[1:])
    assert f == {"nEnvs": "64", "obs_size": "4", "obs_type": "6", "ranges": "4", "act_n": "2", "act_size": "2", "seeds": "64", "first_seed": "7", "is_wrapper": "0"}


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref not built (it is built in the container that mounts the reference and travels with the tree)")
def test_dropin_built_against_the_reference_headers_runs_on_the_gpu():
    """The same binary on hardware: the reference's own VecEnvWrapper<false> forwarding reset() / step(actions) to HipVecEnv,
    which was compiled against the reference's base.h — shapes as sync_env.h:107-112 returns them, one stepWait through the wrapper."""
    r = subprocess.run([DROPIN, "run"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-1000:]
    run = [l for l in r.stdout.strip().splitlines() if l.startswith("run ")][-1]
    assert run.split() == ["run", "obs=[64,4]", "reward=[64,1]", "done=[64,1]", "steps=1"]


HIP_ENV_DEMO = os.path.join(os.path.dirname(G), "cpp", "bin", "hip_env_demo")


@pytest.mark.gpu
@pytest.mark.parametrize("templates", ["reference", "mirror"])
@pytest.mark.parametrize("kind,seed,max_steps", [("cartpole", 3, 500), ("cartpole", 11, 9), ("mountaincar", 0, 200), ("mountaincar", 0, 7), ("cartpole_narrowed", 5, 40)])
def test_reference_monitor_and_timelimit_templates_over_the_gpu_env(oracle, kind, seed, max_steps, templates):
    """The REFERENCE's own wrapper code (wrappers/monitor.h, time_limit.h, clip_reward.h — compiled from its headers) around the
    single-env face, stepping on the GPU: Monitor<TimeLimit<HipCartPoleEnv>> and configs[2]'s per-env stack
    Monitor<ClipRewardEnv<TimeLimit<HipMountainCarEnv>>>.  Every double of every CartPole observation (MountainCar: <= 1e-14),
    rewards, dones, info["TimeLimit.truncated"] and info["episode"] (r, l) equal the oracle's.  "mirror": the same loop
    (tests/cpp/wrapped_probe.h) with the stand-alone gym/wrappers.h templates instead (tests/cpp/hip_env_demo).
    "cartpole_narrowed": the CartPole stack behind an ObservationWrapper2 (common/wrapper.h:231-275) that returns vector<float>:
    every observation equals the oracle's double narrowed to float."""
    from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT
    narrowed = kind == "cartpole_narrowed"
    cart = kind.startswith("cartpole")
    f64 = (lambda row: [float(np.float32(x)) for x in row]) if narrowed else list
    steps, n_act = (450, 2) if cart else (900, 3)
    if templates == "reference":
        if not os.path.exists(DROPIN):
            pytest.skip("oracle/_ref not built (it is built in the container that mounts the reference and travels with the tree)")
        cmd = [DROPIN, "wrapped_narrowed" if narrowed else "wrapped" if cart else "wrapped_mc", str(seed), str(steps), str(max_steps)]
    else:
        assert os.path.exists(HIP_ENV_DEMO), "build tests/cpp first"
        cmd = [HIP_ENV_DEMO, kind, str(seed), str(steps), str(max_steps)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    lines = [l.split() for l in r.stdout.strip().splitlines() if l.startswith("w")]
    v = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, 1, F_TIME_LIMIT | F_MONITOR | (0 if cart else F_CLIP_REWARD), max_steps, threads=1)
    v.seed(seed)
    v.set_auto_reset(False)
    _, o64 = v.reset(want64=True)
    k = 0
    assert lines[k][0] == "wreset" and [float(x) for x in lines[k][1:]] == f64(o64[0]); k += 1
    episodes = truncations = 0
    for t in range(steps):
        a = ((t * 2654435761 & 0xFFFFFFFF) >> 7) % n_act
        o = v.step(np.array([a], np.int32), want64=True)
        tag, tt, aa, rew, done, has_tr, tr, has_ep, ep_r, ep_l, *obs = lines[k]; k += 1
        assert tag == "wstep" and int(tt) == t and int(aa) == a
        assert float(rew) == o["reward"][0] and int(done) == int(o["done"][0]), t
        if cart:
            assert [float(x) for x in obs] == f64(o["obs64"][0]), t
        else:
            assert np.abs(np.array([float(x) for x in obs]) - o["obs64"][0]).max() <= 1e-14, t
        assert int(has_ep) == int(o["done"][0]), t
        if o["done"][0]:
            episodes += 1
            assert float(ep_r) == o["ep_r"][0] and float(ep_l) == o["ep_l"][0], t
            assert int(has_tr) == 1 or o["ep_l"][0] < max_steps, t           # time_limit.h:30-34 sets the key only when the limit fired
            if int(has_tr):
                assert int(tr) == int(o["truncated"][0]), t
                truncations += int(tr)
            _, r64 = v.reset_index(0)
            assert lines[k][0] == "wreset" and [float(x) for x in lines[k][1:]] == f64(r64); k += 1
    assert k == len(lines) and episodes >= 4
    assert truncations > 0 if max_steps < 500 else True
