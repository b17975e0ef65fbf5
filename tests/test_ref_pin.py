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
