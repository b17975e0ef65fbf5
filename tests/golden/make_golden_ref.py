#!/usr/bin/env python3
"""Freezes outputs of the REFERENCE's own code (oracle/_ref/ref_wrappers_driver: the reference's TimeLimit /
ClipRewardEnv / Monitor / RunningMeanStd headers compiled in place against libtorch) into tests/golden/ref_*.npz.

Run in the build container after `make -C oracle _ref`.  Inputs are regenerated from the seeds stored beside the
outputs; only data is committed, never reference source."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_wrappers_driver")
OUT = os.path.dirname(os.path.abspath(__file__))

REC = np.dtype([("reward", "<f4"), ("done", "u1"), ("has_trunc", "u1"), ("trunc", "u1"), ("has_ep", "u1"), ("ep_r", "<f4"), ("ep_l", "<f4")])


def wrapper_script(seed, L):
    rng = np.random.default_rng(seed)
    rew = rng.choice(np.array([0.0, -0.0, 1.0, -1.0, 0.25, -7.5, 3.0, 1e-3], np.float32), L).astype(np.float32)
    done = (rng.random(L) < 0.07).astype(np.uint8)
    return rew, done


def run_wrappers(rew, done, max_steps, steps, driver=None):
    """`driver`: another binary that speaks the same protocol (tests/cpp/bin/wrappers_mirror_demo: the stand-alone mirror)."""
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        with open(fin, "wb") as f:
            f.write(np.array([rew.size, max_steps, steps, 7], np.int64).tobytes()); f.write(rew.tobytes()); f.write(done.tobytes())
        subprocess.run([driver or DRIVER, "wrappers", fin, fout], check=True, capture_output=True)
        return np.fromfile(fout, REC)


UNROLL = np.dtype([("obs", "<f8"), ("reward", "<f4"), ("done", "u1"), ("obs2", "<f4"), ("reward2", "<f4"), ("done2", "u1")])


def run_unroll(rew, done, steps, driver=None):
    """The scripted env under WrapperUnroll<env, ClipRewardEnv stage, observation stage> and under an ObservationWrapper2 adapter
    (tests/cpp/wrappers_probe.h run_unroll) -> (rows[steps + 1] (row 0 = reset), inner env's step count, flag word)."""
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        with open(fin, "wb") as f:
            f.write(np.array([rew.size, 0, steps, 0], np.int64).tobytes()); f.write(rew.tobytes()); f.write(done.tobytes())
        subprocess.run([driver or DRIVER, "unroll", fin, fout], check=True, capture_output=True)
        raw = open(fout, "rb").read()
    n = (steps + 1) * UNROLL.itemsize
    t, flags, *updater = np.frombuffer(raw[n:], np.int64)
    return np.frombuffer(raw[:n], UNROLL), int(t), int(flags), [int(x) for x in updater]


def rms_batches(seed, nb, F):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(nb):
        n = int(rng.integers(1, 3000))
        out.append((rng.normal(size=(n, F)) * [1, 10, 0.01, 3][:F] + [0, 5, -2, 100 * (k % 3)][:F]).astype(np.float32))
    out.append(np.full((512, F), -0.5, np.float32))       # identical rows: batch variance exactly 0
    out.append(np.full((1, F), 3.0, np.float32))          # single-row batch
    return out


def run_rms(batches, F):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        with open(fin, "wb") as f:
            f.write(np.array([len(batches), F], np.int64).tobytes())
            for x in batches:
                f.write(np.array([x.shape[0]], np.int64).tobytes()); f.write(np.ascontiguousarray(x).tobytes())
        subprocess.run([DRIVER, "rms", fin, fout], check=True, capture_output=True)
        return np.fromfile(fout, np.float32).reshape(len(batches), 4 * F + 1)


def ckpt_batch(seed, F, n=37):
    return np.random.default_rng(seed).standard_normal((n, max(F, 1))).astype(np.float32)


def run_ckpt_save(x, F, out_pt):
    """The reference's torch::save(RunningMeanStd) after one update(x) -> file out_pt; returns (mean, var, count) it held."""
    with tempfile.TemporaryDirectory() as d:
        fin = os.path.join(d, "in")
        with open(fin, "wb") as f:
            f.write(np.array([F, x.shape[0]], np.int64).tobytes()); f.write(np.ascontiguousarray(x).tobytes())
        subprocess.run([DRIVER, "ckpt_save", fin, out_pt], check=True, capture_output=True)
    v = np.fromfile(out_pt + ".bin", np.float32)
    os.remove(out_pt + ".bin")
    W = max(F, 1)
    return v[:W], v[W:2 * W], float(v[2 * W])


def run_ckpt_load(in_pt):
    """The reference's torch::load(RunningMeanStd) of a file -> (mean, var, count) flattened."""
    with tempfile.TemporaryDirectory() as d:
        fout = os.path.join(d, "out")
        subprocess.run([DRIVER, "ckpt_load", in_pt, fout], check=True, capture_output=True)
        raw = open(fout, "rb").read()
    n = int(np.frombuffer(raw[:8], np.int64)[0])
    v = np.frombuffer(raw[8:], np.float32)
    return v[:n], v[n:2 * n], float(v[2 * n])


def run_csv(rew, done, max_steps, steps, out_csv, driver=None):
    """The reference's Monitor<TimeLimit<scripted env>> constructed with id "env7" and a file name -> its CSV log."""
    with tempfile.TemporaryDirectory() as d:
        fin = os.path.join(d, "in")
        with open(fin, "wb") as f:
            f.write(np.array([rew.size, max_steps, steps, 5], np.int64).tobytes()); f.write(rew.tobytes()); f.write(done.tobytes())
        subprocess.run([driver or DRIVER, "csv", fin, out_csv], check=True, capture_output=True)


def main():
    rew, done = wrapper_script(11, 997)
    run_csv(rew, done, 13, 1500, os.path.join(OUT, "ref_monitor.csv"))
    # checkpoints written by the reference's own save(): observation statistics (F = 4) and the scalar return statistics
    for name, F, seed in (("observation", 4, 21), ("returns", 0, 22)):
        pt = os.path.join(OUT, f"ref_rms_checkpoint_{name}.pt")
        mean, var, cnt = run_ckpt_save(ckpt_batch(seed, F), F, pt)
        np.savez(os.path.join(OUT, f"ref_rms_checkpoint_{name}.npz"), F=F, batch_seed=seed, mean=mean, var=var, count=np.float32(cnt))
    rew, done = wrapper_script(11, 997)
    rec = run_wrappers(rew, done, 13, 4000)
    np.savez_compressed(os.path.join(OUT, "ref_wrappers.npz"), script_seed=11, script_len=997, max_steps=13, steps=4000,
                        **{k: rec[k] for k in REC.names})
    rew, done = wrapper_script(12, 211)
    rows, t, flags, updater = run_unroll(rew, done, 1000)
    np.savez_compressed(os.path.join(OUT, "ref_unroll.npz"), script_seed=12, script_len=211, steps=1000, t=t, flags=flags, updater=np.array(updater, np.int64),
                        **{k: rows[k] for k in UNROLL.names})
    F = 4
    rows = run_rms(rms_batches(5, 30, F), F)
    np.savez_compressed(os.path.join(OUT, "ref_rms.npz"), batch_seed=5, n_batches=30, F=F, rows=rows)
    print("reference-built golden vectors written")


if __name__ == "__main__":
    main()


def freeze_spaces():
    """tests/golden/ref_spaces.json: tests/cpp/spaces_probe.h over the reference's own spaces/*.h, spaces/*.cpp and env.h
    (oracle/_ref/ref_spaces_driver).  The sampled values inside are range checks only, so the file is deterministic."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_spaces_driver")
    subprocess.run([drv, os.path.join(OUT, "ref_spaces.json")], check=True)


if __name__ == "__main__" and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_spaces_driver")):
    freeze_spaces()
