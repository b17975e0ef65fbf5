#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per run) into HBM bytes per step_kernel launch.

    python tools/pmc_to_traffic.py <key> <fetch_counter_collection.csv> <write_counter_collection.csv> \
        <read_bytes_per_env> <write_bytes_per_env> <envs_per_launch>  [--out profiles/traffic.json] [--copy-prefix profiles/r01_pmc]

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters count KiB; on gfx950
FETCH_SIZE reports half of the streamed read bytes, so it is doubled.  The correction is CHECKED here, not assumed: the
first step_kernel dispatches after a reset finish no episode (no CartPole episode is shorter than 8 steps), so they
move exactly the algorithmic bytes; the ratio counter/known of those dispatches is stored as `calibration`.
The steady-state figure is the mean over the second half of the dispatches (episodes ending every step).
"""
import argparse
import csv
import json
import os


def kernel_rows(path, needle="step_kernel"):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if needle in r["Kernel_Name"]:
                rows.append(r)
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("key")
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("read_bytes_per_env", type=float)
    ap.add_argument("write_bytes_per_env", type=float)
    ap.add_argument("envs", type=int)
    ap.add_argument("--out", default="profiles/traffic.json")
    ap.add_argument("--copy-prefix", default=None, help="also write the step_kernel rows of both passes to <prefix>_{fetch,write}_<key>.csv")
    a = ap.parse_args()

    out = {}
    raw = {}
    for name, path, per_env, scale in (("FETCH_SIZE", a.fetch_csv, a.read_bytes_per_env, 2.0), ("WRITE_SIZE", a.write_csv, a.write_bytes_per_env, 1.0)):
        rows = [r for r in kernel_rows(path) if r["Counter_Name"] == name]
        vals = [float(r["Counter_Value"]) for r in rows]
        assert len(vals) >= 8, f"{path}: only {len(vals)} step_kernel dispatches with {name}"
        steady = vals[len(vals) // 2:]
        known = per_env * a.envs
        first = vals[0] * 1024.0 * scale
        raw[name] = {"first_no_done_KB": vals[0], "steady_mean_KB": sum(steady) / len(steady), "dispatches": len(vals)}
        out[name] = {"bytes": sum(steady) / len(steady) * 1024.0 * scale, "calibration": first / known}
        if a.copy_prefix:
            dst = f"{a.copy_prefix}_{'fetch' if name == 'FETCH_SIZE' else 'write'}_{a.key}.csv"
            with open(dst, "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows)

    rec = {
        "hbm_bytes_per_launch": round(out["FETCH_SIZE"]["bytes"] + out["WRITE_SIZE"]["bytes"]),
        "read_bytes": round(out["FETCH_SIZE"]["bytes"]),
        "write_bytes": round(out["WRITE_SIZE"]["bytes"]),
        "algorithmic_bytes_per_launch": round((a.read_bytes_per_env + a.write_bytes_per_env) * a.envs),
        "envs_per_launch": a.envs,
        "calibration": {
            "note": f"dispatches before any env is done move exactly (read,write) = ({a.read_bytes_per_env:g},{a.write_bytes_per_env:g}) B/env; ratio counter/known",
            "fetch_x2_over_known": round(out["FETCH_SIZE"]["calibration"], 4),
            "write_over_known": round(out["WRITE_SIZE"]["calibration"], 4),
        },
        "raw": raw,
    }
    allrec = {}
    if os.path.exists(a.out):
        allrec = json.load(open(a.out))
    allrec[a.key] = rec
    json.dump(allrec, open(a.out, "w"), indent=1)
    print(a.key, json.dumps({k: rec[k] for k in ("hbm_bytes_per_launch", "read_bytes", "write_bytes", "algorithmic_bytes_per_launch", "calibration")}))


if __name__ == "__main__":
    main()
