#!/usr/bin/env python3
"""Per-kernel HBM traffic of the secondary kernels from two rocprofv3 --pmc passes over tools/profile_extra.py.

    python tools/pmc_kernels.py <fetch_counter_collection.csv> <write_counter_collection.csv> [--out profiles/traffic.json]

Same units and gfx950 correction as tools/pmc_to_traffic.py (KiB; FETCH_SIZE doubled).  For every kernel in KERNELS the
mean over the second half of its dispatches is stored next to the algorithmic bytes of one launch of
tools/profile_extra.py's workload (DESIGN.md's per-unit figures x the units in that launch).
"""
import argparse
import csv
import json
import os

N3 = 1 << 24        # config 3 envs in profile_extra.py
N4 = 4096           # config 4 envs
# kernel-name needle -> (key, algorithmic read bytes, algorithmic write bytes) per launch
KERNELS = {
    "step_kernel<1, true, int, true,": ("config3_step_kernel_mountaincar_lockstep_norm", N3 * (8 + 8 + 4 + 4 + 4), N3 * (8 + 8 + 4 + 4 + 4 + 4)),
    "normalize_kernel<2,": ("config3_normalize_kernel", N3 * (8 + 4), N3 * (8 + 4)),
    "atari_preprocess_kernel": ("config4_atari_preprocess_rgb", N4 * 2 * 100800, N4 * 7056),
    "materialize_nchw_kernel<0,": ("config4_materialize_nchw_f32", N4 * 28224, N4 * 28224 * 4),
    "materialize_nchw_kernel<1, 1,": ("config4_materialize_nchw_bf16", N4 * 28224, N4 * 28224 * 2),
}


def per_kernel(path, counter):
    acc = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for needle in KERNELS:
                if needle in r["Kernel_Name"]:
                    acc.setdefault(needle, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = {}
    for needle, rows in acc.items():
        rows.sort()
        vals = [v for _, v in rows]
        steady = vals[len(vals) // 2:]
        out[needle] = (sum(steady) / len(steady), len(vals))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--out", default="profiles/traffic.json")
    a = ap.parse_args()
    fetch = per_kernel(a.fetch_csv, "FETCH_SIZE")
    write = per_kernel(a.write_csv, "WRITE_SIZE")
    allrec = json.load(open(a.out)) if os.path.exists(a.out) else {}
    for needle, (key, rd, wr) in KERNELS.items():
        if needle not in fetch or needle not in write:
            print("missing", needle)
            continue
        rb = fetch[needle][0] * 1024.0 * 2.0
        wb = write[needle][0] * 1024.0
        allrec[key] = {
            "hbm_bytes_per_launch": round(rb + wb), "read_bytes": round(rb), "write_bytes": round(wb),
            "algorithmic_bytes_per_launch": rd + wr, "algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr,
            "traffic_over_algorithmic": round((rb + wb) / (rd + wr), 4),
            "raw": {"FETCH_SIZE_steady_mean_KB": fetch[needle][0], "WRITE_SIZE_steady_mean_KB": write[needle][0], "dispatches": fetch[needle][1]},
        }
        print(key, json.dumps({k: allrec[key][k] for k in ("read_bytes", "write_bytes", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}))
    json.dump(allrec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
