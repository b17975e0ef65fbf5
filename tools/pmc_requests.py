#!/usr/bin/env python3
"""L2 -> fabric read requests of the step kernel by size (one rocprofv3 pass with TCC_EA0_RDREQ_{sum,32B_sum,64B_sum,128B_sum}):
exact fetched bytes = 32 a + 64 b + 128 c, for the first dispatch (no env done yet) and the steady-state mean.
usage: pmc_requests.py <key> <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

key, path = sys.argv[1], sys.argv[2]
d = defaultdict(dict)
for r in csv.DictReader(open(path)):
    if "step_kernel" in r["Kernel_Name"]:
        d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
def summary(rows):
    n = len(rows)
    a = sum(r.get("TCC_EA0_RDREQ_32B_sum", 0.0) for r in rows) / n
    b = sum(r.get("TCC_EA0_RDREQ_64B_sum", 0.0) for r in rows) / n
    c = sum(r.get("TCC_EA0_RDREQ_128B_sum", 0.0) for r in rows) / n
    t = sum(r.get("TCC_EA0_RDREQ_sum", 0.0) for r in rows) / n
    return f"requests {t:.0f} (32B {a:.0f}, 64B {b:.0f}, 128B {c:.0f}) = {32 * a + 64 * b + 128 * c:.0f} B"
print(f"{key}: first dispatch {summary([d[ids[0]]])}; steady mean over {len(ids) - len(ids) // 2} dispatches {summary([d[i] for i in ids[len(ids) // 2:]])}")
