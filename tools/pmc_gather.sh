# Calibrates the L2's memory-side read counters for the scattered 16-B reset-row gather (tools/gather_probe.hip).
# Pass 1 per layout: FETCH_SIZE (what profiles/traffic.json is built from).  Pass 2: the request counters by size
# (TCC_EA0_RDREQ_{32B,64B,128B}), which give the fetched bytes exactly: 32 a + 64 b + 128 c.
# Run on the GPU box from the repo root.
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/r3
cd /tmp
for m in dense tape mt wave; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3/prof_gather_$m -- $R/tools/bin/gather_probe $m 16777216 6 > $R/gpurun_out/r3/gather_$m.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $R/gpurun_out/r3/prof_gather_req_$m -- $R/tools/bin/gather_probe $m 16777216 6 > $R/gpurun_out/r3/gather_req_$m.log 2>&1 || exit 1
  grep rows_per_launch $R/gpurun_out/r3/gather_$m.log
  python3 - "$(find $R/gpurun_out/r3/prof_gather_$m -name '*counter_collection.csv' | head -1)" "$(find $R/gpurun_out/r3/prof_gather_req_$m -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys
from collections import defaultdict
def per_dispatch(path):
    d = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if "gather_kernel" in r.get("Kernel_Name", ""):
            d[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    return [d[k] for k in sorted(d, key=int)]
f = per_dispatch(sys.argv[1]); q = per_dispatch(sys.argv[2])
print("  FETCH_SIZE KB per dispatch (last):", round(f[-1]["FETCH_SIZE"], 1))
last = q[-1]
a, b, c, tot = (last.get(k, 0.0) for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum"))
print(f"  requests (last dispatch): total {tot:.0f}, 32B {a:.0f}, 64B {b:.0f}, 128B {c:.0f} -> {32*a + 64*b + 128*c:.0f} bytes")
PY
done
