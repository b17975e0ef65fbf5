# VALU occupancy of the headline step kernel (run on the GPU box from the repo root): one PMC pass of SQ counters.
export TMPDIR=/tmp
R=$PWD
cd /tmp
export GYM_BENCH_PREWARM_S=0
B="python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra"
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_valu -- $B > $R/gpurun_out/prof_valu.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_valu/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "step_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    s = v[len(v) // 2:]
    print(k, len(v), sum(s) / len(s))
PY
