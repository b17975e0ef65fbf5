# LDS-side SQ counters of the RGB Atari preprocess kernel (run on the GPU box from the repo root): how busy the LDS pipe is, how much
# of that is bank conflicts, how long waves wait for it.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r3
mkdir -p $O
cd /tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/prof_atari_lds -- python3 $R/tools/profile_atari_rgb.py > $O/prof_atari_lds.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $O/prof_atari_act -- python3 $R/tools/profile_atari_rgb.py > $O/prof_atari_act.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("prof_atari_lds", "prof_atari_act"):
    f = glob.glob(f"gpurun_out/r3/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "atari_preprocess" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        s = v[len(v) // 2:]
        print(d, k, len(v), f"{sum(s) / len(s):,.0f}")
PY
