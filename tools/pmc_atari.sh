# SQ counters of the RGB Atari preprocess kernel (run on the GPU box from the repo root)
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout -k 10 100 python3 $R/tools/profile_atari_rgb.py || exit 1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_atari_sq -- python3 $R/tools/profile_atari_rgb.py > $R/gpurun_out/prof_atari_sq.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/prof_atari_mem -- python3 $R/tools/profile_atari_rgb.py > $R/gpurun_out/prof_atari_mem.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("prof_atari_sq", "prof_atari_mem"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "atari_preprocess" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        s = v[len(v) // 2:]
        print(d, k, len(v), sum(s) / len(s))
PY
