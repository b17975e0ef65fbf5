# A/B of library variants at small env counts (working set inside the 256 MiB Infinity Cache): tools/ab_small.sh <label>=<lib> ...
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for n in ${SIZES:-262144 1048576 2097152 4194304}; do
    GYMHIP_LIB=$lib timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 400 --warmup 50 --envs-per-gpu $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label n=$n', '%.3e'%d['value'], 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', r['kernel_ms'], r['achieved'])"
  done
done
