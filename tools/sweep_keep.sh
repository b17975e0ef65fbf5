# Residency-plan sweep (tuning knobs GYMHIP_KEEP_SMALL / GYMHIP_KEEP_ENVS): tools/sweep_keep.sh <envs-per-gpu> "<small masks>" "<keep env counts>" [bench args]
n=$1; smalls=$2; keeps=$3; shift 3
for sm in $smalls; do for ke in $keeps; do
  GYMHIP_KEEP_SMALL=$sm GYMHIP_KEEP_ENVS=$ke timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 60 --envs-per-gpu $n "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('n=$n $* small=$sm keep_envs=$ke', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])" || exit 1
done; done
