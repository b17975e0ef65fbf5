# Small-stream cache policy vs env count: tools/sweep_nts.sh <label>=<lib-or-empty> ...   (cartpole lockstep rng + mountaincar lockstep)
for n in 8388608 12582912 16777216 20971520 25165824 33554432 67108864; do
  for spec in "$@"; do
    label=${spec%%=*}; lib=${spec#*=}
    for env in cartpole mountaincar; do
      GYMHIP_LIB=$lib timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 60 --envs-per-gpu $n --env $env 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n $label $env', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])" || exit 1
    done
  done
done
