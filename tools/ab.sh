# A/B timing of library variants: tools/ab.sh <label>=<path-or-empty> ...   (each: lockstep+f32, tape and rng resets)
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for m in lockstep f32; do for r in tape rng; do
    GYMHIP_LIB=$lib timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 --mode $m --reset-mode $r 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label $m $r', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])"
  done; done
done
