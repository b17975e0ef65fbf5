# GYMHIP_GRID sweep of the reset-tape variants (the gather's latency wants more waves in flight than the counter-RNG form)
for m in f32 lockstep; do for g in 512 1024 2048 4096; do
  GYMHIP_GRID=$g timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 --mode $m --reset-mode tape 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cartpole $m tape grid $g', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])" || exit 1
done; done
