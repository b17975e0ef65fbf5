# Does a grid whose stride is not a power of two help (every block of a 1024-block grid-stride loop always hits the same
# HBM channel set)?  Microbench and the real lockstep/f32 kernels.
timeout -k 10 120 tools/bin/stream_ceiling 1000 1023 1280 1536 1792 3000 || exit 1
for g in 768 1000 1023 1024 1280 1536 1792 2000 3000; do
  for m in lockstep f32; do
    GYMHIP_GRID=$g timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 --mode $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('grid $g $m', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])"
  done
done
