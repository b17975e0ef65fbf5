# Diagnostic sweep: what bounds the CartPole step kernel?  (tools/stream_ceiling.hip = plain streams; GYMHIP_DIAG builds)
run() { # label, env assignments...
  label=$1; shift
  for m in lockstep f32; do
    env "$@" timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --mode $m --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', '$m', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])"
  done
}
V=$PWD/gym_amd/lib/variants
timeout -k 10 120 tools/bin/stream_ceiling || exit 1
run default
run diag1_streams_only GYMHIP_LIB=$V/libgymhip_diag1.so
run diag2_no_sincos GYMHIP_LIB=$V/libgymhip_diag2.so
run diag3_no_divide GYMHIP_LIB=$V/libgymhip_diag3.so
