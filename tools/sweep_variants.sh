# Tuning sweep: GYMHIP_LIB picks a variant build (gym_amd/csrc: make OUT=... VARIANT="-D..."), GYMHIP_GRID the grid cap.
run() { # label, env assignments...
  label=$1; shift
  for m in lockstep f32; do
    env "$@" timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --mode $m --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', '$m', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])"
  done
}
V=$PWD/gym_amd/lib/variants
for g in 512 1024 2048; do
  run nt_g$g GYMHIP_LIB=$V/libgymhip_nt.so GYMHIP_GRID=$g
  run ilp2nt_g$g GYMHIP_LIB=$V/libgymhip_ilp2nt.so GYMHIP_GRID=$g
  run ilp4nt_g$g GYMHIP_LIB=$V/libgymhip_ilp4nt.so GYMHIP_GRID=$g
done
