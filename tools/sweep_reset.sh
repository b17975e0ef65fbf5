# Which part of the CartPole step is the done path?  tape gather vs stateless counter RNG, and grid caps, per mode.
run() { # label, bench args..., env assignments via GRID
  label=$1; shift
  timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])"
}
for g in 1024 2048 4096; do
  export GYMHIP_GRID=$g
  run "g$g lockstep tape" --mode lockstep --reset-mode tape
  run "g$g lockstep rng " --mode lockstep --reset-mode rng
  run "g$g f32 tape" --mode f32 --reset-mode tape
  run "g$g f32 rng " --mode f32 --reset-mode rng
done
