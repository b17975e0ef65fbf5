# GYMHIP_GRID sweep after the residency plans (16.7 M envs): frac of 8 TB/s per variant and grid
b() { timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 60 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_ms'], r['frac'], end='   ')" || exit 1; }
for cfg in "--env cartpole --mode lockstep" "--env cartpole --mode f32" "--env mountaincar --mode lockstep" "--env mountaincar --mode f32" "--env cartpole --mode lockstep --reset-mode tape" "--env cartpole --mode f32 --reset-mode tape"; do
  echo "== $cfg"
  for g in 256 512 768 1024 1536 2048 4096; do echo -n "grid $g: "; GYMHIP_GRID=$g b $cfg; done; echo
done
