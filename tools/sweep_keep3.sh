# Residency plans of the step kernel (GYMHIP_KEEP_SMALL = one of 15 / 11 / 1 / 0; default = plan_residency's choice), 16.7 M envs per GPU.
b() { timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 60 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])" || exit 1; }
for env in cartpole mountaincar; do for cfg in "--mode lockstep --reset-mode rng" "--mode f32 --reset-mode rng" "--mode f32 --reset-mode tape" "--mode lockstep --reset-mode tape"; do
  [ $env = mountaincar ] && [ "${cfg##* }" = tape ] && continue
  echo "== $env $cfg =="
  echo -n "default plan  "; b --env $env $cfg
  for k in 15 11 1 0; do echo -n "plan $k  "; GYMHIP_KEEP_SMALL=$k b --env $env $cfg; done
done; done
