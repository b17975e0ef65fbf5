# Residency plan vs env count: all streams non-temporal (GYMHIP_KEEP_SMALL=0) against the plan plan_residency picks, CartPole and MountainCar lockstep.
for n in 8388608 12582912 16777216 20971520 25165824 33554432 67108864; do
  for spec in none=0 plan=; do
    label=${spec%%=*}; k=${spec#*=}
    for env in cartpole mountaincar; do
      ( [ -n "$k" ] && export GYMHIP_KEEP_SMALL=$k; timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 60 --envs-per-gpu $n --env $env 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n $label $env', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])" || exit 1
    done
  done
done
