# Round-2 residency-plan measurements (one gpurun call): budget cliff, default plan in every mode, the NT threshold, large env counts.
b() { timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 60 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])" || exit 1; }
echo "== cliff at 16.7M, all small kept, obs/state prefix =="
for ke in 0 1048576 1572864 2097152 2621440 3145728; do echo -n "keep_envs=$ke  "; GYMHIP_KEEP_SMALL=15 GYMHIP_KEEP_ENVS=$ke b; done
echo "== default plan, 16.7M =="
for m in lockstep f32; do for r in tape rng; do for env in cartpole mountaincar; do echo -n "$env $m $r  "; b --mode $m --reset-mode $r --env $env; done; done; done
echo "== NT threshold: all-temporal (default below 336 MiB) vs NT + plan =="
for n in 3145728 4194304 5242880 6291456 7340032; do for env in cartpole mountaincar; do
  echo -n "n=$n $env default  "; b --envs-per-gpu $n --env $env
  echo -n "n=$n $env NT+plan  "; GYMHIP_NT_MB=0 b --envs-per-gpu $n --env $env
done; done
echo "== large env counts: default plan vs no plan =="
for n in 25165824 33554432 67108864; do for env in cartpole mountaincar; do
  echo -n "n=$n $env plan  "; b --envs-per-gpu $n --env $env
  echo -n "n=$n $env none  "; GYMHIP_KEEP_SMALL=0 GYMHIP_KEEP_ENVS=0 b --envs-per-gpu $n --env $env
done; done
