b() { timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_ms'], r['frac'], end='   ')" || exit 1; }
for env in cartpole mountaincar; do for n in 4194304 5242880 5767168 6291456 6815744 7340032; do
  echo -n "$env n=$n default: "; b --envs-per-gpu $n --env $env; echo -n " NT+plan: "; GYMHIP_NT_MB=0 b --envs-per-gpu $n --env $env; echo
done; done
