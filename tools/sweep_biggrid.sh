# CartPole lockstep (the 84-B kernel): the shipped launch shape (<= 8 envs per thread, cap 2048) against 2 envs per thread, cap 16384, per env count.
b() { timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_ms'], r['frac'], end='   ')" || exit 1; }
for n in 262144 1048576 4194304 8388608 16777216 33554432; do
  cur=$(( n / 4096 )); [ $cur -lt 256 ] && cur=256; [ $cur -gt 2048 ] && cur=2048
  new=$(( n / 1024 )); [ $new -lt 256 ] && new=256; [ $new -gt 16384 ] && new=16384
  echo -n "n=$n  shipped grid $cur: "; GYMHIP_GRID=$cur b --envs-per-gpu $n; echo -n " | grid $new: "; GYMHIP_GRID=$new b --envs-per-gpu $n
  echo -n " | again $cur: "; GYMHIP_GRID=$cur b --envs-per-gpu $n; echo -n " | again $new: "; GYMHIP_GRID=$new b --envs-per-gpu $n; echo
done
