# config 3 (step + normalise with the folded reduction) and the headline kernel across block sizes / grids
for spec in "256:1024:" "256:512:" "512:512:_b512" "512:256:_b512" "1024:256:_b1024" "1024:512:_b1024"; do IFS=: read b g v <<< "$spec"
  L=$PWD/gym_amd/lib/libgymhip$v.so
  for n in 131072 1048576 16777216; do echo -n "block=$b grid=$g "; GYMHIP_LIB=$L GYMHIP_GRID=$g timeout -k 10 100 python tools/profile_config3.py $n 1000 | tail -1 || exit 1; done
  GYMHIP_LIB=$L GYMHIP_GRID=$g timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('block $b grid $g headline', r['kernel_ms'], r['achieved'])" || exit 1
done
