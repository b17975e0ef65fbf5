# step kernel compiled for more waves per SIMD (fewer VGPRs), with the grid sized to the resident set: tools/ab_waves.sh
run() { label=$1; lib=$2; grid=$3
  GYMHIP_GRID=$grid GYMHIP_LIB=$lib timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label grid=$grid', '%.3e'%d['value'], r['kernel_ms'], r['achieved'], r['frac'])"; }
run base "" 1024
run w5 tools/bin/libgymhip_w5.so 1024
run w5 tools/bin/libgymhip_w5.so 1280
run w6 tools/bin/libgymhip_w6.so 1024
run w6 tools/bin/libgymhip_w6.so 1536
run w8 tools/bin/libgymhip_w8.so 1024
run w8 tools/bin/libgymhip_w8.so 2048
