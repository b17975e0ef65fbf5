# Cache-policy bits on the step kernel's streaming stores (variant builds: make -C gym_amd/csrc OUT=../lib/variants/libgymhip_<name>.so
# VARIANT="-DGYMHIP_ST_MODS='\"sc1 nt\"'"): wall ms per step and kernel ms, headline configuration.
V=$PWD/gym_amd/lib/variants
for name in default sc1nt sc0sc1nt sc1 sc0sc1; do
  lib=""; [ $name != default ] && lib=$V/libgymhip_$name.so
  GYMHIP_LIB=$lib timeout -k 10 120 python -u bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', 'value %.4e' % d['value'], 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'err', d['config']['error_flag'])" || exit 1
done
