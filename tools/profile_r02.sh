# Round-2 profile recipe (run on the GPU box from the repo root): bench, kernel trace, one PMC pass per counter, the
# secondary kernels and config 3 at the sizes BASELINE configs[2] specifies.  Outputs land in gpurun_out/; copy the r02_* files
# and traffic.json into profiles/.
set -x
export TMPDIR=/tmp
R=$PWD
timeout -k 10 500 python -u bench.py > gpurun_out/r02_bench_1gpu.json 2> gpurun_out/bench_r02.err || exit 1
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r02_bench_driver_style.json 2>> gpurun_out/bench_r02.err || exit 1
cd /tmp
# kernel trace of the default bench command's headline leg in steady state (the pre-warm's launches are part of the trace: same kernel, same state)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/prof_trace.log 2>&1 || exit 1
# PMC passes: no pre-warm, so that the first profiled dispatches are the first steps after the reset (no env done yet: known bytes, used as calibration)
export GYM_BENCH_PREWARM_S=0
B="python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- $B > $R/gpurun_out/prof_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- $B > $R/gpurun_out/prof_write.log 2>&1 || exit 1
BT="$B --reset-mode tape"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_tape -- $BT > $R/gpurun_out/prof_fetch_tape.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_tape -- $BT > $R/gpurun_out/prof_write_tape.log 2>&1 || exit 1
BM="$B --env mountaincar"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_mc -- $BM > $R/gpurun_out/prof_fetch_mc.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_mc -- $BM > $R/gpurun_out/prof_write_mc.log 2>&1 || exit 1
B2="python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra --mode f32"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_f32 -- $B2 > $R/gpurun_out/prof_fetch_f32.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_f32 -- $B2 > $R/gpurun_out/prof_write_f32.log 2>&1 || exit 1
# secondary kernels: normaliser pair (config 3, 16.7 M envs) and the Atari preprocess / materialise kernels (config 4)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace_extra -- python3 $R/tools/profile_extra.py > $R/gpurun_out/prof_trace_extra.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_extra -- python3 $R/tools/profile_extra.py > $R/gpurun_out/prof_fetch_extra.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_extra -- python3 $R/tools/profile_extra.py > $R/gpurun_out/prof_write_extra.log 2>&1 || exit 1
# config 3 at its per-GPU shard (131,072 envs) and at 1,048,576 envs: wall time per step + per-kernel durations
for n in 131072 1048576; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3_$n -- python3 $R/tools/profile_config3.py $n 2000 > $R/gpurun_out/prof_c3_$n.log 2>&1 || exit 1
  timeout -k 10 100 python3 $R/tools/profile_config3.py $n 2000 | tail -1 >> $R/gpurun_out/r02_config3_small.txt || exit 1
done
cd $R
f() { find gpurun_out/$1 -name "*$2" | head -1; }
python3 tools/pmc_to_traffic.py cartpole_lockstep "$(f prof_fetch counter_collection.csv)" "$(f prof_write counter_collection.csv)" 40 44 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r02_pmc || exit 1
python3 tools/pmc_to_traffic.py cartpole_lockstep_tape "$(f prof_fetch_tape counter_collection.csv)" "$(f prof_write_tape counter_collection.csv)" 40 44 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r02_pmc || exit 1
python3 tools/pmc_to_traffic.py mountaincar_lockstep "$(f prof_fetch_mc counter_collection.csv)" "$(f prof_write_mc counter_collection.csv)" 24 28 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r02_pmc || exit 1
python3 tools/pmc_to_traffic.py cartpole_f32 "$(f prof_fetch_f32 counter_collection.csv)" "$(f prof_write_f32 counter_collection.csv)" 24 28 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r02_pmc || exit 1
python3 tools/pmc_kernels.py "$(f prof_fetch_extra counter_collection.csv)" "$(f prof_write_extra counter_collection.csv)" --out gpurun_out/traffic.json || exit 1
cp "$(f prof_trace kernel_stats.csv)" gpurun_out/r02_kernel_stats_cartpole_lockstep.csv
cp "$(f prof_trace_extra kernel_stats.csv)" gpurun_out/r02_kernel_stats_normalizer_atari.csv
cp "$(f prof_c3_131072 kernel_stats.csv)" gpurun_out/r02_kernel_stats_config3_131072.csv
cp "$(f prof_c3_1048576 kernel_stats.csv)" gpurun_out/r02_kernel_stats_config3_1M.csv
echo PROFILE_OK
