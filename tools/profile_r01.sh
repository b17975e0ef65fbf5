# Round-1 profile recipe (run on the GPU box from the repo root): bench, kernel trace, and one PMC pass per counter.
# Outputs land in gpurun_out/; tools/pmc_to_traffic.py then writes profiles/traffic.json and the trimmed CSVs.
set -x
export TMPDIR=/tmp
R=$PWD
timeout -k 10 400 python -u bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err || exit 1
cd /tmp
B="python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- $B > $R/gpurun_out/prof_trace.log 2>&1 || exit 1
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
# secondary kernels: normaliser pair (config 3) and the Atari preprocess / materialise kernels (config 4)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace_extra -- python3 $R/tools/profile_extra.py > $R/gpurun_out/prof_trace_extra.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_extra -- python3 $R/tools/profile_extra.py > $R/gpurun_out/prof_fetch_extra.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_extra -- python3 $R/tools/profile_extra.py > $R/gpurun_out/prof_write_extra.log 2>&1 || exit 1
cd $R
f() { find gpurun_out/$1 -name "*$2" | head -1; }
python3 tools/pmc_to_traffic.py cartpole_lockstep "$(f prof_fetch counter_collection.csv)" "$(f prof_write counter_collection.csv)" 40 44 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r01_pmc || exit 1
python3 tools/pmc_to_traffic.py cartpole_lockstep_tape "$(f prof_fetch_tape counter_collection.csv)" "$(f prof_write_tape counter_collection.csv)" 40 44 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r01_pmc || exit 1
python3 tools/pmc_to_traffic.py mountaincar_lockstep "$(f prof_fetch_mc counter_collection.csv)" "$(f prof_write_mc counter_collection.csv)" 24 28 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r01_pmc || exit 1
python3 tools/pmc_to_traffic.py cartpole_f32 "$(f prof_fetch_f32 counter_collection.csv)" "$(f prof_write_f32 counter_collection.csv)" 24 28 16777216 --out gpurun_out/traffic.json --copy-prefix gpurun_out/r01_pmc || exit 1
python3 tools/pmc_kernels.py "$(f prof_fetch_extra counter_collection.csv)" "$(f prof_write_extra counter_collection.csv)" --out gpurun_out/traffic.json || exit 1
cp "$(f prof_trace kernel_stats.csv)" gpurun_out/r01_kernel_stats_cartpole_lockstep.csv
cp "$(f prof_trace_extra kernel_stats.csv)" gpurun_out/r01_kernel_stats_normalizer_atari.csv
echo PROFILE_OK
