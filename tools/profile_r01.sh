# Round-1 profile recipe (run on the GPU box from the repo root): bench, kernel trace, and one PMC pass per counter.
set -x
export TMPDIR=/tmp
R=$PWD
timeout -k 10 400 python -u bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err || exit 1
cd /tmp
B="python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- $B > $R/gpurun_out/prof_trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- $B > $R/gpurun_out/prof_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- $B > $R/gpurun_out/prof_write.log 2>&1 || exit 1
B2="python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra --mode f32"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_f32 -- $B2 > $R/gpurun_out/prof_fetch_f32.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_f32 -- $B2 > $R/gpurun_out/prof_write_f32.log 2>&1 || exit 1
echo PROFILE_OK
