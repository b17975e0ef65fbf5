# Round-3 profile recipe (run on the GPU box from the repo root): bench, kernel traces, one PMC pass per counter for the three
# CartPole reset sources, the secondary kernels and config 3 at the sizes BASELINE configs[2] specifies.  Outputs land in
# gpurun_out/r3p/; copy the r03_* files and traffic.json into profiles/.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r3p
mkdir -p $O
timeout -k 10 500 python -u bench.py > $O/r03_bench_1gpu.json 2> $O/bench.err || exit 1
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/r03_bench_driver_style.json 2>> $O/bench.err || exit 1
GYM_BENCH_FORCE_DIST=1 timeout -k 10 400 python -u bench.py --steps 20 --warmup 5 > $O/r03_bench_forced_1rank_rccl.json 2>> $O/bench.err || exit 1
cd /tmp
# kernel traces: the default bench command's headline leg, and the same with per-env device mt19937 resets over a whole regeneration period
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $R/bench.py --no-cpu-baseline --no-extra > $O/prof_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace_mt -- python3 $R/bench.py --no-cpu-baseline --no-extra --reset-mode mt19937 --steps 4000 > $O/prof_trace_mt.log 2>&1 || exit 1
# PMC passes: no pre-warm, so that the first profiled dispatches are the first steps after the reset (no env done yet: known bytes, used as calibration)
export GYM_BENCH_PREWARM_S=0
for v in rng tape mt19937; do
  B="python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra --reset-mode $v"
  timeout -s KILL 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch_$v -- $B > $O/prof_fetch_$v.log 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write_$v -- $B > $O/prof_write_$v.log 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/prof_req_$v -- $B > $O/prof_req_$v.log 2>&1 || exit 1
done
unset GYM_BENCH_PREWARM_S
# secondary kernels: normaliser pair (config 3, 16.7 M envs) and the Atari preprocess / materialise kernels (config 4)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace_extra -- python3 $R/tools/profile_extra.py > $O/prof_trace_extra.log 2>&1 || exit 1
for n in 131072 1048576; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_$n -- python3 $R/tools/profile_config3.py $n 2000 > $O/prof_c3_$n.log 2>&1 || exit 1
done
cd $R
f() { find $O/$1 -name "*$2" | head -1; }
for v in rng tape mt19937; do
  key=cartpole_lockstep; [ $v = rng ] || key=cartpole_lockstep_$v
  python3 tools/pmc_to_traffic.py $key "$(f prof_fetch_$v counter_collection.csv)" "$(f prof_write_$v counter_collection.csv)" 40 44 16777216 --out $O/traffic.json --copy-prefix $O/r03_pmc || exit 1
  python3 tools/pmc_requests.py $key "$(f prof_req_$v counter_collection.csv)" >> $O/r03_read_requests_by_size.txt || exit 1
done
cp "$(f prof_trace kernel_stats.csv)" $O/r03_kernel_stats_cartpole_lockstep.csv
cp "$(f prof_trace_mt kernel_stats.csv)" $O/r03_kernel_stats_cartpole_lockstep_mt19937.csv
cp "$(f prof_trace_extra kernel_stats.csv)" $O/r03_kernel_stats_normalizer_atari.csv
cp "$(f prof_c3_131072 kernel_stats.csv)" $O/r03_kernel_stats_config3_131072.csv
cp "$(f prof_c3_1048576 kernel_stats.csv)" $O/r03_kernel_stats_config3_1M.csv
echo PROFILE_OK
