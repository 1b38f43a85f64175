#!/bin/bash
# Round-3 profiles: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of bench.py's default workload (single-launch
# kernel, N = 1e6 + 1) and of the 1e7 two-launch path; every step chained with &&.  Program directly after `--`.
set -x
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03p
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_1e6 -- python3 $R/bench.py --no-extras --no-cpu-baseline > $R/$O/prof_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/pmc_write_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/pmc_fetch_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/prof_1e7.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $R/$O/pmc_write_1e7.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $R/$O/pmc_fetch_1e7.log 2>&1
rc=$?
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err && python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_20steps.json 2> $O/bench_20steps.err
tail -1 $O/prof_1e6.log | cut -c1-400
cat $O/prof_1e6/*/*kernel_stats.csv | cut -c1-200
cat $O/prof_1e7/*/*kernel_stats.csv | cut -c1-200
cut -c1-300 $O/bench_20steps.json
exit $rc
