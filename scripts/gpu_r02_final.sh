#!/bin/bash
# Round-2 final evidence batch: full GPU suite, bench (default and the driver's 20-step form), rocprofv3 stats + PMC of the
# final code, the final microbench tables and the hand-off latency probe.  Steps chained with &&.
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02z
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err && \
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err && \
(cd cav-hoomd_amd/csrc && for n in "100001 64" "300001 22" "1000001 7" "4000001 2"; do set -- $n; timeout -k 10 200 ./microbench_persistent $1 $2 9 20 > $R/$O/mbp_final_$1.txt 2>&1 || exit 1; done) && \
timeout -k 10 60 scripts/dev/pingpong 2000 > $O/pingpong.txt 2>&1 && \
(cd cav-hoomd_amd/csrc && for n in 1000001 100001 3001 2049; do timeout -k 10 60 ./microbench_persistent_late $n 1 1 1 | tail -1 || exit 1; done > $R/$O/fault_late.txt 2>&1 && for n in 1000001 3001; do timeout -k 10 60 ./microbench_persistent_fault $n 1 1 1 | tail -1 || exit 1; done > $R/$O/fault_silent.txt 2>&1) && \
cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_1e6 -- python3 $R/bench.py --no-extras --no-cpu-baseline > $R/$O/prof_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/pmc_write_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/pmc_fetch_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/prof_1e7.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $R/$O/pmc_write_1e7.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $R/$O/pmc_fetch_1e7.log 2>&1
rc=$?
cd $R
for f in $O/prof_1e6 $O/prof_1e7; do awk -F'",' '{print substr($1,1,50), $0}' $f/*/*kernel_stats.csv | awk '{print $1,$2,$3, $NF}' | cut -c1-160; done
grep -h "two launches\|single launch," $O/mbp_final_*.txt
cat $O/pingpong.txt
exit $rc
