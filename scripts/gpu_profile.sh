#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, bench, rocprofv3 kernel stats and the FETCH_SIZE / WRITE_SIZE PMC passes.
# Outputs land in gpurun_out/; the summaries that are judged are copied into profiles/ afterwards.
set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu4.log 2>&1; tail -2 gpurun_out/pytest_gpu4.log
timeout -k 10 400 python bench.py > gpurun_out/bench3.json 2> gpurun_out/bench3.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof3 $R/gpurun_out/pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -- python3 $R/bench.py --no-extras --no-cpu-baseline > $R/gpurun_out/prof3.log 2>&1; echo "prof rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/pmc_fetch_1e6.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/pmc_write_1e6.log 2>&1; echo "pmc2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $R/gpurun_out/pmc_fetch_1e7.log 2>&1; echo "pmc3 rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $R/gpurun_out/pmc_write_1e7.log 2>&1; echo "pmc4 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_1e7 -- python3 $R/bench.py --n-molecular 10000000 --frames 2 --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/prof3_1e7.log 2>&1; echo "prof1e7 rc=$?"
cd $R
ls gpurun_out/pmc_fetch_1e6/*/ | head
cat gpurun_out/prof3/*/*kernel_stats.csv | cut -c1-120
cat gpurun_out/prof3_1e7/*/*kernel_stats.csv | cut -c1-120
