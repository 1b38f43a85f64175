#!/bin/bash
# Round-2 step A on the GPU box: parity suite, the driver's 20-step line vs a 200-step line, the plain --gpus 2 form,
# rocprofv3 kernel stats and the WRITE_SIZE / FETCH_SIZE passes after the scratch removal.  Steps are chained with &&.
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err && \
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > $O/bench_200.json 2> $O/bench_200.err && \
CAVMD_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 3 --n-molecular 200000 > $O/bench_gpus2.json 2> $O/bench_gpus2.err && \
cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_1e6 -- python3 $R/bench.py --no-extras --no-cpu-baseline > $R/$O/prof_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/pmc_write_1e6.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch_1e6 -- python3 $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/pmc_fetch_1e6.log 2>&1
rc=$?
cd $R
python3 -c "
import json
for f in ('bench_20','bench_200','bench_gpus2'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels'])
    except Exception as e: print(f, 'ERR', e)
"
cat $O/prof_1e6/*/*kernel_stats.csv | cut -c1-150
exit $rc
