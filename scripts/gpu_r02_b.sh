#!/bin/bash
# Round-2 step B: first runs of the single-launch kernel: its parity tests, then the A/B against two launches.
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_launch or graph or tunables" > $O/pytest_single.log 2>&1; rc=$?; tail -15 $O/pytest_single.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/ab_tunable.py persistent 0 1 30000 100000 300000 1000000 2000000 4000000 > $O/ab_persistent.txt 2>&1; rc=$?; cat $O/ab_persistent.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200.json 2> $O/bench_200.err; rc=$?
python3 -c "
import json
d=json.loads(open('$O/bench_200.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels']); print({k:(v.get('evals_per_s'),v.get('us_per_eval')) for k,v in d.get('extras',{}).items()})
"
exit $rc
