#!/bin/bash
# the same kernels at N = 1e7: the latency-bound tails (ticket, fold, publication) are constant, the streaming scales
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03obs_1e7
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/scripts/profile_observables.py 10000000 30 > $R/$O/prof.log 2>&1
rc=$?
tail -1 $R/$O/prof.log | cut -c1-900
cat $R/$O/prof/*/*kernel_stats.csv | cut -c1-200
exit $rc
