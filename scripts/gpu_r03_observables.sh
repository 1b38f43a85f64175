#!/bin/bash
# rocprofv3 kernel stats of the kernels next to the force path (rows f2-f4) at N = 1e6, plus the un-profiled wall times.
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03obs${1:-}
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/profile_observables.py > $R/$O/wall.json 2> $R/$O/wall.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/scripts/profile_observables.py > $R/$O/prof.log 2>&1
rc=$?
cat $R/$O/wall.json
cat $R/$O/prof/*/*kernel_stats.csv | cut -c1-220
exit $rc
