#!/bin/bash
# rocprofv3 kernel stats of the observable kernels (row f3): 1e6 particles x 50 wavevectors.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_obs
cat > /tmp/obs_run.py <<'PY'
import sys, os
R = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import torch, cavitymd
from cavitymd import synthetic, observables
cfg = synthetic.config3(seed=1)
pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
f = observables.DensityField(pd, observables.generate_fibonacci_sphere(50))
for _ in range(30):
    f.enqueue()
torch.cuda.synchronize()
print(abs(f.result()).max())
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_obs -- python3 /tmp/obs_run.py > $R/gpurun_out/prof_obs.log 2>&1; echo "rc=$?"
cat $R/gpurun_out/prof_obs/*/*kernel_stats.csv | cut -c1-160
