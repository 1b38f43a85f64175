"""Executes cavitymd/hoomd_plugin.py against a STAND-IN hoomd package (tests/stubs/hoomd: not HOOMD-blue, no claim of API
compatibility) so that its control flow -- attach ladder, the hoomd.md.force.Custom route, the energy cache, the loggable
properties -- runs at least once and is checked against the oracle.  What this proves: the module is free of typos and its
own logic is consistent.  What it does not prove: anything about a real HOOMD-blue; f1 stays blocked.

Runs in a subprocess because the stand-in must be on sys.path BEFORE cavitymd is imported (cavitymd.forces decides at import
time whether hoomd is there)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
STUBS = os.path.join(ROOT, "tests", "stubs")

CHILD = r'''
import contextlib, sys
import numpy as np, torch
import hoomd
assert getattr(hoomd, "IS_STUB", False)
import cavitymd
from cavitymd import hoomd_plugin, marshal, synthetic
import oracle
assert cavitymd.CavityForce is hoomd_plugin.HoomdCavityForce       # the HOOMD flavour was selected at import time

class Particles:  pass
class Snap:       pass
class Box:        pass
class State:
    def __init__(self, cfg, dev):
        self.particle_types = list(cfg["types"])
        self._cpp_sys_def = object()
        self.pos4 = torch.from_numpy(oracle.pack_pos(cfg["position"], cfg["typeid"])).to(dev)
        self.charge = torch.from_numpy(cfg["charge"]).to(dev)
        self.image = torch.from_numpy(cfg["image"]).to(dev)
        self.box = cfg["box"]
        self._forces = {}
    def _force4_for(self, force):
        n = self.pos4.shape[0]
        return self._forces.setdefault(id(force), torch.full((n, 4), float("nan"), dtype=torch.float64, device=self.pos4.device))
    @property
    @contextlib.contextmanager
    def gpu_local_snapshot(self):
        s, p, b = Snap(), Particles(), Box()
        p.position, p.typeid = self.pos4[:, :3], self.pos4.view(torch.int32)[:, 6]
        p.image, p.charge = self.image, self.charge
        b.L = self.box
        s.particles, s.global_box = p, b
        yield s
class Sim:
    def __init__(self, cfg, device):
        self.device, self.state = device, State(cfg, "cuda")

cfg = synthetic.config1(seed=4)
p = cfg["params"]
ref = oracle.RefOracle()
want = ref.compute(oracle.pack_pos(cfg["position"], cfg["typeid"]), cfg["charge"], cfg["image"], cfg["box"], cfg["L_typeid"],
                   ref.make_params(p["omegac"], p["couplstr"], p["phmass"]))

# no CPU implementation: attaching on hoomd.device.CPU must raise
f = cavitymd.CavityForce(kvector=[0, 0, 1], couplstr=p["couplstr"], omegac=p["omegac"], phmass=p["phmass"])
try:
    f._attach(Sim(cfg, hoomd.device.CPU()))
except RuntimeError as e:
    assert "no CPU implementation" in str(e)
else:
    raise SystemExit("attach on a CPU device did not raise")

# the compiled rung is absent here (no HOOMD headers) -> the hoomd.md.force.Custom route
f = cavitymd.CavityForce(kvector=[0, 0, 1], couplstr=p["couplstr"], omegac=p["omegac"], phmass=p["phmass"])
assert [float(v) for v in f.kvector] == [0.0, 0.0, 1.0]
assert f.couplstr == p["couplstr"] and f.omegac == p["omegac"] and f.phmass == p["phmass"]
sim = Sim(cfg, hoomd.device.GPU())
f._attach(sim)
assert f.implementation == "hip_custom" and f._cpp_obj is f._force_impl._cpp_obj
assert f.energy == 0.0                             # before the first step the workspace does not exist yet
for step in range(3):
    f._cpp_obj.compute(step)                        # what HOOMD's integrator does every step
torch.cuda.synchronize()
inner = f._force_impl
frc = sim.state._force4_for(inner).cpu().numpy()
scale = np.abs(want["force"]).max()
assert np.abs(frc - want["force"]).max() <= 1e-12 * scale and not np.isnan(frc).any()
assert np.allclose([f.harmonic_energy, f.coupling_energy, f.dipole_self_energy], want["energies"], rtol=1e-12, atol=0)
assert f.total_cavity_energy == f.harmonic_energy + f.coupling_energy + f.dipole_self_energy == f.energy
assert f.forces is None                            # as the reference for compiled implementations
# setParams at the same "timestep", then a recomputation: the energies must be the new ones (cache keyed on the evaluation)
inner.setParams(p["omegac"], 2 * p["couplstr"], p["phmass"])
f._cpp_obj.compute(2)
assert np.isclose(f.coupling_energy, 2 * want["energies"][1], rtol=1e-12)
f._detach()
assert f._force_impl is None and f.harmonic_energy == 0.0
print("STUB-PLUGIN-OK")
'''


@pytest.mark.gpu
def test_hoomd_plugin_control_flow_against_stand_in_hoomd():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUBS, ROOT, os.path.join(ROOT, "cav-hoomd_amd")]))
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "STUB-PLUGIN-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_plugin_imports_and_selects_the_hoomd_flavour_with_the_stand_in():
    """CPU part: with the stand-in on the path the package picks HoomdCavityForce, and attaching without a GPU device refuses."""
    code = ("import hoomd, cavitymd; from cavitymd import hoomd_plugin; assert hoomd.IS_STUB; "
            "assert cavitymd.CavityForce is hoomd_plugin.HoomdCavityForce; "
            "f = cavitymd.CavityForce([0, 0, 1], 1e-3, 0.0091); "
            "assert f.implementation == 'hip' and f.energy == 0.0 and f.forces is None; print('OK')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUBS, ROOT, os.path.join(ROOT, "cav-hoomd_amd")]))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
