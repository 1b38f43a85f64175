"""The C++ half of the drop-in, cav-hoomd_amd/csrc/hoomd_shim/CavityForceComputeHIP.{h,cc}, COMPILED (unchanged) against the
stand-in HOOMD declarations of tests/stubs/hoomd_cpp and EXECUTED on the GPU box through a pybind11 module built from them.

What this checks: the file parses and links against libcavmd; the exported Python names are the reference's
(src/CavityForceComputeGPU.cc:257-264); ArrayHandle scopes are balanced (every array acquired once per step, none held on
return); the energy cache is keyed on the evaluation (setParams + a recomputation at the SAME timestep gives new energies);
N = 0; a system that grows past the workspace's capacity; no type named 'L'; forces and energies against the oracle.
What it does NOT check: anything about a real HOOMD-blue -- the stand-ins are this repository's own, written from the calls the
shim makes.  SURVEY.md row f1 stays "blocked: no HOOMD in the image"."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
STANDIN = os.path.join(ROOT, "tests", "stubs", "hoomd_cpp")


@pytest.fixture(scope="module")
def shim(capi):
    """Builds (idempotent) and imports the stand-in module; `capi` has built libcavmd.so first."""
    subprocess.run(["make", "-C", STANDIN, "-s", "syntax"], check=True)      # the shim alone, -fsyntax-only, module.cc included
    subprocess.run(["make", "-C", STANDIN, "-s", "all"], check=True)
    # loaded by file path: the stand-in directory holds a folder named "hoomd" (C++ headers) and must not get onto sys.path,
    # where Python would take it for a (namespace) package of that name
    import glob
    import importlib.util
    path = glob.glob(os.path.join(STANDIN, "_cavitymd_hip_standin*.so"))[0]
    spec = importlib.util.spec_from_file_location("_cavitymd_hip_standin", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.IS_STAND_IN
    return mod


def test_shim_compiles_and_exports_the_reference_names(shim):
    cls = shim.CavityForceComputeHIP
    for name in ("setParams", "getParams", "getHarmonicEnergy", "getCouplingEnergy", "getDipoleSelfEnergy", "compute"):
        assert hasattr(cls, name), name
    assert issubclass(cls, shim.ForceCompute)
    # a CPU execution configuration is refused (the reference throws "GPU computation required but not available",
    # src/CavityForceComputeGPU.cc:106-109); N = 0 so that the stand-in allocates nothing and this runs without a GPU
    pd = shim.ParticleData(0, (10.0, 10.0, 10.0), ["A", "L"], shim.ExecutionConfiguration(False))
    with pytest.raises(RuntimeError, match="GPU execution configuration is required"):
        cls(shim.SystemDefinition(pd), 0.0091, 1e-3, 1.0)


def _system(shim, cfg):
    n = len(cfg["charge"])
    pd = shim.ParticleData(n, tuple(float(x) for x in cfg["box"]), list(cfg["types"]), shim.ExecutionConfiguration(True))
    return pd


def _load(pd, oracle_mod, cfg):
    pd.set_arrays(oracle_mod.pack_pos(cfg["position"], cfg["typeid"]), cfg["charge"], cfg["image"].astype(np.int32))


def _want(ref, oracle_mod, cfg, couplstr=None):
    p = cfg["params"]
    prm = ref.make_params(p["omegac"], p["couplstr"] if couplstr is None else couplstr, p["phmass"])
    return ref.compute(oracle_mod.pack_pos(cfg["position"], cfg["typeid"]), cfg["charge"], cfg["image"], cfg["box"],
                       cfg["types"].index("L") if "L" in cfg["types"] else -1, prm)


def _close(got, want, cfg):
    scale = np.abs(want["force"]).max()
    assert not np.isnan(got).any()
    assert np.abs(got - want["force"]).max() <= 1e-12 * scale


@pytest.mark.gpu
def test_shim_runs_the_steps_hoomd_would_ask_for(shim, ref, oracle_mod):
    from cavitymd import synthetic
    cfg = synthetic.config1(seed=3)
    p = cfg["params"]
    pd = _system(shim, cfg)
    _load(pd, oracle_mod, cfg)
    base = pd.acquisitions()
    fc = shim.CavityForceComputeHIP(shim.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
    assert fc.getParams() == {"omegac": p["omegac"], "couplstr": p["couplstr"], "K": p["phmass"] * p["omegac"] ** 2, "phmass": p["phmass"]}
    assert (fc.getHarmonicEnergy(), fc.getCouplingEnergy(), fc.getDipoleSelfEnergy()) == (0.0, 0.0, 0.0)   # before any step
    want = _want(ref, oracle_mod, cfg)
    for step in range(3):
        fc.fill_force(float("nan"))
        fc.compute(step)
        # every particle array acquired exactly once per step, the force array too, and none left held
        assert pd.acquisitions() == tuple(b + step + 1 for b in base)
        assert not pd.any_handle_held() and not fc.force_handle_held()
        _close(fc.force(), want, cfg)
        e = np.array([fc.getHarmonicEnergy(), fc.getCouplingEnergy(), fc.getDipoleSelfEnergy()])
        assert np.allclose(e, want["energies"], rtol=1e-12, atol=0)
    # setParams, then a recomputation at the SAME timestep (what sim.run(0) does): new forces, new energies -- not the cache
    fc.setParams(p["omegac"], 2 * p["couplstr"], p["phmass"])
    fc.compute(2)
    want2 = _want(ref, oracle_mod, cfg, couplstr=2 * p["couplstr"])
    _close(fc.force(), want2, cfg)
    assert np.isclose(fc.getCouplingEnergy(), want2["energies"][1], rtol=1e-12) and not np.isclose(fc.getCouplingEnergy(), want["energies"][1])
    # the system grows past the capacity the workspace was created for (HOOMD-blue: particle insertion): new workspace, same answers
    big = synthetic.random_charged_box(20_000, seed=8)
    pd.setN(len(big["charge"]))
    _load(pd, oracle_mod, big)
    fc.setParams(big["params"]["omegac"], big["params"]["couplstr"], big["params"]["phmass"])
    fc.compute(3)
    # (the stand-in system keeps its box and type list: evaluate the oracle on what the shim actually saw)
    big_seen = dict(big, box=cfg["box"], types=cfg["types"])
    _close(fc.force(), _want(ref, oracle_mod, big_seen), big_seen)
    # N = 0: nothing enqueued, energies zero (src/CavityForceCompute.cc:148-156)
    pd.setN(0)
    fc.compute(4)
    assert fc.force().shape == (0, 4)
    assert (fc.getHarmonicEnergy(), fc.getCouplingEnergy(), fc.getDipoleSelfEnergy()) == (0.0, 0.0, 0.0)


@pytest.mark.gpu
def test_shim_without_a_type_named_L_gives_zeros(shim, ref, oracle_mod):
    """The CPU reference lets getTypeByName throw; its GPU class catches that and zeroes the energies
    (src/CavityForceComputeGPU.cc:114-123).  The shim passes -1, which matches no particle."""
    from cavitymd import synthetic
    cfg = synthetic.config1(seed=4)
    cfg = dict(cfg, types=["O", "N", "X"])
    pd = _system(shim, cfg)
    _load(pd, oracle_mod, cfg)
    p = cfg["params"]
    fc = shim.CavityForceComputeHIP(shim.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
    fc.fill_force(float("nan"))
    fc.compute(0)
    assert not fc.force().any()
    assert (fc.getHarmonicEnergy(), fc.getCouplingEnergy(), fc.getDipoleSelfEnergy()) == (0.0, 0.0, 0.0)
