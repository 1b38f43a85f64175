"""The pybind11 flavour of the shim (cavitymd._cavitymd): builds in-tree, exposes the reference's method names, refuses to
run without a GPU (CPU test) and gives the same bits as the ctypes route (GPU test)."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def ext(capi):
    from cavitymd import _cavitymd
    return _cavitymd


def test_surface_matches_reference_exports(ext):
    cls = ext.CavityForceComputeHIP
    # src/CavityForceCompute.cc:212-224: setParams, getParams, getHarmonicEnergy, getCouplingEnergy, getDipoleSelfEnergy
    for name in ("setParams", "getParams", "getHarmonicEnergy", "getCouplingEnergy", "getDipoleSelfEnergy",
                 "computeForces"):
        assert hasattr(cls, name)
    assert ext.version() == 2


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_is_a_runtime_error(ext):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.CavityForceComputeHIP(100, 0.0091, 1e-3)


@pytest.mark.gpu
def test_pybind_route_equals_ctypes_route(ext, ref):
    import cavitymd
    from cavitymd import synthetic
    cfg = synthetic.random_charged_box(50_000, seed=11)
    p = cfg["params"]
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                           cfg["box"], device="cuda")
    a = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
    a.compute(0)
    n = pd.getN()
    b = ext.CavityForceComputeHIP(n, p["omegac"], p["couplstr"], p["phmass"])
    assert b.getParams() == a.getParams() == ref.make_params(p["omegac"], p["couplstr"], p["phmass"])
    force = torch.full((n, 4), float("nan"), dtype=torch.float64, device="cuda")
    L = cfg["box"]
    b.computeForces(pd.getPositions().data_ptr(), pd.getCharges().data_ptr(), pd.getImages().data_ptr(), n, L[0], L[1], L[2],
                    2, force.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(force.cpu().numpy(), a.getForceArray().cpu().numpy())
    assert b.getEnergies() == a.getEnergies()
    assert (b.getHarmonicEnergy(), b.getCouplingEnergy(), b.getDipoleSelfEnergy()) == a.getEnergies()
    r = b.getResult()
    assert r["photon_idx"] == n - 1 and r["n_photon_typed"] == 1 and r["dipole"] == tuple(a.getResult().dipole[:])
    b.setParams(p["omegac"], 2 * p["couplstr"], p["phmass"])
    assert b.getParams()["couplstr"] == 2 * p["couplstr"]
    with pytest.raises(RuntimeError):
        b.computeForces(0, 0, 0, n, 1.0, 1.0, 1.0, 2, 0, 0)


@pytest.mark.gpu
def test_compute_uses_either_binding_with_identical_results(ext):
    """CavityForceComputeHIP.compute goes through the pybind11 free function when the module is built; the ctypes route
    on the same workspace must give the same bits."""
    import cavitymd
    from cavitymd import compute as compute_mod, synthetic
    cfg = synthetic.config1(seed=9)
    p = cfg["params"]
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                           cfg["box"], device="cuda")
    comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
    assert compute_mod._ext is ext or os.environ.get("CAVMD_BINDING", "").lower() == "ctypes"
    comp.compute(0)
    a = comp.getForceArray().cpu().numpy().copy()
    ea = comp.getEnergies()
    saved, compute_mod._ext = compute_mod._ext, None
    try:
        comp.getForceArray().fill_(float("nan"))
        comp.compute(1)
        assert np.array_equal(comp.getForceArray().cpu().numpy(), a) and comp.getEnergies() == ea
    finally:
        compute_mod._ext = saved
