"""GPU parity of the observables next to the force path against oracle/observables.py.

Tolerances: the density field sums N unit-modulus terms whose phases the reference evaluates with numpy (BLAS dot for
k.r, libm cos/sin) -- the GPU evaluates k.r = (x kx + y ky) + z kz without FMA and uses the device library's sincos
(<= 2 ulp), so term by term the two differ by a few 1e-16 * (1 + |k.r|).  Stated bound: |rho_gpu - rho_ref| <= 1e-12 * N
per component, and <= 1e-13 * N against the exactly summed evaluation of the same phases."""
import numpy as np
import pytest
import torch

import cavitymd
from cavitymd import _capi, observables as prod, synthetic
from oracle import observables as obs

pytestmark = pytest.mark.gpu


def _pdata(cfg):
    return cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                             cfg["box"], device="cuda")


@pytest.mark.parametrize("n_mol,n_k,kmag", [(500, 50, 1.0), (4096, 50, 1.0), (100_000, 50, 1.0), (10_001, 7, 0.1),
                                           (3_000, 64, 3.0), (3_000, 65, 3.0), (2_000, 150, 1.0), (63, 1, 1.0)])
def test_density_field_matches_reference_expression(n_mol, n_k, kmag):
    cfg = synthetic.diatomic_box(n_mol + (n_mol % 2), seed=n_mol + n_k)
    k = (obs.fibonacci_sphere(n_k) if n_k > 1 else np.array([[0.3, -0.4, 1.2]])) * kmag
    pd = _pdata(cfg)
    field = prod.DensityField(pd, k)
    n = pd.getN()
    want = obs.density_field(cfg["position"], k)            # the reference's expression (numpy)
    exact = obs.density_field_exact(cfg["position"], k)     # same phases, exactly rounded sums
    # every mapping of the kernel: chosen automatically, lane = wavevector, lane = particle with 25 / 10 / 5 per chunk
    for mapping in (-1, 0, 1, 2, 3):
        field._ws.set_tunable("rho_lane_particle", mapping)
        got = field.compute()
        assert got.shape == (n_k,)
        assert np.abs(got - want).max() <= 1e-12 * n, mapping
        assert np.abs(got - exact).max() <= 1e-13 * n, mapping
        # repeatable bit for bit, and usable on a side stream
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        assert np.array_equal(field.compute(stream=side), got)


def test_sincos_term_accuracy_through_single_particle():
    """One particle at (1, 0, 0) and wavevectors (x_i, 0, 0): rho(k_i) = exp(i x_i) exactly one term, so this checks the
    kernel's own sin/cos (Cody-Waite + fdlibm polynomials below 1e8, device-library path above) value by value."""
    rng = np.random.default_rng(7)
    pos = torch.tensor([[1.0, 0.0, 0.0]], dtype=torch.float64, device="cuda")
    ws = _capi.Workspace(1)
    for span, tol in ((1.0, 4.5e-16), (30.0, 4.5e-16), (3.0e4, 4.5e-16), (9.0e7, 4.5e-16), (1.0e12, 4.5e-16)):
        x = rng.uniform(-span, span, 4000)
        x[:5] = [0.0, np.pi / 2, -np.pi, 3 * np.pi / 4, np.nextafter(np.pi / 4, 1)][:5] if span >= 3 else x[:5]
        k = np.zeros((x.size, 3))
        k[:, 0] = x
        ws.set_wavevectors(k)
        ws.density_field(0, 1, pos.data_ptr(), 24)
        got = ws.density_field_read()
        assert np.abs(got.real - np.cos(x)).max() <= tol, span
        assert np.abs(got.imag - np.sin(x)).max() <= tol, span


def test_density_field_packed_positions_and_edge_cases():
    rng = np.random.default_rng(3)
    k = obs.fibonacci_sphere(50)
    ws = _capi.Workspace(1)
    with pytest.raises(_capi.CavmdError):                       # no wavevectors stored yet
        ws.density_field(0, 10, 8, 24)
    ws.set_wavevectors(k)
    with pytest.raises(_capi.CavmdError):                       # nothing computed yet
        ws.density_field_read()
    for n in (1, 64, 65, 1000):
        pos = rng.uniform(-20, 20, (n, 3))
        d = torch.from_numpy(pos).cuda()                         # packed (N,3), stride 24
        ws.density_field(0, n, d.data_ptr(), 24)
        got = ws.density_field_read()
        assert np.abs(got - obs.density_field_exact(pos, k)).max() <= 1e-13 * n
    # N = 0: rho = 0
    ws.density_field(0, 0, d.data_ptr(), 24)
    assert not ws.density_field_read().any()
    # large |k.r| takes the device library's big-argument reduction
    pos = rng.uniform(-1e6, 1e6, (500, 3))
    d = torch.from_numpy(pos).cuda()
    ws.density_field(0, 500, d.data_ptr(), 24)
    assert np.abs(ws.density_field_read() - obs.density_field_exact(pos, k)).max() <= 1e-9 * 500
    with pytest.raises(ValueError):
        ws.set_wavevectors(np.zeros((3, 2)))


def test_cavity_mode_and_total_dipole_without_a_snapshot():
    cfg = synthetic.config1(seed=5)
    pd = _pdata(cfg)
    p = cfg["params"]
    comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
    rng = np.random.default_rng(1)
    n = pd.getN()
    vel = rng.normal(size=(n, 3)) * 1e-3
    mass = rng.uniform(1.0, 30.0, n)
    vel4 = torch.from_numpy(np.concatenate([vel, mass[:, None]], axis=1)).cuda()
    with pytest.raises(_capi.CavmdError):                        # needs an evaluation first
        prod.cavity_mode(comp, vel4)
    comp.compute(0)
    got = prod.cavity_mode(comp, vel4)
    want = obs.cavity_mode(vel, mass, cfg["typeid"], comp.getHarmonicEnergy(), L_typeid=2)
    assert got == pytest.approx(want, rel=1e-14)
    d = prod.compute_total_dipole_moment(comp)
    want_d = obs.total_dipole_moment(cfg["position"], cfg["image"], cfg["charge"], cfg["box"])
    assert np.allclose(d, want_d, rtol=1e-12, atol=1e-12)
    # no photon -> zeros, like the reference's tracker
    cfg2 = dict(cfg)
    cfg2["typeid"] = np.zeros(n, dtype=np.int32)
    comp2 = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(_pdata(cfg2)), p["omegac"], p["couplstr"], p["phmass"])
    comp2.compute(0)
    assert prod.cavity_mode(comp2, vel4) == (0.0, 0.0, 0.0, 0.0)
