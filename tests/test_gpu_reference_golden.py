"""The HIP path against numbers the REFERENCE'S OWN PYTHON computed (tests/golden/reference_python_golden.npz; generator:
tests/golden/make_reference_python_golden.py, build container only).  Every call goes through the C ABI (libcavmd.so); the
oracle is not involved here at all -- these are reference outputs, not restatement outputs.

Tolerances (the fixtures' sums are numpy's: np.dot in BLAS order, np.sum pairwise; the kernels sum compensated in a fixed tree)
  force path     dipole |d_hip - d_ref| <= 2 N eps sum|c_i r_i|  (a-priori bound for two orders of the same addends; the
                 kernel's own error is <= 2 ulp of the exact sum) ; E_h equal to 4 eps relative (no sum in it) ; E_c, E_d and
                 the forces <= 1e-12 x the pre-cancellation scales of the parity contract -- two orders
                 of magnitude tighter than the contract's 1e-10.
  rho(k)         <= 1e-12 N per component (as tests/test_gpu_observables.py) ; F(k,t) from the GPU fields <= 1e-12 N^2
  cavity mode    rel 1e-14 ; sum |F|/m rel 1e-13, dt rel 1e-13
"""
import os

import numpy as np
import pytest
import torch

import cavitymd
from cavitymd import _capi, observables as prod

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps
COINCIDE = ["n3", "n50_first", "n501_stand_in", "n2000_last", "n257_heavy_photon"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    with np.load(os.path.join(golden_dir, "reference_python_golden.npz")) as z:
        return {k: z[k] for k in z.files}


def case(gold, name):
    pre = f"force/{name}/"
    return {k[len(pre):]: v for k, v in gold.items() if k.startswith(pre)}


def _class_route(c, tunables=None):
    """Through the mirror of the reference's compiled class (HOOMD's Scalar4 layouts); the cavity type is named 'L' and
    has type id 1, the id the reference's Python force hard-codes."""
    omegac, g, m, _ = c["params"]
    pd = cavitymd.ParticleData.from_arrays(c["position"], c["typeid"], c["charge"], c["image"], ["A", "L"], tuple(c["box"]),
                                           device="cuda")
    comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), omegac, g, m)
    for k, v in (tunables or {}).items():
        comp.workspace.set_tunable(k, v)
    comp.getForceArray().fill_(float("nan"))
    comp.compute(0)
    torch.cuda.synchronize()
    res = comp.getResult()
    return {"force": comp.getForceArray().cpu().numpy(), "energies": np.array(comp.getEnergies()),
            "total_dipole": np.array(res.total_dipole[:]), "dipole": np.array(res.dipole[:]), "photon_idx": res.photon_idx,
            "K": comp.getParams()["K"], "comp": comp}


def _custom_route(c):
    """Through cavmd_compute_soa, the entry point of the hoomd.md.force.Custom surface: packed (N,3) position, (N,) typeid,
    (N,3) force and (N,) potential_energy -- the very arrays the reference's set_forces reads and writes."""
    omegac, g, m, _ = c["params"]
    n = len(c["charge"])
    ws = _capi.Workspace(n)
    pos = torch.from_numpy(np.ascontiguousarray(c["position"])).cuda()
    tid = torch.from_numpy(c["typeid"].astype(np.int32)).cuda()
    img = torch.from_numpy(np.ascontiguousarray(c["image"], dtype=np.int32)).cuda()
    chg = torch.from_numpy(np.ascontiguousarray(c["charge"])).cuda()
    frc = torch.full((n, 3), float("nan"), dtype=torch.float64, device="cuda")
    pe = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    ws.compute_soa(0, n, (pos.data_ptr(), 24), (tid.data_ptr(), 4), (img.data_ptr(), 12), (chg.data_ptr(), 8),
                   tuple(c["box"]), 1, _capi.make_params(omegac, g, m), (frc.data_ptr(), 24), (pe.data_ptr(), 8))
    torch.cuda.synchronize()
    res = ws.result()
    return {"force": frc.cpu().numpy(), "potential_energy": pe.cpu().numpy(), "energies": np.array(ws.energies()),
            "total_dipole": np.array(res.total_dipole[:]), "photon_idx": res.photon_idx}


def _check_force_path(c, got):
    omegac, g, m, K = c["params"]
    n = len(c["charge"])
    cav = int(np.flatnonzero(c["typeid"] == 1)[0])
    assert got["photon_idx"] == cav
    unwrapped = c["position"] + c["image"] * c["box"][None, :]
    dscale = np.abs(c["charge"][:, None] * unwrapped).sum(axis=0)
    assert np.all(np.abs(got["total_dipole"] - c["total_dipole"]) <= 2 * n * EPS * dscale + 1e-300)
    q = unwrapped[cav]
    dxy = np.abs(c["total_dipole"][:2]).max()
    assert abs(got["energies"][0] - c["energies"][0]) <= 4 * EPS * c["energies"][0]
    assert abs(got["energies"][1] - c["energies"][1]) <= 1e-12 * g * np.abs(q[:2]).max() * dxy
    assert abs(got["energies"][2] - c["energies"][2]) <= 1e-12 * c["energies"][2]
    S = g * np.abs(c["charge"]) * (np.abs(q[:2]).max() + g / K * dxy)
    S[cav] = K * np.abs(q).max() + g * dxy
    f = got["force"][:, :3]
    assert not np.isnan(got["force"]).any()
    assert np.all(np.abs(f - c["force"]) <= 1e-12 * S[:, None])
    mol = np.arange(n) != cav
    assert np.all(f[mol, 2] == 0.0)
    # given the SAME dipole the reference's formulas are reproduced bit for bit: check it by feeding the kernel's own total
    # back through the reference's expressions (cavity_force_python.py:135-145)
    d = got["total_dipole"]
    Dq = np.array([q[0] + (g / K) * d[0], q[1] + (g / K) * d[1]])
    F = np.zeros((n, 3))
    F[:, 0] = ((-g) * c["charge"]) * Dq[0]
    F[:, 1] = ((-g) * c["charge"]) * Dq[1]
    F[cav] = [-K * q[0] - g * d[0], -K * q[1] - g * d[1], -K * q[2]]
    return F


@pytest.mark.parametrize("name", COINCIDE)
def test_forces_and_energies_match_the_executed_reference(gold, name):
    c = case(gold, name)
    a = _class_route(c)
    assert a["K"] == c["params"][3]
    F = _check_force_path(c, a)
    assert np.all(a["force"][:, 3] == 0.0) and np.all(c["potential_energy"] == 0.0)
    # the photon carries charge 0 here, so the kernel's molecular dipole and its total dipole are the same number, and the
    # forces are bit-identical to the reference's expressions evaluated on it
    assert np.array_equal(a["dipole"], a["total_dipole"]) and np.array_equal(a["force"][:, :3], F)
    b = _custom_route(c)
    F = _check_force_path(c, b)
    assert np.array_equal(b["force"], F) and np.all(b["potential_energy"] == 0.0)
    # every launch shape the library has gives the same answer on these inputs (single block, one launch, two launches)
    for tun in ({"small_system_max_n": 0}, {"small_system_max_n": 0, "persistent": 0}):
        _check_force_path(c, _class_route(c, tun))


def test_the_documented_divergences_follow_the_cpp_class_not_the_fallback(gold):
    """The product implements the C++ class: a charged cavity particle stays OUT of the molecular dipole and a
    second particle of the cavity type gets NO force -- where the reference's Python fallback does otherwise.  What must
    still agree with the executed fallback: the TOTAL dipole observable (all particles), and all forces in the second case
    except that one particle."""
    c = case(gold, "div_charged_cavity")
    a = _class_route(c)
    n = len(c["charge"])
    dscale = np.abs(c["charge"][:, None] * (c["position"] + c["image"] * c["box"][None, :])).sum(axis=0)
    assert np.all(np.abs(a["total_dipole"] - c["total_dipole"]) <= 2 * n * EPS * dscale)
    assert not np.allclose(a["dipole"], c["total_dipole"], rtol=1e-6)
    c = case(gold, "div_two_cavity_typed")
    a = _class_route(c)
    first, second = np.flatnonzero(c["typeid"] == 1)
    assert a["photon_idx"] == first and np.all(a["force"][second] == 0.0) and np.any(c["force"][second] != 0.0)
    others = np.setdiff1d(np.arange(n), [second])
    assert np.abs(a["force"][others, :3] - c["force"][others]).max() <= 1e-12 * np.abs(c["force"][others]).max()
    c = case(gold, "no_cavity")
    a = _class_route(c)
    assert a["photon_idx"] == -1 and not a["force"].any() and not a["energies"].any()
    b = _custom_route(c)
    assert not b["force"].any() and not b["potential_energy"].any() and not b["energies"].any()


def test_density_field_and_F_kt_match_the_executed_reference(gold):
    frames = gold["trajectory/frames"]
    n = frames.shape[1]
    ws = _capi.Workspace(n)
    for key in ("density/k1.0_n50", "density/k0.35_n17", "density/k2.5_n64"):
        k = gold[key + "/wavevectors"]
        ws.set_wavevectors(k)
        want = gold[key + "/rho_k"]
        for mapping in (-1, 0, 1):
            ws.set_tunable("rho_lane_particle", mapping)
            rho = []
            for f in frames:
                d = torch.from_numpy(np.ascontiguousarray(f)).cuda()
                ws.density_field(0, n, d.data_ptr(), 24)
                rho.append(ws.density_field_read())
            rho = np.array(rho)
            assert np.abs(rho - want).max() <= 1e-12 * n
            fkt = np.array([np.mean(np.real(rho[0] * np.conj(rho[t]))) for t in range(1, len(frames))])
            assert np.abs(fkt - gold[key + "/F_kt"][1:]).max() <= 1e-12 * n * n
    # the wavevector construction the product exports is the reference's, to the last bit but one
    assert np.abs(prod.generate_fibonacci_sphere(50) - gold["fibonacci/50"]).max() <= 2.3e-16
    c = case(gold, "n2000_last")
    ws2 = _capi.Workspace(2000)
    ws2.set_wavevectors(gold["density/n2000/wavevectors"])
    d = torch.from_numpy(np.ascontiguousarray(c["position"])).cuda()
    ws2.density_field(0, 2000, d.data_ptr(), 24)
    assert np.abs(ws2.density_field_read() - gold["density/n2000/rho_k"]).max() <= 1e-12 * 2000


def test_total_dipole_series_and_C_t_match_the_executed_reference(gold):
    """DipoleAutocorrelation's observable from the force evaluation's own reduction, frame by frame."""
    c = case(gold, "n501_stand_in")
    frames = gold["trajectory/frames"]
    n = frames.shape[1]
    omegac, g, m, _ = c["params"]
    d_t = []
    for f in frames:
        pd = cavitymd.ParticleData.from_arrays(f, c["typeid"], c["charge"], c["image"], ["A", "L"], tuple(c["box"]), device="cuda")
        comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), omegac, g, m)
        comp.compute(0)
        d_t.append(prod.compute_total_dipole_moment(comp))
    d_t = np.array(d_t)
    want = gold["dipole_acf/dipole_t"]
    for t, f in enumerate(frames):
        dscale = np.abs(c["charge"][:, None] * (f + c["image"] * c["box"][None, :])).sum(axis=0)
        assert np.all(np.abs(d_t[t] - want[t]) <= 2 * n * EPS * dscale)
    C = np.array([np.dot(d_t[0], x) for x in d_t])
    assert np.allclose(C, gold["dipole_acf/C_t"], rtol=1e-11, atol=0)


def test_cavity_mode_matches_the_executed_reference(gold):
    for i in range(3):
        g = {k: gold[f"cavity_mode/{i}/{k}"] for k in ("typeid", "mass", "velocity", "harmonic_energy", "properties")}
        n = len(g["mass"])
        where = int(np.flatnonzero(g["typeid"] == 2)[0])
        # an evaluation whose harmonic energy is the fixture's: photon alone at q with 0.5 K q^2 = harmonic (no molecules
        # charged), the other particles neutral; cavity_mode reads the photon index and E_h from that evaluation
        omegac, gcoup, m = 0.25, 1e-3, 2.0
        K = m * omegac * omegac
        pos = np.zeros((n, 3))
        pos[where, 0] = np.sqrt(2.0 * float(g["harmonic_energy"]) / K)
        pd = cavitymd.ParticleData.from_arrays(pos, g["typeid"], np.zeros(n), np.zeros((n, 3), dtype=np.int32), ["A", "B", "L"],
                                               (1e3, 1e3, 1e3), device="cuda")
        comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), omegac, gcoup, m)
        comp.compute(0)
        vel4 = torch.from_numpy(np.concatenate([g["velocity"], g["mass"][:, None]], axis=1)).cuda()
        ke, pe, tot, T = prod.cavity_mode(comp, vel4)
        want = g["properties"]
        assert ke == pytest.approx(want[0], rel=1e-14) and T == pytest.approx(want[3], rel=1e-14)
        assert pe == comp.getHarmonicEnergy() == pytest.approx(want[1], rel=1e-14, abs=0)
        assert tot == pytest.approx(ke + pe, rel=1e-15)


def test_force_mass_sum_and_timestep_rule_match_the_executed_reference(gold):
    for i in range(3):
        g = {k: gold[f"adaptive_dt/{i}/{k}"] for k in ("mass", "force_a", "force_b", "error_tolerance", "dt")}
        n = len(g["mass"])
        net = np.zeros((n, 4))
        net[:, :3] = g["force_a"] + g["force_b"]              # HOOMD's net force array is the sum over force objects
        vel4 = np.zeros((n, 4))
        vel4[:, 3] = g["mass"]
        ws = _capi.Workspace(n)
        S = prod.force_mass_sum(ws, torch.from_numpy(net).cuda(), torch.from_numpy(vel4).cuda())
        tol, dt = float(g["error_tolerance"]), float(g["dt"])
        assert S == pytest.approx(tol / dt**2, rel=1e-13)
        assert prod.adaptive_timestep(tol, S) == pytest.approx(dt, rel=1e-13)


def test_group_kinetic_energy_matches_the_executed_reference(gold):
    """cavmd_kinetic_energy over the molecular group (device index list = every particle whose typeid is not 2) and over the
    cavity particle alone against EnergyTracker's internal kinetic energies, computed by the reference's own code
    (analysis.py:524-598).  numpy sums pairwise, the kernel compensated in a fixed tree: 1e-13 relative."""
    for i in range(3):
        g = {k: gold[f"kinetic/{i}/{k}"] for k in ("typeid", "mass", "velocity", "molecular", "cavity")}
        n = len(g["mass"])
        vel4 = torch.from_numpy(np.concatenate([g["velocity"], g["mass"][:, None]], axis=1)).cuda()
        ws = _capi.Workspace(n)
        mol = torch.from_numpy(np.flatnonzero(g["typeid"] != 2).astype(np.int32)).cuda()
        ke = ws.kinetic_energy(0, vel4.data_ptr(), mol.data_ptr(), mol.numel())
        assert ke == pytest.approx(g["molecular"][0], rel=1e-13)
        cav = torch.from_numpy(np.flatnonzero(g["typeid"] == 2).astype(np.int32)).cuda()
        assert ws.kinetic_energy(0, vel4.data_ptr(), cav.data_ptr(), 1) == pytest.approx(float(g["cavity"]), rel=1e-15)
        # total = molecular + cavity, all particles without an index list
        assert ws.kinetic_energy(0, vel4.data_ptr(), None, n) == pytest.approx(g["molecular"][0] + float(g["cavity"]), rel=1e-13)
