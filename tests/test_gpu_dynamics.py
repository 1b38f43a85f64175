"""End-to-end check on BASELINE config 1 ("1k steps"): a velocity-Verlet integration of the cavity-coupled system in
which the ONLY force is the cavity force from the HIP kernels, run on the GPU, beside the same integration with the CPU
oracle's forces.  HOOMD-blue (the reference's integrator, bonds, LJ, PPPM, thermostats) is not available here, so this
is not the reference's simulation; what it does test is what a simulation needs from this path over many steps:
  - forces and energies are consistent: H = KE + E_h + E_c + E_d is conserved to the integrator's order, no drift;
  - the GPU trajectory stays on the oracle's trajectory (the 1e-14-level force differences do not grow);
  - the adaptive-timestep reduction (row f4) and the cavity-mode observable (row f2) agree with the oracle every time
    they are polled."""
import numpy as np
import pytest
import torch

import cavitymd
from cavitymd import observables as prod, synthetic
from oracle import observables as obs

pytestmark = pytest.mark.gpu


def _wrap(r, L):
    img = np.floor((r + L / 2) / L)
    return r - img * L, img.astype(np.int32)


def test_thousand_step_velocity_verlet(ref, oracle_mod):
    cfg = synthetic.config1(seed=1)
    n = len(cfg["charge"])
    L = np.asarray(cfg["box"])
    p = cfg["params"]
    rng = np.random.default_rng(42)
    mass = np.where(cfg["typeid"] == 2, 1.0, rng.uniform(2.5e4, 3.0e4, n))
    r0 = cfg["position"] + cfg["image"] * L[None, :]
    v0 = rng.normal(size=(n, 3)) * np.sqrt(3.167e-4 / mass)[:, None]
    dt, steps = 5.0, 1000

    # ---- GPU run: state lives in device memory, the force comes from libcavmd
    dev = "cuda"
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                           cfg["box"], device=dev)
    comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
    Lg = torch.tensor(L, device=dev)
    r = torch.from_numpy(r0).to(dev)
    vel4 = torch.from_numpy(np.concatenate([v0, mass[:, None]], axis=1)).to(dev)   # HOOMD Scalar4 vel: mass in .w
    m = vel4[:, 3:4]

    def gpu_force(step):
        img = torch.floor((r + Lg / 2) / Lg)
        pd.getPositions()[:, :3] = r - img * Lg
        pd.getImages().copy_(img.to(torch.int32))
        comp.compute(step)
        return comp.getForceArray()

    # ---- CPU twin with the oracle's forces
    prm = ref.make_params(p["omegac"], p["couplstr"], p["phmass"])
    rc, vc = r0.copy(), v0.copy()

    def cpu_force():
        pos, img = _wrap(rc, L)
        out = ref.compute(oracle_mod.pack_pos(pos, cfg["typeid"]), cfg["charge"], img, cfg["box"], 2, prm)
        return out["force"], out["energies"]

    F = gpu_force(0)
    Fc, Ec = cpu_force()
    H = []
    for step in range(1, steps + 1):
        vel4[:, :3] += 0.5 * dt * F[:, :3] / m
        r += dt * vel4[:, :3]
        F = gpu_force(step)
        vel4[:, :3] += 0.5 * dt * F[:, :3] / m
        vc += 0.5 * dt * Fc[:, :3] / mass[:, None]
        rc += dt * vc
        Fc, Ec = cpu_force()
        vc += 0.5 * dt * Fc[:, :3] / mass[:, None]
        if step % 10 == 0:
            e = comp.getEnergies()
            ke = float(0.5 * (m[:, 0] * (vel4[:, :3]**2).sum(dim=1)).sum())
            H.append(ke + e[0] + e[1] + e[2])
        if step % 250 == 0:
            # observables polled the way the reference's trackers/updaters would, but without a snapshot
            S = prod.force_mass_sum(comp.workspace, F, vel4)
            assert S == pytest.approx(obs.force_mass_sum(Fc[:, :3], mass), rel=1e-9)
            assert S == pytest.approx(obs.force_mass_sum_exact(F[:, :3].cpu().numpy(), mass), rel=1e-14)
            assert prod.adaptive_timestep(1e-3, S) == pytest.approx(np.sqrt(1e-3 / S), rel=1e-15)
            ke_g = prod.cavity_mode(comp, vel4)
            ke_c = obs.cavity_mode(vc, mass, cfg["typeid"], float(Ec[0]))
            assert ke_g == pytest.approx(ke_c, rel=1e-8)

    # GPU trajectory == oracle trajectory
    rg, vg = r.cpu().numpy(), vel4[:, :3].cpu().numpy()
    assert np.abs(rg - rc).max() <= 1e-9 * np.abs(rc - r0).max()
    assert np.abs(vg - vc).max() <= 1e-9 * np.abs(vc).max()
    # energy: bounded oscillation of velocity Verlet ((omega dt)^2 ~ 2e-3 of the oscillator energy), no drift
    H = np.array(H)
    scale = abs(Ec).max() + 0.5 * float((mass * (vc**2).sum(axis=1)).sum())
    assert np.abs(H - H[0]).max() <= 5e-3 * scale
    assert abs(H[-20:].mean() - H[:20].mean()) <= 5e-4 * scale
    # the photon really moved through several periods, so the test exercised changing q, d and image flags
    assert np.abs(rc[-1] - r0[-1]).max() > 0.1


def test_force_mass_sum_sizes_and_errors():
    from cavitymd import _capi
    rng = np.random.default_rng(9)
    ws = _capi.Workspace(1)
    for n in (1, 255, 1024, 1025, 300_001):
        f = rng.normal(size=(n, 4))
        v = rng.normal(size=(n, 4))
        v[:, 3] = rng.uniform(0.5, 50.0, n)
        fg, vg = torch.from_numpy(f).cuda(), torch.from_numpy(v).cuda()
        got = prod.force_mass_sum(ws, fg, vg)
        assert got == pytest.approx(obs.force_mass_sum_exact(f[:, :3], v[:, 3]), rel=1e-15)
        assert got == pytest.approx(obs.force_mass_sum(f[:, :3], v[:, 3]), rel=1e-12)
        assert prod.force_mass_sum(ws, fg, vg) == got          # bit-reproducible
    assert prod.adaptive_timestep(1e-3, 0.0) is None
    with pytest.raises(ValueError):
        prod.force_mass_sum(ws, fg[:, :3].contiguous(), vg)
    assert ws.force_mass_sum(0, 0, fg.data_ptr(), vg.data_ptr()) == 0.0
