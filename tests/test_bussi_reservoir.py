"""Row f4: the Bussi reservoir thermostat step (src/BussiReservoirThermostat.h:43-98, 177-225) with injected variates.

CPU part (no GPU): the product's scalar rule (host arithmetic inside libcavmd) against the oracle restatement bit for bit,
closed-form known answers, both branches of the sign rule, the edge cases of the reference (Nf = 0, Nf = 1, tau = 0,
deltaT = 0, zero momenta) and the reservoir bookkeeping.  GPU part (-m gpu): the group kinetic energy and a 1000-step run.

PARITY UNPINNED by the reference (its own test, src/pytest/test_bussi_reservoir.py, asserts only that the counters start at
zero, move and reset -- mirrored in test_counters_start_at_zero_move_and_reset).  Variate GENERATION (HOOMD's RandomGenerator)
is outside this comparison: the variates are inputs here.
"""
import math

import numpy as np
import pytest
import torch

import cavitymd
from cavitymd import _capi, thermostats


@pytest.fixture(scope="module")
def bussi(oracle_mod):
    return oracle_mod.BussiOracle()


def test_scalar_rule_matches_the_oracle_bit_for_bit(bussi, capi):
    rng = np.random.default_rng(7)
    neg = 0
    for _ in range(20000):
        dof = float(rng.choice([1.0, 2.0, 3.0, 297.0, 2999997.0]))
        set_T = float(10.0 ** rng.uniform(-4, 1))
        K = float(0.5 * dof * set_T * 10.0 ** rng.uniform(-3, 3))
        dt = float(10.0 ** rng.uniform(-3, 0))
        tau = float(rng.choice([0.0, dt * 20, 10.0 ** rng.uniform(-3, 3)]))
        r = float(rng.standard_normal() * rng.choice([1.0, 4.0]))
        g = float(rng.gamma((dof - 1) / 2, 1.0)) if dof > 1 else 0.0
        want = bussi.rescale_factor(K, dof, dt, set_T, tau, r, g)
        got = _capi.bussi_rescale_factor(K, dof, dt, set_T, tau, r, g)
        assert got == want or (math.isnan(got) and math.isnan(want))
        neg += got < 0
    assert neg > 50   # the negative branch of eq. (A8) was exercised


def test_known_answers(capi):
    f = _capi.bussi_rescale_factor
    # Nf = 0: no degrees of freedom, factor 1, no variate consumed (:183-184)
    assert f(5.0, 0.0, 0.1, 1.5, 1.0, 123.0, 456.0) == 1.0
    # tau = 0 (c = 0): alpha^2 = (kT / 2K) (2 g + R^2), sign = sign(R) (:186-190, 201-213)
    K, T, R, g = 3.0, 1.5, -0.75, 2.25
    a = f(K, 4.0, 0.01, T, 0.0, R, g)
    assert a == -math.sqrt(T / 2.0 / K * (1.0 - 0.0) * (2.0 * g + R * R)) and a < 0
    # Nf = 1: the gamma variate is ignored, KE_new = alpha^2 K = kT R^2 / 2 at tau = 0
    a = f(K, 1.0, 0.01, T, 0.0, 0.5, 1e9)
    assert a * a * K == pytest.approx(T * 0.25 / 2.0, rel=1e-15)
    # R = 0, no other degrees of freedom: alpha^2 = c, alpha = exp(-dt / 2 tau) > 0
    a = f(K, 1.0, 0.2, T, 0.8, 0.0, 0.0)
    assert a == math.sqrt(math.exp(-0.2 / 0.8))
    # dt / tau -> 0: c = 1, nothing happens
    assert f(K, 3.0, 1e-300, T, 1e300, 0.3, 0.9) == 1.0
    # the sign flips only for R < -sqrt(c Nf K / ((1 - c) Kbar)) = -sqrt(2 c K / ((1 - c) kT))
    c = math.exp(-0.5)
    edge = -math.sqrt(2.0 * c * K / ((1.0 - c) * T))
    assert f(K, 3.0, 0.5, T, 1.0, edge * (1 - 1e-12), 0.1) > 0 > f(K, 3.0, 0.5, T, 1.0, edge * (1 + 1e-12), 0.1)
    # large Nf: E[alpha^2 K] relaxes towards Kbar; with the mean variates (R^2 -> 1, 2g -> Nf - 1) and R = 0 cross term:
    Nf, Kbar = 2999997.0, 0.5 * 2999997.0 * T
    a = f(2 * Kbar, Nf, 0.1, T, 1.0, 0.0, (Nf - 1) / 2)
    cc = math.exp(-0.1)
    assert a * a * 2 * Kbar == pytest.approx(cc * 2 * Kbar + (1 - cc) * Kbar * (Nf - 1) / Nf, rel=1e-12)


def test_step_bookkeeping_matches_oracle_and_reference_edge_cases(bussi, capi):
    rng = np.random.default_rng(11)
    st = _capi.BussiReservoirState()
    ref_state = np.zeros(4)
    for step in range(1000):
        Kt, Kr = float(rng.uniform(0.5, 50)), float(rng.uniform(0.5, 50))
        var = [float(rng.standard_normal()), float(rng.gamma(148.0)), float(rng.standard_normal()), float(rng.gamma(99.0))]
        got = _capi.bussi_step(st, Kt, 297.0, Kr, 199.0, 0.01, 0.3, 0.2, var)
        want = bussi.step(ref_state, Kt, 297.0, Kr, 199.0, 0.01, 0.3, 0.2, var)
        assert got == want
        assert (st.reservoir_translational, st.reservoir_rotational, st.instantaneous_translational,
                st.instantaneous_rotational) == tuple(ref_state)
        assert st.instantaneous_translational == Kt * (1.0 - got[0] * got[0])          # :88
    # deltaT == 0: factors {1, 1}, counters untouched (:45-48)
    before = (st.reservoir_translational, st.reservoir_rotational)
    assert _capi.bussi_step(st, 1.0, 3.0, 1.0, 3.0, 0.0, 0.3, 0.2, [9, 9, 9, 9]) == (1.0, 1.0)
    assert (st.reservoir_translational, st.reservoir_rotational) == before
    # zero momenta with degrees of freedom: the reference throws (:57-61)
    with pytest.raises(_capi.CavmdError):
        _capi.bussi_step(st, 0.0, 3.0, 0.0, 0.0, 0.01, 0.3, 0.2, [0, 0, 0, 0])
    with pytest.raises(RuntimeError):
        bussi.step(ref_state, 0.0, 3.0, 0.0, 0.0, 0.01, 0.3, 0.2, [0, 0, 0, 0])
    # no rotational degrees of freedom (the cavity driver's case): rotational factor 1, counter stays 0
    st2 = _capi.BussiReservoirState()
    assert _capi.bussi_step(st2, 2.0, 297.0, 0.0, 0.0, 0.01, 1.5, 0.2, [0.1, 140.0, 0, 0])[1] == 1.0
    assert st2.reservoir_rotational == 0.0 and st2.reservoir_translational != 0.0


def test_python_surface_matches_reference_names():
    # src/thermostats.py: BussiReservoir(kT, tau=0.0) and its loggables; unattached -> 0.0 (reference: `if not self._attached`)
    b = thermostats.BussiReservoir(kT=1.5, tau=1.0)
    assert b.kT == 1.5 and b.tau == 1.0
    for name in ("reservoir_energy_translational", "reservoir_energy_rotational", "total_reservoir_energy",
                 "instantaneous_reservoir_translational", "instantaneous_reservoir_rotational",
                 "instantaneous_reservoir_total"):
        assert getattr(b, name) == 0.0
    b.reset_reservoir_energy()
    assert thermostats.BussiReservoir(kT=2.0).tau == 0.0
    v = thermostats.draw_variates(np.random.default_rng(1), 1.0, 0.0)
    assert v[1] == 0.0 and v[2:] == [0.0, 0.0] and v[0] != 0.0


# ---- GPU part ---------------------------------------------------------------------------------------------------------------
def _velocities(n, seed, masses=(15.999, 14.007)):
    rng = np.random.default_rng(seed)
    vel = np.empty((n, 4))
    vel[:, :3] = rng.normal(0, 1e-3, (n, 3))
    vel[:, 3] = np.where(np.arange(n) % 2 == 0, masses[0], masses[1]) * 1822.888
    return vel


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 255, 1024, 1025, 100_003, 1_000_001])
def test_group_kinetic_energy(bussi, n):
    vel = _velocities(n, seed=n)
    dvel = torch.from_numpy(vel).cuda()
    ws = _capi.Workspace(n)
    for members in (None, np.arange(0, n, 2, dtype=np.uint32), np.array(sorted(set(np.random.default_rng(n).integers(0, n, 500))), dtype=np.uint32)):
        if members is None:
            got = ws.kinetic_energy(0, dvel.data_ptr(), None, n)
        else:
            dm = torch.from_numpy(members.view(np.int32).copy()).cuda()
            got = ws.kinetic_energy(0, dvel.data_ptr(), dm.data_ptr(), len(members))
        want = bussi.kinetic_energy(vel, members)
        hi, lo = bussi.kinetic_energy(vel, members, exact=True)
        assert abs(got - want) <= 1e-12 * abs(want)                  # vs the reference-order sum
        assert abs(got - hi) <= 2 * np.spacing(abs(hi))              # vs the exactly rounded sum
    assert ws.kinetic_energy(0, dvel.data_ptr(), None, 0) == 0.0


@pytest.mark.gpu
def test_counters_start_at_zero_move_and_reset_over_1000_steps(bussi):
    """The reference's own test (src/pytest/test_bussi_reservoir.py:11-80) in this package's terms, plus what it cannot check:
    with no forces the kinetic energy lost by the group is exactly what the reservoir gained, step by step and in total,
    and the GPU run tracks the oracle driven by the same variates."""
    n = 20_001
    vel = _velocities(n, seed=5)
    members = np.arange(n - 1, dtype=np.uint32)           # the molecular group: everything but the photon (last particle)
    dvel = torch.from_numpy(vel.copy()).cuda()
    th = thermostats.BussiReservoir(kT=3.167e-6 * 100, tau=0.5)
    assert th.total_reservoir_energy == 0.0
    th.attach(n, members)
    dof = 3.0 * (n - 1) - 3.0
    ke0 = th.kinetic_energy(dvel)
    assert ke0 == pytest.approx(bussi.kinetic_energy(vel, members), rel=1e-12)
    rng = np.random.default_rng(99)
    ref_state, ref_vel = np.zeros(4), vel.copy()
    flips = 0
    for step in range(1000):
        var = thermostats.draw_variates(rng, dof)
        # now and then an instantaneous thermalisation (tau = 0: c = 0, sign[alpha] = sign[R]) with R < 0: the negative
        # branch of the sign rule, every velocity of the group reverses
        th.tau = 0.0 if step % 97 == 5 else 0.5
        if step % 97 == 5:
            var[0] = -abs(var[0]) - 0.1
        at, ar = th.step(step, 0.02, dvel, dof, variates=var)
        flips += at < 0
        ke_ref = bussi.kinetic_energy(ref_vel, members)
        want = bussi.step(ref_state, ke_ref, dof, 0.0, 0.0, 0.02, th.kT, th.tau, var)
        ref_vel[members, :3] *= want[0]
        assert ar == 1.0 and at == pytest.approx(want[0], rel=1e-9)
        assert th.instantaneous_reservoir_translational == pytest.approx(ref_state[2], rel=1e-6, abs=1e-12 * ke0)
    assert flips >= 5
    torch.cuda.synchronize()
    got_vel = dvel.cpu().numpy()
    assert np.array_equal(got_vel[-1], vel[-1])                                   # the photon is not in the group
    assert np.allclose(got_vel[:-1, :3], ref_vel[:-1, :3], rtol=1e-8, atol=0)
    assert np.array_equal(got_vel[:, 3], vel[:, 3])                               # masses untouched
    ke_end = th.kinetic_energy(dvel)
    assert th.total_reservoir_energy == pytest.approx(ke0 - ke_end, rel=1e-9, abs=1e-12 * ke0)   # energy bookkeeping closes
    assert th.total_reservoir_energy == pytest.approx(ref_state[0], rel=1e-7)
    assert th.reservoir_energy_rotational == 0.0 and th.total_reservoir_energy != 0.0
    th.reset_reservoir_energy()
    assert th.total_reservoir_energy == 0.0 and th.instantaneous_reservoir_total == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,grouped", [(1_025, False), (20_001, True), (300_001, False)])
def test_on_device_step_is_the_host_step_bit_for_bit(bussi, n, grouped):
    """cavmd_bussi_step_device (two kernels, no host round trip) against the host-driven step of the same library and
    against the oracle: same kinetic energy bits (same kernel body and grid), same alpha bits (the rule is ONE source function
    for host and device, c = exp(-dt/tau) taken on the host), same velocities, same counters -- step by step, including the
    negative-alpha branch (tau = 0, R < 0) and dt = 0."""
    vel = _velocities(n, seed=n + 1)
    members = np.arange(n - 1, dtype=np.uint32) if grouped else None
    nm = n - 1 if grouped else n
    dof = 3.0 * nm - 3.0
    a_vel, b_vel = torch.from_numpy(vel.copy()).cuda(), torch.from_numpy(vel.copy()).cuda()
    host = thermostats.BussiReservoir(kT=3.167e-6 * 100, tau=0.5)
    dev = thermostats.BussiReservoir(kT=3.167e-6 * 100, tau=0.5)
    host.attach(n, members)
    dev.attach(n, members)
    rng = np.random.default_rng(7)
    ref_state = np.zeros(4)
    for step in range(60):
        var = thermostats.draw_variates(rng, dof)
        tau = 0.0 if step % 13 == 5 else 0.5
        if step % 13 == 5:
            var[0] = -abs(var[0]) - 0.1
        dt = 0.0 if step == 20 else 0.02
        host.tau = dev.tau = tau
        ke_before = host.kinetic_energy(a_vel)
        at, _ = host.step(step, dt, a_vel, dof, variates=var)
        dev.step_async(step, dt, b_vel, dof, variates=var)
        if step % 7 == 0 or step == 20:
            st = dev.device_state()
            if dt != 0.0:
                assert st.last_kinetic_energy == ke_before and st.last_alpha == at          # bits
                assert st.last_alpha == _capi.bussi_rescale_factor(ke_before, dof, dt, host.kT, tau, var[0], var[1])
                want = bussi.step(ref_state.copy(), ke_before, dof, 0.0, 0.0, dt, host.kT, tau, var)
                assert st.last_alpha == want[0]
            assert dev.instantaneous_reservoir_translational == host.instantaneous_reservoir_translational or dt == 0.0
            assert dev.reservoir_energy_translational == host.reservoir_energy_translational
    torch.cuda.synchronize()
    assert torch.equal(a_vel, b_vel)
    st = dev.device_state()
    assert st.steps == 59 and st.refused == 0                                                 # the dt = 0 step enqueued nothing
    assert dev.total_reservoir_energy == host.total_reservoir_energy != 0.0
    dev.reset_reservoir_energy()
    assert dev.total_reservoir_energy == 0.0 and dev.device_state().steps == 0


@pytest.mark.gpu
def test_on_device_step_refuses_zero_momenta_and_keeps_the_array():
    """Degrees of freedom without kinetic energy: the reference throws "requires non-zero initial momenta"
    (src/BussiReservoirThermostat.h:57-61).  On the device the step is refused (alpha = 1, nothing rescaled) and the next read
    of the state reports it once."""
    n = 5_000
    vel = np.zeros((n, 4))
    vel[:, 3] = 2.0
    dvel = torch.from_numpy(vel.copy()).cuda()
    ws = _capi.Workspace(n)
    ws.bussi_step_device(0, dvel.data_ptr(), None, n, 3.0 * n, 0.02, 3e-4, 0.5, 0.3, 7000.0)
    with pytest.raises(_capi.CavmdError) as e:
        ws.bussi_device_read()
    assert e.value.status == _capi.CAVMD_ERR_BAD_PARAMS
    st = ws.bussi_device_read()                                   # reported once
    assert st.refused == 1 and st.steps == 0 and st.last_alpha == 1.0 and st.reservoir_translational == 0.0
    torch.cuda.synchronize()
    assert np.array_equal(dvel.cpu().numpy(), vel)
    # dof == 0: alpha = 1 by the rule itself (:183-184), not an error
    ws.bussi_step_device(0, dvel.data_ptr(), None, n, 0.0, 0.02, 3e-4, 0.5, 0.3, 0.0)
    st = ws.bussi_device_read()
    assert st.steps == 1 and st.last_alpha == 1.0
    # argument validation, as the other entry points
    with pytest.raises(_capi.CavmdError):
        ws.bussi_step_device(0, dvel.data_ptr() + 8, None, n, 3.0, 0.02, 3e-4, 0.5, 0.3, 0.0)
