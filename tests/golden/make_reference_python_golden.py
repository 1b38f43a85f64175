#!/usr/bin/env python3
"""REFERENCE-EXECUTED fixtures for the force formulas and the observables next to them.

Run in the BUILD container only (`python tests/golden/make_reference_python_golden.py`); /root/reference does not exist on
the GPU box and nothing of it travels: this script writes OUTPUT NUMBERS (and the seeded inputs they belong to) into
tests/golden/reference_python_golden.npz and nothing else.

What executes: the reference's own files, loaded by path as a package whose __init__ is not run --
    src/cavitymd/cavity_force_python.py   CavityForcePython.set_forces                 (:65-145)
    src/cavitymd/forces.py                CavityForce(force_python=True) -> the same, through the user-facing class
    src/cavitymd/analysis.py              compute_total_dipole_moment (:18-31), compute_density_field (:34-47),
                                          generate_fibonacci_sphere (:50-64), FieldAutocorrelationTracker (wavevectors
                                          :304-311, F(k,t) :359-364), DipoleAutocorrelation (:190-195, :211-240),
                                          CavityModeTracker.compute_cavity_properties (:1324-1368),
                                          EnergyTracker._compute_molecular_kinetic_energy / _compute_cavity_kinetic_energy (:524-598)
    src/cavitymd/simulation.py            AdaptiveTimestepUpdater.act (:33-127): error-tolerance ramp, sum |f_i|/m_i, dt rule
-- on top of tests/stubs/hoomd, which supplies `import hoomd` with CONTAINERS ONLY (base classes, context managers, one
decorator; no arithmetic).  The state/snapshot objects below are likewise plain holders of numpy arrays.  Every number
written out is therefore computed by the reference's own bytes with this container's numpy.

What these fixtures pin: the formulas of the reference's Python force path (energies, molecular and photon forces, the
unwrap + np.dot dipole) and of its observables.  The reference's C++ class stays a restatement (oracle/cavity_ref.c): the
two paths are documented to coincide when there is exactly one cavity particle, of typeid 1, with charge 0,
which is how the "coincide" cases below are built; two cases exercise the documented divergences.  np.dot sums in BLAS
order, not left to right, so the pins on summed quantities hold at the rounding level (tests state the tolerance).
"""
import contextlib
import importlib
import io
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
REF_PKG_DIR = "/root/reference/src/cavitymd"
OUT = os.path.join(HERE, "reference_python_golden.npz")


def load_reference_package():
    """The reference's cavitymd package under the name `refcavitymd`, WITHOUT running its __init__ (which would try the
    compiled module); submodules resolve their relative imports (`from .utils import ...`) through __path__."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "stubs"))
    import hoomd
    assert getattr(hoomd, "IS_STUB", False)
    pkg = types.ModuleType("refcavitymd")
    pkg.__path__ = [REF_PKG_DIR]
    sys.modules["refcavitymd"] = pkg
    mods = {}
    for name in ("utils", "cavity_force_python", "analysis", "simulation", "forces"):
        with contextlib.redirect_stdout(io.StringIO()):
            mods[name] = importlib.import_module("refcavitymd." + name)
        assert mods[name].__file__.startswith(REF_PKG_DIR), mods[name].__file__
    return mods


# ---- holders of numpy arrays (no arithmetic) -------------------------------------------------------------------------
class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class State:
    def __init__(self, position, typeid, image, charge, box, mass=None, velocity=None):
        self.p = _NS(position=position, typeid=typeid, image=image, charge=charge, mass=mass, velocity=velocity)
        self.box = _NS(L=np.asarray(box, dtype=np.float64))
        self._cpp_sys_def = object()
        self._f4 = {}

    def _force4_for(self, force):
        return self._f4.setdefault(id(force), np.full((len(self.p.position), 4), np.nan))

    @property
    @contextlib.contextmanager
    def cpu_local_snapshot(self):
        yield _NS(particles=self.p, global_box=self.box)

    def get_snapshot(self):
        return _NS(particles=self.p)


class Sim:
    def __init__(self, state, hoomd, dt=1.0):
        self.state, self.device, self.timestep = state, hoomd.device.CPU(), 0
        self.operations = _NS(integrator=_NS(dt=dt))


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


# ---- seeded inputs ---------------------------------------------------------------------------------------------------
OMEGAC = 2000.0 / 219474.63
G = 1e-3


def random_case(seed, n, cav_pos="random", box=(40.0, 37.5, 43.25)):
    rng = np.random.default_rng(seed)
    L = np.asarray(box)
    pos = rng.uniform(-0.5, 0.5, (n, 3)) * L
    charge = rng.uniform(-1, 1, n)
    image = rng.integers(-2, 3, (n, 3)).astype(np.int32)
    tid = np.zeros(n, dtype=np.int32)
    cav = {"random": int(rng.integers(0, n)), "first": 0, "last": n - 1}[cav_pos]
    tid[cav] = 1
    charge[cav] = 0.0
    return dict(position=pos, typeid=tid, image=image, charge=charge, box=L)


def diatomic_stand_in(seed, n_mol=250, box_L=40.0):
    """N = 501 in the schema of the missing examples/init-0.gsd (250 diatomics, net-neutral pairs), cavity particle LAST;
    molecules typeid 0 and the cavity particle typeid 1 -- the id the reference's Python force looks for."""
    rng = np.random.default_rng(seed)
    L = np.full(3, box_L)
    centre = rng.uniform(-0.5, 0.5, (n_mol, 3)) * L
    axis = rng.normal(size=(n_mol, 3))
    axis /= np.linalg.norm(axis, axis=1)[:, None]
    bond = np.where(np.arange(n_mol) % 2 == 0, 2.2817, 2.0744)[:, None]
    r = np.empty((2 * n_mol, 3))
    r[0::2] = centre - 0.5 * bond * axis
    r[1::2] = centre + 0.5 * bond * axis
    delta = rng.uniform(0.1, 0.5, n_mol)
    charge = np.empty(2 * n_mol)
    charge[0::2], charge[1::2] = delta, -delta
    r += rng.integers(-1, 2, (2 * n_mol, 3)) * L          # put some molecules in neighbouring images
    q = rng.normal(scale=np.sqrt(3.167e-6 * 100.0 / OMEGAC**2), size=3)
    r = np.vstack([r, q[None, :]])
    charge = np.append(charge, 0.0)
    img = np.floor((r + L / 2) / L)
    pos = r - img * L
    tid = np.zeros(2 * n_mol + 1, dtype=np.int32)
    tid[-1] = 1
    return dict(position=pos, typeid=tid, image=img.astype(np.int32), charge=charge, box=L)


def main():
    ref = load_reference_package()
    import hoomd
    cfp, analysis, simulation, forces = ref["cavity_force_python"], ref["analysis"], ref["simulation"], ref["forces"]
    out = {}
    log = []

    # ---- 1. the force: CavityForcePython.set_forces, driven the way HOOMD's integrator drives a force.Custom --------
    cases = {
        "n3": (random_case(10, 3), OMEGAC, G, 1.0),
        "n50_first": (random_case(11, 50, "first"), OMEGAC, G, 1.0),
        "n501_stand_in": (diatomic_stand_in(12), OMEGAC, G, 1.0),
        "n2000_last": (random_case(13, 2000, "last", box=(215.4, 215.4, 215.4)), OMEGAC, G, 1.0),
        "n257_heavy_photon": (random_case(14, 257), 0.02, 5e-3, 2.5),
        # documented divergence 1: a CHARGED cavity particle enters the fallback's dipole (cavity_force_python.py:101)
        "div_charged_cavity": (None, OMEGAC, G, 1.0),
        # documented divergence 2: TWO particles of typeid 1: the first is the cavity, the second gets a molecular force
        "div_two_cavity_typed": (None, OMEGAC, G, 1.0),
        # no particle of typeid 1: zeros (cavity_force_python.py:78-81)
        "no_cavity": (None, OMEGAC, G, 1.0),
    }
    c = random_case(15, 40)
    c["charge"][np.flatnonzero(c["typeid"] == 1)[0]] = 0.75
    cases["div_charged_cavity"] = (c, OMEGAC, G, 1.0)
    c = random_case(16, 40, "first")
    c["typeid"][17] = 1
    cases["div_two_cavity_typed"] = (c, OMEGAC, G, 1.0)
    c = random_case(17, 12)
    c["typeid"][:] = 0
    cases["no_cavity"] = (c, OMEGAC, G, 1.0)

    for name, (c, omegac, g, phmass) in cases.items():
        st = State(c["position"], c["typeid"], c["image"], c["charge"], c["box"])
        sim = Sim(st, hoomd)
        f, _ = quiet(cfp.CavityForcePython, kvector=[0, 0, 1], couplstr=g, omegac=omegac, phmass=phmass)
        f._attach(sim)
        _, printed = quiet(f._cpp_obj.compute, 0)
        assert "Error in cavity force calculation" not in printed, printed
        f4 = st._force4_for(f)
        # the same evaluation through the user-facing class (forces.py:45-173, force_python=True): must be the same bits
        st2 = State(c["position"], c["typeid"], c["image"], c["charge"], c["box"])
        sim2 = Sim(st2, hoomd)
        top, _ = quiet(forces.CavityForce, kvector=[0, 0, 1], couplstr=g, omegac=omegac, phmass=phmass, force_python=True)
        assert top.implementation == "python"
        quiet(top._attach, sim2)
        _, printed2 = quiet(top._cpp_obj.compute, 0)
        assert "Error in cavity force calculation" not in printed2, printed2
        f4_top = st2._force4_for(top._force_impl)
        assert np.array_equal(f4, f4_top, equal_nan=True), name
        e = np.array([f.harmonic_energy, f.coupling_energy, f.dipole_self_energy])
        assert np.array_equal(e, [top.harmonic_energy, top.coupling_energy, top.dipole_self_energy])
        assert top.total_cavity_energy == f.total_cavity_energy == top.energy
        with st.cpu_local_snapshot as snap:
            dip = analysis.compute_total_dipole_moment(snap)
        for k, v in c.items():
            out[f"force/{name}/{k}"] = v
        out[f"force/{name}/params"] = np.array([omegac, g, phmass, f.K])
        out[f"force/{name}/force"] = f4[:, :3].copy()
        out[f"force/{name}/potential_energy"] = f4[:, 3].copy()
        out[f"force/{name}/energies"] = e
        out[f"force/{name}/total_cavity_energy"] = np.float64(f.total_cavity_energy)
        out[f"force/{name}/total_dipole"] = dip
        log.append(f"set_forces {name}: N={len(c['charge'])} energies={e}")
    out["force/names"] = np.array(list(cases.keys()))

    # ---- 2. fibonacci sphere + density field -------------------------------------------------------------------------
    for n in (2, 10, 50, 100):
        out[f"fibonacci/{n}"] = analysis.generate_fibonacci_sphere(n)
    c = cases["n501_stand_in"][0]
    frames = [c["position"]]
    rng = np.random.default_rng(18)
    for _ in range(3):
        frames.append(frames[-1] + 0.05 * rng.normal(size=frames[-1].shape))
    out["trajectory/frames"] = np.array(frames)          # wrapped positions of 4 frames; typeid/image/charge/box as n501_stand_in
    tmp = tempfile.mkdtemp(prefix="refgolden_")
    cwd = os.getcwd()
    os.chdir(tmp)                          # the trackers write their text files into the working directory
    try:
        for kmag, nk in ((1.0, 50), (0.35, 17), (2.5, 64)):
            st = State(frames[0], c["typeid"], c["image"], c["charge"], c["box"])
            sim = Sim(st, hoomd, dt=2.0)
            tr, _ = quiet(analysis.FieldAutocorrelationTracker, sim, "density_correlation", kmag=kmag, num_wavevectors=nk,
                          output_period_steps=1, max_references=1)
            key = f"density/k{kmag}_n{nk}"
            out[key + "/wavevectors"] = np.array(tr.wavevectors)
            fields, fkt = [np.array(tr.references[0]["field"])], [np.nan]
            for t in range(1, len(frames)):
                st.p.position = frames[t]
                sim.timestep = t
                quiet(tr.act, t)
                with st.cpu_local_snapshot as snap:
                    fields.append(analysis.compute_density_field(snap, tr.wavevectors))
                fkt.append(tr.current_autocorr_value)
            out[key + "/rho_k"] = np.array(fields)
            out[key + "/F_kt"] = np.array(fkt)
            log.append(f"density {key}: |rho_k[0][:3]| = {np.abs(fields[0][:3])}")
        # a bigger, denser field: 2000 particles in the 215.4 box, default 50 wavevectors at |k| = 1
        c2 = cases["n2000_last"][0]
        with State(c2["position"], c2["typeid"], c2["image"], c2["charge"], c2["box"]).cpu_local_snapshot as snap:
            wv = analysis.generate_fibonacci_sphere(50) * 1.0
            out["density/n2000/wavevectors"] = wv
            out["density/n2000/rho_k"] = analysis.compute_density_field(snap, wv)

        # ---- 3. dipole autocorrelation C(t) = d(0).d(t) ---------------------------------------------------------------
        st = State(frames[0], c["typeid"], c["image"], c["charge"], c["box"])
        sim = Sim(st, hoomd, dt=2.0)
        da, _ = quiet(analysis.DipoleAutocorrelation, sim, output_period_steps=1)
        dip_t, c_t = [np.array(da.reference_value)], [da.current_autocorr_value]
        for t in range(1, len(frames)):
            st.p.position = frames[t]
            quiet(da.act, t)
            with st.cpu_local_snapshot as snap:
                dip_t.append(analysis.compute_total_dipole_moment(snap))
            c_t.append(da.current_autocorr_value)
        out["dipole_acf/dipole_t"] = np.array(dip_t)
        out["dipole_acf/C_t"] = np.array(c_t)

        # ---- 4. cavity mode (photon = typeid 2 here: analysis.py:1329) -----------------------------------------------
        rng = np.random.default_rng(19)
        for i, (n, where) in enumerate(((501, 500), (64, 0), (33, 20))):
            cm = random_case(20 + i, n)
            tid = np.zeros(n, dtype=np.int32)
            tid[where] = 2
            mass = rng.uniform(1.0, 30.0, n)
            mass[where] = [1.0, 2.5, 0.125][i]
            vel = rng.normal(scale=1e-2, size=(n, 3))
            st = State(cm["position"], tid, cm["image"], cm["charge"], cm["box"], mass=mass, velocity=vel)
            sim = Sim(st, hoomd)
            harmonic = [3.25e-4, 0.0, 1.5][i]
            tr, _ = quiet(analysis.CavityModeTracker, sim, _NS(harmonic_energy=harmonic))
            (props, _) = quiet(tr.compute_cavity_properties)
            out[f"cavity_mode/{i}/typeid"], out[f"cavity_mode/{i}/mass"], out[f"cavity_mode/{i}/velocity"] = tid, mass, vel
            out[f"cavity_mode/{i}/harmonic_energy"] = np.float64(harmonic)
            out[f"cavity_mode/{i}/properties"] = np.array(props)       # kinetic, potential, total, temperature
        # no photon at all -> four zeros (analysis.py:1331-1332)
        st = State(cm["position"], np.zeros(n, dtype=np.int32), cm["image"], cm["charge"], cm["box"], mass=mass, velocity=vel)
        tr, _ = quiet(analysis.CavityModeTracker, Sim(st, hoomd), _NS(harmonic_energy=1.0))
        out["cavity_mode/no_photon/properties"] = np.array(quiet(tr.compute_cavity_properties)[0])
    finally:
        os.chdir(cwd)
        for fn in os.listdir(tmp):
            os.remove(os.path.join(tmp, fn))
        os.rmdir(tmp)

    # ---- 4b. EnergyTracker's internal kinetic energies (analysis.py:524-598): molecular = typeid != 2, cavity = typeid == 2 ----
    # (the methods are run on a plain holder carrying the three attributes they read; the tracker's constructor only opens
    # text files)
    rng = np.random.default_rng(25)
    for i, (n, where) in enumerate(((501, 500), (4096, 17), (8193, 8192))):
        tid = (np.arange(n) % 2).astype(np.int32)
        tid[where] = 2
        mass = np.where(tid == 0, 15.999, 14.007) * 1822.888
        mass[where] = 1.0
        vel = rng.normal(scale=1e-3, size=(n, 3))
        st = State(np.zeros((n, 3)), tid, np.zeros((n, 3), dtype=np.int32), np.zeros(n), (1, 1, 1), mass=mass, velocity=vel)
        holder = _NS(sim=Sim(st, hoomd), verbose="normal", cavity_mode_tracker=None)
        (ke_mol, temp), printed = quiet(analysis.EnergyTracker._compute_molecular_kinetic_energy, holder)
        assert "Error" not in printed, printed
        ke_cav, printed = quiet(analysis.EnergyTracker._compute_cavity_kinetic_energy, holder)
        assert "Error" not in printed, printed
        out[f"kinetic/{i}/typeid"], out[f"kinetic/{i}/mass"], out[f"kinetic/{i}/velocity"] = tid, mass, vel
        out[f"kinetic/{i}/molecular"] = np.array([ke_mol, temp])
        out[f"kinetic/{i}/cavity"] = np.float64(ke_cav)
        log.append(f"kinetic {i}: N={n} KE_mol={ke_mol:.9e} T={temp:.6f} KE_cav={ke_cav:.9e}")

    # ---- 5. adaptive timestep: error-tolerance ramp, sum |f_i| / m_i, dt = sqrt(tol / S) -------------------------------
    rng = np.random.default_rng(30)
    for i, n in enumerate((5, 501, 4096)):
        mass = rng.uniform(1.0, 30.0, n)
        fa = rng.normal(scale=1e-3, size=(n, 3))
        fb = rng.normal(scale=1e-4, size=(n, 3))
        st = State(np.zeros((n, 3)), np.zeros(n, dtype=np.int32), np.zeros((n, 3), dtype=np.int32), np.zeros(n), (1, 1, 1),
                   mass=mass)
        thermo_m, thermo_c = _NS(tau=None), _NS(tau=None)
        integ = _NS(dt=0.5, forces=[_NS(forces=fa), _NS(forces=fb), _NS(forces=None)],
                    methods=[_NS(thermostat=thermo_m), _NS(thermostat=thermo_c)])
        tt = _NS(elapsed_time=[0.0, 12.5, 400.0][i])
        upd, _ = quiet(simulation.AdaptiveTimestepUpdater, st, integ, error_tolerance=1e-3, time_constant_ps=50.0,
                       initial_fraction=0.01, adaptiveerror=True, molecular_thermostat_tau=5.0, cavity_thermostat_tau=0.5,
                       time_tracker=tt)
        quiet(upd.act, 7)
        out[f"adaptive_dt/{i}/mass"], out[f"adaptive_dt/{i}/force_a"], out[f"adaptive_dt/{i}/force_b"] = mass, fa, fb
        out[f"adaptive_dt/{i}/elapsed_ps"] = np.float64(tt.elapsed_time)
        out[f"adaptive_dt/{i}/error_tolerance"] = np.float64(upd.current_error_tolerance)
        out[f"adaptive_dt/{i}/dt"] = np.float64(integ.dt)
        out[f"adaptive_dt/{i}/tau"] = np.array([thermo_m.tau, thermo_c.tau])
        log.append(f"adaptive dt {i}: N={n} tol={upd.current_error_tolerance:.6e} dt={integ.dt:.9f}")
    # all-zero forces: dt untouched (simulation.py:88)
    integ = _NS(dt=0.5, forces=[_NS(forces=np.zeros((4, 3)))], methods=[_NS(thermostat=_NS(tau=None))])
    st = State(np.zeros((4, 3)), np.zeros(4, dtype=np.int32), np.zeros((4, 3), dtype=np.int32), np.zeros(4), (1, 1, 1),
               mass=np.ones(4))
    upd, _ = quiet(simulation.AdaptiveTimestepUpdater, st, integ, error_tolerance=1e-3, adaptiveerror=False)
    quiet(upd.act, 1)
    out["adaptive_dt/zero_force/dt"] = np.float64(integ.dt)

    out["generated_by"] = np.array("tests/golden/make_reference_python_golden.py executing /root/reference/src/cavitymd/"
                                   "{cavity_force_python,forces,analysis,simulation}.py on tests/stubs/hoomd (containers "
                                   f"only); numpy {np.__version__}")
    np.savez_compressed(OUT, **out)
    print("\n".join(log))
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
