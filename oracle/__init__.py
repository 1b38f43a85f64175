"""CPU oracle for the cavity-force hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this
package.  The product (``cav-hoomd_amd/``) never does and fails loudly without its HIP library.

Contents
    cavity_ref.c   C restatement of the reference CPU path (src/CavityForceCompute.cc:73-208 of
                   muhammadhasyim/cav-hoomd), built by ``oracle/Makefile`` into ``libcavref.so``.
    RefOracle      ctypes front end to it (this file).
    bussi_ref.c    C restatement of the Bussi reservoir thermostat step with injected variates
                   (src/BussiReservoirThermostat.h:43-98, 177-225) -> ``libbussiref.so`` / BussiOracle.
    numpy_mirror   second, independent restatement in numpy (vectorised maths, sequential sums
                   via math.fsum-free python loops only for tiny N) + exactly rounded sums.

Pinning status: PARITY UNPINNED by the reference's own tests (it has none for this path, and it
cannot be built or imported here because HOOMD-blue is absent).  See the header of cavity_ref.c
for what pins the oracle instead.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# numpy views of the HOOMD layouts (double build)
SCALAR4 = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("w", "<f8")])
INT3 = np.dtype([("x", "<i4"), ("y", "<i4"), ("z", "<i4")])


class _RefParams(ctypes.Structure):
    _fields_ = [("omegac", ctypes.c_double), ("couplstr", ctypes.c_double), ("K", ctypes.c_double),
                ("phmass", ctypes.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent).  Returns the path of libcavref.so."""
    so = os.path.join(_HERE, "libcavref.so")
    src = os.path.join(_HERE, "cavity_ref.c")
    bso, bsrc = os.path.join(_HERE, "libbussiref.so"), os.path.join(_HERE, "bussi_ref.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src) \
        or not os.path.exists(os.path.join(_HERE, "libcavref_O3.so")) or not os.path.exists(os.path.join(_HERE, "libcavomp.so")) \
        or not os.path.exists(bso) or os.path.getmtime(bso) < os.path.getmtime(bsrc)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)
    return so


def type_tag_as_double(typeid) -> np.ndarray:
    """pos.w for a given type id: HOOMD's __int_as_scalar puts the int in the low 4 bytes of the double."""
    t = np.asarray(typeid, dtype=np.int64) & 0xFFFFFFFF
    return t.astype(np.uint64).view(np.float64)


def pack_pos(position: np.ndarray, typeid: np.ndarray) -> np.ndarray:
    """(N,3) positions + (N,) type ids -> (N,4) float64 array laid out like HOOMD's Scalar4 pos."""
    n = position.shape[0]
    out = np.empty((n, 4), dtype=np.float64)
    out[:, :3] = position
    out[:, 3] = type_tag_as_double(typeid)
    return out


class RefOracle:
    """ctypes front end to libcavref.so (optimisation level 'O2' or 'O3')."""

    def __init__(self, opt: str = "O2"):
        build()
        name = "libcavref.so" if opt == "O2" else "libcavref_O3.so"
        self.opt = opt
        self.lib = ctypes.CDLL(os.path.join(_HERE, name))
        L = self.lib
        vp, dbl, ci, cu = ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_uint
        L.cavref_make_params.argtypes = [dbl, dbl, dbl, ctypes.POINTER(_RefParams)]
        L.cavref_make_params.restype = None
        L.cavref_find_photon.argtypes = [vp, cu, ci]
        L.cavref_find_photon.restype = ci
        L.cavref_unwrap.argtypes = [vp, vp, vp, dbl, dbl, dbl, cu]
        L.cavref_unwrap.restype = None
        L.cavref_dipole.argtypes = [vp, vp, cu, ci, vp]
        L.cavref_dipole.restype = None
        L.cavref_compute_forces.argtypes = [cu, vp, vp, vp, dbl, dbl, dbl, ci, ctypes.POINTER(_RefParams), vp, vp,
                                            vp, vp]
        L.cavref_compute_forces.restype = ci
        L.cavref_dipole_exact.argtypes = [cu, vp, vp, vp, dbl, dbl, dbl, ci, vp, vp]
        L.cavref_dipole_exact.restype = None
        L.cavref_time_evaluations.argtypes = [cu, vp, vp, vp, dbl, dbl, dbl, ci, ctypes.POINTER(_RefParams), vp, ci]
        L.cavref_time_evaluations.restype = dbl
        L.cavref_layout_sizes.argtypes = [vp]
        L.cavref_layout_sizes.restype = ci

    # -- helpers ------------------------------------------------------------------------------
    @staticmethod
    def _c(a: np.ndarray, dtype) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=dtype)
        return a

    def make_params(self, omegac: float, couplstr: float, phmass: float = 1.0) -> dict:
        p = _RefParams()
        self.lib.cavref_make_params(omegac, couplstr, phmass, ctypes.byref(p))
        return {"omegac": p.omegac, "couplstr": p.couplstr, "K": p.K, "phmass": p.phmass}

    def _params(self, params: dict) -> _RefParams:
        return _RefParams(params["omegac"], params["couplstr"], params["K"], params["phmass"])

    def layout_sizes(self):
        out = np.zeros(4, dtype=np.int32)
        self.lib.cavref_layout_sizes(out.ctypes.data)
        return tuple(int(v) for v in out)

    # -- the path -------------------------------------------------------------------------------
    def find_photon(self, pos4: np.ndarray, L_typeid: int) -> int:
        pos4 = self._c(pos4, np.float64)
        return int(self.lib.cavref_find_photon(pos4.ctypes.data, pos4.shape[0], int(L_typeid)))

    def unwrap(self, pos4, image, box_L) -> np.ndarray:
        pos4 = self._c(pos4, np.float64)
        image = self._c(image, np.int32)
        n = pos4.shape[0]
        out = np.empty((n, 3), dtype=np.float64)
        self.lib.cavref_unwrap(out.ctypes.data, pos4.ctypes.data, image.ctypes.data, float(box_L[0]), float(box_L[1]),
                               float(box_L[2]), n)
        return out

    def compute(self, pos4, charge, image, box_L, L_typeid: int, params: dict) -> dict:
        """One evaluation.  pos4 (N,4) f64 with type tags in column 3, charge (N,), image (N,3) i32."""
        pos4 = self._c(pos4, np.float64)
        charge = self._c(charge, np.float64)
        image = self._c(image, np.int32)
        n = pos4.shape[0]
        assert pos4.shape == (n, 4) and charge.shape == (n,) and image.shape == (n, 3)
        force = np.full((n, 4), np.nan, dtype=np.float64)  # the oracle must overwrite every entry
        energies = np.zeros(3, dtype=np.float64)
        dipole = np.zeros(3, dtype=np.float64)
        pidx = ctypes.c_int(-2)
        rc = self.lib.cavref_compute_forces(n, pos4.ctypes.data, charge.ctypes.data, image.ctypes.data, float(box_L[0]),
                                            float(box_L[1]), float(box_L[2]), int(L_typeid),
                                            ctypes.byref(self._params(params)), force.ctypes.data, energies.ctypes.data,
                                            dipole.ctypes.data, ctypes.addressof(pidx))
        if rc != 0:
            raise MemoryError("oracle could not allocate its temporary array")
        return {"force": force, "energies": energies, "dipole": dipole, "photon_idx": int(pidx.value)}

    def dipole_exact(self, pos4, charge, image, box_L, photon_idx: int):
        """Exactly rounded sum of the same fp64 terms (double-double): returns (hi[3], lo[3])."""
        pos4 = self._c(pos4, np.float64)
        charge = self._c(charge, np.float64)
        image = self._c(image, np.int32)
        hi = np.zeros(3)
        lo = np.zeros(3)
        self.lib.cavref_dipole_exact(pos4.shape[0], pos4.ctypes.data, charge.ctypes.data, image.ctypes.data,
                                     float(box_L[0]), float(box_L[1]), float(box_L[2]), int(photon_idx),
                                     hi.ctypes.data, lo.ctypes.data)
        return hi, lo

    def time_evaluations(self, pos4, charge, image, box_L, L_typeid: int, params: dict, iters: int) -> float:
        """Seconds for `iters` back-to-back evaluations on one host thread (bench.py cpu_baseline)."""
        pos4 = self._c(pos4, np.float64)
        charge = self._c(charge, np.float64)
        image = self._c(image, np.int32)
        force = np.empty((pos4.shape[0], 4), dtype=np.float64)
        return float(
            self.lib.cavref_time_evaluations(pos4.shape[0], pos4.ctypes.data, charge.ctypes.data, image.ctypes.data,
                                             float(box_L[0]), float(box_L[1]), float(box_L[2]), int(L_typeid),
                                             ctypes.byref(self._params(params)), force.ctypes.data, int(iters)))


class BussiOracle:
    """ctypes front end to libbussiref.so: the reference's Bussi reservoir step with injected variates
    (src/BussiReservoirThermostat.h:43-98, 177-225) and the kinetic energy it consumes."""

    def __init__(self):
        build()
        self.lib = ctypes.CDLL(os.path.join(_HERE, "libbussiref.so"))
        dbl, vp, sz = ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
        self.lib.bussiref_rescale_factor.argtypes = [dbl] * 7
        self.lib.bussiref_rescale_factor.restype = dbl
        self.lib.bussiref_step.argtypes = [vp, dbl, dbl, dbl, dbl, dbl, dbl, dbl, vp, vp]
        self.lib.bussiref_step.restype = ctypes.c_int
        self.lib.bussiref_kinetic_energy.argtypes = [vp, vp, sz]
        self.lib.bussiref_kinetic_energy.restype = dbl
        self.lib.bussiref_kinetic_energy_exact.argtypes = [vp, vp, sz, vp]
        self.lib.bussiref_kinetic_energy_exact.restype = dbl

    def rescale_factor(self, K, dof, deltaT, set_T, tau, r_normal, gamma_variate) -> float:
        return float(self.lib.bussiref_rescale_factor(K, dof, deltaT, set_T, tau, r_normal, gamma_variate))

    def step(self, state: np.ndarray, K_trans, dof_trans, K_rot, dof_rot, deltaT, set_T, tau, variates):
        """state: float64[4], updated in place.  Returns (alpha_translational, alpha_rotational) or raises where the
        reference throws."""
        v = np.ascontiguousarray(variates, dtype=np.float64)
        f = np.zeros(2)
        assert state.dtype == np.float64 and state.shape == (4,) and state.flags.c_contiguous
        rc = self.lib.bussiref_step(state.ctypes.data, K_trans, dof_trans, K_rot, dof_rot, deltaT, set_T, tau, v.ctypes.data,
                                    f.ctypes.data)
        if rc != 0:
            raise RuntimeError("Bussi thermostat requires non-zero initial momenta.")
        return float(f[0]), float(f[1])

    def kinetic_energy(self, vel4: np.ndarray, members=None, exact: bool = False):
        vel4 = np.ascontiguousarray(vel4, dtype=np.float64)
        if members is None:
            mp, n = None, vel4.shape[0]
        else:
            members = np.ascontiguousarray(members, dtype=np.uint32)
            mp, n = members.ctypes.data, members.shape[0]
        if exact:
            lo = ctypes.c_double()
            hi = self.lib.bussiref_kinetic_energy_exact(vel4.ctypes.data, mp, n, ctypes.addressof(lo))
            return float(hi), float(lo.value)
        return float(self.lib.bussiref_kinetic_energy(vel4.ctypes.data, mp, n))


def allowed_cpus() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(quota / int(g.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


class AllCoresCourtesy:
    """OpenMP variant of the same passes (oracle/cavity_omp.c): a courtesy 'every core of the host' figure for bench.py.
    NOT the reference's algorithm (parallel reduction order) and never used as an oracle."""

    def __init__(self):
        build()
        # libgomp reads these when it is loaded: idle threads must sleep, not spin (shared hosts are oversubscribed)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("OMP_PROC_BIND", "false")
        os.environ.setdefault("OMP_NUM_THREADS", str(min(allowed_cpus(), 64)))
        self.lib = ctypes.CDLL(os.path.join(_HERE, "libcavomp.so"))
        vp, dbl, ci, cu = ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_uint
        self.lib.cavomp_threads.restype = ci
        self.lib.cavomp_set_threads.argtypes = [ci]
        self.lib.cavomp_set_threads.restype = None
        # libgomp may already have been initialised by another library (torch): size the team through the API as well
        self.lib.cavomp_set_threads(min(allowed_cpus(), 64))
        self.lib.cavomp_compute.argtypes = [cu, vp, vp, vp, dbl, dbl, dbl, ci, ctypes.POINTER(_RefParams), vp, vp]
        self.lib.cavomp_compute.restype = None
        self.lib.cavomp_time.argtypes = [cu, vp, vp, vp, dbl, dbl, dbl, ci, ctypes.POINTER(_RefParams), vp, ci]
        self.lib.cavomp_time.restype = dbl

    def threads(self) -> int:
        return int(self.lib.cavomp_threads())

    def compute(self, pos4, charge, image, box_L, L_typeid, params):
        pos4 = np.ascontiguousarray(pos4, dtype=np.float64)
        charge = np.ascontiguousarray(charge, dtype=np.float64)
        image = np.ascontiguousarray(image, dtype=np.int32)
        force = np.empty_like(pos4)
        e = np.zeros(3)
        p = _RefParams(params["omegac"], params["couplstr"], params["K"], params["phmass"])
        self.lib.cavomp_compute(pos4.shape[0], pos4.ctypes.data, charge.ctypes.data, image.ctypes.data, float(box_L[0]),
                                float(box_L[1]), float(box_L[2]), int(L_typeid), ctypes.byref(p), force.ctypes.data,
                                e.ctypes.data)
        return {"force": force, "energies": e}

    def time_evaluations(self, pos4, charge, image, box_L, L_typeid, params, iters: int) -> float:
        pos4 = np.ascontiguousarray(pos4, dtype=np.float64)
        charge = np.ascontiguousarray(charge, dtype=np.float64)
        image = np.ascontiguousarray(image, dtype=np.int32)
        force = np.empty_like(pos4)
        p = _RefParams(params["omegac"], params["couplstr"], params["K"], params["phmass"])
        return float(self.lib.cavomp_time(pos4.shape[0], pos4.ctypes.data, charge.ctypes.data, image.ctypes.data,
                                          float(box_L[0]), float(box_L[1]), float(box_L[2]), int(L_typeid),
                                          ctypes.byref(p), force.ctypes.data, int(iters)))
