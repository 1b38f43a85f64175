"""ctypes binding of the C ABI declared in ``include/cavmd.h`` (``libcavmd.so``).

This is the only place where Python meets the HIP library.  There is deliberately no CPU or
PyTorch fallback: if the shared library is missing, cannot be loaded, or finds no HIP device,
the caller gets an exception that says so.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
# in-tree build (this repository) first; next to this file when installed into HOOMD's python tree by
# csrc/hoomd_shim/CMakeLists.txt
_INSTALLED = os.path.join(_HERE, "libcavmd.so")
LIB_PATH = _INSTALLED if (os.path.exists(_INSTALLED) and not os.path.isdir(CSRC_DIR)) else os.path.join(CSRC_DIR, "libcavmd.so")

CAVMD_OK = 0
CAVMD_ERR_INVALID_VALUE = -1
CAVMD_ERR_NO_DEVICE = -2
CAVMD_ERR_CAPACITY = -3
CAVMD_ERR_BAD_PARAMS = -4
CAVMD_ERR_NOT_COMPUTED = -5
CAVMD_ERR_SYNC_TIMEOUT = -6


class CavmdError(RuntimeError):
    """A non-zero status from libcavmd (the reference throws std::runtime_error in the same places,
    src/CavityForceComputeGPU.cc:106-109, 188-192)."""

    def __init__(self, status: int, message: str, where: str = ""):
        self.status = status
        super().__init__(f"libcavmd {where}: [{status}] {message}")


class Params(ctypes.Structure):
    """cavmd_params == the reference's cavity_force_params (src/CavityForceCompute.h:28-54)."""
    _fields_ = [("omegac", ctypes.c_double), ("couplstr", ctypes.c_double), ("K", ctypes.c_double),
                ("phmass", ctypes.c_double)]

    def as_dict(self) -> dict:
        return {"omegac": self.omegac, "couplstr": self.couplstr, "K": self.K, "phmass": self.phmass}


class Result(ctypes.Structure):
    """cavmd_result (192 bytes)."""
    _fields_ = [("dipole", ctypes.c_double * 3), ("q", ctypes.c_double * 3), ("Dq", ctypes.c_double * 2),
                ("energy", ctypes.c_double * 3), ("photon_force", ctypes.c_double * 3),
                ("dipole_lo", ctypes.c_double * 3), ("photon_idx", ctypes.c_int32),
                ("n_photon_typed", ctypes.c_int32), ("n_particles", ctypes.c_uint32),
                ("n_partials", ctypes.c_uint32), ("sequence", ctypes.c_uint64), ("total_dipole", ctypes.c_double * 3),
                ("reserved", ctypes.c_double)]


class BussiReservoirState(ctypes.Structure):
    """cavmd_bussi_reservoir (src/BussiReservoirThermostat.h:160-165)."""
    _fields_ = [("reservoir_translational", ctypes.c_double), ("reservoir_rotational", ctypes.c_double),
                ("instantaneous_translational", ctypes.c_double), ("instantaneous_rotational", ctypes.c_double)]


class BussiDeviceState(ctypes.Structure):
    """cavmd_bussi_device_state: the on-device thermostat's counters after its last step."""
    _fields_ = [("reservoir_translational", ctypes.c_double), ("instantaneous_translational", ctypes.c_double),
                ("last_alpha", ctypes.c_double), ("last_kinetic_energy", ctypes.c_double), ("steps", ctypes.c_uint64),
                ("refused", ctypes.c_uint64)]


# every symbol include/cavmd.h exports; tests check the header and the library against this list
EXPORTED_SYMBOLS = (
    "cavmd_make_params", "cavmd_create", "cavmd_destroy", "cavmd_compute_hoomd", "cavmd_compute_soa",
    "cavmd_energies", "cavmd_result_read", "cavmd_result_device_ptr", "cavmd_set_wavevectors", "cavmd_density_field",
    "cavmd_density_field_read", "cavmd_cavity_mode", "cavmd_force_mass_sum", "cavmd_kinetic_energy", "cavmd_scale_velocities",
    "cavmd_bussi_step_device", "cavmd_bussi_device_read", "cavmd_bussi_device_reset",
    "cavmd_bussi_rescale_factor", "cavmd_bussi_step", "cavmd_profile_enable", "cavmd_profile_read", "cavmd_profile_samples",
    "cavmd_set_tunable", "cavmd_get_tunable", "cavmd_device_info", "cavmd_error_string", "cavmd_version",
)

_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile libcavmd.so for gfx950 with hipcc (cross-compiles without a GPU).  Idempotent."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".hpp")) and not f.startswith("microbench")]
    srcs.append(os.path.normpath(os.path.join(CSRC_DIR, "..", "..", "include", "cavmd.h")))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", CSRC_DIR, "-s", "libcavmd.so"] + (["-B"] if force else []), check=True)
    # the same library with the test hooks compiled in (fault injection; loaded by tests only, see load_hooks_build)
    subprocess.run(["make", "-C", CSRC_DIR, "-s", "libcavmd_hooks.so"], check=True)
    # the pybind11 flavour of the shim (cavitymd._cavitymd); make rebuilds it only when stale
    subprocess.run(["make", "-C", CSRC_DIR, "-s", "pymod"], check=True)
    return LIB_PATH


def load():
    """Load libcavmd.so and declare its prototypes.  Raises if the library is not there."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC_DIR}` (or `python -c 'import __graft_entry__ as g; "
                "g.build()'`).  The cavity force has no CPU/PyTorch fallback in this package.")
        _lib = _declare(ctypes.CDLL(LIB_PATH))
        return _lib


_hooks_lib = None
HOOKS_LIB_PATH = os.path.join(CSRC_DIR, "libcavmd_hooks.so")


def load_hooks_build():
    """TESTS ONLY: libcavmd_hooks.so, the same sources compiled with -DCAVMD_TEST_HOOKS (fault-injecting instantiations of the
    single-launch kernel and the debug_* tunables that drive them).  The product library has neither."""
    global _hooks_lib
    with _lock:
        if _hooks_lib is None:
            if not os.path.exists(HOOKS_LIB_PATH):
                raise ImportError(f"{HOOKS_LIB_PATH} is missing: build it with `make -C {CSRC_DIR} hooks`")
            _hooks_lib = _declare(ctypes.CDLL(HOOKS_LIB_PATH))
        return _hooks_lib


def _declare(lib):
    if True:
        vp, sz, dbl, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int
        P = ctypes.POINTER
        lib.cavmd_make_params.argtypes = [dbl, dbl, dbl]
        lib.cavmd_make_params.restype = Params
        lib.cavmd_create.argtypes = [ci, sz, P(vp)]
        lib.cavmd_create.restype = ci
        lib.cavmd_destroy.argtypes = [vp]
        lib.cavmd_destroy.restype = ci
        lib.cavmd_compute_hoomd.argtypes = [vp, vp, sz, vp, vp, vp, dbl, dbl, dbl, ci, P(Params), vp]
        lib.cavmd_compute_hoomd.restype = ci
        lib.cavmd_compute_soa.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, dbl, dbl, dbl, ci, P(Params), vp,
                                          sz, vp, sz]
        lib.cavmd_compute_soa.restype = ci
        lib.cavmd_energies.argtypes = [vp, P(dbl * 3)]
        lib.cavmd_energies.restype = ci
        lib.cavmd_result_read.argtypes = [vp, P(Result)]
        lib.cavmd_result_read.restype = ci
        lib.cavmd_result_device_ptr.argtypes = [vp, P(vp)]
        lib.cavmd_result_device_ptr.restype = ci
        lib.cavmd_set_wavevectors.argtypes = [vp, sz, vp]
        lib.cavmd_set_wavevectors.restype = ci
        lib.cavmd_density_field.argtypes = [vp, vp, sz, vp, sz]
        lib.cavmd_density_field.restype = ci
        lib.cavmd_density_field_read.argtypes = [vp, vp]
        lib.cavmd_density_field_read.restype = ci
        lib.cavmd_cavity_mode.argtypes = [vp, vp, vp, dbl, P(dbl * 4)]
        lib.cavmd_cavity_mode.restype = ci
        lib.cavmd_force_mass_sum.argtypes = [vp, vp, sz, vp, vp, P(dbl)]
        lib.cavmd_force_mass_sum.restype = ci
        lib.cavmd_kinetic_energy.argtypes = [vp, vp, vp, vp, sz, P(dbl)]
        lib.cavmd_kinetic_energy.restype = ci
        lib.cavmd_scale_velocities.argtypes = [vp, vp, vp, vp, sz, dbl]
        lib.cavmd_scale_velocities.restype = ci
        lib.cavmd_bussi_rescale_factor.argtypes = [dbl] * 7 + [P(dbl)]
        lib.cavmd_bussi_rescale_factor.restype = ci
        lib.cavmd_bussi_step.argtypes = [P(BussiReservoirState), dbl, dbl, dbl, dbl, dbl, dbl, dbl, P(dbl * 4), P(dbl * 2)]
        lib.cavmd_bussi_step.restype = ci
        lib.cavmd_bussi_step_device.argtypes = [vp, vp, vp, vp, sz] + [dbl] * 6
        lib.cavmd_bussi_step_device.restype = ci
        lib.cavmd_bussi_device_read.argtypes = [vp, P(BussiDeviceState)]
        lib.cavmd_bussi_device_read.restype = ci
        lib.cavmd_bussi_device_reset.argtypes = [vp, vp]
        lib.cavmd_bussi_device_reset.restype = ci
        lib.cavmd_profile_enable.argtypes = [vp, ci]
        lib.cavmd_profile_enable.restype = ci
        lib.cavmd_profile_read.argtypes = [vp, P(dbl * 3), P(ctypes.c_uint64)]
        lib.cavmd_profile_read.restype = ci
        lib.cavmd_profile_samples.argtypes = [vp, vp, sz, P(sz)]
        lib.cavmd_profile_samples.restype = ci
        lib.cavmd_set_tunable.argtypes = [vp, ctypes.c_char_p, ci]
        lib.cavmd_set_tunable.restype = ci
        lib.cavmd_get_tunable.argtypes = [vp, ctypes.c_char_p, P(ci)]
        lib.cavmd_get_tunable.restype = ci
        lib.cavmd_device_info.argtypes = [vp, P(ci), P(ci), ctypes.c_char_p, sz]
        lib.cavmd_device_info.restype = ci
        lib.cavmd_error_string.argtypes = [ci]
        lib.cavmd_error_string.restype = ctypes.c_char_p
        lib.cavmd_version.argtypes = []
        lib.cavmd_version.restype = ci
        return lib


def error_string(status: int) -> str:
    return load().cavmd_error_string(int(status)).decode()


def check(status: int, where: str = "") -> None:
    if status != CAVMD_OK:
        raise CavmdError(status, error_string(status), where)


def bussi_rescale_factor(K, degrees_of_freedom, deltaT, set_T, tau, normal_variate, gamma_variate) -> float:
    out = ctypes.c_double()
    check(load().cavmd_bussi_rescale_factor(float(K), float(degrees_of_freedom), float(deltaT), float(set_T), float(tau),
                                            float(normal_variate), float(gamma_variate), ctypes.byref(out)),
          "cavmd_bussi_rescale_factor")
    return float(out.value)


def bussi_step(state: BussiReservoirState, K_trans, dof_trans, K_rot, dof_rot, deltaT, set_T, tau, variates):
    v = (ctypes.c_double * 4)(*[float(x) for x in variates])
    f = (ctypes.c_double * 2)()
    check(load().cavmd_bussi_step(ctypes.byref(state), float(K_trans), float(dof_trans), float(K_rot), float(dof_rot),
                                  float(deltaT), float(set_T), float(tau), ctypes.byref(v), ctypes.byref(f)), "cavmd_bussi_step")
    return float(f[0]), float(f[1])


def make_params(omegac: float, couplstr: float, phmass: float = 1.0) -> Params:
    return load().cavmd_make_params(float(omegac), float(couplstr), float(phmass))


class Workspace:
    """Owns one cavmd_workspace (scratch for partial sums + the 192-byte result block)."""

    def __init__(self, max_N: int, device: int = -1, hooks: bool = False):
        self._lib = load_hooks_build() if hooks else load()
        self._h = ctypes.c_void_p()
        check(self._lib.cavmd_create(int(device), int(max_N), ctypes.byref(self._h)), "cavmd_create")
        self.max_N = int(max_N)

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.cavmd_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hot path -----------------------------------------------------------------------------------
    def compute_hoomd(self, stream: int, N: int, pos_ptr: int, charge_ptr: int, image_ptr: int, box_L, L_typeid: int,
                      params: Params, force_ptr: int) -> None:
        check(
            self._lib.cavmd_compute_hoomd(self._h, ctypes.c_void_p(stream), int(N), ctypes.c_void_p(pos_ptr),
                                          ctypes.c_void_p(charge_ptr), ctypes.c_void_p(image_ptr), float(box_L[0]),
                                          float(box_L[1]), float(box_L[2]), int(L_typeid), ctypes.byref(params),
                                          ctypes.c_void_p(force_ptr)), "cavmd_compute_hoomd")

    def compute_soa(self, stream: int, N: int, position, typeid, image, charge, box_L, L_typeid: int, params: Params,
                    force, potential_energy=None) -> None:
        """Each array argument is a (device_pointer, stride_in_bytes) pair."""
        pe_ptr, pe_stride = potential_energy if potential_energy is not None else (None, 0)
        check(
            self._lib.cavmd_compute_soa(self._h, ctypes.c_void_p(stream), int(N), ctypes.c_void_p(position[0]),
                                        int(position[1]), ctypes.c_void_p(typeid[0]), int(typeid[1]),
                                        ctypes.c_void_p(image[0]), int(image[1]), ctypes.c_void_p(charge[0]),
                                        int(charge[1]), float(box_L[0]), float(box_L[1]), float(box_L[2]),
                                        int(L_typeid), ctypes.byref(params), ctypes.c_void_p(force[0]), int(force[1]),
                                        ctypes.c_void_p(pe_ptr), int(pe_stride)), "cavmd_compute_soa")

    # -- results ------------------------------------------------------------------------------------
    def energies(self):
        out = (ctypes.c_double * 3)()
        check(self._lib.cavmd_energies(self._h, ctypes.byref(out)), "cavmd_energies")
        return float(out[0]), float(out[1]), float(out[2])

    def result(self) -> Result:
        r = Result()
        check(self._lib.cavmd_result_read(self._h, ctypes.byref(r)), "cavmd_result_read")
        return r

    def result_device_ptr(self) -> int:
        p = ctypes.c_void_p()
        check(self._lib.cavmd_result_device_ptr(self._h, ctypes.byref(p)), "cavmd_result_device_ptr")
        return int(p.value)

    # -- observables -----------------------------------------------------
    def set_wavevectors(self, wavevectors) -> None:
        import numpy as np
        k = np.ascontiguousarray(wavevectors, dtype=np.float64)
        if k.ndim != 2 or k.shape[1] != 3:
            raise ValueError("wavevectors must have shape (n_k, 3)")
        self._n_k = int(k.shape[0])
        check(self._lib.cavmd_set_wavevectors(self._h, self._n_k, ctypes.c_void_p(k.ctypes.data)), "cavmd_set_wavevectors")

    def density_field(self, stream: int, N: int, position_ptr: int, position_stride: int) -> None:
        check(self._lib.cavmd_density_field(self._h, ctypes.c_void_p(stream), int(N), ctypes.c_void_p(position_ptr),
                                            int(position_stride)), "cavmd_density_field")

    def density_field_read(self):
        import numpy as np
        out = np.empty(2 * self._n_k, dtype=np.float64)
        check(self._lib.cavmd_density_field_read(self._h, ctypes.c_void_p(out.ctypes.data)), "cavmd_density_field_read")
        return out[0::2] + 1j * out[1::2]

    def cavity_mode(self, stream: int, vel_ptr: int, kB: float):
        out = (ctypes.c_double * 4)()
        check(self._lib.cavmd_cavity_mode(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(vel_ptr), float(kB),
                                          ctypes.byref(out)), "cavmd_cavity_mode")
        return float(out[0]), float(out[1]), float(out[2]), float(out[3])

    def force_mass_sum(self, stream: int, N: int, force_ptr: int, vel_ptr: int) -> float:
        out = ctypes.c_double()
        check(self._lib.cavmd_force_mass_sum(self._h, ctypes.c_void_p(stream), int(N), ctypes.c_void_p(force_ptr),
                                             ctypes.c_void_p(vel_ptr), ctypes.byref(out)), "cavmd_force_mass_sum")
        return float(out.value)

    def kinetic_energy(self, stream: int, vel_ptr: int, members_ptr, n_members: int) -> float:
        out = ctypes.c_double()
        check(self._lib.cavmd_kinetic_energy(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(vel_ptr),
                                             ctypes.c_void_p(members_ptr) if members_ptr else None, int(n_members),
                                             ctypes.byref(out)), "cavmd_kinetic_energy")
        return float(out.value)

    def scale_velocities(self, stream: int, vel_ptr: int, members_ptr, n_members: int, alpha: float) -> None:
        check(self._lib.cavmd_scale_velocities(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(vel_ptr),
                                               ctypes.c_void_p(members_ptr) if members_ptr else None, int(n_members),
                                               float(alpha)), "cavmd_scale_velocities")

    def bussi_step_device(self, stream: int, vel_ptr: int, members_ptr, n_members: int, dof: float, deltaT: float, set_T: float,
                          tau: float, normal_variate: float, gamma_variate: float) -> None:
        """One translational thermostat step on the device, asynchronous (two kernels, no host round trip)."""
        check(self._lib.cavmd_bussi_step_device(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(vel_ptr),
                                                ctypes.c_void_p(members_ptr) if members_ptr else None, int(n_members),
                                                float(dof), float(deltaT), float(set_T), float(tau), float(normal_variate),
                                                float(gamma_variate)), "cavmd_bussi_step_device")

    def bussi_device_read(self) -> "BussiDeviceState":
        out = BussiDeviceState()
        check(self._lib.cavmd_bussi_device_read(self._h, ctypes.byref(out)), "cavmd_bussi_device_read")
        return out

    def bussi_device_reset(self, stream: int = 0) -> None:
        check(self._lib.cavmd_bussi_device_reset(self._h, ctypes.c_void_p(stream)), "cavmd_bussi_device_reset")

    # -- measurement / tuning -----------------------------------------------------------------------
    def profile_enable(self, on: bool) -> None:
        check(self._lib.cavmd_profile_enable(self._h, 1 if on else 0), "cavmd_profile_enable")

    def profile_read(self):
        ms = (ctypes.c_double * 3)()
        n = ctypes.c_uint64()
        check(self._lib.cavmd_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n)), "cavmd_profile_read")
        return [float(ms[0]), float(ms[1]), float(ms[2])], int(n.value)

    def profile_samples(self, cap: int = 4096):
        """(n, 3) array of per-evaluation kernel times in ms {reduce, finalize, map}; call before profile_read()."""
        import numpy as np
        out = np.empty((cap, 3), dtype=np.float64)
        n = ctypes.c_size_t()
        check(self._lib.cavmd_profile_samples(self._h, ctypes.c_void_p(out.ctypes.data), int(cap), ctypes.byref(n)),
              "cavmd_profile_samples")
        return out[:n.value].copy()

    def set_tunable(self, name: str, value: int) -> None:
        check(self._lib.cavmd_set_tunable(self._h, name.encode(), int(value)), f"cavmd_set_tunable({name})")

    def get_tunable(self, name: str) -> int:
        v = ctypes.c_int()
        check(self._lib.cavmd_get_tunable(self._h, name.encode(), ctypes.byref(v)), f"cavmd_get_tunable({name})")
        return int(v.value)

    def device_info(self) -> dict:
        dev, cu = ctypes.c_int(), ctypes.c_int()
        buf = ctypes.create_string_buffer(64)
        check(self._lib.cavmd_device_info(self._h, ctypes.byref(dev), ctypes.byref(cu), buf, 64), "cavmd_device_info")
        return {"device": dev.value, "compute_units": cu.value, "arch": buf.value.decode()}
