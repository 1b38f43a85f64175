"""``CavityForce`` -- the user-facing force object, same surface as ``hoomd.cavitymd.CavityForce``.

Surface preserved from the reference (src/cavitymd/forces.py:21-232):
    CavityForce(kvector, couplstr, omegac, phmass=1.0, force_python=False)
    attributes   kvector, couplstr, omegac, phmass
    properties   implementation, harmonic_energy, coupling_energy, dipole_self_energy,
                 total_cavity_energy, energy, forces
    methods      set_forces(timestep)

With HOOMD-blue importable the class is the ``hoomd.md.force.Force`` subclass defined in
``hoomd_plugin.py`` (attach ladder: hip compute class -> hip through force.Custom); without it (this
environment) the class below is used with an explicit ``attach(sysdef)`` and ``compute(timestep)``.
Either way the numbers come from the HIP kernels; the implementation string is ``"hip"``.
"""
from __future__ import annotations

import numpy as np

from .compute import CavityForceComputeHIP

try:  # HOOMD-blue is optional at import time
    import hoomd  # noqa: F401
    import hoomd.md  # noqa: F401  (a real package, not a directory that merely happens to be called "hoomd" on sys.path)
    _HAVE_HOOMD = True
except ImportError:  # an ordinary ModuleNotFoundError in this environment
    _HAVE_HOOMD = False


class CavityForceSurface:
    """Everything that does not depend on how the object is attached to a simulation."""

    def _init_surface(self, kvector, couplstr, omegac, phmass, force_python):
        kv = np.array(kvector, dtype=float)
        if kv.shape != (3,):
            raise ValueError("kvector must have three components")
        self.kvector = kv          # accepted and stored, never used by the force (reference forces.py:33-34)
        self.couplstr = float(couplstr)
        self.omegac = float(omegac)
        self.phmass = float(phmass)
        self._force_python = bool(force_python)
        self._force_impl = None
        self._implementation = "hip"

    @property
    def implementation(self) -> str:
        """'hip' (compute class on device arrays) or 'hip_custom' (through hoomd.md.force.Custom)."""
        return self._implementation

    def _energy_component(self, k: int) -> float:
        impl = self._force_impl
        if impl is None:
            return 0.0
        return impl.getEnergies()[k]

    def _harmonic_energy(self) -> float:
        return self._energy_component(0)

    def _coupling_energy(self) -> float:
        return self._energy_component(1)

    def _dipole_self_energy(self) -> float:
        return self._energy_component(2)

    def _total_cavity_energy(self) -> float:
        impl = self._force_impl
        if impl is None:
            return 0.0
        e = impl.getEnergies()
        # same association as the reference: (harmonic + coupling) + dipole_self (forces.py:204-207)
        return e[0] + e[1] + e[2]

    def set_forces(self, timestep):
        impl = self._force_impl
        if impl is not None and hasattr(impl, "set_forces"):
            return impl.set_forces(timestep)
        return None


class StandaloneCavityForce(CavityForceSurface):
    """CavityForce for use without HOOMD-blue: attach to a ``cavitymd.state.SystemDefinition``."""

    def __init__(self, kvector, couplstr, omegac, phmass=1.0, force_python=False):
        self._init_surface(kvector, couplstr, omegac, phmass, force_python)
        if self._force_python:
            raise NotImplementedError(
                "force_python=True selects the reference's pure-Python hoomd.md.force.Custom path, which needs "
                "HOOMD-blue; this package computes on the GPU only and has no CPU fallback")

    # reference: _attach_hook builds the C++ object from sim.state._cpp_sys_def (forces.py:97-173)
    def attach(self, sysdef):
        self._force_impl = CavityForceComputeHIP(sysdef, self.omegac, self.couplstr, self.phmass)
        self._cpp_obj = self._force_impl
        return self

    def detach(self):
        self._force_impl = None
        self._cpp_obj = None

    def compute(self, timestep: int = 0, stream=None):
        if self._force_impl is None:
            raise RuntimeError("CavityForce is not attached")
        self._force_impl.compute(timestep, stream=stream)

    harmonic_energy = property(CavityForceSurface._harmonic_energy, doc="(1/2) K q.q")
    coupling_energy = property(CavityForceSurface._coupling_energy, doc="g (q_xy . d_xy)")
    dipole_self_energy = property(CavityForceSurface._dipole_self_energy, doc="(g^2 / 2K) d_xy . d_xy")
    total_cavity_energy = property(CavityForceSurface._total_cavity_energy, doc="sum of the three components")

    @property
    def energy(self) -> float:
        return self._total_cavity_energy()

    @property
    def forces(self):
        """(N,3) host array of the forces of the last evaluation (like hoomd.md.force.Force.forces)."""
        if self._force_impl is None:
            return None
        return self._force_impl.getForceArray()[:, :3].cpu().numpy()


if _HAVE_HOOMD:  # pragma: no cover - HOOMD-blue is absent from the build/test image
    from .hoomd_plugin import HoomdCavityForce as CavityForce
else:
    CavityForce = StandaloneCavityForce
