"""All HOOMD-blue-touching Python of the package lives in this one file.

UNVERIFIED AGAINST A REAL HOOMD-blue: the build/test image has no HOOMD (``import hoomd`` raises
ModuleNotFoundError), so this module is only ever imported when a user installs the package next to a
HOOMD-blue 4.x ROCm build.  It is written against the HOOMD 4 Python API the reference itself uses
(src/cavitymd/forces.py:45-173, src/cavitymd/cavity_force_python.py:31-149).  Its control flow is executed by
tests/test_hoomd_plugin_stub.py against a STAND-IN hoomd package (tests/stubs/hoomd) -- that catches typos and checks
this module's own logic, it says nothing about a real HOOMD-blue.  See INTEGRATION.md.

Two attach routes, tried in this order (reference ladder: cuda -> cpp -> python, forces.py:97-173):

  1. ``_cavitymd_hip.CavityForceComputeHIP`` -- the compile-gated C++ ``ForceCompute`` subclass in
     ``csrc/hoomd_shim/`` (exists only when the package was built against HOOMD's headers); HOOMD calls
     its ``computeForces`` directly, no Python in the step loop.   implementation == "hip"
  2. ``CavityForceCustomHIP`` -- a ``hoomd.md.force.Custom`` whose ``set_forces`` hands the device
     pointers of ``gpu_local_snapshot`` / ``gpu_local_force_arrays`` to ``cavmd_compute_soa`` through
     ctypes.                                                       implementation == "hip_custom"

Neither route ever computes on the CPU; if no GPU device is in use the attach raises.
"""
from __future__ import annotations

import hoomd
from hoomd.logging import log

from . import _capi, marshal
from .forces import CavityForceSurface

try:  # built only when HOOMD's headers were available at compile time
    from . import _cavitymd_hip  # type: ignore
except ImportError:
    _cavitymd_hip = None


class CavityForceCustomHIP(hoomd.md.force.Custom):
    """``hoomd.md.force.Custom`` front end of the HIP kernels (replaces CavityForcePython,
    src/cavitymd/cavity_force_python.py:31-149, with the C++ compute's semantics)."""

    def __init__(self, couplstr, omegac, phmass=1.0):
        super().__init__(aniso=False)
        self._params = _capi.make_params(omegac, couplstr, phmass)
        self._ws = None
        self._energies = marshal.EnergyCache()

    def setParams(self, omegac, couplstr, phmass=1.0):
        self._params = _capi.make_params(omegac, couplstr, phmass)

    def getParams(self):
        return self._params.as_dict()

    def getEnergies(self):
        if self._ws is None:
            return (0.0, 0.0, 0.0)
        return self._energies.get(self._ws.energies)

    def set_forces(self, timestep):
        # all marshalling is HOOMD-free and tested: cavitymd/marshal.py
        state = self._state
        with state.gpu_local_snapshot as snap, self.gpu_local_force_arrays as arrays:
            n = int(snap.particles.position.shape[0])
            if n == 0:
                self._energies.clear()          # the reference zeroes its energies when there is nothing to do
                return
            if self._ws is None or n > self._ws.max_N:
                self._ws = _capi.Workspace(max(n, 1))
            marshal.set_forces_custom(self._ws, self._params, 0,  # HOOMD-blue works on the null stream
                                      n, snap.particles.position, snap.particles.typeid, snap.particles.image,
                                      snap.particles.charge, snap.global_box.L, state.particle_types, arrays.force,
                                      arrays.potential_energy)
            self._energies.bump()


class HoomdCavityForce(CavityForceSurface, hoomd.md.force.Force):
    """``hoomd.cavitymd.CavityForce`` with the HIP engine underneath."""

    def __init__(self, kvector, couplstr, omegac, phmass=1.0, force_python=False):
        hoomd.md.force.Force.__init__(self)
        param_dict = hoomd.data.parameterdicts.ParameterDict(
            kvector=hoomd.data.typeconverter.to_type_converter([float, float, float]), couplstr=float, omegac=float,
            phmass=float, force_python=bool)
        param_dict["kvector"] = [float(v) for v in kvector]
        param_dict["couplstr"] = float(couplstr)
        param_dict["omegac"] = float(omegac)
        param_dict["phmass"] = float(phmass)
        param_dict["force_python"] = bool(force_python)
        self._param_dict.update(param_dict)
        self._init_surface(kvector, couplstr, omegac, phmass, force_python)

    def _attach_hook(self):
        sim = self._simulation
        route = marshal.choose_route(_cavitymd_hip is not None, self._force_python, isinstance(sim.device, hoomd.device.GPU))
        sysdef = sim.state._cpp_sys_def
        if route == marshal.ROUTE_HIP:
            self._force_impl = _cavitymd_hip.CavityForceComputeHIP(sysdef, self.omegac, self.couplstr, self.phmass)
            self._cpp_obj = self._force_impl
        else:
            # The Custom force is attached through HOOMD's own machinery, so that ITS _simulation / _cpp_obj are the ones
            # gpu_local_force_arrays looks at (hoomd.md.force.Custom builds its CustomForceCompute in _attach_hook); this
            # object then shares the inner C++ compute.  UNVERIFIED against a real HOOMD-blue, like the rest of this file.
            self._force_impl = CavityForceCustomHIP(self.couplstr, self.omegac, self.phmass)
            self._force_impl._attach(sim)
            self._cpp_obj = self._force_impl._cpp_obj
        self._implementation = route
        super()._attach_hook()

    def _detach_hook(self):
        impl, self._force_impl = self._force_impl, None
        if impl is not None and hasattr(impl, "_detach"):
            impl._detach()

    def _energy_component(self, k):
        impl = self._force_impl
        if impl is None:
            return 0.0
        if hasattr(impl, "getEnergies"):
            return impl.getEnergies()[k]
        return (impl.getHarmonicEnergy, impl.getCouplingEnergy, impl.getDipoleSelfEnergy)[k]()

    @log(requires_run=True)
    def harmonic_energy(self):
        """(1/2) K q.q"""
        return self._energy_component(0)

    @log(requires_run=True)
    def coupling_energy(self):
        """g (q_xy . d_xy)"""
        return self._energy_component(1)

    @log(requires_run=True)
    def dipole_self_energy(self):
        """(g^2 / 2K) d_xy . d_xy"""
        return self._energy_component(2)

    @log(requires_run=True)
    def total_cavity_energy(self):
        return self.harmonic_energy + self.coupling_energy + self.dipole_self_energy

    @property
    def energy(self):
        return self.total_cavity_energy

    @property
    def forces(self):
        return None  # as the reference for compiled implementations (forces.py:214-221)
