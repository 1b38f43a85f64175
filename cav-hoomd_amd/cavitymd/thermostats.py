"""``BussiReservoir`` -- host-side mirror of the reference's extended Bussi thermostat
(src/thermostats.py / src/bussi_reservoir/thermostats.py: class BussiReservoir; C++ side
src/BussiReservoirThermostat.h).

Same constructor and read-only properties as the reference:

    BussiReservoir(kT, tau=0.0)
    .kT .tau
    .reservoir_energy_translational  .reservoir_energy_rotational  .total_reservoir_energy
    .instantaneous_reservoir_translational  .instantaneous_reservoir_rotational  .instantaneous_reservoir_total
    .reset_reservoir_energy()

Inside HOOMD the thermostat is driven by the integration method (``getRescalingFactorsOne``).  HOOMD-blue is absent here,
so ``step`` below plays that caller for the standalone harness: kinetic energy of the group on the GPU
(``cavmd_kinetic_energy``), the scalar rule with its sign handling and the reservoir bookkeeping (``cavmd_bussi_step``),
and the velocity rescaling HOOMD's integration method would apply (``cavmd_scale_velocities``).  ``step_async`` does the same
translational step entirely on the device (``cavmd_bussi_step_device``): two kernels, no host round trip, counters fetched lazily.

The two random variates per degree-of-freedom class come from the caller (``variates=``) or from a
``numpy.random.Generator``.  The reference draws them from HOOMD's RandomGenerator seeded by (timestep, simulation seed,
first member tag), which is not vendored: variate GENERATION is therefore not bit-comparable with a HOOMD run, everything
after the draw is (tests/test_bussi_reservoir.py).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _capi


class BussiReservoir:
    def __init__(self, kT, tau: float = 0.0):
        self.kT = kT           # a number or a callable of the timestep (hoomd.variant-like)
        self.tau = float(tau)
        self._state = _capi.BussiReservoirState()
        self._ws = None
        self._members = None   # device tensor of member indices, or None for all particles
        self._n_members = 0
        self._attached = False
        self._dev_used = False        # some step ran on the device: its counters live there (cavmd_bussi_device_read)
        self._dev_stream = 0          # stream of the last on-device step
        self._last_on_device = False  # which path the "instantaneous" counters belong to

    # -- attachment (reference: _attach_hook builds the C++ object from the method's filter group) --------------------------
    def attach(self, n_particles: int, members=None, device="cuda") -> None:
        """`members`: particle indices of the thermostatted group (None = all)."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("BussiReservoir needs the velocity array in GPU memory; no CPU fallback exists in this package")
        self._ws = _capi.Workspace(max(int(n_particles), 1), device=dev.index if dev.index is not None else -1)
        if members is None:
            self._members, self._n_members = None, int(n_particles)
        else:
            m = np.ascontiguousarray(members, dtype=np.uint32)
            self._members = torch.from_numpy(m.view(np.int32).copy()).to(dev)
            self._n_members = int(m.shape[0])
        self._attached = True

    def detach(self) -> None:
        if self._ws is not None:
            self._ws.close()
        self._ws = None
        self._attached = False
        self._dev_used = self._last_on_device = False

    def _set_T(self, timestep: int) -> float:
        return float(self.kT(timestep)) if callable(self.kT) else float(self.kT)

    # -- one thermostat step -----------------------------------------------------------------------------------------------
    def kinetic_energy(self, velocity: torch.Tensor, stream=None) -> float:
        handle = self._stream(velocity, stream)
        mp = self._members.data_ptr() if self._members is not None else None
        return self._ws.kinetic_energy(handle, velocity.data_ptr(), mp, self._n_members)

    def step(self, timestep: int, deltaT: float, velocity: torch.Tensor, translational_dof: float, variates=None, rng=None,
             rotational_kinetic_energy: float = 0.0, rotational_dof: float = 0.0, rescale: bool = True, stream=None):
        """KE of the group -> alpha (translational, rotational) -> reservoir counters -> velocities *= alpha_translational.
        velocity: (N,4) float64 device tensor, HOOMD's Scalar4 layout (mass in column 3)."""
        if not self._attached:
            raise RuntimeError("BussiReservoir.step before attach()")
        if velocity.dtype != torch.float64 or velocity.dim() != 2 or velocity.shape[1] != 4 or not velocity.is_contiguous():
            raise ValueError("velocity must be a contiguous (N,4) float64 tensor (HOOMD Scalar4, mass in .w)")
        ke = self.kinetic_energy(velocity, stream)
        if variates is None:
            rng = rng if rng is not None else np.random.default_rng()
            variates = draw_variates(rng, translational_dof, rotational_dof)
        at, ar = _capi.bussi_step(self._state, ke, translational_dof, rotational_kinetic_energy, rotational_dof, deltaT,
                                  self._set_T(timestep), self.tau, variates)
        self._last_on_device = False
        if rescale and deltaT != 0.0:
            mp = self._members.data_ptr() if self._members is not None else None
            self._ws.scale_velocities(self._stream(velocity, stream), velocity.data_ptr(), mp, self._n_members, at)
        return at, ar

    def step_async(self, timestep: int, deltaT: float, velocity: torch.Tensor, translational_dof: float, variates=None,
                   rng=None, stream=None) -> None:
        """The translational step without a host round trip: the kernel that folds the kinetic energy evaluates the rule on
        the device and leaves alpha for the rescale kernel enqueued right behind it; nothing is waited for.  The counters are
        fetched when a property is read.  (Rotational degrees of freedom are not handled on this path: use ``step``.)"""
        if not self._attached:
            raise RuntimeError("BussiReservoir.step_async before attach()")
        if velocity.dtype != torch.float64 or velocity.dim() != 2 or velocity.shape[1] != 4 or not velocity.is_contiguous():
            raise ValueError("velocity must be a contiguous (N,4) float64 tensor (HOOMD Scalar4, mass in .w)")
        if variates is None:
            rng = rng if rng is not None else np.random.default_rng()
            variates = draw_variates(rng, translational_dof, 0.0)
        mp = self._members.data_ptr() if self._members is not None else None
        self._dev_stream = self._stream(velocity, stream)
        self._ws.bussi_step_device(self._dev_stream, velocity.data_ptr(), mp, self._n_members, translational_dof, deltaT,
                                   self._set_T(timestep), self.tau, variates[0], variates[1])
        if deltaT != 0.0 and self._n_members:
            self._dev_used = True
            self._last_on_device = True

    def device_state(self):
        """Counters of the on-device path after its last enqueued step (waits for that step's flag, nothing else)."""
        return self._ws.bussi_device_read()

    def _dev(self, field: str) -> float:
        return getattr(self.device_state(), field) if self._dev_used else 0.0

    @staticmethod
    def _stream(t: torch.Tensor, stream) -> int:
        if stream is None:
            return torch.cuda.current_stream(t.device).cuda_stream
        return stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)

    # -- the reference's loggable quantities ---------------------------------------------------------------------------------
    @property
    def reservoir_energy_translational(self) -> float:
        return (self._state.reservoir_translational + self._dev("reservoir_translational")) if self._attached else 0.0

    @property
    def reservoir_energy_rotational(self) -> float:
        return self._state.reservoir_rotational if self._attached else 0.0

    @property
    def total_reservoir_energy(self) -> float:
        return (self.reservoir_energy_translational + self._state.reservoir_rotational) if self._attached else 0.0

    @property
    def instantaneous_reservoir_translational(self) -> float:
        if not self._attached:
            return 0.0
        if self._last_on_device:
            return self._dev("instantaneous_translational")
        return self._state.instantaneous_translational

    @property
    def instantaneous_reservoir_rotational(self) -> float:
        return self._state.instantaneous_rotational if self._attached else 0.0

    @property
    def instantaneous_reservoir_total(self) -> float:
        return (self.instantaneous_reservoir_translational + self.instantaneous_reservoir_rotational) if self._attached else 0.0

    def reset_reservoir_energy(self) -> None:
        self._state = _capi.BussiReservoirState()
        if self._dev_used:
            self._ws.bussi_device_reset(self._dev_stream)


def draw_variates(rng: np.random.Generator, translational_dof: float, rotational_dof: float = 0.0):
    """{normal_t, gamma_t, normal_r, gamma_r} in the order the reference consumes its generator
    (src/BussiReservoirThermostat.h:73-82, 192-199): nothing is drawn for a class with 0 degrees of freedom, the gamma variate
    only for more than one."""
    out = [0.0, 0.0, 0.0, 0.0]
    for k, dof in enumerate((translational_dof, rotational_dof)):
        if dof == 0:
            continue
        out[2 * k] = float(rng.standard_normal())
        if dof > 1.0:
            out[2 * k + 1] = float(rng.gamma((dof - 1.0) / 2.0, 1.0))
    return out
