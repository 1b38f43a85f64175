"""On-device observables that sit next to the cavity-force path (the reference computes them from a full CPU snapshot
every step in Python; here nothing but the few result bytes leaves the GPU).

Names follow the reference's observable library (src/cavitymd/analysis.py:14-64):

    generate_fibonacci_sphere(samples)                 the default wavevector directions of the F(k,t) tracker
    compute_total_dipole_moment(compute)               sum_i q_i r_i over ALL particles, from the force's own reduction
    DensityField(pdata, wavevectors).compute()         rho(k) = sum_j exp(i k . r_j), wrapped positions, all particles
    cavity_mode(compute, velocity)                     (KE, harmonic PE, KE + PE, T) of the cavity oscillator
    force_mass_sum(workspace, net_force, velocity)     sum_i |F_i| / m_i, what AdaptiveTimestepUpdater reduces on the host
    adaptive_timestep(error_tolerance, S)              dt = sqrt(tol / S)   (src/cavitymd/simulation.py:88-91)
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _capi
from .utils import PhysicalConstants


def generate_fibonacci_sphere(samples: int = 100) -> np.ndarray:
    """`samples` points on the unit sphere along a golden-angle spiral, y running from +1 to -1
    (same construction as src/cavitymd/analysis.py:50-64, so wavevector sets are interchangeable)."""
    i = np.arange(samples, dtype=np.float64)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    y = 1.0 - (i / float(samples - 1)) * 2.0
    radius = np.sqrt(1.0 - y * y)
    theta = golden * i
    return np.stack([np.cos(theta) * radius, y, np.sin(theta) * radius], axis=1)


def compute_total_dipole_moment(compute) -> np.ndarray:
    """Total dipole (photon included) of the last force evaluation of a ``CavityForceComputeHIP``: the reduction the
    force needs anyway, so the DipoleAutocorrelation tracker costs one 192-byte read instead of a snapshot."""
    return np.array(compute.getResult().total_dipole[:])


class DensityField:
    """rho(k) on the GPU for a fixed set of wavevectors (FieldAutocorrelationTracker's observable)."""

    def __init__(self, pdata, wavevectors):
        if pdata.device.type != "cuda":
            raise RuntimeError("DensityField needs particle data in GPU memory; there is no CPU fallback in this package")
        self._pdata = pdata
        self.wavevectors = np.ascontiguousarray(wavevectors, dtype=np.float64)
        self._ws = _capi.Workspace(max(pdata.getN(), 1), device=pdata.device.index if pdata.device.index is not None else -1)
        self._ws.set_wavevectors(self.wavevectors)

    def enqueue(self, stream=None) -> None:
        pd = self._pdata
        if stream is None:
            stream = torch.cuda.current_stream(pd.device)
        handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        pos = pd.getPositions()  # (N,4) Scalar4: x, y, z at stride 32
        self._ws.density_field(handle, pd.getN(), pos.data_ptr(), pos.stride(0) * pos.element_size())

    def result(self) -> np.ndarray:
        return self._ws.density_field_read()

    def compute(self, stream=None) -> np.ndarray:
        self.enqueue(stream)
        return self.result()


def cavity_mode(compute, velocity: torch.Tensor, stream=None):
    """(kinetic, potential, total, temperature) of the photon the last evaluation of `compute` found.
    `velocity` is HOOMD's (N,4) Scalar4 velocity array on the GPU (mass in the 4th column)."""
    if velocity.dtype != torch.float64 or velocity.dim() != 2 or velocity.shape[1] != 4 or not velocity.is_contiguous():
        raise ValueError("velocity must be a contiguous (N,4) float64 tensor (HOOMD Scalar4 vel, mass in column 3)")
    if stream is None:
        stream = torch.cuda.current_stream(velocity.device)
    handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    return compute.workspace.cavity_mode(handle, velocity.data_ptr(), PhysicalConstants.KB_HARTREE_PER_K)


def force_mass_sum(workspace: _capi.Workspace, net_force: torch.Tensor, velocity: torch.Tensor, stream=None) -> float:
    """S = sum_i |F_i| / m_i over HOOMD's (N,4) net-force and velocity arrays (mass = velocity[:, 3]) on the GPU."""
    for name, t in (("net_force", net_force), ("velocity", velocity)):
        if t.dtype != torch.float64 or t.dim() != 2 or t.shape[1] != 4 or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous (N,4) float64 tensor (HOOMD Scalar4)")
    if net_force.shape[0] != velocity.shape[0]:
        raise ValueError("net_force and velocity disagree on N")
    if stream is None:
        stream = torch.cuda.current_stream(net_force.device)
    handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    return workspace.force_mass_sum(handle, net_force.shape[0], net_force.data_ptr(), velocity.data_ptr())


def adaptive_timestep(error_tolerance: float, force_mass_sum_value: float):
    """dt = sqrt(tol / S); None when S == 0 (the reference then leaves dt unchanged, simulation.py:88)."""
    if not force_mass_sum_value > 0:
        return None
    return math.sqrt(error_tolerance / force_mass_sum_value)
