"""``CavityForceComputeHIP`` -- host-side mirror of the reference's C++ compute classes.

Same constructor and method names as ``_cavitymd.CavityForceCompute[GPU]``
(src/CavityForceCompute.cc:212-224, src/CavityForceComputeGPU.cc:257-264 of the reference):

    CavityForceComputeHIP(sysdef, omegac, couplstr, phmass=1.0)
    .setParams(omegac, couplstr, phmass) / .getParams() -> {omegac, couplstr, K, phmass}
    .getHarmonicEnergy() / .getCouplingEnergy() / .getDipoleSelfEnergy()
    .compute(timestep)            # HOOMD: ForceCompute::compute -> computeForces(timestep)

The work itself is three HIP kernels behind the C ABI (``include/cavmd.h``); this class only owns the
force array (HOOMD: ``m_force``), the workspace and the parameter block.  There is no CPU fallback.
"""
from __future__ import annotations

import torch

from . import _capi


class CavityForceComputeHIP:
    def __init__(self, sysdef, omegac: float, couplstr: float, phmass: float = 1.0):
        self._sysdef = sysdef
        self._pdata = sysdef.getParticleData()
        dev = self._pdata.device
        if dev.type != "cuda":
            # the reference throws std::runtime_error("GPU computation required but not available")
            # (src/CavityForceComputeGPU.cc:106-109); there is no silent fallback here either
            raise RuntimeError("CavityForceComputeHIP requires particle data in GPU memory (got device "
                               f"'{dev}'); no CPU fallback exists in this package")
        self._params = _capi.make_params(omegac, couplstr, phmass)
        n = self._pdata.getN()
        self._ws = _capi.Workspace(max(n, 1), device=dev.index if dev.index is not None else -1)
        self._force = torch.empty((n, 4), dtype=torch.float64, device=dev)
        self._virial = None  # never written: the reference leaves m_virial zero
        self._last_timestep = None

    # -- parameters (src/CavityForceCompute.cc:48-56) ----------------------------------------------------
    def setParams(self, omegac: float, couplstr: float, phmass: float = 1.0) -> None:
        self._params = _capi.make_params(omegac, couplstr, phmass)

    def getParams(self) -> dict:
        return self._params.as_dict()

    # -- the per-step entry point --------------------------------------------------------------------------
    def compute(self, timestep: int = 0, stream=None) -> None:
        """Enqueue one force evaluation on ``stream`` (default: torch's current stream).  Asynchronous."""
        pd = self._pdata
        n = pd.getN()
        if self._force.shape[0] != n:
            self._force = torch.empty((n, 4), dtype=torch.float64, device=pd.device)
        if n > self._ws.max_N:
            self._ws.close()
            self._ws = _capi.Workspace(n, device=pd.device.index if pd.device.index is not None else -1)
        try:
            L_typeid = pd.getTypeByName("L")
        except RuntimeError:
            # no type named 'L': the reference's GPU class zeroes the energies and returns
            # (src/CavityForceComputeGPU.cc:114-123); -1 matches no particle and takes the no-photon path
            L_typeid = -1
        if stream is None:
            stream = torch.cuda.current_stream(pd.device)
        handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        box_L = pd.getGlobalBox().getL()
        self._ws.compute_hoomd(handle, n, pd.getPositions().data_ptr(), pd.getCharges().data_ptr(),
                               pd.getImages().data_ptr(), box_L, L_typeid, self._params, self._force.data_ptr())
        self._last_timestep = timestep

    # alias with the reference's protected virtual name
    computeForces = compute

    # -- results ---------------------------------------------------------------------------------------------
    def getHarmonicEnergy(self) -> float:
        return self._ws.energies()[0]

    def getCouplingEnergy(self) -> float:
        return self._ws.energies()[1]

    def getDipoleSelfEnergy(self) -> float:
        return self._ws.energies()[2]

    def getEnergies(self):
        """(harmonic, coupling, dipole_self) with a single device round trip."""
        return self._ws.energies()

    def getResult(self) -> _capi.Result:
        return self._ws.result()

    def getForceArray(self) -> torch.Tensor:
        """(N,4) float64 device tensor laid out like HOOMD's ``m_force`` (x, y, z, per-particle energy)."""
        return self._force

    @property
    def workspace(self) -> _capi.Workspace:
        return self._ws
