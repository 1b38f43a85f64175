"""``CavityForceComputeHIP`` -- host-side mirror of the reference's C++ compute classes.

Same constructor and method names as ``_cavitymd.CavityForceCompute[GPU]``
(src/CavityForceCompute.cc:212-224, src/CavityForceComputeGPU.cc:257-264 of the reference):

    CavityForceComputeHIP(sysdef, omegac, couplstr, phmass=1.0)
    .setParams(omegac, couplstr, phmass) / .getParams() -> {omegac, couplstr, K, phmass}
    .getHarmonicEnergy() / .getCouplingEnergy() / .getDipoleSelfEnergy()
    .compute(timestep)            # HOOMD: ForceCompute::compute -> computeForces(timestep)

The work itself is one HIP kernel launch (two above ~2.4e6 particles) behind the C ABI
(``include/cavmd.h``); this class only owns the force array (HOOMD: ``m_force``), the workspace and the parameter
block.  There is no CPU fallback.  The call goes through the pybind11 module when it is built (less host overhead
per call), else through ctypes; ``CAVMD_BINDING=ctypes`` forces the latter.
"""
from __future__ import annotations

import os

import torch

from . import _capi

try:  # pybind11 flavour of the shim: ~1.5 us less host overhead per call than ctypes (csrc/pybind/module.cc)
    from . import _cavitymd as _ext
except ImportError:  # not built: the ctypes route does the same work
    _ext = None
if os.environ.get("CAVMD_BINDING", "").lower() == "ctypes":
    _ext = None

# torch's current raw stream handle without building a Stream object (saves ~1 us per call at N = 501, where the whole
# evaluation takes 5 us on the GPU); falls back to the public API
_raw_current_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


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
        self._cache_key = None   # identity of the arrays the cached pointers belong to
        self._cached = None

    # -- parameters (src/CavityForceCompute.cc:48-56) ----------------------------------------------------
    def setParams(self, omegac: float, couplstr: float, phmass: float = 1.0) -> None:
        self._params = _capi.make_params(omegac, couplstr, phmass)

    def getParams(self) -> dict:
        return self._params.as_dict()

    # -- the per-step entry point --------------------------------------------------------------------------
    def compute(self, timestep: int = 0, stream=None) -> None:
        """Enqueue one force evaluation on ``stream`` (default: torch's current stream).  Asynchronous."""
        pd = self._pdata
        pos, chg, img = pd._pos, pd._charge, pd._image
        k = self._cache_key  # the very objects the cached pointers belong to (held, so ids cannot be recycled)
        if k is None or pos is not k[0] or chg is not k[1] or img is not k[2] or pd._types is not k[3] \
                or len(pd._types) != k[4] or pd._box is not k[5]:
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
            box_L = pd.getGlobalBox().getL()
            self._cached = (n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), box_L, L_typeid, self._force.data_ptr(),
                            pd.device.index if pd.device.index is not None else torch.cuda.current_device())
            self._cache_key = (pos, chg, img, pd._types, len(pd._types), pd._box)
        n, p_pos, p_chg, p_img, box_L, L_typeid, p_force, dev_index = self._cached
        if stream is None:
            handle = _raw_current_stream(dev_index) if _raw_current_stream is not None \
                else torch.cuda.current_stream(pd.device).cuda_stream
        else:
            handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        prm = self._params
        if _ext is not None:
            _ext.compute_hoomd(self._ws.handle.value, handle, n, p_pos, p_chg, p_img, box_L[0], box_L[1], box_L[2], L_typeid,
                               prm.omegac, prm.couplstr, prm.K, prm.phmass, p_force)
        else:
            self._ws.compute_hoomd(handle, n, p_pos, p_chg, p_img, box_L, L_typeid, prm, p_force)
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
