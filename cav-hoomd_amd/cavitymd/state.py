"""Device-resident particle arrays in HOOMD-blue's native layouts.

HOOMD-blue is not available in the build/test environment, so the compute classes of this package
take, in place of HOOMD's ``SystemDefinition``, the small stand-in below.  It exposes exactly the
accessors the reference's compute reads (src/CavityForceCompute.cc:79, 137-142, 159):
``getParticleData()``, ``getN()``, ``getTypeByName()``, ``getGlobalBox().getL()`` and the pos / charge /
image arrays -- with the arrays held as torch tensors in device memory (PyTorch-ROCm is used for
memory and streams only).

Layouts (double-precision HOOMD build):
    pos     (N, 4) float64   x, y, z, and the type id in the low 32 bits of the 4th double (Scalar4)
    charge  (N,)   float64
    image   (N, 3) int32     (int3, 12-byte stride)
"""
from __future__ import annotations

import numpy as np
import torch


def type_tag_as_double(typeid) -> np.ndarray:
    """Bit pattern HOOMD stores in pos.w for a type id (``__int_as_scalar``): the int in the low 4 bytes."""
    t = np.asarray(typeid, dtype=np.int64) & 0xFFFFFFFF
    return t.astype(np.uint64).view(np.float64)


class BoxDim:
    """Orthorhombic box; only ``getL()`` is used by the cavity force (tilt is ignored by the reference too,
    src/CavityForceCompute.cc:97-109)."""

    def __init__(self, Lx: float, Ly: float, Lz: float):
        self._L = (float(Lx), float(Ly), float(Lz))

    def getL(self):
        return self._L


class ParticleData:
    def __init__(self, pos: torch.Tensor, charge: torch.Tensor, image: torch.Tensor, types, box):
        n = pos.shape[0]
        if pos.dtype != torch.float64 or pos.shape != (n, 4) or not pos.is_contiguous():
            raise ValueError("pos must be a contiguous (N,4) float64 tensor (HOOMD Scalar4)")
        if charge.dtype != torch.float64 or charge.shape != (n,) or not charge.is_contiguous():
            raise ValueError("charge must be a contiguous (N,) float64 tensor")
        if image.dtype != torch.int32 or image.shape != (n, 3) or not image.is_contiguous():
            raise ValueError("image must be a contiguous (N,3) int32 tensor (HOOMD int3)")
        if not (pos.device == charge.device == image.device):
            raise ValueError("pos, charge and image must live on the same device")
        self._pos, self._charge, self._image = pos, charge, image
        self._types = list(types)
        self._box = box if isinstance(box, BoxDim) else BoxDim(*box)

    # -- construction ------------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, position, typeid, charge, image, types, box, device="cuda"):
        """Build from host arrays in snapshot form: position (N,3), typeid (N,), charge (N,), image (N,3)."""
        position = np.asarray(position, dtype=np.float64)
        n = position.shape[0]
        pos4 = np.empty((n, 4), dtype=np.float64)
        pos4[:, :3] = position
        pos4[:, 3] = type_tag_as_double(typeid)
        dev = torch.device(device)
        return cls(
            torch.from_numpy(pos4).to(dev),
            torch.from_numpy(np.ascontiguousarray(charge, dtype=np.float64)).to(dev),
            torch.from_numpy(np.ascontiguousarray(image, dtype=np.int32)).to(dev), types, box)

    # -- the accessors the compute uses (names as in HOOMD's ParticleData) ---------------------------
    def getN(self) -> int:
        return int(self._pos.shape[0])

    def getTypeByName(self, name: str) -> int:
        try:
            return self._types.index(name)
        except ValueError:
            # HOOMD throws std::runtime_error("Type <name> not found!") here
            raise RuntimeError(f"Type {name} not found!") from None

    def getNameByType(self, typeid: int) -> str:
        return self._types[typeid]

    def getGlobalBox(self) -> BoxDim:
        return self._box

    def setGlobalBox(self, box):
        self._box = box if isinstance(box, BoxDim) else BoxDim(*box)

    def getPositions(self) -> torch.Tensor:
        return self._pos

    def getCharges(self) -> torch.Tensor:
        return self._charge

    def getImages(self) -> torch.Tensor:
        return self._image

    @property
    def device(self) -> torch.device:
        return self._pos.device

    @property
    def types(self):
        return list(self._types)


class SystemDefinition:
    """What the compute constructors receive as ``sysdef`` (HOOMD: ``sim.state._cpp_sys_def``)."""

    def __init__(self, pdata: ParticleData):
        self._pdata = pdata

    def getParticleData(self) -> ParticleData:
        return self._pdata
