"""HOOMD-free half of the HOOMD attachment: everything ``hoomd_plugin.py`` does that needs no ``import hoomd``.

The build/test image has no HOOMD-blue, so the HOOMD-touching classes in ``hoomd_plugin.py`` cannot run here; what they
compute from plain Python objects can, and lives in this module so that it is tested (tests/test_hoomd_marshalling.py):

    cai_pointer_stride(arr)       (device pointer, byte stride between particles) of anything that exposes
                                  ``__cuda_array_interface__`` -- HOOMD's GPU local-snapshot arrays do, so do torch tensors
    photon_typeid(types)          index of the particle type named 'L', -1 if there is none
                                  (reference: getTypeByName("L"), src/CavityForceCompute.cc:79; GPU class catch-all
                                  src/CavityForceComputeGPU.cc:114-123)
    choose_route(...)             which rung of the attach ladder applies (reference ladder: src/cavitymd/forces.py:97-173)
    custom_force_arguments(...)   the argument list of cavmd_compute_soa for a ``hoomd.md.force.Custom``-style call
                                  (reference: CavityForcePython.set_forces, src/cavitymd/cavity_force_python.py:72-145)
    set_forces_custom(...)        that call
    EnergyCache                   the lazily fetched energies of one evaluation, keyed on the evaluation COUNTER (not the
                                  timestep: setParams + sim.run(0) recomputes at the same timestep)
"""
from __future__ import annotations

ROUTE_HIP = "hip"                # compiled ForceCompute subclass (csrc/hoomd_shim), no Python in the step loop
ROUTE_HIP_CUSTOM = "hip_custom"  # hoomd.md.force.Custom -> cavmd_compute_soa through ctypes


def cai_pointer_stride(arr):
    """(device pointer, byte stride between consecutive particles) of a ``__cuda_array_interface__`` exporter.
    ``strides`` is None for C-contiguous arrays (version-2+ of the interface): the stride is then the row size."""
    d = arr.__cuda_array_interface__
    typestr = d["typestr"]
    itemsize = int(typestr[2:])
    shape = tuple(d["shape"])
    strides = d.get("strides")
    if len(shape) == 0:
        raise ValueError("a per-particle array has at least one dimension")
    if strides is None:
        row = itemsize
        for s in shape[1:]:
            row *= int(s)
        stride0 = row
    else:
        stride0 = int(strides[0])
        # the inner dimension must be dense: the kernels read x, y, z at consecutive addresses
        if len(shape) > 1 and int(strides[1]) != itemsize:
            raise ValueError(f"inner stride {strides[1]} != itemsize {itemsize}: components must be contiguous")
    ptr = d["data"][0]
    if ptr is None or int(ptr) == 0:
        if shape[0] != 0:
            raise ValueError("null device pointer for a non-empty array")
        ptr = 0
    return int(ptr), int(stride0)


def photon_typeid(types) -> int:
    """Index of type 'L' in the simulation's type list, -1 when no type has that name (forces and energies then become
    zero, as in the reference's GPU class, src/CavityForceComputeGPU.cc:114-123)."""
    types = list(types)
    return types.index("L") if "L" in types else -1


def choose_route(have_compiled_class: bool, force_python: bool, device_is_gpu: bool) -> str:
    """The attach ladder.  The reference tries cuda -> cpp -> python and falls back with a warning
    (src/cavitymd/forces.py:97-173); here both rungs run the same HIP kernels and there is no CPU rung."""
    if not device_is_gpu:
        raise RuntimeError("cavitymd (HIP build) needs hoomd.device.GPU; it has no CPU implementation")
    if have_compiled_class and not force_python:
        return ROUTE_HIP
    return ROUTE_HIP_CUSTOM


def custom_force_arguments(n, position, typeid, image, charge, box_L, types, force, potential_energy=None):
    """Arguments of ``Workspace.compute_soa`` after the stream: every array as (device pointer, byte stride).
    HOOMD's local snapshot hands out strided VIEWS of its Scalar4 buffers (position = pos[:, :3], typeid = the int in
    pos.w, force = force4[:, :3], potential_energy = force4[:, 3]); cavmd_compute_soa recognises exactly that pattern
    (strides 32/32/12/8/32/32, typeid 24 bytes after position, potential_energy 24 bytes after force) and takes the
    HOOMD-native kernels."""
    pe = cai_pointer_stride(potential_energy) if potential_energy is not None else None
    return (int(n), cai_pointer_stride(position), cai_pointer_stride(typeid), cai_pointer_stride(image),
            cai_pointer_stride(charge), (float(box_L[0]), float(box_L[1]), float(box_L[2])), photon_typeid(types)), \
           (cai_pointer_stride(force), pe)


def takes_native_route(args, outs) -> bool:
    """True when custom_force_arguments describes HOOMD's own Scalar4 views, i.e. cavmd_compute_soa will forward to the
    HOOMD-native entry point (mirrors the test in csrc/cavmd_capi.hip, cavmd_compute_soa)."""
    n, pos, tid, img, chg, _, _ = args
    frc, pe = outs
    return (pos[1] == 32 and tid[1] == 32 and tid[0] == pos[0] + 24 and img[1] == 12 and chg[1] == 8 and frc[1] == 32
            and pe is not None and pe[1] == 32 and pe[0] == frc[0] + 24 and pos[0] % 16 == 0 and frc[0] % 16 == 0)


def set_forces_custom(workspace, params, stream, n, position, typeid, image, charge, box_L, types, force,
                      potential_energy=None) -> None:
    """One evaluation through the ``force.Custom`` surface (HOOMD-blue works on the null stream: stream = 0)."""
    args, outs = custom_force_arguments(n, position, typeid, image, charge, box_L, types, force, potential_energy)
    n, pos, tid, img, chg, box, L_typeid = args
    workspace.compute_soa(stream, n, pos, tid, img, chg, box, L_typeid, params, outs[0], outs[1])


class EnergyCache:
    """Energies of the most recent evaluation, fetched from the workspace at most once per evaluation.  `bump()` is called
    by whoever enqueues an evaluation; keying on a counter instead of the timestep keeps setParams(...) followed by a
    recomputation at the SAME timestep (sim.run(0) twice) from returning stale numbers, and `clear()` reproduces the
    reference's zeroing when there is nothing to compute (src/CavityForceCompute.cc:148-156)."""

    def __init__(self):
        self._seq = 0
        self._have = -1
        self._e = (0.0, 0.0, 0.0)

    def bump(self) -> None:
        self._seq += 1

    def clear(self) -> None:
        self._seq += 1
        self._have = self._seq
        self._e = (0.0, 0.0, 0.0)

    def get(self, fetch):
        if self._have != self._seq:
            self._e = tuple(fetch())
            self._have = self._seq
        return self._e
