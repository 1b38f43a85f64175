"""Row f1, the part that is possible without HOOMD-blue: the HOOMD-free marshalling the plugin uses (cavitymd/marshal.py).

CPU part: pointer / stride extraction from ``__cuda_array_interface__`` exporters (the protocol HOOMD's GPU local-snapshot
arrays and torch CUDA tensors share), the 'L' type lookup, the attach ladder, the energy cache keyed on the evaluation
counter.  GPU part (-m gpu): torch tensors laid out exactly as HOOMD hands them out -- strided VIEWS of Scalar4 buffers --
and as packed arrays, marshalled as ``CavityForceCustomHIP.set_forces`` does and checked against the oracle.

hoomd_plugin.py itself still cannot be imported here (no HOOMD-blue): it stays UNVERIFIED, see INTEGRATION.md.
"""
import numpy as np
import pytest
import torch

import cavitymd
from cavitymd import _capi, marshal


class FakeCAI:
    """Anything with a __cuda_array_interface__ dict (no GPU needed to test the arithmetic on it)."""

    def __init__(self, ptr, shape, typestr, strides=None):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": typestr, "strides": strides,
                                         "version": 3}


def test_pointer_and_stride_extraction():
    # C-contiguous arrays export strides=None: the stride is the row size
    assert marshal.cai_pointer_stride(FakeCAI(0x1000, (10, 3), "<f8")) == (0x1000, 24)
    assert marshal.cai_pointer_stride(FakeCAI(0x1000, (10,), "<f8")) == (0x1000, 8)
    assert marshal.cai_pointer_stride(FakeCAI(0x1000, (10, 3), "<i4")) == (0x1000, 12)
    assert marshal.cai_pointer_stride(FakeCAI(0x1000, (10,), "<u4")) == (0x1000, 4)
    # HOOMD's views of its Scalar4 buffers: position = pos4[:, :3], typeid = int view of pos4.w, pe = force4[:, 3]
    assert marshal.cai_pointer_stride(FakeCAI(0x2000, (10, 3), "<f8", (32, 8))) == (0x2000, 32)
    assert marshal.cai_pointer_stride(FakeCAI(0x2018, (10,), "<i4", (32,))) == (0x2018, 32)
    # an empty system may export a null pointer
    assert marshal.cai_pointer_stride(FakeCAI(0, (0, 3), "<f8")) == (0, 24)
    with pytest.raises(ValueError):
        marshal.cai_pointer_stride(FakeCAI(0, (5, 3), "<f8"))                  # null pointer, non-empty
    with pytest.raises(ValueError):
        marshal.cai_pointer_stride(FakeCAI(0x1000, (10, 3), "<f8", (24, 16)))  # components not contiguous
    with pytest.raises(ValueError):
        marshal.cai_pointer_stride(FakeCAI(0x1000, (), "<f8"))


def test_type_lookup_and_attach_ladder():
    assert marshal.photon_typeid(["O", "N", "L"]) == 2 and marshal.photon_typeid(("L", "A")) == 0
    assert marshal.photon_typeid(["O", "N"]) == -1                       # no type named L: zeros, not an error
    assert marshal.choose_route(True, False, True) == marshal.ROUTE_HIP == "hip"
    assert marshal.choose_route(True, True, True) == marshal.ROUTE_HIP_CUSTOM == "hip_custom"   # force_python=True
    assert marshal.choose_route(False, False, True) == "hip_custom"     # shim not built against HOOMD's headers
    with pytest.raises(RuntimeError):
        marshal.choose_route(True, False, False)                         # hoomd.device.CPU(): no CPU implementation


def test_native_route_detection_on_fake_views():
    pos, frc = 0x10000, 0x90000
    args, outs = marshal.custom_force_arguments(
        7, FakeCAI(pos, (7, 3), "<f8", (32, 8)), FakeCAI(pos + 24, (7,), "<i4", (32,)), FakeCAI(0x50000, (7, 3), "<i4"),
        FakeCAI(0x60000, (7,), "<f8"), (10.0, 11.0, 12.0), ["O", "N", "L"], FakeCAI(frc, (7, 3), "<f8", (32, 8)),
        FakeCAI(frc + 24, (7,), "<f8", (32,)))
    assert args == (7, (pos, 32), (pos + 24, 32), (0x50000, 12), (0x60000, 8), (10.0, 11.0, 12.0), 2)
    assert outs == ((frc, 32), (frc + 24, 32)) and marshal.takes_native_route(args, outs)
    args, outs = marshal.custom_force_arguments(
        7, FakeCAI(pos, (7, 3), "<f8"), FakeCAI(0x40000, (7,), "<u4"), FakeCAI(0x50000, (7, 3), "<i4"),
        FakeCAI(0x60000, (7,), "<f8"), (10.0, 11.0, 12.0), ["O", "N"], FakeCAI(frc, (7, 3), "<f8"), None)
    assert args[1] == (pos, 24) and args[6] == -1 and outs == ((frc, 24), None) and not marshal.takes_native_route(args, outs)


def test_energy_cache_is_keyed_on_the_evaluation_not_the_timestep():
    calls = []

    def fetch():
        calls.append(1)
        return (1.0 * len(calls), 2.0, 3.0)

    c = marshal.EnergyCache()
    assert c.get(fetch) == (1.0, 2.0, 3.0) and c.get(fetch) == (1.0, 2.0, 3.0) and len(calls) == 1   # one fetch per evaluation
    c.bump()                                     # setParams + sim.run(0): same timestep, new evaluation
    assert c.get(fetch) == (2.0, 2.0, 3.0) and len(calls) == 2
    c.clear()                                    # N == 0: the reference zeroes its energies
    assert c.get(fetch) == (0.0, 0.0, 0.0) and len(calls) == 2


# ---- GPU part ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["hoomd_views", "packed"])
def test_custom_route_marshalling_on_cuda_array_interface_exporters(ref, oracle_mod, layout):
    from test_gpu_parity import _random_cfg, check_parity, ref_eval
    for n, photon_at in ((3000, 2999), (70_001, 5)):
        cfg = _random_cfg(n, seed=n + 17, photon_at=photon_at)
        dev = "cuda"
        chg = torch.from_numpy(cfg["charge"]).to(dev)
        img = torch.from_numpy(cfg["image"]).to(dev)
        if layout == "hoomd_views":
            pos4 = torch.from_numpy(oracle_mod.pack_pos(cfg["position"], cfg["typeid"])).to(dev)
            frc4 = torch.full((n, 4), float("nan"), dtype=torch.float64, device=dev)
            position, typeid = pos4[:, :3], pos4.view(torch.int32)[:, 6]         # strided views, as HOOMD exports them
            force, pe = frc4[:, :3], frc4[:, 3]
        else:
            position = torch.from_numpy(np.ascontiguousarray(cfg["position"])).to(dev)
            typeid = torch.from_numpy(cfg["typeid"].astype(np.int32)).to(dev)
            force = torch.full((n, 3), float("nan"), dtype=torch.float64, device=dev)
            pe = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
        assert all(hasattr(t, "__cuda_array_interface__") for t in (position, typeid, img, chg, force, pe))
        args, outs = marshal.custom_force_arguments(n, position, typeid, img, chg, cfg["box"], cfg["types"], force, pe)
        assert marshal.takes_native_route(args, outs) == (layout == "hoomd_views")
        ws = _capi.Workspace(n)
        prm = _capi.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
        cache = marshal.EnergyCache()
        marshal.set_forces_custom(ws, prm, 0, n, position, typeid, img, chg, cfg["box"], cfg["types"], force, pe)
        cache.bump()
        torch.cuda.synchronize()
        res = ws.result()
        f4 = np.concatenate([force.cpu().numpy(), pe.cpu().numpy()[:, None]], axis=1)
        gpu = {"force": f4, "energies": np.array(cache.get(ws.energies)), "dipole": np.array(res.dipole[:]),
               "photon_idx": res.photon_idx}
        check_parity(cfg, gpu, ref_eval(ref, oracle_mod, cfg))
        # setParams at the same "timestep": the cache must hand out the NEW energies after the re-evaluation
        prm2 = _capi.make_params(cfg["params"]["omegac"], 2 * cfg["params"]["couplstr"], cfg["params"]["phmass"])
        marshal.set_forces_custom(ws, prm2, 0, n, position, typeid, img, chg, cfg["box"], cfg["types"], force, pe)
        cache.bump()
        assert cache.get(ws.energies)[1] == pytest.approx(2 * gpu["energies"][1], rel=1e-12)
