"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.  Run with `-m gpu` on an MI355X.

Tolerances (stated here as the contract):
  P1 dipole         |d_gpu - d_ref|_inf <= 1e-10 * |d_ref|_inf       (ref = sequential fp64 sum, the reference order)
  P2 energies       each of E_h, E_c, E_d: relative <= 1e-10          (never only their sum: it cancels at finite q)
  P3 forces         |F_gpu - F_ref| <= 1e-10 * S_i,  S_i = g|c_i|(|q_xy|_inf + (g/K)|d_xy|_inf) for molecules,
                    S_L = K|q|_inf + g|d_xy|_inf for the photon (the scale before the cancellation in Dq)
  P4 accuracy       |F_gpu - F_exact| <= |F_ref - F_exact| + 1e-14 * S_i   (exact = correctly rounded dipole)
  P5 structure      F.z == 0 and F.w == 0 exactly for molecules, every entry written, no-photon -> all zeros,
                    bit-identical results from run to run
The GPU dipole itself is additionally required to be within 2 ulp of the correctly rounded sum.
"""
import ctypes
import json
import math
import os
import time

import numpy as np
import pytest
import torch

import cavitymd
from cavitymd import _capi, synthetic
from oracle import numpy_mirror as nm

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps


# ---- helpers ---------------------------------------------------------------------------------------------------
def to_device(cfg, device="cuda"):
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                           cfg["box"], device=device)
    return cavitymd.SystemDefinition(pd)


def gpu_eval(cfg, tunables=None):
    sysdef = to_device(cfg)
    p = cfg["params"]
    comp = cavitymd.CavityForceComputeHIP(sysdef, p["omegac"], p["couplstr"], p["phmass"])
    for k, v in (tunables or {}).items():
        comp.workspace.set_tunable(k, v)
    comp.getForceArray().fill_(float("nan"))  # every entry must be overwritten
    comp.compute(0)
    torch.cuda.synchronize()
    res = comp.getResult()
    return {"force": comp.getForceArray().cpu().numpy(), "energies": np.array(comp.getEnergies()),
            "dipole": np.array(res.dipole[:]), "dipole_lo": np.array(res.dipole_lo[:]), "photon_idx": res.photon_idx,
            "n_L": res.n_photon_typed, "q": np.array(res.q[:]), "Dq": np.array(res.Dq[:]), "result": res, "comp": comp}


def ref_eval(ref, oracle_mod, cfg):
    p = ref.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
    pos4 = oracle_mod.pack_pos(cfg["position"], cfg["typeid"])
    out = ref.compute(pos4, cfg["charge"], cfg["image"], cfg["box"], cfg["L_typeid"], p)
    out["params"] = p
    out["pos4"] = pos4
    if out["photon_idx"] >= 0:
        hi, lo = ref.dipole_exact(pos4, cfg["charge"], cfg["image"], cfg["box"], out["photon_idx"])
        out["dipole_exact"] = hi
    return out


def force_scales(cfg, refout):
    p = refout["params"]
    g, K = p["couplstr"], p["K"]
    pidx = refout["photon_idx"]
    box = np.asarray(cfg["box"])
    q = cfg["position"][pidx] + cfg["image"][pidx] * box
    d = refout["dipole"]
    s_mol = g * np.abs(cfg["charge"]) * (np.abs(q[:2]).max() + (g / K) * np.abs(d[:2]).max())
    s_L = K * np.abs(q).max() + g * np.abs(d[:2]).max()
    S = s_mol.copy()
    S[pidx] = s_L
    return S


def forces_from_dipole(cfg, refout, d):
    """Forces the reference formulas give for a prescribed dipole (used with the exactly rounded one)."""
    p = refout["params"]
    g, K = p["couplstr"], p["K"]
    pidx = refout["photon_idx"]
    box = np.asarray(cfg["box"])
    q = cfg["position"][pidx] + cfg["image"][pidx] * box
    Dq = np.array([q[0] + (g / K) * d[0], q[1] + (g / K) * d[1]])
    F = np.zeros((len(cfg["charge"]), 4))
    s = (-g) * cfg["charge"]
    F[:, 0] = s * Dq[0]
    F[:, 1] = s * Dq[1]
    F[cfg["typeid"] == cfg["L_typeid"]] = 0.0
    F[pidx, :3] = [-K * q[0] - g * d[0], -K * q[1] - g * d[1], -K * q[2] - g * 0.0]
    return F


def check_parity(cfg, gpu, refout, tol=1e-10):
    assert gpu["photon_idx"] == refout["photon_idx"]
    assert not np.isnan(gpu["force"]).any(), "force entries left unwritten"
    if refout["photon_idx"] < 0:
        assert not gpu["force"].any() and not gpu["energies"].any()
        return {}
    d_ref, d_gpu, d_exact = refout["dipole"], gpu["dipole"], refout["dipole_exact"]
    # P1
    assert np.abs(d_gpu - d_ref).max() <= tol * np.abs(d_ref).max() + 1e-300
    # GPU dipole is the correctly rounded sum to within 2 ulp
    assert np.all(np.abs(d_gpu - d_exact) <= 2 * np.spacing(np.abs(d_exact)) + 1e-300)
    # P2
    for k in range(3):
        e_ref, e_gpu = refout["energies"][k], gpu["energies"][k]
        assert abs(e_gpu - e_ref) <= tol * abs(e_ref) + 1e-300, ("energy", k, e_gpu, e_ref)
    # P3
    S = force_scales(cfg, refout)
    diff = np.abs(gpu["force"][:, :3] - refout["force"][:, :3])
    assert np.all(diff <= tol * S[:, None] + 1e-300), float((diff / (S[:, None] + 1e-300)).max())
    # P4
    F_exact = forces_from_dipole(cfg, refout, d_exact)
    err_gpu = np.abs(gpu["force"][:, :3] - F_exact[:, :3])
    err_ref = np.abs(refout["force"][:, :3] - F_exact[:, :3])
    assert np.all(err_gpu <= err_ref + 1e-14 * S[:, None] + 1e-300)
    # P5
    mol = np.ones(len(S), dtype=bool)
    mol[refout["photon_idx"]] = False
    assert np.all(gpu["force"][mol, 2] == 0.0) and np.all(gpu["force"][:, 3] == 0.0)
    raw_rel = diff[mol, :2] / (np.abs(refout["force"][mol, :2]) + 1e-300)
    return {"max_scaled_force_err": float((diff / (S[:, None] + 1e-300)).max()),
            "max_raw_rel_force_err": float(raw_rel.max()) if raw_rel.size else 0.0,
            "dipole_rel_err_vs_ref": float(np.abs(d_gpu - d_ref).max() / max(np.abs(d_ref).max(), 1e-300))}


# ---- known answers and golden vectors ----------------------------------------------------------------------------
def test_known_answers_bit_exact(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "kat_golden.json")))
    for case in kat["cases"]:
        cfg = {"position": np.array(case["position"], dtype=float), "typeid": np.array(case["typeid"], dtype=np.int32),
               "charge": np.array(case["charge"], dtype=float), "image": np.array(case["image"], dtype=np.int32),
               "types": ["A", "B", "C"], "box": tuple(case["box"]), "L_typeid": case["L_typeid"],
               "params": {"omegac": case["omegac"], "couplstr": case["couplstr"], "phmass": case["phmass"]}}
        cfg["types"][case["L_typeid"]] = "L"
        out = gpu_eval(cfg)
        assert out["photon_idx"] == case["photon_idx"], case["name"]
        assert np.array_equal(out["force"], np.array(case["force"], dtype=float)), case["name"]
        assert np.array_equal(out["energies"], np.array(case["energies"], dtype=float)), case["name"]
        assert np.array_equal(out["dipole"], np.array(case["dipole"], dtype=float)), case["name"]


def test_config1_against_committed_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "config1_oracle.npz"))
    cfg = synthetic.config1(seed=1)
    out = gpu_eval(cfg)
    assert out["photon_idx"] == 500 == int(g["photon_idx"])
    assert np.all(np.abs(out["dipole"] - g["dipole_exact_hi"]) <= 2 * np.spacing(np.abs(g["dipole_exact_hi"])))
    assert np.allclose(out["energies"], g["energies"], rtol=1e-12, atol=0)
    scale = np.abs(g["force"]).max()
    assert np.abs(out["force"] - g["force"]).max() <= 1e-12 * scale


def test_config1_thousand_step_pseudo_trajectory(ref, oracle_mod):
    """BASELINE config 1: '1k steps' on the init-0.gsd stand-in.  HOOMD integration is impossible here, so the
    positions follow a fixed pseudo-trajectory (r += 1e-3 N(0,1) per step) and GPU and oracle are compared at
    every step, reusing ONE compute object (workspace, force array) as a simulation would."""
    cfg = synthetic.config1(seed=1)
    sysdef = to_device(cfg)
    pd = sysdef.getParticleData()
    p = cfg["params"]
    comp = cavitymd.CavityForceComputeHIP(sysdef, p["omegac"], p["couplstr"], p["phmass"])
    worst = 0.0
    for step in range(1000):
        if step:
            cfg = synthetic.perturb(cfg, step)
            pos4 = oracle_mod.pack_pos(cfg["position"], cfg["typeid"])
            pd.getPositions().copy_(torch.from_numpy(pos4))
            pd.getImages().copy_(torch.from_numpy(cfg["image"]))
        comp.compute(step)
        gpu = {"force": comp.getForceArray().cpu().numpy(), "energies": np.array(comp.getEnergies())}
        res = comp.getResult()
        gpu.update(dipole=np.array(res.dipole[:]), photon_idx=res.photon_idx)
        assert res.sequence == step + 1
        stats = check_parity(cfg, gpu, ref_eval(ref, oracle_mod, cfg))
        worst = max(worst, stats["max_scaled_force_err"])
    assert worst <= 1e-10


# ---- the BASELINE sizes -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("maker,kwargs", [(synthetic.config2, {}), (synthetic.config3, {}), (synthetic.config4, {})],
                         ids=["config2_1e5", "config3_1e6_finite_q", "config4_1e7"])
def test_full_size_parity_and_properties(ref, oracle_mod, maker, kwargs):
    cfg = maker(**kwargs)
    n = len(cfg["charge"])
    gpu = gpu_eval(cfg)
    refout = ref_eval(ref, oracle_mod, cfg)
    stats = check_parity(cfg, gpu, refout)
    print(f"\n{cfg['name']}: N={n} {stats}")
    comp = gpu["comp"]
    # determinism: same bits on a second and third evaluation
    for _ in range(2):
        comp.compute(1)
        torch.cuda.synchronize()
        assert np.array_equal(comp.getForceArray().cpu().numpy(), gpu["force"])
        assert np.array_equal(np.array(comp.getResult().dipole[:]), gpu["dipole"])
    # exact linearity: doubling every charge doubles every addend exactly, so d doubles bit for bit
    pd = comp._pdata
    pd.getCharges().mul_(2.0)
    comp.compute(2)
    torch.cuda.synchronize()
    assert np.array_equal(np.array(comp.getResult().dipole[:]), 2.0 * gpu["dipole"])
    pd.getCharges().mul_(0.5)
    # sum rule: sum_i F_i,xy over molecules = -g Q_tot Dq  (Q_tot ~ 0 for neutral systems -> compare absolutely)
    g = cfg["params"]["couplstr"]
    mol = np.ones(n, dtype=bool)
    mol[gpu["photon_idx"]] = False
    lhs = gpu["force"][mol, :2].sum(axis=0)
    rhs = -g * cfg["charge"][mol].sum() * gpu["Dq"]
    assert np.all(np.abs(lhs - rhs) <= 1e-9 * g * np.abs(cfg["charge"]).sum() * np.abs(gpu["Dq"]).max() + 1e-300)
    # image shift: moving every molecule one box length along x changes d_x by Q_tot * Lx
    pd.getImages()[:-1, 0] += 1
    comp.compute(3)
    torch.cuda.synchronize()
    d_shift = np.array(comp.getResult().dipole[:])
    term_scale = np.abs(cfg["charge"]).sum() * (cfg["box"][0] * 3)
    assert abs((d_shift[0] - gpu["dipole"][0]) - cfg["charge"][mol].sum() * cfg["box"][0]) <= 1e-12 * term_scale
    assert d_shift[1] == gpu["dipole"][1] and d_shift[2] == gpu["dipole"][2]


@pytest.mark.parametrize("replica", list(range(8)), ids=[f"seed{r + 1}" for r in range(8)])
def test_config5_replicas_seeds_1_to_8(ref, oracle_mod, replica):
    """BASELINE config 5: the eight replicas (seeds 1-8) of the 1e6 + photon, finite-q configuration, one after the other
    on the one GPU of this box (on the 8-GPU node each runs on its own GPU: bench.py --gpus 8), each against the oracle
    with the full P1-P5 contract, through the default path (one launch) and through two launches, built exactly as
    bench.py builds rank r's workload."""
    from cavitymd import replicas
    p = synthetic.default_params()
    cfg = synthetic.diatomic_box(1_000_000, seed=replicas.replica_seed(replica + 1, 0), finite_q=True, image_range=1, params=p,
                                 name=f"config5_replica{replica + 1}")
    assert np.array_equal(cfg["position"], synthetic.config5_replica(replica)["position"])   # the same replica both ways
    refout = ref_eval(ref, oracle_mod, cfg)
    one = gpu_eval(cfg)
    assert one["result"].n_particles == 1_000_001 and one["photon_idx"] == 1_000_000
    stats = check_parity(cfg, one, refout)
    two = gpu_eval(cfg, {"persistent": 0})
    check_parity(cfg, two, refout)
    strided = gpu_eval(cfg, {"persistent": 1, "persistent_balanced": 0})
    assert np.array_equal(strided["force"], two["force"]) and np.array_equal(strided["energies"], two["energies"])
    assert np.all(np.abs(one["dipole"] - two["dipole"]) <= np.spacing(np.abs(two["dipole"])))
    print(f"\nreplica {replica + 1}: {stats}")


def test_150_million_particles_64bit_offsets():
    """N = 150 000 001: the pos and force arrays are 4.8 GB each, so every byte offset beyond 2^32 is exercised.
    Inputs are generated on the device; the check uses device-side fp64 reductions (torch) for the dipole and an
    exact element-wise recomputation of the force map from the kernel's own Dq."""
    n = 150_000_001
    dev = "cuda"
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    L = (531.0, 531.0, 531.0)
    pos = torch.empty((n, 4), dtype=torch.float64, device=dev)
    pos[:, :3] = (torch.rand((n, 3), dtype=torch.float64, device=dev, generator=gen) - 0.5) * L[0]
    tags = torch.zeros(n, dtype=torch.int64, device=dev)
    tags[1::2] = 1
    tags[-1] = 2                                              # photon last, as the driver appends it
    pos[:, 3] = tags.view(torch.float64)
    del tags
    charge = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    charge -= charge[:-1].mean()
    charge[-1] = 0.0
    image = torch.randint(-2, 3, (n, 3), dtype=torch.int32, device=dev, generator=gen)
    force = torch.full((n, 4), float("nan"), dtype=torch.float64, device=dev)
    ws = _capi.Workspace(n)
    prm = _capi.make_params(2000.0 / 219474.63, 1e-3, 1.0)
    ws.compute_hoomd(0, n, pos.data_ptr(), charge.data_ptr(), image.data_ptr(), L, 2, prm, force.data_ptr())
    torch.cuda.synchronize()
    res = ws.result()
    assert res.photon_idx == n - 1 and res.n_photon_typed == 1 and res.n_particles == n
    # dipole: torch's tree sum is accurate to ~log2(N) eps relative to sum|t|; ours is compensated
    d = np.array(res.dipole[:])
    Lt = torch.tensor(L, device=dev)
    for k in range(3):
        t = charge[:-1] * (pos[:-1, k] + image[:-1, k].to(torch.float64) * Lt[k])
        want, scale = float(t.sum()), float(t.abs().sum())
        assert abs(d[k] - want) <= 64 * EPS * scale, (k, d[k], want)
        del t
    # force map, exactly: F = ((-g) c) Dq with the kernel's own Dq; z and w zero; photon row = F_L
    g = prm.couplstr
    s = (-g) * charge
    for k in range(2):
        assert torch.equal(force[:-1, k], s[:-1] * res.Dq[k])
    assert not force[:-1, 2:].any() and not torch.isnan(force).any()
    assert force[-1].tolist() == [res.photon_force[0], res.photon_force[1], res.photon_force[2], 0.0]
    # the tail of the arrays (offsets > 4 GiB) really was reached
    assert (n - 1) * 32 > 2**32 and float(force[-2, 0]) == float(s[-2] * res.Dq[0])


# ---- edge cases -------------------------------------------------------------------------------------------------------------
def _random_cfg(n, seed, photon_at=None, L=(31.0, 17.5, 23.25), image_range=3, photon_charge=0.0):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-0.5, 0.5, (n, 3)) * np.asarray(L)
    tid = rng.integers(0, 2, n).astype(np.int32)
    charge = rng.uniform(-1, 1, n)
    if photon_at is not None:
        tid[photon_at] = 2
        charge[photon_at] = photon_charge
    image = rng.integers(-image_range, image_range + 1, (n, 3)).astype(np.int32)
    return {"name": f"rand{n}", "seed": seed, "position": pos, "typeid": tid, "charge": charge, "image": image,
            "types": ["O", "N", "L"], "box": L, "L_typeid": 2,
            "params": {"omegac": 0.0091, "couplstr": 1e-3, "phmass": 1.0}}


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049,
                               4095, 4097, 10_000, 65_537, 262_145, 300_001])
def test_ragged_sizes_and_photon_positions(ref, oracle_mod, n):
    """Every tile-boundary neighbourhood of both kernels, and the sizes where the reference's own GPU path breaks
    (N > 65 536 partial-sum truncation, N > 100 000 early-out, N > 262 144 buffer overrun)."""
    for photon_at in sorted({0, n // 2, n - 1}):
        cfg = _random_cfg(n, seed=n * 7 + photon_at, photon_at=photon_at)
        refout = ref_eval(ref, oracle_mod, cfg)
        check_parity(cfg, gpu_eval(cfg), refout)
        if n <= 4097:
            # every code path at small N: the single-block launch (default up to 1024), several blocks (one launch or two)
            check_parity(cfg, gpu_eval(cfg, {"small_system_max_n": 0}), refout)
            check_parity(cfg, gpu_eval(cfg, {"small_system_max_n": 0, "persistent": 0}), refout)
            check_parity(cfg, gpu_eval(cfg, {"small_system_max_n": 8192}), refout)


def test_no_photon_zeroes_everything(ref, oracle_mod):
    for n in (1, 777, 5000):
        cfg = _random_cfg(n, seed=n)
        out = gpu_eval(cfg)
        assert out["photon_idx"] == -1 and not out["force"].any() and not out["energies"].any()
        assert not out["dipole"].any()
    # no type named 'L' at all: the compute passes -1 (reference GPU class: energies zero, src/CavityForceComputeGPU.cc:114-123)
    cfg = _random_cfg(100, seed=1, photon_at=5)
    cfg["types"] = ["O", "N", "X"]
    out = gpu_eval(cfg)
    assert out["photon_idx"] == -1 and not out["force"].any()


def test_charged_photon_is_excluded_from_dipole(ref, oracle_mod):
    cfg = _random_cfg(3000, seed=3, photon_at=1234, photon_charge=7.5)
    check_parity(cfg, gpu_eval(cfg), ref_eval(ref, oracle_mod, cfg))


def test_several_L_typed_particles(ref, oracle_mod):
    """Degenerate input: only the FIRST 'L' is the photon; later ones enter the dipole but get no force
    (src/CavityForceCompute.cc:122 vs :191).  The reference's GPU kernel gets this wrong (Appendix A.4)."""
    cfg = _random_cfg(5000, seed=9, photon_at=100)
    for extra in (0, 99, 101, 2500, 4999):
        if extra != 100:
            cfg["typeid"][extra] = 2
    cfg["typeid"][0] = 0  # so the first L-typed particle (= the photon) is index 99, not 0
    gpu = gpu_eval(cfg)
    refout = ref_eval(ref, oracle_mod, cfg)
    assert gpu["photon_idx"] == refout["photon_idx"] == 99
    assert gpu["n_L"] == 5
    # the L-sum detour costs one extra rounding: compare at 1e-12 of the scale instead of the 2-ulp dipole test
    assert np.abs(gpu["dipole"] - refout["dipole"]).max() <= 1e-12 * np.abs(refout["dipole"]).max()
    S = force_scales(cfg, refout)
    assert np.all(np.abs(gpu["force"][:, :3] - refout["force"][:, :3]) <= 1e-10 * S[:, None] + 1e-300)
    for i in (101, 2500, 4999):
        assert not gpu["force"][i].any()


def test_large_images_and_garbage_type_high_word(ref, oracle_mod):
    cfg = _random_cfg(4000, seed=21, photon_at=3999, image_range=1 << 20)
    check_parity(cfg, gpu_eval(cfg), ref_eval(ref, oracle_mod, cfg))
    # HOOMD's __int_as_scalar leaves the high word of pos.w unspecified: only the low 32 bits are the type
    sysdef = to_device(cfg)
    pd = sysdef.getParticleData()
    w = pd.getPositions()[:, 3].view(torch.int64)
    w |= (0x5EADBEEF << 32)
    comp = cavitymd.CavityForceComputeHIP(sysdef, 0.0091, 1e-3, 1.0)
    comp.compute(0)
    assert comp.getResult().photon_idx == 3999


def test_non_finite_inputs_stay_non_finite(ref, oracle_mod):
    """A NaN or infinite coordinate of a charged molecule poisons the dipole in the reference and here alike, hence every
    molecular x/y force; the exact flavour is NOT preserved (TwoSum turns an infinite addend into NaN, a plain sum keeps
    Inf).  Stated so that nobody relies on it; finite inputs never produce non-finite outputs."""
    for bad in (float("nan"), float("inf")):
        cfg = _random_cfg(2000, seed=55, photon_at=1999)
        cfg["position"][17, 0] = bad
        gpu = gpu_eval(cfg)
        refout = ref_eval(ref, oracle_mod, cfg)
        assert not np.isfinite(refout["dipole"][0]) and not np.isfinite(gpu["dipole"][0])
        assert np.isfinite(refout["dipole"][1]) and gpu["dipole"][1] == pytest.approx(refout["dipole"][1], rel=1e-12)
        mol = cfg["charge"] != 0
        mol[1999] = False
        assert not np.isfinite(gpu["force"][mol, 0]).any() and not np.isfinite(refout["force"][mol, 0]).any()
        assert np.isfinite(gpu["force"][mol, 1]).all() and np.all(gpu["force"][:1999, 2:] == 0.0)


def test_tunables_do_not_change_the_physics(ref, oracle_mod):
    cfg = _random_cfg(200_003, seed=5, photon_at=200_002)
    base = gpu_eval(cfg)
    refout = ref_eval(ref, oracle_mod, cfg)
    check_parity(cfg, base, refout)
    for tun in ({"reduce_blocks_per_cu": 1}, {"reduce_blocks_per_cu": 16}, {"map_blocks_per_cu": 1},
                {"map_blocks_per_cu": 16, "map_nt_store": 1}, {"map_nt_store": 0}, {"fused_finalize": 0}, {"fused_finalize": 0, "reduce_blocks_per_cu": 8},
                {"reduce_nt_load": 0}, {"reduce_nt_load": 2}, {"reduce_blocks_per_cu": 3}, {"small_system_max_n": 1 << 20}, {"map_reverse": 1}, {"map_reverse": 0, "map_blocks_per_cu": 1},
                {"map_reverse": 1, "map_blocks_per_cu": 7}, {"fused_finalize": 1, "reduce_blocks_per_cu": 5, "map_nt_store": 1},
                {"persistent": 0}, {"persistent": 1, "persistent_balanced": 0}, {"persistent": 1, "reduce_blocks_per_cu": 2},
                {"persistent": 1, "reduce_blocks_per_cu": 4, "map_nt_store": 0}):
        out = gpu_eval(cfg, tun)
        check_parity(cfg, out, refout)
        assert np.all(np.abs(out["dipole"] - base["dipole"]) <= np.spacing(np.abs(base["dipole"])))


# ---- the ABI itself ----------------------------------------------------------------------------------------------------------
def test_abi_argument_validation():
    lib = _capi.load()
    ws = _capi.Workspace(1000)
    n = 1000
    pos = torch.zeros(n, 4, dtype=torch.float64, device="cuda")
    chg = torch.zeros(n, dtype=torch.float64, device="cuda")
    img = torch.zeros(n, 3, dtype=torch.int32, device="cuda")
    frc = torch.full((n, 4), 7.0, dtype=torch.float64, device="cuda")
    prm = _capi.make_params(0.0091, 1e-3, 1.0)

    def call(N, pos_p, chg_p, img_p, frc_p, params):
        return lib.cavmd_compute_hoomd(ws.handle, None, N, pos_p, chg_p, img_p, 10.0, 10.0, 10.0, 2,
                                       ctypes.byref(params) if params is not None else None, frc_p)

    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert call(n, None, P(chg), P(img), P(frc), prm) == _capi.CAVMD_ERR_INVALID_VALUE
    assert call(n, P(pos), P(chg), P(img), None, prm) == _capi.CAVMD_ERR_INVALID_VALUE
    assert call(n, P(pos), P(chg), P(img), P(frc), None) == _capi.CAVMD_ERR_INVALID_VALUE
    assert call(n, ctypes.c_void_p(pos.data_ptr() + 8), P(chg), P(img), P(frc), prm) == _capi.CAVMD_ERR_INVALID_VALUE
    assert call(n + 1, P(pos), P(chg), P(img), P(frc), prm) == _capi.CAVMD_ERR_CAPACITY
    bad = _capi.make_params(0.0, 1e-3, 1.0)  # K == 0
    assert call(n, P(pos), P(chg), P(img), P(frc), bad) == _capi.CAVMD_ERR_BAD_PARAMS
    # N == 0 succeeds and touches nothing (src/CavityForceComputeGPU.cu:530-532)
    assert call(0, P(pos), P(chg), P(img), P(frc), prm) == 0
    torch.cuda.synchronize()
    assert torch.all(frc == 7.0)
    # energies before any evaluation read 0.0 like the reference's freshly constructed compute
    assert ws.energies() == (0.0, 0.0, 0.0)
    with pytest.raises(_capi.CavmdError):
        ws.result()
    info = ws.device_info()
    assert info["arch"].startswith("gfx950") and info["compute_units"] == 256
    with pytest.raises(_capi.CavmdError):
        ws.set_tunable("reduce_blocks_per_cu", 0)
    with pytest.raises(_capi.CavmdError):
        ws.set_tunable("nonsense", 1)


def test_side_stream_and_result_device_pointer(ref, oracle_mod):
    cfg = _random_cfg(50_000, seed=8, photon_at=49_999)
    sysdef = to_device(cfg)
    comp = cavitymd.CavityForceComputeHIP(sysdef, 0.0091, 1e-3, 1.0)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    comp.compute(0, stream=side)
    e = comp.getEnergies()  # syncs `side`, not the device
    refout = ref_eval(ref, oracle_mod, cfg)
    assert np.allclose(e, refout["energies"], rtol=1e-10, atol=0)
    assert comp.workspace.result_device_ptr() != 0


def test_graph_capture_and_replay(ref, oracle_mod):
    """cavmd_compute_* only enqueues kernels, so it can be captured into a hipGraph and replayed on new data."""
    cfg = _random_cfg(30_000, seed=12, photon_at=29_999)
    sysdef = to_device(cfg)
    pd = sysdef.getParticleData()
    comp = cavitymd.CavityForceComputeHIP(sysdef, 0.0091, 1e-3, 1.0)
    comp.compute(0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        comp.compute(1)
    cfg2 = synthetic.perturb(cfg, 1, amplitude=0.5)
    pd.getPositions().copy_(torch.from_numpy(oracle_mod.pack_pos(cfg2["position"], cfg2["typeid"])))
    pd.getImages().copy_(torch.from_numpy(cfg2["image"]))
    comp.getForceArray().fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    gpu = {"force": comp.getForceArray().cpu().numpy(), "energies": np.array(comp.getEnergies())}
    res = comp.getResult()
    gpu.update(dipole=np.array(res.dipole[:]), photon_idx=res.photon_idx)
    check_parity(cfg2, gpu, ref_eval(ref, oracle_mod, cfg2))


# ---- the snapshot-layout entry point (hoomd.md.force.Custom surface) -----------------------------------------------------------
def _soa_eval(cfg, hoomd_views, with_pe=True):
    n = len(cfg["charge"])
    prm = _capi.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
    ws = _capi.Workspace(n)
    dev = "cuda"
    chg = torch.from_numpy(cfg["charge"]).to(dev)
    img = torch.from_numpy(cfg["image"]).to(dev)
    if hoomd_views:
        # HOOMD's local snapshot hands out strided views of its Scalar4 buffers: position = pos[:, :3] (stride 32),
        # typeid = int view of pos.w (stride 32), force = force4[:, :3] (stride 32), potential_energy = force4[:, 3]
        pos4 = torch.from_numpy(np.ascontiguousarray(
            np.concatenate([cfg["position"], cavitymd.state.type_tag_as_double(cfg["typeid"])[:, None]], axis=1))).to(dev)
        frc4 = torch.full((n, 4), float("nan"), dtype=torch.float64, device=dev)
        position, typeid = (pos4.data_ptr(), 32), (pos4.data_ptr() + 24, 32)
        force, pe = (frc4.data_ptr(), 32), (frc4.data_ptr() + 24, 32)
        keep = (pos4, frc4)
    else:
        pos3 = torch.from_numpy(np.ascontiguousarray(cfg["position"])).to(dev)
        tid = torch.from_numpy(cfg["typeid"].astype(np.int32)).to(dev)
        frc3 = torch.full((n, 3), float("nan"), dtype=torch.float64, device=dev)
        pe1 = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
        position, typeid = (pos3.data_ptr(), 24), (tid.data_ptr(), 4)
        force, pe = (frc3.data_ptr(), 24), (pe1.data_ptr(), 8)
        keep = (pos3, tid, frc3, pe1)
    ws.compute_soa(0, n, position, typeid, (img.data_ptr(), 12), (chg.data_ptr(), 8), cfg["box"], cfg["L_typeid"], prm,
                   force, pe if with_pe else None)
    torch.cuda.synchronize()
    res = ws.result()
    if hoomd_views:
        f4 = keep[1].cpu().numpy()
    else:
        f4 = np.concatenate([keep[2].cpu().numpy(), keep[3].cpu().numpy()[:, None]], axis=1)
    if not with_pe:
        assert np.isnan(f4[:, 3]).all()
        f4[:, 3] = 0.0
    return {"force": f4, "energies": np.array(ws.energies()), "dipole": np.array(res.dipole[:]),
            "photon_idx": res.photon_idx}


@pytest.mark.parametrize("hoomd_views", [False, True], ids=["packed", "hoomd_strided_views"])
def test_snapshot_layout_entry_point(ref, oracle_mod, hoomd_views):
    for n, photon_at in ((1, 0), (1025, 7), (70_001, 70_000)):
        cfg = _random_cfg(n, seed=n + 3, photon_at=photon_at)
        check_parity(cfg, _soa_eval(cfg, hoomd_views), ref_eval(ref, oracle_mod, cfg))
    cfg = _random_cfg(999, seed=2)  # no photon
    out = _soa_eval(cfg, hoomd_views)
    assert out["photon_idx"] == -1 and not out["force"].any()
    cfg = _random_cfg(5000, seed=4, photon_at=4000)
    check_parity(cfg, _soa_eval(cfg, hoomd_views, with_pe=False), ref_eval(ref, oracle_mod, cfg))


def test_both_layouts_give_identical_bits():
    for n, tun in ((123_457, {}), (20_001, {}), (1_500, {"small_system_max_n": 0})):
        cfg = _random_cfg(n, seed=77, photon_at=n - 1)
        # (the single-block path exists for the AoS layout only: switched off for the small case; the snapshot layout
        # runs as two launches, so the AoS side uses the same partition of the particles)
        a = gpu_eval(cfg, dict(tun, persistent_balanced=0))
        b = _soa_eval(cfg, hoomd_views=False)
        assert np.array_equal(a["dipole"], b["dipole"]) and np.array_equal(a["energies"], b["energies"])
        assert np.array_equal(a["force"], b["force"])


def test_randomised_small_systems_one_workspace(ref, oracle_mod):
    """200 seeded random systems through ONE workspace and ONE set of device buffers (as a long simulation reuses them):
    sizes 1..3000, photon anywhere or absent, magnitudes from 1e-150 to 1e+6, zero and negative-zero charges."""
    cap = 3000
    ws = _capi.Workspace(cap)
    dev = "cuda"
    pos_d = torch.empty((cap, 4), dtype=torch.float64, device=dev)
    chg_d = torch.empty((cap,), dtype=torch.float64, device=dev)
    img_d = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    frc_d = torch.empty((cap, 4), dtype=torch.float64, device=dev)
    rng = np.random.default_rng(2025)
    for case in range(200):
        n = int(rng.integers(1, cap + 1))
        scale = 10.0 ** rng.integers(-3, 7)
        L = tuple(float(v) for v in scale * rng.uniform(0.5, 2.0, 3))
        photon_at = int(rng.integers(0, n)) if rng.random() < 0.85 else None
        cfg = _random_cfg(n, seed=10_000 + case, photon_at=photon_at, L=L, image_range=int(rng.integers(0, 6)))
        mag = rng.choice([1.0, 1e-150, 1e-8, 1e3])
        cfg["charge"] = cfg["charge"] * mag
        zero = rng.random(n) < 0.1
        cfg["charge"][zero] = np.where(rng.random(zero.sum()) < 0.5, 0.0, -0.0)
        if photon_at is not None:
            cfg["charge"][photon_at] = 0.0
        cfg["params"] = {"omegac": float(10.0 ** rng.uniform(-3, 0)), "couplstr": float(10.0 ** rng.uniform(-4, 0)),
                         "phmass": float(rng.uniform(0.5, 2.0))}
        pos_d[:n].copy_(torch.from_numpy(oracle_mod.pack_pos(cfg["position"], cfg["typeid"])))
        chg_d[:n].copy_(torch.from_numpy(cfg["charge"]))
        img_d[:n].copy_(torch.from_numpy(cfg["image"]))
        frc_d.fill_(float("nan"))
        prm = _capi.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
        ws.compute_hoomd(0, n, pos_d.data_ptr(), chg_d.data_ptr(), img_d.data_ptr(), cfg["box"], 2, prm, frc_d.data_ptr())
        torch.cuda.synchronize()
        res = ws.result()
        gpu = {"force": frc_d[:n].cpu().numpy(), "energies": np.array(ws.energies()), "dipole": np.array(res.dipole[:]),
               "photon_idx": res.photon_idx}
        assert np.isnan(frc_d[n:].cpu().numpy()).all(), "wrote past N"
        check_parity(cfg, gpu, ref_eval(ref, oracle_mod, cfg))


def test_randomised_mid_size_systems_one_workspace(ref, oracle_mod):
    """60 seeded random systems of 2 049 ... 400 000 particles through ONE workspace and ONE set of device buffers: the
    single-launch kernel's grid, tile depth, ragged tail and LDS footprint change from call to call while its granule slabs
    and epoch word live on (a stale record of an earlier, larger grid must never be mistaken for a fresh one); photon
    anywhere or absent, sometimes several L-typed particles, charge magnitudes from 1e-150 to 1e3."""
    cap = 400_000
    ws = _capi.Workspace(cap)
    dev = "cuda"
    pos_d = torch.empty((cap, 4), dtype=torch.float64, device=dev)
    chg_d = torch.empty((cap,), dtype=torch.float64, device=dev)
    img_d = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    frc_d = torch.empty((cap, 4), dtype=torch.float64, device=dev)
    rng = np.random.default_rng(4242)
    for case in range(60):
        n = int(rng.choice([rng.integers(2049, 9000), rng.integers(9000, 140_000), rng.integers(140_000, cap + 1)]))
        L = tuple(float(v) for v in 10.0 ** rng.integers(0, 4) * rng.uniform(0.5, 2.0, 3))
        photon_at = int(rng.integers(0, n)) if rng.random() < 0.9 else None
        cfg = _random_cfg(n, seed=77_000 + case, photon_at=photon_at, L=L, image_range=int(rng.integers(0, 4)))
        cfg["charge"] = cfg["charge"] * rng.choice([1.0, 1e-150, 1e-8, 1e3])
        if photon_at is not None:
            cfg["charge"][photon_at] = 0.0
            if rng.random() < 0.15:                      # degenerate: more L-typed particles after the photon
                extra = rng.integers(photon_at, n, 3)
                cfg["typeid"][extra] = 2
        cfg["params"] = {"omegac": float(10.0 ** rng.uniform(-3, 0)), "couplstr": float(10.0 ** rng.uniform(-4, 0)),
                         "phmass": float(rng.uniform(0.5, 2.0))}
        pos_d[:n].copy_(torch.from_numpy(oracle_mod.pack_pos(cfg["position"], cfg["typeid"])))
        chg_d[:n].copy_(torch.from_numpy(cfg["charge"]))
        img_d[:n].copy_(torch.from_numpy(cfg["image"]))
        frc_d.fill_(float("nan"))
        prm = _capi.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
        ws.compute_hoomd(0, n, pos_d.data_ptr(), chg_d.data_ptr(), img_d.data_ptr(), cfg["box"], 2, prm, frc_d.data_ptr())
        torch.cuda.synchronize()
        res = ws.result()
        assert res.sequence == case + 1 and res.n_particles == n
        assert np.isnan(frc_d[n:n + 64].cpu().numpy()).all(), "wrote past N"
        refout = ref_eval(ref, oracle_mod, cfg)
        gpu = {"force": frc_d[:n].cpu().numpy(), "energies": np.array(ws.energies()), "dipole": np.array(res.dipole[:]),
               "photon_idx": res.photon_idx}
        if res.n_photon_typed > 1:
            # the L-sum detour costs one extra rounding: scale-relative bounds instead of the 2-ulp dipole test
            assert gpu["photon_idx"] == refout["photon_idx"]
            assert np.abs(gpu["dipole"] - refout["dipole"]).max() <= 1e-12 * np.abs(refout["dipole"]).max() + 1e-300
            S = force_scales(cfg, refout)
            assert np.all(np.abs(gpu["force"][:, :3] - refout["force"][:, :3]) <= 1e-10 * S[:, None] + 1e-300)
        else:
            check_parity(cfg, gpu, refout)


# ---- observable that reuses the reduction ------------------------------------------------------------
def test_total_dipole_observable(oracle_mod):
    """cavmd_result.total_dipole = the reference's compute_total_dipole_moment (src/cavitymd/analysis.py:18-31):
    np.dot(charge, unwrapped_positions) over ALL particles, photon included -- without a snapshot round trip."""
    for cfg in (_random_cfg(100_000, seed=31, photon_at=99_999, photon_charge=0.0),
                _random_cfg(77_777, seed=32, photon_at=123, photon_charge=2.5),   # a charged photon counts here
                _random_cfg(5_000, seed=33)):                                      # no photon at all
        out = gpu_eval(cfg)
        unwrapped = cavitymd.unwrap_positions(cfg["position"], cfg["image"], np.asarray(cfg["box"]))
        want = np.dot(cfg["charge"], unwrapped)          # the reference's expression
        terms = np.abs(cfg["charge"][:, None] * unwrapped).sum(axis=0)
        got = np.array(out["result"].total_dipole[:])
        assert np.all(np.abs(got - want) <= 1e-13 * terms)
        exact = np.array([math.fsum((cfg["charge"] * unwrapped[:, k]).tolist()) for k in range(3)])
        # total = the double-double molecular sum plus the L-typed sum, rounded once: within 4 ulp of the exact total
        assert np.all(np.abs(got - exact) <= 4 * np.spacing(np.abs(exact)) + 1e-300)
        if out["photon_idx"] < 0:
            assert not out["dipole"].any() and got.any()   # force-path dipole is zeroed, the observable is not


# ---- the user-facing object ---------------------------------------------------------------------------------------------------
def test_cavity_force_object_end_to_end(ref, oracle_mod):
    cfg = synthetic.config1(seed=3)
    p = cfg["params"]
    f = cavitymd.CavityForce(kvector=[0, 0, 1], couplstr=p["couplstr"], omegac=p["omegac"], phmass=p["phmass"])
    f.attach(to_device(cfg))
    f.compute(0)
    refout = ref_eval(ref, oracle_mod, cfg)
    assert f.implementation == "hip"
    assert f.harmonic_energy == pytest.approx(refout["energies"][0], rel=1e-10)
    assert f.coupling_energy == pytest.approx(refout["energies"][1], rel=1e-10)
    assert f.dipole_self_energy == pytest.approx(refout["energies"][2], rel=1e-10)
    assert f.total_cavity_energy == f.harmonic_energy + f.coupling_energy + f.dipole_self_energy == f.energy
    assert f.forces.shape == (501, 3)
    S = force_scales(cfg, refout)
    assert np.all(np.abs(f.forces - refout["force"][:, :3]) <= 1e-10 * S[:, None])
    # setParams at run time (src/CavityForceCompute.cc:48-51)
    f._force_impl.setParams(p["omegac"], 2 * p["couplstr"], p["phmass"])
    assert f._force_impl.getParams()["couplstr"] == 2 * p["couplstr"]
    f.compute(1)
    assert f.coupling_energy == pytest.approx(2 * refout["energies"][1], rel=1e-10)
    f.detach()
    assert f.energy == 0.0


# ---- the single-launch evaluation (cavmd_persistent_kernel.hpp) ----------------------------------------------------------------
@pytest.mark.parametrize("n", [2049, 3000, 10_000, 65_537, 131_073, 262_145, 1_000_001, 3_000_017, 6_000_001])
def test_single_launch_gives_the_bits_of_two_launches(ref, oracle_mod, n):
    """With the tiles dealt round-robin (the two-launch path's partition) the one-launch kernel has the same grid, the same
    tiles and the same fold order: it must reproduce the two-launch path bit for bit (forces, dipole high and low words,
    energies, photon) -- any stale, torn or misplaced granule of the in-launch all-reduce would change bits.  With balanced
    contiguous shares (a tunable; measured slower, off by default) the partials differ, so the dipole may move by an ulp; that variant is checked against
    the oracle with the full contract.  6e6 particles: more tiles than a block's LDS holds, the overflow is re-read."""
    for photon_at in sorted({0, n // 3, n - 1}):
        cfg = _random_cfg(n, seed=n + photon_at, photon_at=photon_at)
        two = gpu_eval(cfg, {"persistent": 0})
        one = gpu_eval(cfg, {"persistent": 1, "persistent_balanced": 0})
        assert one["result"].n_partials == two["result"].n_partials
        assert np.array_equal(one["force"], two["force"])
        assert np.array_equal(one["dipole"], two["dipole"]) and np.array_equal(one["dipole_lo"], two["dipole_lo"])
        assert np.array_equal(one["energies"], two["energies"]) and one["photon_idx"] == two["photon_idx"] == photon_at
        assert np.array_equal(np.array(one["result"].total_dipole[:]), np.array(two["result"].total_dipole[:]))
        bal = gpu_eval(cfg, {"persistent": 1, "persistent_balanced": 1})
        assert bal["photon_idx"] == photon_at
        assert np.all(np.abs(bal["dipole"] - two["dipole"]) <= np.spacing(np.abs(two["dipole"])))
        if photon_at == n - 1 and n <= 3_000_017:
            check_parity(cfg, bal, ref_eval(ref, oracle_mod, cfg))


def test_single_launch_degenerate_inputs(ref, oracle_mod):
    # no photon -> zeros; several L-typed particles -> the slow path that reads the type tags
    cfg = _random_cfg(50_000, seed=5)
    out = gpu_eval(cfg, {"persistent": 1})
    assert out["photon_idx"] == -1 and not out["force"].any() and not out["energies"].any()
    cfg = _random_cfg(50_000, seed=9, photon_at=100)
    for extra in (99, 101, 25_000, 49_999):
        cfg["typeid"][extra] = 2
    one, two = gpu_eval(cfg, {"persistent": 1, "persistent_balanced": 0}), gpu_eval(cfg, {"persistent": 0})
    assert one["photon_idx"] == 99 and one["n_L"] == 5
    assert np.array_equal(one["force"], two["force"]) and np.array_equal(one["dipole"], two["dipole"])
    refout = ref_eval(ref, oracle_mod, cfg)
    S = force_scales(cfg, refout)
    assert np.all(np.abs(one["force"][:, :3] - refout["force"][:, :3]) <= 1e-10 * S[:, None] + 1e-300)
    bal = gpu_eval(cfg, {"persistent": 1, "persistent_balanced": 1})
    assert bal["photon_idx"] == 99 and bal["n_L"] == 5
    assert np.all(np.abs(bal["force"][:, :3] - refout["force"][:, :3]) <= 1e-10 * S[:, None] + 1e-300)
    for i in (101, 25_000, 49_999):
        assert not bal["force"][i].any()


def test_single_launch_hand_off_under_load_and_reuse():
    """The in-launch gather, hammered: 3000 evaluations through ONE workspace on data that changes every step (a stale
    or torn granule from an earlier evaluation would change bits), with a second stream keeping the memory system busy
    half of the time (uneven load), every result compared bit for bit with the two-launch path on the same frame."""
    n = 300_001
    frames = 6
    cfgs = [_random_cfg(n, seed=900 + f, photon_at=n - 1) for f in range(frames)]
    dev = "cuda"
    pos = [torch.from_numpy(np.concatenate([c["position"], cavitymd.state.type_tag_as_double(c["typeid"])[:, None]], axis=1)).to(dev) for c in cfgs]
    chg = [torch.from_numpy(c["charge"]).to(dev) for c in cfgs]
    img = [torch.from_numpy(c["image"]).to(dev) for c in cfgs]
    prm = _capi.make_params(0.0091, 1e-3, 1.0)
    L = cfgs[0]["box"]
    want_f, want_d = [], []
    ws2 = _capi.Workspace(n)
    ws2.set_tunable("persistent", 0)
    frc = torch.empty((n, 4), dtype=torch.float64, device=dev)
    for f in range(frames):
        ws2.compute_hoomd(0, n, pos[f].data_ptr(), chg[f].data_ptr(), img[f].data_ptr(), L, 2, prm, frc.data_ptr())
        torch.cuda.synchronize()
        want_f.append(frc.clone())
        want_d.append(np.array(ws2.result().dipole[:]))
    ws1 = _capi.Workspace(n)
    ws1.set_tunable("persistent", 1)
    ws1.set_tunable("persistent_balanced", 0)   # the two-launch path's partition, so that every bit must agree
    side = torch.cuda.Stream()
    junk = torch.empty(64 * 2**20, dtype=torch.float32, device=dev)
    out = [torch.empty((n, 4), dtype=torch.float64, device=dev) for _ in range(frames)]
    bad = 0
    for it in range(3000):
        f = (it * 5 + it // 7) % frames
        if (it // 50) % 2:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)   # a streaming kernel on another stream: CUs and memory queues are contended
        ws1.compute_hoomd(0, n, pos[f].data_ptr(), chg[f].data_ptr(), img[f].data_ptr(), L, 2, prm, out[f].data_ptr())
        if it % 10 == 9:
            r = ws1.result()             # flag protocol, no device synchronisation
            assert r.sequence == it + 1 and np.array_equal(np.array(r.dipole[:]), want_d[f])
            bad += int(not torch.equal(out[f], want_f[f]))
            out[f].fill_(float("nan"))
    torch.cuda.synchronize()
    assert bad == 0


def test_a_sync_timeout_is_reported_once_and_the_workspace_falls_back_to_two_launches():
    """A single-launch evaluation whose blocks starve each other ends in a time-out (a real one is provoked by
    `make microbench_persistent_fault`, profiles/r02/fault_injection.txt).  What the library does with it, exercised through the
    fault-injection tunable: the NEXT call (enqueue or result read) reports CAVMD_ERR_SYNC_TIMEOUT exactly once, enqueues
    nothing, invalidates the stale result, and from then on the workspace evaluates with two launches -- same bits."""
    n = 60_001
    cfg = _random_cfg(n, seed=77, photon_at=n - 1)
    dev = "cuda"
    pos = torch.from_numpy(np.concatenate([cfg["position"], cavitymd.state.type_tag_as_double(cfg["typeid"])[:, None]], axis=1)).to(dev)
    chg = torch.from_numpy(cfg["charge"]).to(dev)
    img = torch.from_numpy(cfg["image"]).to(dev)
    prm = _capi.make_params(0.0091, 1e-3, 1.0)
    L = cfg["box"]
    ws = _capi.Workspace(n, hooks=True)              # libcavmd_hooks.so: the product library cannot raise the flag by hand
    assert ws.get_tunable("test_hooks") == 1
    assert ws.get_tunable("persistent") == -1 and ws.get_tunable("sync_timeout_seen") == 0
    frc = torch.empty((n, 4), dtype=torch.float64, device=dev)
    ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())
    want_d = np.array(ws.result().dipole[:])
    want_f = frc.clone()

    for notice_in in ("compute", "result"):
        ws.set_tunable("persistent", -1)
        ws.set_tunable("sync_timeout_seen", 0)
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())
        ws.set_tunable("sync_timeout_seen", 1)          # as if that evaluation's kernel had given up
        frc.fill_(float("nan"))
        with pytest.raises(_capi.CavmdError) as ei:
            if notice_in == "compute":
                ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())
            else:
                ws.result()
        assert ei.value.status == _capi.CAVMD_ERR_SYNC_TIMEOUT
        torch.cuda.synchronize()
        assert bool(torch.isnan(frc).all())              # the reporting call enqueued nothing
        with pytest.raises(_capi.CavmdError) as ei:
            ws.result()                                  # the stale result is not handed out as current
        assert ei.value.status == _capi.CAVMD_ERR_NOT_COMPUTED
        assert ws.get_tunable("persistent_suspended") == 2 and ws.get_tunable("sync_timeout_seen") == 1   # for good
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())   # two launches now
        assert np.array_equal(np.array(ws.result().dipole[:]), want_d)
        torch.cuda.synchronize()
        assert torch.equal(frc, want_f)


@pytest.mark.parametrize("n", [2049, 4097, 60_001, 1_000_001])
def test_a_starved_evaluation_is_completed_by_its_last_block(n):
    """A REAL starved single-launch evaluation, provoked through the library's test hooks: one block of the grid starts 20 ms
    late while the others' bounded waits are cut to a few ms, so they all give up and leave, as they would if another grid held
    that block's CU.  The late block gives up last and completes the evaluation alone.  Expected: no error from any call, the
    result and EVERY force entry bit for bit those of the two-launch path, and the single launch suspended afterwards.
    Late block = the first (a group leader: it would otherwise find all group totals and finish on its own), one in the
    middle, and the last (the block that normally publishes the result)."""
    cfg = _random_cfg(n, seed=4100 + n % 97, photon_at=n // 3)
    dev = "cuda"
    pos = torch.from_numpy(np.concatenate([cfg["position"], cavitymd.state.type_tag_as_double(cfg["typeid"])[:, None]], axis=1)).to(dev)
    chg = torch.from_numpy(cfg["charge"]).to(dev)
    img = torch.from_numpy(cfg["image"]).to(dev)
    prm = _capi.make_params(0.0091, 1e-3, 1.0)
    L = cfg["box"]
    frc = torch.empty((n, 4), dtype=torch.float64, device=dev)
    ref_ws = _capi.Workspace(n)
    ref_ws.set_tunable("persistent", 0)
    ref_ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())
    want = ref_ws.result()
    torch.cuda.synchronize()
    want_f = frc.clone()
    grid = want.n_partials                      # the single-launch grid is the two-launch path's reduction grid
    for late in sorted({0, min(5, grid - 1), grid - 1}):
        ws = _capi.Workspace(n, hooks=True)
        ws.set_tunable("persistent", 1)
        ws.set_tunable("debug_spin_limit", 5000)
        ws.set_tunable("debug_late_block", late)
        ws.set_tunable("debug_late_ticks", 2_000_000)
        frc.fill_(float("nan"))
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())
        got = ws.result()                        # waits for the repaired evaluation; no error
        assert got.sequence == 1 and got.n_partials == grid
        for field in ("dipole", "dipole_lo", "total_dipole", "energy", "photon_force", "q"):
            if hasattr(want, field):
                assert np.array_equal(np.array(getattr(got, field)[:]), np.array(getattr(want, field)[:])), (late, field)
        torch.cuda.synchronize()
        assert torch.equal(frc.view(torch.int64), want_f.view(torch.int64)), f"late block {late} of {grid}"
        assert ws.get_tunable("persistent_suspended") == 1 and ws.get_tunable("sync_timeout_seen") == 1
        # two launches for now, no fault hook in that path: same bits again
        frc.fill_(float("nan"))
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())
        assert np.array_equal(np.array(ws.result().dipole[:]), np.array(want.dipole[:]))
        torch.cuda.synchronize()
        assert torch.equal(frc.view(torch.int64), want_f.view(torch.int64))


def test_a_suspended_single_launch_is_probed_again_with_back_off():
    """After a starved (repaired) evaluation the workspace evaluates with two launches for a pause, then tries the single
    launch again; a probe that starves again is repaired again and the pause grows eightfold.  Which path an evaluation took
    is read from the per-kernel timers (the force-map slot is used by the two-launch path only)."""
    n = 60_001
    cfg = _random_cfg(n, seed=515, photon_at=n - 1)
    dev = "cuda"
    pos = torch.from_numpy(np.concatenate([cfg["position"], cavitymd.state.type_tag_as_double(cfg["typeid"])[:, None]], axis=1)).to(dev)
    chg = torch.from_numpy(cfg["charge"]).to(dev)
    img = torch.from_numpy(cfg["image"]).to(dev)
    prm = _capi.make_params(0.0091, 1e-3, 1.0)
    L = cfg["box"]
    frc = torch.empty((n, 4), dtype=torch.float64, device=dev)
    ws = _capi.Workspace(n, hooks=True)
    ws.set_tunable("persistent", 1)
    ws.profile_enable(True)

    def evaluate():
        """-> (launches used, dipole)"""
        frc.fill_(float("nan"))
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc.data_ptr())
        d = np.array(ws.result().dipole[:])
        ms, count = ws.profile_read()
        assert count == 1
        torch.cuda.synchronize()
        assert not bool(torch.isnan(frc).any())
        return (2 if ms[2] > 0 else 1), d

    launches, want = evaluate()
    assert launches == 1 and ws.get_tunable("persistent_suspended") == 0
    # evaluation 2 starves (block 3 starts late) and is repaired; first pause cut from 2^16 to 4 evaluations
    for k, v in (("debug_spin_limit", 5000), ("debug_late_block", 3), ("debug_late_ticks", 2_000_000), ("debug_suspend_first", 4)):
        ws.set_tunable(k, v)
    launches, d = evaluate()
    assert launches == 1 and np.array_equal(d, want) and ws.get_tunable("persistent_suspended") == 1
    ws.set_tunable("debug_late_block", -1)            # whoever held the CU has gone
    path = []
    for _ in range(6):
        launches, d = evaluate()
        assert np.array_equal(d, want)
        path.append(launches)
    assert path == [2, 2, 2, 2, 1, 1], path           # four evaluations on two launches, then the probe, which stays
    assert ws.get_tunable("persistent_suspended") == 0
    # the next starvation comes soon after the probe: the pause is eight times as long
    ws.set_tunable("debug_late_block", 3)
    launches, d = evaluate()
    assert launches == 1 and np.array_equal(d, want) and ws.get_tunable("persistent_suspended") == 1
    ws.set_tunable("debug_late_block", -1)
    path = [evaluate()[0] for _ in range(34)]
    assert path == [2] * 32 + [1, 1], path


def test_environment_switch_for_shared_gpus(monkeypatch):
    """CAVMD_PERSISTENT is read by cavmd_create: 0 = two launches (GPUs shared by more processes than fit), 1 = always one."""
    for env, want in (("0", 0), ("1", 1), ("bogus", -1)):
        monkeypatch.setenv("CAVMD_PERSISTENT", env)
        assert _capi.Workspace(1000).get_tunable("persistent") == want
    monkeypatch.delenv("CAVMD_PERSISTENT")
    assert _capi.Workspace(1000).get_tunable("persistent") == -1


def test_graph_replays_on_changing_data_read_without_sync(ref, oracle_mod):
    """Two captures (single-launch and two-launch), several replays each on data that changes between replays, results read
    straight after graph.replay() with NO device synchronisation by the caller: a captured workspace must not trust the
    host-visible flag (the sequence argument is frozen in the graph) and waits for the device itself."""
    for tun in ({"persistent": 1}, {"persistent": 0}):
        cfg = _random_cfg(40_000, seed=12, photon_at=39_999)
        sysdef = to_device(cfg)
        pd = sysdef.getParticleData()
        comp = cavitymd.CavityForceComputeHIP(sysdef, 0.0091, 1e-3, 1.0)
        for k, v in tun.items():
            comp.workspace.set_tunable(k, v)
        comp.compute(0)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            comp.compute(1)
        cur = cfg
        for rep in range(1, 5):
            cur = synthetic.perturb(cur, rep, amplitude=0.5)
            pd.getPositions().copy_(torch.from_numpy(oracle_mod.pack_pos(cur["position"], cur["typeid"])))
            pd.getImages().copy_(torch.from_numpy(cur["image"]))
            graph.replay()
            e = np.array(comp.getEnergies())      # no torch.cuda.synchronize() in between
            res = comp.getResult()
            gpu = {"force": comp.getForceArray().cpu().numpy(), "energies": e, "dipole": np.array(res.dipole[:]),
                   "photon_idx": res.photon_idx}
            check_parity(cur, gpu, ref_eval(ref, oracle_mod, cur))


def _device_inputs(n, seed, photon_at):
    cfg = _random_cfg(n, seed=seed, photon_at=photon_at)
    dev = "cuda"
    pos = torch.from_numpy(np.concatenate([cfg["position"], cavitymd.state.type_tag_as_double(cfg["typeid"])[:, None]], axis=1)).to(dev)
    return cfg, pos, torch.from_numpy(cfg["charge"]).to(dev), torch.from_numpy(cfg["image"]).to(dev)


def test_result_read_reports_an_evaluation_that_never_published():
    """The stream has drained and the host-visible flag does not carry the evaluation's sequence: its launch failed on the
    device.  cavmd_result_read must say so (hipErrorLaunchFailure, as the scalar reductions' wait does) and invalidate the
    block -- NOT hand out the previous evaluation's numbers with CAVMD_OK.  Provoked through the hooks build: the kernels
    publish into a scratch block the host never reads.  All three launch shapes."""
    for n, tun in ((501, {}), (60_001, {}), (60_001, {"persistent": 0})):
        cfg, pos, chg, img = _device_inputs(n, seed=n, photon_at=n - 1)
        prm = _capi.make_params(0.0091, 1e-3, 1.0)
        frc = torch.empty((n, 4), dtype=torch.float64, device="cuda")
        ws = _capi.Workspace(n, hooks=True)
        for k, v in tun.items():
            ws.set_tunable(k, v)
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), cfg["box"], 2, prm, frc.data_ptr())
        first = np.array(ws.result().dipole[:])
        ws.set_tunable("debug_skip_publish", 1)
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), cfg["box"], 2, prm, frc.data_ptr())
        with pytest.raises(_capi.CavmdError) as e:
            ws.result()
        assert e.value.status > 0                                   # a hipError_t, not a CAVMD_ERR_* code
        with pytest.raises(_capi.CavmdError) as e:
            ws.result()                                             # and the stale block is not handed out afterwards either
        assert e.value.status == _capi.CAVMD_ERR_NOT_COMPUTED
        assert ws.energies() == (0.0, 0.0, 0.0) or list(ws.energies()) == [0.0, 0.0, 0.0]
        ws.set_tunable("debug_skip_publish", 0)
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), cfg["box"], 2, prm, frc.data_ptr())
        assert np.array_equal(np.array(ws.result().dipole[:]), first)


@pytest.mark.parametrize("n", [3_001, 60_001])
def test_launches_queued_behind_an_evaluation_nobody_could_complete_fail_as_a_whole(n):
    """a starved evaluation that ends UNREPAIRED used to leave the in-kernel hand-off state dirty (stuck
    give-up count, epoch not advanced) with only the host to clean it -- launches already queued behind it ran on that state
    and could end with a result flagged valid.  Now the failing evaluation leaves a poison word that every block of every
    later launch reads at entry: those launches fail at once and as a whole.  Provoked for real: one block of the grid never
    publishes its record (hooks build), spin limit cut; then TWO more evaluations are enqueued with no host call in between
    that could notice anything (the host flag is raised only when the first kernel's waits run out, milliseconds later).
    n = 3001: a one-hop grid (the silent block completes on its own, the count can never reach G: the advisor's case);
    n = 60001: two levels."""
    cfg, pos, chg, img = _device_inputs(n, seed=n + 7, photon_at=n // 2)
    prm = _capi.make_params(0.0091, 1e-3, 1.0)
    L = cfg["box"]
    frc = [torch.full((n, 4), 7.0, dtype=torch.float64, device="cuda") for _ in range(3)]
    ref_ws = _capi.Workspace(n)
    want = torch.empty((n, 4), dtype=torch.float64, device="cuda")
    ref_ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, want.data_ptr())
    want_d = np.array(ref_ws.result().dipole[:])
    ws = _capi.Workspace(n, hooks=True)
    ws.set_tunable("persistent", 1)
    ws.set_tunable("small_system_max_n", 0)
    ws.set_tunable("debug_spin_limit", 5000)
    ws.set_tunable("debug_silent_block", 1)
    t0 = time.perf_counter()
    for k in range(3):                                   # all three enqueued back to back, asynchronously
        ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc[k].data_ptr())
    assert time.perf_counter() - t0 < 0.5
    ws.set_tunable("debug_silent_block", -1)
    with pytest.raises(_capi.CavmdError) as e:
        ws.result()
    assert e.value.status == _capi.CAVMD_ERR_SYNC_TIMEOUT
    torch.cuda.synchronize()
    # evaluation 1: the blocks that gave up poisoned their tiles; 2 and 3 ran behind it on the poisoned state: NaN everywhere
    assert bool(torch.isnan(frc[0]).any())
    assert bool(torch.isnan(frc[1]).all()) and bool(torch.isnan(frc[2]).all())
    with pytest.raises(_capi.CavmdError) as e:
        ws.result()                                      # nothing of this is ever handed out as a result
    assert e.value.status == _capi.CAVMD_ERR_NOT_COMPUTED
    assert ws.get_tunable("persistent_suspended") == 2
    # the caller switches the single launch on again: the host wipes slabs, count and poison before it runs; same bits as ever
    ws.set_tunable("persistent", 1)
    ws.compute_hoomd(0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), L, 2, prm, frc[0].data_ptr())
    assert np.array_equal(np.array(ws.result().dipole[:]), want_d)
    torch.cuda.synchronize()
    assert torch.equal(frc[0].view(torch.int64), want.view(torch.int64))


def test_replays_of_a_captured_graph_behind_an_uncompleted_evaluation_fail_as_a_whole():
    """The same for a captured graph: the host-side recovery does not reach into it, so after a replay that nobody could
    complete every further replay has to fail by itself -- at once, as a whole, with nothing published -- until the caller
    captures again.  What tells a replay from the failed one is the launch nonce (the address of the replay's own AQL
    packet): kernel arguments, sequence number included, are frozen in a graph."""
    n = 3_001                                                       # one-hop grid: the silent block completes on its own,
    cfg, pos, chg, img = _device_inputs(n, seed=91, photon_at=7)    # the give-up count can never reach G
    prm = _capi.make_params(0.0091, 1e-3, 1.0)
    frc = torch.full((n, 4), 7.0, dtype=torch.float64, device="cuda")
    ws = _capi.Workspace(n, hooks=True)
    for k, v in (("persistent", 1), ("small_system_max_n", 0), ("debug_spin_limit", 5000), ("debug_silent_block", 1)):
        ws.set_tunable(k, v)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ws.compute_hoomd(torch.cuda.current_stream().cuda_stream, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), cfg["box"], 2,
                         prm, frc.data_ptr())
    graph.replay()                                                  # starves itself, ends uncompleted: poison stays
    torch.cuda.synchronize()
    assert bool(torch.isnan(frc).any()) and not bool(torch.isnan(frc).all())
    for _ in range(3):
        frc.fill_(7.0)
        t0 = time.perf_counter()
        graph.replay()
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.05                      # no bounded wait: the replay gives up at entry
        assert bool(torch.isnan(frc).all())
    with pytest.raises(_capi.CavmdError) as e:
        ws.result()
    assert e.value.status == _capi.CAVMD_ERR_SYNC_TIMEOUT
