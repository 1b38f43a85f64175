"""Second, independent CPU restatement of the cavity force in numpy -- TEST INFRASTRUCTURE.

Purpose: cross-check oracle/cavity_ref.c with code that shares nothing with it, and provide
exactly rounded (math.fsum) and exact rational (fractions.Fraction) evaluations as accuracy
yardsticks.  Follows src/CavityForceCompute.cc:73-208 of the reference; the three places where
the reference's *Python* fallback (src/cavitymd/cavity_force_python.py:75,101,137-141) diverges
from its C++ are NOT reproduced (photon found by type id of 'L', photon excluded from the
dipole, L-typed particles get no molecular force).
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np


def type_ids(pos4: np.ndarray) -> np.ndarray:
    """Type id = low 32 bits of the bit pattern of pos.w (HOOMD __scalar_as_int)."""
    w = np.ascontiguousarray(pos4[:, 3], dtype=np.float64)
    return (w.view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)


def unwrap(pos4: np.ndarray, image: np.ndarray, box_L) -> np.ndarray:
    """r = p + img * L, rounding after the multiply and after the add (numpy never fuses)."""
    L = np.asarray(box_L, dtype=np.float64)
    return pos4[:, :3] + image.astype(np.float64) * L[None, :]


def terms(pos4, charge, image, box_L) -> np.ndarray:
    """The fp64-rounded addends charge_i * r_i, shape (N,3)."""
    return np.asarray(charge, dtype=np.float64)[:, None] * unwrap(pos4, image, box_L)


def compute(pos4, charge, image, box_L, L_typeid: int, params: dict, summation: str = "sequential") -> dict:
    """One evaluation.  summation: 'sequential' (reference order, python loop -- small N only),
    'fsum' (exactly rounded sum of the same addends) or 'pairwise' (numpy's tree)."""
    pos4 = np.asarray(pos4, dtype=np.float64)
    charge = np.asarray(charge, dtype=np.float64)
    image = np.asarray(image, dtype=np.int32)
    n = pos4.shape[0]
    g, K = params["couplstr"], params["K"]
    force = np.zeros((n, 4), dtype=np.float64)
    tid = type_ids(pos4)
    hits = np.nonzero(tid == np.int32(L_typeid))[0] if n else np.array([], dtype=np.int64)
    if hits.size == 0:
        return {"force": force, "energies": np.zeros(3), "dipole": np.zeros(3), "photon_idx": -1}
    pidx = int(hits[0])
    r = unwrap(pos4, image, box_L)
    t = charge[:, None] * r
    keep = np.ones(n, dtype=bool)
    keep[pidx] = False
    tk = t[keep]
    if summation == "sequential":
        d = [0.0, 0.0, 0.0]
        for row in tk.tolist():
            d[0] += row[0]
            d[1] += row[1]
            d[2] += row[2]
        d = np.array(d)
    elif summation == "fsum":
        d = np.array([math.fsum(tk[:, k].tolist()) for k in range(3)])
    elif summation == "pairwise":
        d = tk.sum(axis=0)
    else:
        raise ValueError(summation)
    q = r[pidx]
    e_h = 0.5 * K * (q[0] * q[0] + q[1] * q[1] + q[2] * q[2])
    e_c = g * (d[0] * q[0] + d[1] * q[1] + 0.0 * 0.0)
    e_d = 0.5 * (g * g / K) * (d[0] * d[0] + d[1] * d[1] + 0.0 * 0.0)
    gK = g / K
    Dq = np.array([q[0] + gK * d[0], q[1] + gK * d[1]])
    mol = tid != np.int32(L_typeid)
    s = (-g) * charge
    force[mol, 0] = (s * Dq[0])[mol]
    force[mol, 1] = (s * Dq[1])[mol]
    force[pidx, 0] = -K * q[0] - g * d[0]
    force[pidx, 1] = -K * q[1] - g * d[1]
    force[pidx, 2] = -K * q[2] - g * 0.0
    return {"force": force, "energies": np.array([e_h, e_c, e_d]), "dipole": d, "photon_idx": pidx, "Dq": Dq,
            "q": q.copy()}


def dipole_rational(pos4, charge, image, box_L, photon_idx: int):
    """Exact real-arithmetic dipole sum_i c_i (p_i + img_i L) as Fractions (tiny N only): no rounding at all."""
    L = [Fraction(float(v)) for v in box_L]
    d = [Fraction(0), Fraction(0), Fraction(0)]
    for i in range(pos4.shape[0]):
        if i == photon_idx:
            continue
        c = Fraction(float(charge[i]))
        for k in range(3):
            d[k] += c * (Fraction(float(pos4[i, k])) + int(image[i, k]) * L[k])
    return d


def hamiltonian(pos4, charge, image, box_L, L_typeid: int, params: dict) -> float:
    """H = 1/2 K q.q + g q_xy.d_xy + (g^2/2K) d_xy.d_xy in extended precision (longdouble), for the
    finite-difference force check (docs/theory.rst of the reference; src/CavityForceCompute.cc:174-176)."""
    ld = np.longdouble
    tid = type_ids(np.asarray(pos4, dtype=np.float64))
    pidx = int(np.nonzero(tid == np.int32(L_typeid))[0][0])
    r = np.asarray(pos4[:, :3], dtype=ld) + np.asarray(image, dtype=ld) * np.asarray(box_L, dtype=ld)[None, :]
    c = np.asarray(charge, dtype=ld)
    keep = np.ones(len(c), dtype=bool)
    keep[pidx] = False
    d = (c[keep, None] * r[keep]).sum(axis=0)
    q = r[pidx]
    g, K = ld(params["couplstr"]), ld(params["K"])
    return float(ld(0.5) * K * (q @ q) + g * (q[0] * d[0] + q[1] * d[1]) + ld(0.5) * g * g / K * (d[0] * d[0] + d[1] * d[1]))
