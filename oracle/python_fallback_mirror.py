"""Restatement of the reference's pure-Python force, `CavityForcePython.set_forces`
(src/cavitymd/cavity_force_python.py:65-149) -- TEST INFRASTRUCTURE.

The reference ships two implementations of the same physics: the C++ class (our oracle, cavity_ref.c) and this Python
fallback.  They differ in three documented ways: the fallback finds the cavity particle by
`typeid == 1` (:75), leaves the photon row IN the dipole sum (:101, relying on its charge being 0), and gives every
non-cavity particle a force regardless of type (:137-141).  On well-formed inputs (exactly one cavity particle, with
charge 0) the two must agree -- which cross-validates the oracle's formulas against a second, independently written
source inside the reference.  `cavity_typeid` is a parameter here (the reference hard-codes 1).
"""
from __future__ import annotations

import numpy as np


def set_forces(position, typeid, image, charge, box_L, couplstr, omegac, phmass=1.0, cavity_typeid=1):
    position = np.asarray(position, dtype=np.float64)
    image = np.asarray(image)
    charge = np.asarray(charge, dtype=np.float64)
    K = phmass * omegac**2                                                     # :46
    n = len(position)
    force = np.zeros((n, 3))
    cavity_indices = np.where(np.asarray(typeid) == cavity_typeid)[0]           # :75-76
    if len(cavity_indices) == 0:                                                # :78-81
        return {"force": force, "energies": np.zeros(3), "cavity_idx": -1}
    cavity_idx = cavity_indices[0]                                              # :85
    unwrapped = position + image * np.asarray(box_L, dtype=np.float64)[None, :]  # :94-98, utils.unwrap_positions
    dipole_moment = np.dot(charge, unwrapped)                                   # :101 (photon row included)
    dipole_xy = dipole_moment.copy()
    dipole_xy[2] = 0.0
    cavity_position = unwrapped[cavity_idx]
    cavity_xy = cavity_position.copy()
    cavity_xy[2] = 0.0
    harmonic = 0.5 * K * np.dot(cavity_position, cavity_position)               # :112
    coupling = couplstr * np.dot(cavity_xy, dipole_xy)                          # :115
    dipole_self = 0.5 * (couplstr**2 / K) * np.dot(dipole_xy, dipole_xy)        # :118
    force_factor = cavity_xy + (couplstr / K) * dipole_xy                       # :135
    for i in range(n):                                                          # :137-141
        if i != cavity_idx:
            force[i] = -couplstr * charge[i] * force_factor
    force[cavity_idx] = -K * cavity_position - couplstr * dipole_xy             # :144-145
    return {"force": force, "energies": np.array([harmonic, coupling, dipole_self]), "cavity_idx": int(cavity_idx),
            "dipole": dipole_moment}
