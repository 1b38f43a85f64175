"""cavitymd -- MI355X-native cavity-MD force engine (the cavity-force hot path of
muhammadhasyim/cav-hoomd behind the same Python surface).

Installed into a HOOMD-blue tree this package takes the place of ``hoomd.cavitymd`` for the cavity
force; standalone (no HOOMD, as in the build/test image) it is imported as ``cavitymd`` from the
``cav-hoomd_amd/`` directory.

    from cavitymd import CavityForce, PhysicalConstants, unwrap_positions
"""
from .utils import PhysicalConstants, unwrap_positions
from .forces import CavityForce
from .compute import CavityForceComputeHIP
from .state import BoxDim, ParticleData, SystemDefinition
from . import _capi, observables, replicas, synthetic, thermostats

__all__ = [
    "CavityForce", "CavityForceComputeHIP", "PhysicalConstants", "unwrap_positions", "BoxDim", "ParticleData",
    "SystemDefinition", "observables", "replicas", "synthetic", "thermostats",
]
__version__ = "0.1.0"
