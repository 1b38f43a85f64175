// Python module _cavitymd_hip (imported by cavitymd/hoomd_plugin.py).  UNBUILT without HOOMD-blue.
#include "CavityForceComputeHIP.h"

PYBIND11_MODULE(_cavitymd_hip, m)
    {
    hoomd::cavitymd::detail::export_CavityForceComputeHIP(m);
    }
