// Python module _cavitymd_hip_standin: the repository's HOOMD shim (cav-hoomd_amd/csrc/hoomd_shim/CavityForceComputeHIP.cc,
// compiled UNCHANGED) on top of the STAND-IN declarations in this directory -- NOT HOOMD-blue (banner: hoomd/HOOMDMath.h).
// Exposes just enough of the stand-in containers for tests/test_hoomd_cpp_shim.py to drive the shim the way HOOMD-blue's
// integrator would: build a system from numpy arrays, call ForceCompute::compute(timestep), read m_force back.
#include "hoomd/ForceCompute.h"

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>

#include "CavityForceComputeHIP.h"

namespace py = pybind11;
using namespace hoomd;

PYBIND11_MODULE(_cavitymd_hip_standin, m)
    {
    m.attr("IS_STAND_IN") = true;
    py::class_<ExecutionConfiguration, std::shared_ptr<ExecutionConfiguration>>(m, "ExecutionConfiguration")
        .def(py::init<bool>(), py::arg("gpu"));
    py::class_<ParticleData, std::shared_ptr<ParticleData>>(m, "ParticleData")
        .def(py::init([](unsigned int N, std::array<double, 3> L, std::vector<std::string> types,
                         std::shared_ptr<ExecutionConfiguration> exec)
                      { return std::make_shared<ParticleData>(N, BoxDim(L[0], L[1], L[2]), std::move(types), std::move(exec)); }))
        .def("getN", &ParticleData::getN)
        .def("setN", &ParticleData::setN)
        .def("set_arrays",
             [](ParticleData& pd, py::array_t<double, py::array::c_style | py::array::forcecast> pos4,
                py::array_t<double, py::array::c_style | py::array::forcecast> charge,
                py::array_t<int, py::array::c_style | py::array::forcecast> image)
             {
                 const size_t n = pd.getN();
                 if ((size_t)pos4.size() != 4 * n || (size_t)charge.size() != n || (size_t)image.size() != 3 * n)
                     throw std::runtime_error("stand-in set_arrays: shapes do not match N");
                 ArrayHandle<Scalar4> h_pos(pd.getPositions(), access_location::host, access_mode::overwrite);
                 ArrayHandle<Scalar> h_charge(pd.getCharges(), access_location::host, access_mode::overwrite);
                 ArrayHandle<int3> h_image(pd.getImages(), access_location::host, access_mode::overwrite);
                 if (n)
                     {
                     std::memcpy(h_pos.data, pos4.data(), 32 * n);
                     std::memcpy(h_charge.data, charge.data(), 8 * n);
                     std::memcpy(h_image.data, image.data(), 12 * n);
                     }
             })
        .def("acquisitions",
             [](const ParticleData& pd)
             {
                 return py::make_tuple(pd.getPositions().acquisitions(), pd.getCharges().acquisitions(),
                                       pd.getImages().acquisitions());
             })
        .def("any_handle_held",
             [](const ParticleData& pd)
             { return pd.getPositions().held() || pd.getCharges().held() || pd.getImages().held(); });
    py::class_<SystemDefinition, std::shared_ptr<SystemDefinition>>(m, "SystemDefinition")
        .def(py::init<std::shared_ptr<ParticleData>>());
    py::class_<ForceCompute, std::shared_ptr<ForceCompute>>(m, "ForceCompute")
        .def("compute", &ForceCompute::compute)
        .def("force",
             [](ForceCompute& fc)
             {
                 const size_t n = fc.getForceArray().getNumElements();
                 py::array_t<double> out({n, (size_t)4});
                 ArrayHandle<Scalar4> h(fc.getForceArray(), access_location::host, access_mode::read);
                 if (n)
                     std::memcpy(out.mutable_data(), h.data, 32 * n);
                 return out;
             })
        .def("fill_force",
             [](ForceCompute& fc, double v)
             {
                 ArrayHandle<Scalar4> h(fc.getForceArray(), access_location::host, access_mode::overwrite);
                 for (size_t i = 0; i < fc.getForceArray().getNumElements(); ++i)
                     h.data[i] = make_double4(v, v, v, v);
             })
        .def("force_handle_held", [](const ForceCompute& fc) { return fc.getForceArray().held(); });
    hoomd::cavitymd::detail::export_CavityForceComputeHIP(m);
    }
