// STAND-IN for hoomd/ForceCompute.h (and what it pulls in: GlobalArray, ArrayHandle, ParticleData, SystemDefinition, BoxDim,
// ExecutionConfiguration) -- NOT HOOMD-blue; see the banner in HOOMDMath.h.  Containers and scopes only, no physics.
#ifndef STANDIN_HOOMD_FORCE_COMPUTE_H_
#define STANDIN_HOOMD_FORCE_COMPUTE_H_

#include "HOOMDMath.h"

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace hoomd
    {
struct access_location
    {
    enum Enum { host, device };
    };
struct access_mode
    {
    enum Enum { read, readwrite, overwrite };
    };

//! Device buffer with acquire/release bookkeeping: at most one ArrayHandle at a time, as HOOMD-blue's GlobalArray enforces
template<class T> class GlobalArray
    {
    public:
    GlobalArray() = default;
    explicit GlobalArray(size_t n) { resize(n); }
    GlobalArray(const GlobalArray&) = delete;
    GlobalArray& operator=(const GlobalArray&) = delete;
    ~GlobalArray()
        {
        if (m_dev)
            (void)hipFree(m_dev);
        }
    void resize(size_t n)
        {
        if (m_acquired)
            throw std::runtime_error("stand-in GlobalArray: resize while a handle is held");
        if (m_dev)
            (void)hipFree(m_dev);
        m_dev = nullptr;
        m_n = n;
        if (n && hipMalloc((void**)&m_dev, n * sizeof(T)) != hipSuccess)
            throw std::runtime_error("stand-in GlobalArray: hipMalloc failed");
        }
    size_t getNumElements() const { return m_n; }
    // bookkeeping the tests read
    unsigned acquisitions() const { return m_acquisitions; }
    bool held() const { return m_acquired; }

    private:
    template<class U> friend class ArrayHandle;
    T* m_dev = nullptr;
    size_t m_n = 0;
    bool m_acquired = false;
    unsigned m_acquisitions = 0;
    };

template<class T> class ArrayHandle
    {
    public:
    ArrayHandle(const GlobalArray<T>& array, access_location::Enum location, access_mode::Enum mode)
        : m_array(const_cast<GlobalArray<T>&>(array)), m_location(location), m_mode(mode)
        {
        if (m_array.m_acquired)
            throw std::runtime_error("stand-in ArrayHandle: array already acquired (a handle scope leaked)");
        m_array.m_acquired = true;
        m_array.m_acquisitions += 1;
        if (location == access_location::device)
            data = m_array.m_dev;
        else
            {
            m_host.resize(m_array.m_n);
            if (mode != access_mode::overwrite && m_array.m_n)
                (void)hipMemcpy(m_host.data(), m_array.m_dev, m_array.m_n * sizeof(T), hipMemcpyDeviceToHost);
            data = m_host.data();
            }
        }
    ~ArrayHandle()
        {
        if (m_location == access_location::host && m_mode != access_mode::read && m_array.m_n)
            (void)hipMemcpy(m_array.m_dev, m_host.data(), m_array.m_n * sizeof(T), hipMemcpyHostToDevice);
        m_array.m_acquired = false;
        }
    ArrayHandle(const ArrayHandle&) = delete;
    T* data = nullptr;

    private:
    GlobalArray<T>& m_array;
    access_location::Enum m_location;
    access_mode::Enum m_mode;
    std::vector<T> m_host;
    };

class BoxDim
    {
    public:
    BoxDim() : m_L {0, 0, 0} { }
    BoxDim(Scalar Lx, Scalar Ly, Scalar Lz) : m_L {Lx, Ly, Lz} { }
    Scalar3 getL() const { return m_L; }

    private:
    Scalar3 m_L;
    };

class ExecutionConfiguration
    {
    public:
    explicit ExecutionConfiguration(bool gpu) : m_gpu(gpu) { }
    bool isCUDAEnabled() const { return m_gpu; } // HOOMD-blue's name for "a GPU (CUDA or HIP) execution configuration"

    private:
    bool m_gpu;
    };

class ParticleData
    {
    public:
    ParticleData(unsigned int N, const BoxDim& box, std::vector<std::string> types, std::shared_ptr<ExecutionConfiguration> exec)
        : m_N(N), m_box(box), m_types(std::move(types)), m_exec(std::move(exec)), m_pos(N), m_charge(N), m_image(N)
        {
        }
    unsigned int getN() const { return m_N; }
    unsigned int getMaxN() const { return m_N; }
    const BoxDim& getGlobalBox() const { return m_box; }
    unsigned int getTypeByName(const std::string& name) const
        {
        for (unsigned int i = 0; i < m_types.size(); ++i)
            if (m_types[i] == name)
                return i;
        throw std::runtime_error("Type " + name + " not found!"); // HOOMD-blue throws here as well
        }
    const GlobalArray<Scalar4>& getPositions() const { return m_pos; }
    const GlobalArray<Scalar>& getCharges() const { return m_charge; }
    const GlobalArray<int3>& getImages() const { return m_image; }
    std::shared_ptr<ExecutionConfiguration> getExecConf() const { return m_exec; }
    //! the stand-in's own: shrink / grow the system between steps (HOOMD-blue: particle insertion, domain migration)
    void setN(unsigned int N)
        {
        m_N = N;
        m_pos.resize(N);
        m_charge.resize(N);
        m_image.resize(N);
        }

    private:
    unsigned int m_N;
    BoxDim m_box;
    std::vector<std::string> m_types;
    std::shared_ptr<ExecutionConfiguration> m_exec;
    GlobalArray<Scalar4> m_pos;
    GlobalArray<Scalar> m_charge;
    GlobalArray<int3> m_image;
    };

class SystemDefinition
    {
    public:
    explicit SystemDefinition(std::shared_ptr<ParticleData> pdata) : m_pdata(std::move(pdata)) { }
    std::shared_ptr<ParticleData> getParticleData() const { return m_pdata; }

    private:
    std::shared_ptr<ParticleData> m_pdata;
    };

class ForceCompute
    {
    public:
    explicit ForceCompute(std::shared_ptr<SystemDefinition> sysdef)
        : m_sysdef(std::move(sysdef)), m_pdata(m_sysdef->getParticleData()), m_exec_conf(m_pdata->getExecConf()),
          m_force(m_pdata->getMaxN())
        {
        }
    virtual ~ForceCompute() = default;
    //! what the integrator calls once per step
    void compute(uint64_t timestep)
        {
        if (m_force.getNumElements() != m_pdata->getN())
            m_force.resize(m_pdata->getN()); // HOOMD-blue reallocates through the max-particle-number signal
        computeForces(timestep);
        }
    const GlobalArray<Scalar4>& getForceArray() const { return m_force; }

    protected:
    virtual void computeForces(uint64_t timestep) = 0;
    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<ParticleData> m_pdata;
    std::shared_ptr<ExecutionConfiguration> m_exec_conf;
    GlobalArray<Scalar4> m_force;
    };
    } // namespace hoomd
#endif
