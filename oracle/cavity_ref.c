/* cavity_ref.c -- CPU ORACLE for the cavity-force hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A plain-C restatement of the reference's CPU algorithm, CavityForceCompute::computeForces
 * (reference: src/CavityForceCompute.cc:134-208), kept deliberately close to it: same pass
 * structure, same summation order, same operator association, a fresh heap array for the
 * unwrapped positions on every call, a memset of the force array.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product (libcavmd.so) never does.
 *
 * Pinning status: the reference holds NO golden vector, known-answer test or fixture for this
 * path (its tests cover only the thermostat) and the reference itself can neither be compiled
 * nor imported in this environment (HOOMD-blue is absent), so by the reference's own material
 * PARITY IS UNPINNED.  What pins this file instead (tests/test_oracle_*.py):
 *   - the unwrap convention and the unit constants are checked against outputs of the reference's
 *     own src/cavitymd/utils.py (the one module that loads standalone), committed as
 *     tests/golden/utils_*.json with the generating script;
 *   - K for 2000 cm^-1 reproduces the value the reference's notebook prints (8.30408e-05,
 *     examples/05_advanced_run.ipynb:669);
 *   - closed-form known answers of the formulas at src/CavityForceCompute.cc:174-207;
 *   - F = -dH/dx by finite differences of H = 1/2 K q^2 + g q.d + (g^2/2K) d^2 (docs/theory.rst);
 *   - an exactly rounded (math.fsum / double-double) evaluation of the same sums.
 *
 * Build WITHOUT FMA contraction and without -march flags (see oracle/Makefile): the reference is
 * built flag-less for x86-64, so `p + img*L` and `c*r` round after every operation.
 *
 * The typed layouts are declared locally (not via include/cavmd.h) so that the oracle shares no
 * code with the product; tests assert the two agree on sizes and offsets.
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct
{
    double x, y, z, w;
} ref_scalar4; /* HOOMD Scalar4, double build */
typedef struct
{
    int32_t x, y, z;
} ref_int3; /* HOOMD int3 */
typedef struct
{
    double x, y, z;
} ref_vec3; /* HOOMD vec3<Scalar> */
typedef struct
{
    double omegac, couplstr, K, phmass;
} ref_params; /* src/CavityForceCompute.h:28-54 */

#define REF_API __attribute__((visibility("default")))

/* HOOMD's __scalar_as_int for a double Scalar reads the int that shares storage with the low four
 * bytes of the double (a union {int; Scalar}) -- [HOOMD upstream, HOOMDMath.h]. */
static int ref_scalar_as_int(double w)
{
    int32_t lo;
    memcpy(&lo, &w, sizeof(lo));
    return (int)lo;
}

/* cavity_force_params(omegac, couplstr, phmass): K = phmass * omegac * omegac
 * (src/CavityForceCompute.h:38-42) -- evaluated left to right. */
REF_API void cavref_make_params(double omegac, double couplstr, double phmass, ref_params* out)
{
    out->omegac = omegac;
    out->couplstr = couplstr;
    out->phmass = phmass;
    out->K = phmass * omegac * omegac;
}

/* findPhotonParticle, src/CavityForceCompute.cc:73-89: first match, -1 if none. */
REF_API int cavref_find_photon(const ref_scalar4* pos, unsigned int N, int L_typeid)
{
    for (unsigned int i = 0; i < N; i++)
    {
        if (ref_scalar_as_int(pos[i].w) == L_typeid)
            return (int)i;
    }
    return -1;
}

/* computeUnwrappedPositions, src/CavityForceCompute.cc:91-111. */
REF_API void cavref_unwrap(ref_vec3* unwrapped, const ref_scalar4* pos, const ref_int3* image, double Lx, double Ly,
                           double Lz, unsigned int N)
{
    for (unsigned int i = 0; i < N; i++)
    {
        unwrapped[i].x = pos[i].x + (double)image[i].x * Lx;
        unwrapped[i].y = pos[i].y + (double)image[i].y * Ly;
        unwrapped[i].z = pos[i].z + (double)image[i].z * Lz;
    }
}

/* computeDipoleMoment, src/CavityForceCompute.cc:113-129: strict index order, skips photon_idx only. */
REF_API void cavref_dipole(const ref_vec3* unwrapped, const double* charge, unsigned int N, int photon_idx,
                           double out[3])
{
    double dx = 0.0, dy = 0.0, dz = 0.0;
    for (unsigned int i = 0; i < N; i++)
    {
        if ((int)i != photon_idx)
        {
            dx += charge[i] * unwrapped[i].x;
            dy += charge[i] * unwrapped[i].y;
            dz += charge[i] * unwrapped[i].z;
        }
    }
    out[0] = dx;
    out[1] = dy;
    out[2] = dz;
}

static double ref_dot3(double ax, double ay, double az, double bx, double by, double bz)
{
    /* HOOMD dot(vec3,vec3) = a.x*b.x + a.y*b.y + a.z*b.z */
    return ax * bx + ay * by + az * bz;
}

/* computeForces, src/CavityForceCompute.cc:134-208.
 * energies = {harmonic, coupling, dipole_self}; dipole_out (3) and photon_idx_out may be NULL.
 * Returns 0, or -1 if the temporary array cannot be allocated.  L_typeid < 0 stands for "no type
 * named L" (the reference's getTypeByName would throw there); it takes the no-photon path. */
REF_API int cavref_compute_forces(unsigned int N, const ref_scalar4* pos, const double* charge, const ref_int3* image,
                                  double Lx, double Ly, double Lz, int L_typeid, const ref_params* params,
                                  ref_scalar4* force, double energies[3], double* dipole_out, int* photon_idx_out)
{
    const double g = params->couplstr;
    const double K = params->K;

    /* :145 */
    memset(force, 0, sizeof(ref_scalar4) * (size_t)N);

    /* :148-156 */
    int photon_idx = cavref_find_photon(pos, N, L_typeid);
    if (photon_idx_out)
        *photon_idx_out = photon_idx;
    if (photon_idx == -1)
    {
        energies[0] = 0.0;
        energies[1] = 0.0;
        energies[2] = 0.0;
        if (dipole_out)
            dipole_out[0] = dipole_out[1] = dipole_out[2] = 0.0;
        return 0;
    }

    /* :162-163 -- the reference constructs a std::vector<vec3<Scalar>> of N on every call */
    ref_vec3* unwrapped = (ref_vec3*)malloc(sizeof(ref_vec3) * (size_t)(N ? N : 1));
    if (!unwrapped)
        return -1;
    cavref_unwrap(unwrapped, pos, image, Lx, Ly, Lz, N);

    /* :166 */
    double dipole[3];
    cavref_dipole(unwrapped, charge, N, photon_idx, dipole);
    if (dipole_out)
    {
        dipole_out[0] = dipole[0];
        dipole_out[1] = dipole[1];
        dipole_out[2] = dipole[2];
    }

    /* :169-171 */
    const double qx = unwrapped[photon_idx].x, qy = unwrapped[photon_idx].y, qz = unwrapped[photon_idx].z;
    const double dxy_x = dipole[0], dxy_y = dipole[1], dxy_z = 0.0;
    const double qxy_x = qx, qxy_y = qy, qxy_z = 0.0;

    /* :174-176 */
    energies[0] = 0.5 * K * ref_dot3(qx, qy, qz, qx, qy, qz);
    energies[1] = g * ref_dot3(dxy_x, dxy_y, dxy_z, qxy_x, qxy_y, qxy_z);
    energies[2] = 0.5 * (g * g / K) * ref_dot3(dxy_x, dxy_y, dxy_z, dxy_x, dxy_y, dxy_z);

    /* :180 */
    force[photon_idx].w = 0.0;

    /* :183 */
    const double gK = g / K;
    const double Dq_x = qxy_x + gK * dxy_x;
    const double Dq_y = qxy_y + gK * dxy_y;

    /* :188-200 -- `-g * charge * Dq` associates as ((-g) * charge) * Dq */
    for (unsigned int i = 0; i < N; i++)
    {
        int type = ref_scalar_as_int(pos[i].w);
        if (type != L_typeid)
        {
            const double c = charge[i];
            const double s = -g * c;
            force[i].x = s * Dq_x;
            force[i].y = s * Dq_y;
            force[i].z = 0.0;
        }
    }

    /* :203-207 */
    force[photon_idx].x = -K * qx - g * dxy_x;
    force[photon_idx].y = -K * qy - g * dxy_y;
    force[photon_idx].z = -K * qz - g * dxy_z;

    free(unwrapped);
    return 0;
}

/* ---- exactly-rounded companion (accuracy yardstick, not the reference's algorithm) ------------ */
/* d = sum of the SAME fp64-rounded terms charge_i * (pos_i + image_i * L) the reference adds, but
 * accumulated in double-double (error-free TwoSum), so the result is the correctly rounded sum to
 * well below 1 ulp.  Used to show which of {reference order, GPU tree} is closer to the truth. */
static void dd_add(double* hi, double* lo, double t)
{
    double s = *hi + t;
    double bb = s - *hi;
    double e = (*hi - (s - bb)) + (t - bb);
    *hi = s;
    *lo += e;
}

REF_API void cavref_dipole_exact(unsigned int N, const ref_scalar4* pos, const double* charge, const ref_int3* image,
                                 double Lx, double Ly, double Lz, int photon_idx, double out_hi[3], double out_lo[3])
{
    double hx = 0, lx = 0, hy = 0, ly = 0, hz = 0, lz = 0;
    for (unsigned int i = 0; i < N; i++)
    {
        if ((int)i == photon_idx)
            continue;
        const double rx = pos[i].x + (double)image[i].x * Lx;
        const double ry = pos[i].y + (double)image[i].y * Ly;
        const double rz = pos[i].z + (double)image[i].z * Lz;
        dd_add(&hx, &lx, charge[i] * rx);
        dd_add(&hy, &ly, charge[i] * ry);
        dd_add(&hz, &lz, charge[i] * rz);
    }
    /* renormalise */
    double s;
    s = hx + lx; lx = lx - (s - hx); hx = s;
    s = hy + ly; ly = ly - (s - hy); hy = s;
    s = hz + lz; lz = lz - (s - hz); hz = s;
    out_hi[0] = hx; out_hi[1] = hy; out_hi[2] = hz;
    out_lo[0] = lx; out_lo[1] = ly; out_lo[2] = lz;
}

/* ---- timing leg for bench.py's cpu_baseline ("port", one thread) ------------------------------ */
/* Runs cavref_compute_forces `iters` times back to back and returns the elapsed seconds. */
REF_API double cavref_time_evaluations(unsigned int N, const ref_scalar4* pos, const double* charge,
                                       const ref_int3* image, double Lx, double Ly, double Lz, int L_typeid,
                                       const ref_params* params, ref_scalar4* force, int iters)
{
    double energies[3];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 0; it < iters; it++)
    {
        cavref_compute_forces(N, pos, charge, image, Lx, Ly, Lz, L_typeid, params, force, energies, NULL, NULL);
        /* keep the optimiser from hoisting anything across iterations */
        __asm__ volatile("" : : "r"(force), "r"(energies) : "memory");
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

REF_API int cavref_layout_sizes(int out[4])
{
    out[0] = (int)sizeof(ref_scalar4);
    out[1] = (int)sizeof(ref_int3);
    out[2] = (int)sizeof(ref_params);
    out[3] = (int)sizeof(ref_vec3);
    return 0;
}
