/* cavity_omp.c -- ALL-CORES COURTESY BASELINE.  NOT the reference's algorithm and NOT used as an oracle.
 *
 * The reference's CPU path is single-threaded (no OpenMP/TBB in src/CavityForceCompute.cc; deployed one core per
 * replica, submit.sh:6); oracle/cavity_ref.c restates it faithfully and is what cpu_baseline reports.  This file only
 * answers "what could the same host do with every core": the same passes with OpenMP, fused unwrap + dipole (no
 * temporary array), a parallel reduction (so the summation ORDER differs from the reference) and a parallel map.
 * bench.py reports it under cpu_baseline.all_cores_courtesy, labelled as such.
 */
#define _POSIX_C_SOURCE 199309L
#include <limits.h>
#include <omp.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

typedef struct { double x, y, z, w; } s4;
typedef struct { int32_t x, y, z; } i3;
typedef struct { double omegac, couplstr, K, phmass; } prm_t;
#define API __attribute__((visibility("default")))

static int tag_of(double w) { int32_t lo; memcpy(&lo, &w, 4); return (int)lo; }

API int cavomp_threads(void) { return omp_get_max_threads(); }
API void cavomp_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

API void cavomp_compute(unsigned N, const s4* pos, const double* charge, const i3* image, double Lx, double Ly, double Lz,
                        int L_typeid, const prm_t* p, s4* force, double energies[3])
{
    long photon = LONG_MAX; /* first index of type L = minimum over all matches */
#pragma omp parallel for reduction(min : photon) schedule(static)
    for (long i = 0; i < (long)N; i++)
        if (tag_of(pos[i].w) == L_typeid && i < photon)
            photon = i;
    if (photon == LONG_MAX)
        photon = -1;
    if (photon < 0)
    {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)N; i++) force[i] = (s4) {0, 0, 0, 0};
        energies[0] = energies[1] = energies[2] = 0.0;
        return;
    }
    double dx = 0, dy = 0;
#pragma omp parallel for reduction(+ : dx, dy) schedule(static)
    for (long i = 0; i < (long)N; i++)
        if (i != photon)
        {
            dx += charge[i] * (pos[i].x + (double)image[i].x * Lx);
            dy += charge[i] * (pos[i].y + (double)image[i].y * Ly);
        }
    const double qx = pos[photon].x + (double)image[photon].x * Lx, qy = pos[photon].y + (double)image[photon].y * Ly,
                 qz = pos[photon].z + (double)image[photon].z * Lz;
    const double g = p->couplstr, K = p->K;
    energies[0] = 0.5 * K * (qx * qx + qy * qy + qz * qz);
    energies[1] = g * (dx * qx + dy * qy);
    energies[2] = 0.5 * (g * g / K) * (dx * dx + dy * dy);
    const double Dqx = qx + g / K * dx, Dqy = qy + g / K * dy;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; i++)
    {
        s4 f = {0, 0, 0, 0};
        if (tag_of(pos[i].w) != L_typeid)
        {
            const double s = -g * charge[i];
            f.x = s * Dqx;
            f.y = s * Dqy;
        }
        force[i] = f;
    }
    force[photon] = (s4) {-K * qx - g * dx, -K * qy - g * dy, -K * qz, 0.0};
}

API double cavomp_time(unsigned N, const s4* pos, const double* charge, const i3* image, double Lx, double Ly, double Lz,
                       int L_typeid, const prm_t* p, s4* force, int iters)
{
    double e[3];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 0; it < iters; it++)
    {
        cavomp_compute(N, pos, charge, image, Lx, Ly, Lz, L_typeid, p, force, e);
        __asm__ volatile("" : : "r"(force), "r"(e) : "memory");
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
