/* bussi_ref.c -- CPU ORACLE (test infrastructure, not product) for the Bussi reservoir thermostat step
 * of muhammadhasyim/cav-hoomd: a plain-C restatement of
 *
 *   BussiReservoirThermostat::compute_rescale_factor      src/BussiReservoirThermostat.h:177-225
 *   BussiReservoirThermostat::getRescalingFactorsOne      src/BussiReservoirThermostat.h:43-98
 *   the translational kinetic energy it consumes          [HOOMD upstream: ComputeThermo, sum over the group's members in
 *                                                          index order of mass * (vx*vx + vy*vy + vz*vz), times 0.5]
 *
 * with the random variates INJECTED: the reference draws them from HOOMD's RandomGenerator / NormalDistribution /
 * GammaDistribution (src/BussiReservoirThermostat.h:66, 192-199), which are not vendored and cannot be reproduced here, so
 * variate GENERATION stays unpinned; everything after the draw is deterministic and is what this file pins.
 *
 * Pinning status: PARITY UNPINNED by the reference's own tests (src/pytest/test_bussi_reservoir.py checks only that the
 * counters start at zero, move, and reset).  Pins created here: closed-form known answers in tests/test_bussi_reservoir.py.
 *
 * Built by oracle/Makefile with -ffp-contract=off and no -march (one rounding per operation, as the reference's
 * flag-less build).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define ORACLE_API __attribute__((visibility("default")))

/* src/BussiReservoirThermostat.h:177-225.  `r_normal` is normal(rng) (:193), `gamma_variate` is gamma(rng) of
 * GammaDistribution((dof - 1) / 2, 1) (:195-199; ignored unless dof > 1, exactly as the reference draws it only then). */
ORACLE_API double bussiref_rescale_factor(double K, double degrees_of_freedom, double deltaT, double set_T, double tau,
                                          double r_normal, double gamma_variate)
{
    if (degrees_of_freedom == 0)
        return 1.0;

    double time_decay_factor = 0.0;
    if (tau != 0.0)
    {
        time_decay_factor = exp(-deltaT / tau);
    }

    double r_normal_one = r_normal;

    double r_gamma = 0.0;
    if (degrees_of_freedom > 1.0)
    {
        r_gamma = 2.0 * gamma_variate;
    }

    double v = set_T / 2.0 / K;
    double term1 = v * (1.0 - time_decay_factor) * (r_gamma + r_normal_one * r_normal_one);
    double term2 = 2.0 * r_normal_one * sqrt(v * (1.0 - time_decay_factor) * time_decay_factor);

    double alpha_squared = time_decay_factor + term1 + term2;
    double alpha_magnitude = sqrt(alpha_squared);

    double c = time_decay_factor;
    double K_bar = set_T * degrees_of_freedom / 2.0;
    double sign_term = r_normal_one + sqrt(c * degrees_of_freedom * K / ((1.0 - c) * K_bar));

    if (sign_term >= 0.0)
    {
        return alpha_magnitude;
    }
    else
    {
        return -alpha_magnitude;
    }
}

/* state = {reservoir_translational, reservoir_rotational, instantaneous_translational, instantaneous_rotational}
 * (src/BussiReservoirThermostat.h:160-165).  variates = {normal_t, gamma_t, normal_r, gamma_r} in the order the reference
 * consumes them (:73-82).  Returns 0, or -1 where the reference throws "requires non-zero initial momenta" (:57-61). */
ORACLE_API int bussiref_step(double state[4], double K_trans, double dof_trans, double K_rot, double dof_rot, double deltaT,
                             double set_T, double tau, const double variates[4], double factors[2])
{
    if (deltaT == 0.0)
    {
        factors[0] = 1.0;
        factors[1] = 1.0;
        return 0;
    }
    if ((dof_trans != 0 && K_trans == 0) || (dof_rot != 0 && K_rot == 0))
        return -1;
    const double translational_factor = bussiref_rescale_factor(K_trans, dof_trans, deltaT, set_T, tau, variates[0], variates[1]);
    const double rotational_factor = bussiref_rescale_factor(K_rot, dof_rot, deltaT, set_T, tau, variates[2], variates[3]);
    /* :86-95 */
    double delta_trans = K_trans * (1.0 - translational_factor * translational_factor);
    double delta_rot = K_rot * (1.0 - rotational_factor * rotational_factor);
    state[0] += delta_trans;
    state[1] += delta_rot;
    state[2] = delta_trans;
    state[3] = delta_rot;
    factors[0] = translational_factor;
    factors[1] = rotational_factor;
    return 0;
}

/* Translational kinetic energy of a group [HOOMD upstream ComputeThermo]: members == NULL means all n particles. */
ORACLE_API double bussiref_kinetic_energy(const double* vel4 /* (N,4): vx, vy, vz, mass */, const uint32_t* members, size_t n)
{
    double ke = 0.0;
    for (size_t k = 0; k < n; ++k)
    {
        const size_t j = members ? members[k] : k;
        const double* v = vel4 + 4 * j;
        ke += v[3] * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    }
    return 0.5 * ke;
}

/* The same sum, exactly rounded (double-double accumulation of the same fp64 terms): returns hi; *lo gets the low word. */
ORACLE_API double bussiref_kinetic_energy_exact(const double* vel4, const uint32_t* members, size_t n, double* lo_out)
{
    double hi = 0.0, lo = 0.0;
    for (size_t k = 0; k < n; ++k)
    {
        const size_t j = members ? members[k] : k;
        const double* v = vel4 + 4 * j;
        const double t = v[3] * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double s = hi + t;
        const double bb = s - hi;
        const double e = (hi - (s - bb)) + (t - bb);
        hi = s;
        lo += e;
        const double s2 = hi + lo; /* renormalise */
        lo = lo - (s2 - hi);
        hi = s2;
    }
    if (lo_out)
        *lo_out = 0.5 * lo;
    return 0.5 * hi;
}
