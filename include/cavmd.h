/* cavmd.h -- C ABI of the MI355X (gfx950) cavity-MD force engine.
 *
 * This is the drop-in boundary for ONE hot path of muhammadhasyim/cav-hoomd: the per-step
 * cavity force evaluation.  Every entry point names the reference interface it replaces
 * (paths relative to the reference checkout):
 *
 *   reference                                              this library
 *   ----------------------------------------------------   ---------------------------------
 *   struct cavity_force_params  src/CavityForceCompute.h:28-54      cavmd_params / cavmd_make_params
 *   CavityForceComputeGPU ctor (scratch GPUArrays)
 *                               src/CavityForceComputeGPU.cc:31-93  cavmd_create / cavmd_destroy
 *   kernel::gpu_compute_cavity_force(...)
 *                               src/CavityForceComputeGPU.cuh:28-40 cavmd_compute_hoomd
 *   CavityForcePython.set_forces (snapshot-layout arrays)
 *                               src/cavitymd/cavity_force_python.py:65-145
 *                                                                   cavmd_compute_soa
 *   getHarmonicEnergy/getCouplingEnergy/getDipoleSelfEnergy
 *                               src/CavityForceCompute.cc:58-71     cavmd_energies
 *   BussiReservoirThermostat::getRescalingFactorsOne / compute_rescale_factor
 *                               src/BussiReservoirThermostat.h:43-98, 177-225
 *                                                                   cavmd_bussi_step / cavmd_bussi_rescale_factor,
 *                                                                   cavmd_kinetic_energy, cavmd_scale_velocities,
 *                                                                   cavmd_bussi_step_device (the step without a host round trip)
 *   CavityForceCompute::computeForces (the CPU semantics both follow)
 *                               src/CavityForceCompute.cc:134-208   (semantic contract, see below)
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch/HOOMD types cross this boundary.
 *   - every function returns an int status: CAVMD_OK (0), a CAVMD_ERR_* code (< 0), or a positive
 *     hipError_t forwarded from the HIP runtime.  Nothing throws across the ABI.
 *   - all particle buffers are DEVICE pointers owned by the caller (HOOMD's GlobalArrays, a torch
 *     tensor, ...).  The library owns only its workspace.  No allocation, no host synchronisation
 *     and no host<->device copy happens inside cavmd_compute_*; they only enqueue kernels on
 *     `stream` (NULL = the null stream, which is what HOOMD-blue 4.x uses), so a caller may
 *     capture them into a hipGraph.  One workspace serves one stream at a time (one force object, as in
 *     the reference).  A workspace that has been captured reads its results (cavmd_result_read,
 *     cavmd_energies) behind a hipDeviceSynchronize instead of the host-visible flag: a replayed kernel
 *     carries the sequence number of its capture, so the flag cannot tell replays apart.
 *     A graph whose replay ended in CAVMD_ERR_SYNC_TIMEOUT must be captured again before it is replayed (the
 *     library's own recovery -- two launches, wiped hand-off slabs -- does not reach into a captured graph): until then
 *     every replay of it fails at once and as a whole (NaN forces, nothing published; the poison word of
 *     cavmd_persistent_kernel.hpp), it never yields a result.  A graph captured while the single-launch evaluation was in
 *     use keeps that kernel also after a starved-but-REPAIRED replay (valid results): the back-off that moves a workspace
 *     to two launches acts on enqueues, not on replays, so a graph replayed on a GPU that stays oversubscribed can pay the
 *     bounded wait (~0.3 s) on every replay; capture with the tunable "persistent" = 0, or CAVMD_PERSISTENT=0, there.
 *   - Scalar = double (HOOMD's default HOOMD_LONGREAL_SIZE=64 build).
 *
 * Semantic contract (what "the same result as the reference" means here; file:line = reference)
 *   photon      = first index i whose type tag equals L_typeid (src/CavityForceCompute.cc:73-89);
 *                 the tag is the low 32 bits of the bit pattern of pos[i].w (HOOMD __scalar_as_int).
 *   r_i         = pos_i + image_i * (Lx,Ly,Lz), orthorhombic lengths only (:91-111).
 *   d           = sum_{i != photon} charge_i * r_i  (:113-129).  The reference sums left to right;
 *                 this library uses a fixed-shape compensated tree, so d agrees to ~1 ulp with the
 *                 exactly rounded sum, and is bit-reproducible from run to run.
 *   E_h = 1/2 K (q.q) with the 3-D q;  E_c = g (d_xy . q_xy);  E_d = 1/2 (g^2/K)(d_xy . d_xy) (:174-176)
 *   F_i = (-g c_i Dq_x, -g c_i Dq_y, 0, 0) for every particle whose type is not L,
 *         Dq = q_xy + (g/K) d_xy (:183-200);  L-typed particles other than the photon get 0.
 *   F_L = (-K q_x - g d_x, -K q_y - g d_y, -K q_z, 0) (:203-207).
 *   no particle of type L -> all forces 0, all energies 0, not an error (:148-156).
 *   virial / torque are never written (the reference leaves them zero).
 */
#ifndef CAVMD_H_
#define CAVMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define CAVMD_API
#else
#define CAVMD_API __attribute__((visibility("default")))
#endif

#define CAVMD_VERSION_MAJOR 0
#define CAVMD_VERSION_MINOR 2

/* ---- status codes ------------------------------------------------------------------------- */
#define CAVMD_OK 0
#define CAVMD_ERR_INVALID_VALUE (-1) /* null pointer / bad size; mirrors hipErrorInvalidValue at
                                        src/CavityForceComputeGPU.cu:522-528 */
#define CAVMD_ERR_NO_DEVICE (-2)     /* no HIP device visible: the product path never falls back to a CPU */
#define CAVMD_ERR_CAPACITY (-3)      /* N exceeds the capacity the workspace was created for */
#define CAVMD_ERR_BAD_PARAMS (-4)    /* K == 0 or non-finite parameters */
#define CAVMD_ERR_NOT_COMPUTED (-5)  /* results requested before any cavmd_compute_* call */
#define CAVMD_ERR_SYNC_TIMEOUT (-6)  /* a starved single-launch evaluation that could not be completed (see the "persistent"
                                        tunable): that evaluation's forces are NaN.  Returned by cavmd_result_read /
                                        cavmd_energies or by the next cavmd_compute_* call, whichever comes first. */

/* ---- layouts (bit-compatible with HOOMD-blue's Scalar4 / int3 in a double-precision build) -- */
typedef struct cavmd_double4
{
    double x, y, z, w;
} cavmd_double4; /* 32 B; pos.w carries the type id in its low 32 bits, force.w is the per-particle PE */

typedef struct cavmd_int3
{
    int32_t x, y, z;
} cavmd_int3; /* 12 B, packed */

/* One POD parameter block shared by host, device and this ABI (the reference defines it twice with
 * different alignment: src/CavityForceCompute.h:28-54 and src/CavityForceComputeGPU.cu:24-30). */
typedef struct cavmd_params
{
    double omegac;   /* cavity frequency, atomic units */
    double couplstr; /* coupling strength g, atomic units */
    double K;        /* spring constant = phmass * omegac^2 */
    double phmass;   /* photon mass */
} cavmd_params;

/* Everything one evaluation produces besides the per-particle forces (192 B, lives in the workspace
 * on the device; cavmd_result_read copies it out). */
typedef struct cavmd_result
{
    double dipole[3];        /* molecular dipole d (photon excluded), all three components */
    double q[3];             /* unwrapped photon position */
    double Dq[2];            /* q_xy + (g/K) d_xy */
    double energy[3];        /* harmonic, coupling, dipole-self */
    double photon_force[3];  /* F_L */
    double dipole_lo[3];     /* low words of the compensated dipole sum (diagnostic) */
    int32_t photon_idx;      /* index of the photon, -1 if there is none */
    int32_t n_photon_typed;  /* how many particles carry type L (the driver enforces exactly 1) */
    uint32_t n_particles;    /* N of the evaluation this result belongs to */
    uint32_t n_partials;     /* partial sums that fed the final reduction (diagnostic) */
    uint64_t sequence;       /* evaluation counter of the workspace */
    double total_dipole[3];  /* sum over ALL particles, L-typed included: what the reference's observable
                                compute_total_dipole_moment (src/cavitymd/analysis.py:18-31) returns */
    double reserved;
} cavmd_result;

typedef struct cavmd_workspace cavmd_workspace; /* opaque */

/* ---- parameters ----------------------------------------------------------------------------- */
/* K = phmass * omegac^2, as cavity_force_params(omegac, couplstr, phmass) at src/CavityForceCompute.h:38-42 */
CAVMD_API cavmd_params cavmd_make_params(double omegac, double couplstr, double phmass);

/* ---- life cycle ----------------------------------------------------------------------------- */
/* Binds a workspace to HIP device `device` (-1 = the current device) sized for up to max_N particles.
 * Replaces the four scratch GPUArrays of CavityForceComputeGPU (src/CavityForceComputeGPU.cc:43-54);
 * unlike them the partial-sum buffer is sized from the launch geometry, never from a constant. */
CAVMD_API int cavmd_create(int device, size_t max_N, cavmd_workspace** out_ws);
CAVMD_API int cavmd_destroy(cavmd_workspace* ws);

/* ---- the hot path --------------------------------------------------------------------------- */
/* HOOMD-native AoS layouts.  Replaces kernel::gpu_compute_cavity_force
 * (src/CavityForceComputeGPU.cuh:28-40): d_force/d_pos/d_charge/d_image keep their meaning, `box`
 * becomes the three orthorhombic lengths box.getL() returns, the four scratch pointers become `ws`.
 * N == 0 is a success that touches nothing (src/CavityForceComputeGPU.cu:530-532).
 * Writes all N entries of d_force (x, y, z and w): no separate memset pass is needed or performed. */
CAVMD_API int cavmd_compute_hoomd(cavmd_workspace* ws,
                                  void* stream,
                                  size_t N,
                                  const cavmd_double4* d_pos,
                                  const double* d_charge,
                                  const cavmd_int3* d_image,
                                  double Lx,
                                  double Ly,
                                  double Lz,
                                  int L_typeid,
                                  const cavmd_params* params,
                                  cavmd_double4* d_force);

/* Snapshot ("local snapshot") layouts used by the hoomd.md.force.Custom surface
 * (src/cavitymd/cavity_force_python.py:72-145): position (N,3) f64, typeid (N,) i32, image (N,3) i32,
 * charge (N,) f64 -> force (N,3) f64 and, if not NULL, potential_energy (N,) f64 (set to 0, as the
 * reference does at :126-131).  Strides are in BYTES between consecutive particles, so the same
 * entry point serves packed arrays (24, 4, 12, 8, 24, 8) and HOOMD's strided views of its Scalar4
 * buffers (32, 32, 12, 8, 32, 32).  Same semantics as cavmd_compute_hoomd (the C++ ones: photon =
 * first particle of type L_typeid, photon excluded from d, L-typed particles get no molecular force). */
CAVMD_API int cavmd_compute_soa(cavmd_workspace* ws,
                                void* stream,
                                size_t N,
                                const double* d_position,
                                size_t position_stride,
                                const int32_t* d_typeid,
                                size_t typeid_stride,
                                const int32_t* d_image,
                                size_t image_stride,
                                const double* d_charge,
                                size_t charge_stride,
                                double Lx,
                                double Ly,
                                double Lz,
                                int L_typeid,
                                const cavmd_params* params,
                                double* d_force,
                                size_t force_stride,
                                double* d_potential_energy,
                                size_t potential_energy_stride);

/* ---- results -------------------------------------------------------------------------------- */
/* The three energy getters of src/CavityForceCompute.cc:58-71 in one call:
 * out[0] = harmonic, out[1] = coupling, out[2] = dipole self-energy.  No copy is enqueued: the kernel that
 * computes the scalars also stores the result block and a sequence flag (system-scope release) into mapped
 * pinned host memory; this call spins until it sees the flag of the last evaluation (or the stream idle).
 * Polled after every evaluation (the reference's EnergyTracker at period 1) it adds ~5 us, not a stream sync. */
CAVMD_API int cavmd_energies(cavmd_workspace* ws, double out[3]);
/* Whole result block (dipole, photon position and force, photon index, ...). */
CAVMD_API int cavmd_result_read(cavmd_workspace* ws, cavmd_result* out);
/* Device address of the result block, for consumers that stay on the GPU (trackers, graphs). */
CAVMD_API int cavmd_result_device_ptr(cavmd_workspace* ws, const cavmd_result** out);

/* ---- observables next to the force path ------------------------------------------- */
/* Wavevectors for the density field: n_k rows of (kx, ky, kz) in HOST memory; copied into the workspace once.
 * The reference builds them as kmag * generate_fibonacci_sphere(50) (src/cavitymd/analysis.py:296-306). */
CAVMD_API int cavmd_set_wavevectors(cavmd_workspace* ws, size_t n_k, const double* h_wavevectors);
/* rho(k) = sum_j exp(i k . r_j) over the WRAPPED positions of all N particles, for every stored wavevector;
 * replaces compute_density_field (src/cavitymd/analysis.py:34-47), which pulls a CPU snapshot and loops in numpy.
 * d_position + i * position_stride points at particle i's x, y, z (stride 32 for HOOMD's Scalar4 pos, 24 for a
 * packed (N,3) array).  Enqueues two kernels on `stream`; no host synchronisation. */
CAVMD_API int cavmd_density_field(cavmd_workspace* ws, void* stream, size_t N, const double* d_position,
                                  size_t position_stride);
/* Copies the last density field out: h_out[2k] = Re rho(k_k), h_out[2k+1] = Im rho(k_k).  Synchronises the stream. */
CAVMD_API int cavmd_density_field_read(cavmd_workspace* ws, double* h_out);
/* Cavity-mode properties of CavityModeTracker (src/cavitymd/analysis.py:1324-1368) for the photon the last force
 * evaluation found: out = {KE = 1/2 m v.v, harmonic PE, KE + PE, T = (2/3) KE / k_B}; all zero without a photon.
 * d_vel is HOOMD's Scalar4 velocity array (mass in .w); kB in Hartree/K (the reference uses 3.167e-6).
 * Enqueues one tiny kernel on `stream` and spins on its host-visible flag (no copy, no stream synchronisation). */
CAVMD_API int cavmd_cavity_mode(cavmd_workspace* ws, void* stream, const cavmd_double4* d_vel, double kB, double out[4]);

/* S = sum_i |F_i| / m_i over a Scalar4 net-force array and HOOMD's Scalar4 velocity array (mass in .w): the reduction
 * AdaptiveTimestepUpdater performs on the host every step to set dt = sqrt(tol / S) (src/cavitymd/simulation.py:66-92).
 * Enqueues ONE kernel on `stream` (the block that finishes last folds the partials in a fixed order and hands the number to
 * the host through mapped pinned memory); this call spins on its flag: no copy, no stream synchronisation (the caller needs
 * the number to set dt). */
CAVMD_API int cavmd_force_mass_sum(cavmd_workspace* ws, void* stream, size_t N, const cavmd_double4* d_net_force,
                                   const cavmd_double4* d_vel, double* out);

/* ---- Bussi reservoir thermostat step ------------------------------------------------------- */
/* Translational kinetic energy 1/2 sum_j m_j v_j.v_j of a particle group: what BussiReservoirThermostat reads from
 * ComputeThermo (src/BussiReservoirThermostat.h:49-54).  d_vel is HOOMD's Scalar4 velocity array (mass in .w);
 * d_members is a DEVICE array of n_members particle indices (HOOMD's ParticleGroup index list) or NULL for the particles
 * 0 .. n_members-1.  Enqueues ONE kernel on `stream` and spins on its host-visible flag (no copy, no stream
 * synchronisation). */
CAVMD_API int cavmd_kinetic_energy(cavmd_workspace* ws, void* stream, const cavmd_double4* d_vel, const uint32_t* d_members,
                                   size_t n_members, double* out);
/* v_j.xyz *= alpha for the members of the group: what HOOMD's integration method does with the factor the thermostat
 * returns.  Enqueues one kernel; no host synchronisation. */
CAVMD_API int cavmd_scale_velocities(cavmd_workspace* ws, void* stream, cavmd_double4* d_vel, const uint32_t* d_members,
                                     size_t n_members, double alpha);

/* Reservoir accounting of BussiReservoirThermostat (src/BussiReservoirThermostat.h:160-165). */
typedef struct cavmd_bussi_reservoir
{
    double reservoir_translational;     /* cumulative energy handed to the bath by the translational degrees of freedom */
    double reservoir_rotational;
    double instantaneous_translational; /* the same for the last step only */
    double instantaneous_rotational;
} cavmd_bussi_reservoir;

/* The stochastic velocity-rescaling factor alpha of compute_rescale_factor (src/BussiReservoirThermostat.h:177-225),
 * sign rule included, as a pure function of (K, Nf, dt, kT, tau) and the two random variates the reference draws:
 * normal_variate ~ N(0,1), gamma_variate ~ Gamma((Nf - 1) / 2, 1) (ignored unless Nf > 1; drawn only then).
 * Host arithmetic, no GPU needed.  Variate GENERATION is the caller's (HOOMD's RandomGenerator in the reference). */
CAVMD_API int cavmd_bussi_rescale_factor(double K, double degrees_of_freedom, double deltaT, double set_T, double tau,
                                         double normal_variate, double gamma_variate, double* alpha);
/* One thermostat step, as getRescalingFactorsOne (src/BussiReservoirThermostat.h:43-98): factors[0] / [1] = translational /
 * rotational alpha, reservoir counters updated with KE (1 - alpha^2).  variates = {normal_t, gamma_t, normal_r, gamma_r},
 * in the order the reference consumes them.  deltaT == 0 -> {1, 1}, counters untouched;  a non-zero number of degrees
 * of freedom with zero kinetic energy -> CAVMD_ERR_BAD_PARAMS (the reference throws "requires non-zero initial momenta"). */
CAVMD_API int cavmd_bussi_step(cavmd_bussi_reservoir* state, double K_translational, double dof_translational,
                               double K_rotational, double dof_rotational, double deltaT, double set_T, double tau,
                               const double variates[4], double factors[2]);

/* The same translational step ENTIRELY ON THE DEVICE and asynchronous (round 3): kinetic energy of the group -> alpha ->
 * reservoir counters -> velocities *= alpha, as two kernels enqueued on `stream` with no host round trip in between (the
 * first leaves one partial sum per workgroup; every workgroup of the second folds them in a fixed order, evaluates the rule
 * itself and rescales its share; no atomics, no last-workgroup tail).  What
 * getRescalingFactorsOne + the integration method's rescale do per step (src/BussiReservoirThermostat.h:43-98, 177-225) for the
 * translational degrees of freedom; rotational ones stay on the host path above (cavmd_bussi_step).  The rule runs the same
 * source function as cavmd_bussi_rescale_factor (c = exp(-dt / tau) is taken on the host): same bits for the same kinetic energy.
 * deltaT == 0 enqueues nothing (factors 1, counters untouched, :45-48).  Degrees of freedom with zero kinetic energy: the step
 * is refused on the device (alpha = 1, nothing rescaled) and the NEXT cavmd_bussi_device_read returns CAVMD_ERR_BAD_PARAMS once. */
CAVMD_API int cavmd_bussi_step_device(cavmd_workspace* ws, void* stream, cavmd_double4* d_vel, const uint32_t* d_members,
                                      size_t n_members, double dof_translational, double deltaT, double set_T, double tau,
                                      double normal_variate, double gamma_variate);
typedef struct cavmd_bussi_device_state
{
    double reservoir_translational;     /* cumulative, as cavmd_bussi_reservoir */
    double instantaneous_translational; /* last step */
    double last_alpha;                  /* the factor the last step applied */
    double last_kinetic_energy;         /* the kinetic energy it saw (before rescaling) */
    uint64_t steps;                     /* steps applied since creation / reset */
    uint64_t refused;                   /* steps refused for zero kinetic energy */
} cavmd_bussi_device_state;
/* State after the last enqueued cavmd_bussi_step_device: spins on the flag that step publishes into mapped host memory (no
 * copy, no stream synchronisation; it watches the stream that step was enqueued on); before any step: zeros.
 * CAVMD_ERR_BAD_PARAMS (once) if a step was refused since the last call.  Steps captured into a hipGraph replay correctly on
 * the device (alpha and the counters live there); the flag, however, carries the sequence number frozen at capture, so after
 * replays synchronise the stream (or device) before reading. */
CAVMD_API int cavmd_bussi_device_read(cavmd_workspace* ws, cavmd_bussi_device_state* out);
/* reset_reservoir_energy() of the reference's Python class: zero the counters (ordered on `stream`). */
CAVMD_API int cavmd_bussi_device_reset(cavmd_workspace* ws, void* stream);

/* ---- measurement hooks (bench.py's roofline leg) ---------------------------------------------- */
/* When enabled, every cavmd_compute_* brackets each of its kernels with hipEvents on `stream`. */
CAVMD_API int cavmd_profile_enable(cavmd_workspace* ws, int on);
/* Synchronises and returns the accumulated device time per kernel since the last reset:
 * ms[0] = dipole partial-sum kernel, ms[1] = finalize kernel (0 in the default two-launch mode, where the
 * finalize is the prologue of the force map), ms[2] = force-map kernel;
 * *launches = evaluations accumulated.  Resets the accumulators. */
CAVMD_API int cavmd_profile_read(cavmd_workspace* ws, double ms[3], uint64_t* launches);

/* Per-evaluation samples behind cavmd_profile_read: copies up to `cap` triples {reduce, finalize, map} in
 * milliseconds (oldest first) into out[3 * cap] and stores the count in *n.  Call BEFORE cavmd_profile_read, which
 * clears them.  At most the last 4096 evaluations are kept. */
CAVMD_API int cavmd_profile_samples(cavmd_workspace* ws, double* out, size_t cap, size_t* n);

/* ---- tuning / introspection ------------------------------------------------------------------- */
/* Launch knobs (for A/B measurements; the defaults are the measured best on MI355X).  name is one of
 *   "reduce_blocks_per_cu" 1..16   grid of the reduction = min(tiles, CUs * value); also the number of partials
 *   "map_blocks_per_cu"    1..16   grid of the force map
 *   "map_nt_store"         -1..2   force stores: -1 auto by N, 0 plain, 1 non-temporal, 2 write-through (sc1; measured slower
 *                                  at every size, profiles/r02/ab_store_policy.txt)
 *   "reduce_nt_load"       -1..2   -1 auto by N, 0 plain, 1 pos+image non-temporal, 2 all non-temporal
 *   "fused_finalize"       0/1     1: two launches per evaluation (finalize folded into the force map), 0: three
 *   "map_reverse"          -1..1   -1 auto by N, 1: the force map walks its tiles last-to-first, 0: first-to-last
 *   "small_system_max_n"   0..2^20 at or below this N (default 1024) one single-block launch does the whole evaluation (0 = never)
 *   "reduce_unroll"        -1,1,2  particles per lane and tile of the reduction (-1 auto by N)
 *   "persistent"           -1..1   ONE launch per evaluation (reduction, in-launch all-reduce of the partials, force map from
 *                                  charges kept in LDS): 1 whenever the grid is <= 256 blocks, 0 never (two launches),
 *                                  -1 auto: while a block's charges take at most half a CU's LDS (N <~ 2.4e6), so that two
 *                                  concurrent grids (other streams, other processes on the same GPU) can both be resident.
 *                                  The kernel's workgroups wait for each other inside the launch, which needs them all
 *                                  resident together.  TWO such grids fit side by side (registers: two 4-wave blocks per CU
 *                                  at 2 particles per lane; LDS: the auto rule); when more grids, or foreign kernels,
 *                                  hold the CUs, part of a grid cannot start.  Every wait is bounded (~0.4 s): workgroups
 *                                  that give up poison their share of the forces with NaN and LEAVE, which lets the rest
 *                                  of the grid start; the last workgroup to give up finds every partial in place and
 *                                  completes the whole evaluation alone (same fold, same bits, ~1 ms).  Such an
 *                                  evaluation is late but valid and is not reported as an error.  If it cannot be
 *                                  completed (not observed outside fault injection) the forces stay NaN and the result
 *                                  read or the NEXT cavmd_compute_* call, whichever comes first, returns
 *                                  CAVMD_ERR_SYNC_TIMEOUT (that call enqueues nothing).  After a starved evaluation the
 *                                  workspace SUSPENDS the single launch ("sync_timeout_seen" reads 1): two launches for
 *                                  2^16 evaluations, then one probe -- whoever held the CUs may have gone; a probe that
 *                                  starves again (late but valid, as before) makes the pause 8 times longer, up to 2^31;
 *                                  a pause starts over at 2^16 once the single launch has run healthy for longer than
 *                                  the last pause.  After a FAILED evaluation it stays suspended until the caller
 *                                  writes this tunable again.  A GPU known to be shared by three or more processes that
 *                                  each evaluate N > 1024 can skip the slow steps with the environment variable
 *                                  CAVMD_PERSISTENT=0, read by cavmd_create (=1 forces the single launch).
 *   "persistent_suspended" (read only)  0: no; 1: paused after a starved evaluation; 2: off after a failed one
 *   "sync_timeout_seen"    0..2    read: 1 after a starved evaluation.  Write: fault-injection hook, raises the flag of
 *                                  the host-visible block as a starved kernel would -- 1: failed, 2: completed by its last
 *                                  workgroup (the next call acts on it); write 0: forget it.
 *   "debug_spin_limit", "debug_late_block", "debug_late_ticks", "debug_suspend_first"
 *                                  test hooks of the single-launch kernel: poll rounds of its bounded waits (0 = default),
 *                                  a workgroup (index, -1 = none) that starts late by that many ticks of the 100 MHz
 *                                  clock, as if its CU had been held by another grid -- a real starved evaluation on
 *                                  demand -- and the length of the next pause in evaluations
 *   "rho_lane_particle"    -1..3   density-field mapping: 0 lane = wavevector, 1 / 2 / 3 lane = particle with 25 / 10 / 5
 *                                  wavevectors per chunk, -1 auto by n_k
 *   "persistent_lds_kb"    0..156  LDS budget per block of the single-launch kernel in KiB (0 = default); the charges of tiles
 *                                  beyond it are read a second time
 *   "persistent_balanced"  -1..1   partition of the particles over the blocks of the single-launch kernel: 0 tiles dealt
 *                                  round-robin (the two-launch path's partition: then also its bits), 1 contiguous equal
 *                                  shares, -1 auto
 * Returns CAVMD_ERR_INVALID_VALUE for an unknown name or an out-of-range value.  None of them changes results
 * beyond the last bit of the dipole (different but fixed summation trees). */
CAVMD_API int cavmd_set_tunable(cavmd_workspace* ws, const char* name, int value);
CAVMD_API int cavmd_get_tunable(cavmd_workspace* ws, const char* name, int* value);
CAVMD_API int cavmd_device_info(cavmd_workspace* ws, int* device, int* compute_units, char* arch_name, size_t arch_name_len);
CAVMD_API const char* cavmd_error_string(int status);
CAVMD_API int cavmd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CAVMD_H_ */
