/*
 * nbody_hip.h -- C ABI of libnbody_hip.so: the MI355X (gfx950) all-pairs force + kick/drift path.
 *
 * This is the drop-in boundary for ONE hot path of mathaiml5/NBody-simulation-parallel: the O(N^2)
 * brute-force force evaluation and the two integrator helpers.  The reference has no FFI; its
 * "plugin API" is the free-function template shape of nbody-sim-new/methods.h:29-37 and :85-91,
 * called from run_benchmark<D> (nbody-sim-new/main.cpp:137-140).  Every entry point below names the
 * reference interface it replaces.  C++ wrappers with the reference's exact signatures live in
 * nbody-simulation-parallel_amd/host/methods_hip.h; INTEGRATION.md shows the harness-side binding.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, opaque context handle, int status (0 = NBX_OK).
 *    No C++ types, no exceptions, no torch types cross this boundary.
 *  - Host body arrays are the reference's Body<D> memory (nbody-sim-new/body.h:8-11):
 *    double position[D]; double velocity[D]; double mass;  i.e. stride 56 B (D=3) / 40 B (D=2).
 *    Host force arrays are the reference's Vector<D> memory (vector.h:9-12): double[D] per body.
 *  - The device computes in fp32 on SoA arrays; positions and masses are rounded to fp32 at this
 *    boundary.  Returned forces are F_i = -(G m_i) * sum_j m_j (p_j - p_i)/r^4 with pairs of
 *    r^2 < 1e-10 skipped -- the reference law exactly as written (methods.cpp:21-37), including its
 *    sign and its extra power of r; G*m_i is applied in fp64.
 *  - PRECISION DEFAULT: the reference's arithmetic is fp64 (vector.h:9-12); every entry point below runs the
 *    MIXED MODE by default -- fp32 pair terms for every target, then an fp64 re-evaluation of the targets whose
 *    fp32 sum cannot be trusted to 1e-5 relative (nbx_ctx_set_refine) -- so that every body's force is within
 *    1e-5 relative of brute_force_seq_n_body on the same (fp32-representable) inputs.  nbx_set_default_refine
 *    changes the tolerance for contexts made afterwards; 0 = plain fp32 (a few bodies per 100,000 then miss 1e-5).
 *  - There is NO CPU fallback: without a usable HIP device every compute call fails with
 *    NBX_ERR_NO_DEVICE / NBX_ERR_HIP.
 */
#ifndef NBODY_HIP_H
#define NBODY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define NBX_ABI_VERSION 5   /* 5: + the measurement entries nbx_variant_kernel_symbol, nbx_ctx_enable_clock_stamps, nbx_ctx_shader_clock, nbx_measure_valu_ceiling (additions only); 2: + nbx_leaf_pair_forces, nbx_*_set_softening, nbx_*_set_law, nbx_node_verify_exchange, nbx_ctx_close_set_mode, nbx_ctx_kick_drift2, nbx_*_step_kdk; 3: + nbx_*_set_refine, nbx_ctx_refine_stats, the strict fp64 kernel variant (additions only); 4: mixed mode ON by default (nbx_set_default_refine), nbx_brute_force_forces_ex, nbx_node_refine_stats, the leaf plan (nbx_leaf_plan_*); nbx_leaf_pair_forces no longer reads NBX_LEAF_TIMING_REPS */

/* status codes */
enum {
    NBX_OK = 0,
    NBX_ERR_INVALID = 1,    /* bad argument (null pointer, dim not 2/3, shard out of range ...) */
    NBX_ERR_NO_DEVICE = 2,  /* no HIP device visible */
    NBX_ERR_HIP = 3,        /* a HIP runtime call failed; nbx_last_error_detail() has the text */
    NBX_ERR_ALLOC = 4,      /* device or host allocation failed */
    NBX_ERR_STATE = 5       /* call made in the wrong order (e.g. step before upload) */
};

/* nbody-sim-new/utils.h:21 -- the reference's G, for callers that do not carry their own. */
#define NBX_REFERENCE_G 4.471e-21
/* nbody-sim-new/methods.cpp:24 -- pairs with r^2 below this are skipped. */
#define NBX_R2_SKIP 1e-10

typedef struct nbx_ctx nbx_ctx;

/* ---- library --------------------------------------------------------------------------------- */
int nbx_abi_version(void);
const char* nbx_strerror(int status);
/* Text of the last HIP failure seen by the calling thread ("" if none). */
const char* nbx_last_error_detail(void);
int nbx_device_count(int* count);
/* One-time start-up of `device` (HIP runtime, code-object load, first launches; ~0.15 s) so that a harness
 * that times whole calls, like the reference's safely_execute (utils.h:87-104), does not charge it to the
 * first solver call -- the way OpenMP's thread pool is already up when the reference times its CPU rows. */
int nbx_warmup(int device);
/* The library keeps a few idle HIP streams per device, the RCCL communicators of destroyed nodes, the device allocations of the last
 * two destroyed contexts per device (up to 1 GiB each: a context of N = 2^20 bodies holds ~0.3 GB, most of it the mixed mode's fp64 sums), and the device allocation of
 * the last nbx_leaf_pair_forces call per device (up to 2 GiB), for the next
 * context / node / call on the same devices (creating a stream costs ~1.4 ms and destroying one up to 3 ms on this runtime,
 * a communicator set for 8 GPUs far more -- against a force evaluation of 0.07 ms at N = 1,000).  This gives them
 * back; a harness calls it before it exits.  Safe to call at any time; live contexts and nodes are not touched. */
int nbx_release_cached(void);
/* Process-wide precision default taken by every context, node and one-shot call created AFTER this call (thread-safe):
 * rel_tolerance = 0: plain fp32; otherwise the mixed mode of nbx_ctx_set_refine with this tolerance and sigma factor
 * (0 = the library's calibrated factor).  The library starts with (1e-5, 0): the north star's tolerance for every body.
 * Same argument ranges as nbx_ctx_set_refine (NBX_ERR_INVALID otherwise). */
int nbx_set_default_refine(double rel_tolerance, double sigma_factor);
int nbx_get_default_refine(double* rel_tolerance, double* sigma_factor);
/* The calibrated sigma factor the mixed mode uses for `dim` with the default (three-level) force kernel when the caller passes 0
 * (24 in 3D, 32 in 2D; tests pin it).  The two-level variant "fastpk_t8_w3_u4", selected by name, keeps round 3's 48 / 64. */
double nbx_refine_sigma_default(int dim);

/* ---- one-shot entry points (host memory in, host memory out) --------------------------------- */

/* Replaces brute_force_seq_n_body<D> / brute_force_omp_n_body_{1,2}<D>
 * (nbody-sim-new/methods.h:29-37; methods.cpp:7-42, 45-95, 98-136).
 * bodies: n x Body<dim>, body_stride_bytes = sizeof(Body<dim>) (56 or 40; larger strides allowed).
 * forces_out: n x Vector<dim> (n*dim doubles).  Runs pack -> H2D -> kernel -> D2H on `device`.
 * kernel_ms (optional) receives the force kernel's own duration (hipEvent), for roofline figures;
 * the wall time of the whole call is what the reference's safely_execute (utils.h:87-104) times. */
int nbx_brute_force_forces(const void* bodies, size_t n, int dim, size_t body_stride_bytes,
                           double G, int device, double* forces_out, float* kernel_ms);
/* The same call with the precision spelled out and a report of what ran.  rel_tolerance < 0: the process default
 * (nbx_set_default_refine); 0: plain fp32; > 0: mixed mode with this tolerance.  info (optional) receives:
 *   kernel_ms        the force kernel's device time (hipEvent pair around it);  refine_ms  that of the mixed mode's select /
 *                    fp64 / fold kernels that followed it (0 in plain fp32)
 *   refine_tolerance the tolerance that was in force (0: plain fp32, or the mixed mode does not apply to the kernel that ran)
 *   refine_selected  targets the selection rule listed;  refine_refined  targets re-evaluated in fp64 (always equal: the
 *                    fp64 pass has room for every target of the shard -- a long list costs time, never accuracy)
 *   variant          kernel variant that ran (nbx_variant_name);  close_set_mode  NBX_CLOSE_* of the evaluation */
typedef struct nbx_eval_info {
    float kernel_ms, refine_ms;
    double refine_tolerance;
    unsigned refine_selected, refine_refined;
    int variant, close_set_mode;
} nbx_eval_info;
int nbx_brute_force_forces_ex(const void* bodies, size_t n, int dim, size_t body_stride_bytes, double G, int device,
                              double rel_tolerance, double* forces_out, nbx_eval_info* info);

/* Replaces the loop  { f = brute_force_*_n_body(bodies); update_body_velocities(bodies, f, dt);
 * update_body_positions(bodies, dt); }  repeated nsteps times (methods.h:85-91; methods.cpp:425-450;
 * the reference defines the two helpers but never calls them -- SURVEY F6).  bodies is updated in
 * place (positions and velocities; masses untouched).  State stays on the device between steps. */
int nbx_leapfrog(void* bodies, size_t n, int dim, size_t body_stride_bytes,
                 double G, double dt, int nsteps, int device, float* kernel_ms_total);

/* ---- leaf-pair direct sums (the near-field step of the reference's tree codes) ------------
 * Replaces the pointer-chasing direct sums of FMM_Parlay<D>::p2p_phase (nbody-sim-new/fmm_parlay.cpp:916-1022),
 * of the BVH leaf loop (bvh.cpp:150-176) and of the Barnes-Hut octree's leaf term (octree.cpp:105-125): for every
 * target leaf t and every source leaf s on t's list, every body of t sums the pair terms of every body of s.
 *   leaf_offsets[n_leaves+1], leaf_bodies[leaf_offsets[n_leaves]]: leaf l owns the bodies (indices into `bodies`)
 *       leaf_bodies[leaf_offsets[l] .. leaf_offsets[l+1]); a body belongs to at most one leaf; leaves may be empty.
 *   list_offsets[n_leaves+1], list_sources[list_offsets[n_leaves]]: the source leaves of target leaf t, summed in
 *       list order (include t itself for the leaf's own bodies, like the reference's "self interactions",
 *       fmm_parlay.cpp:973-974).
 *   law: per-pair rule, all of the form  G m_i m_j d / r^4  with d = p_j - p_i:
 *       NBX_LAW_BRUTE      methods.cpp:21-37: force -= ...; pairs with r^2 < 1e-10 skipped (the brute-force law)
 *       NBX_LAW_TREE_LEAF  octree.cpp:105-125 / bvh.cpp:150-176: force += ...; pairs with r^2 < 1e-9 skipped
 *       NBX_LAW_FMM_P2P    fmm_parlay.cpp:992-1020: force += ...; identical positions (|d_k| <= 1e-14) skipped; for
 *                          r^2 < 1e-10 the magnitude uses r^2 + (1e-5)^2 while the direction stays d/|d|
 * forces_out: n x Vector<dim>, zero for bodies in no leaf.  fp32 pair terms on leaf-ordered source pairs, fp64 sums;
 * every index array is validated before the pair kernels follow it (NBX_ERR_INVALID: on the host for small structures; larger
 * ones are laid out ON THE DEVICE, csrc/leaf_plan_device.h, where every index is checked before it is dereferenced and the
 * refusal comes back with the layout's 64-byte summary).  kernel_ms (optional) receives the pair kernel's duration (hipEvent).
 * This one-shot form validates, lays out and uploads the structure on every call (2.4-3.1 ms at N = 2^20, 84 MB of it PCIe);
 * a tree code that evaluates its leaf sums every step from resident bodies uses the PLAN below. */
enum { NBX_LAW_BRUTE = 0, NBX_LAW_TREE_LEAF = 1, NBX_LAW_FMM_P2P = 2 };
int nbx_leaf_pair_forces(const void* bodies, size_t n, int dim, size_t body_stride_bytes,
                         const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves,
                         const uint32_t* list_offsets, const uint32_t* list_sources,
                         int law, double G, int device, double* forces_out, float* kernel_ms);

/* ---- device-resident leaf plan ---------------------------------------------------------------------------------------
 * The reference's tree codes evaluate their leaf sums in EVERY force evaluation, from bodies that stay where they are
 * (bvh.cpp:143-176 BVH::calculate_force per body over the leaves its traversal accepts; fmm_parlay.cpp:916-1022 p2p_phase per
 * step) while the leaf structure changes only when the tree is rebuilt.  A plan is that structure made resident: the CSR
 * arrays are validated ONCE, the launch is laid out once (leaf-ordered slots, copy runs, workgroup tables: on the device,
 * csrc/leaf_plan_device.h, or -- small structures -- on the host, csrc/leaf_plan.h; the two make the same plan word for word)
 * and stays on `device` with the buffers the kernels need; each evaluation then only re-gathers positions and runs the pair
 * kernel.  A caller that rebuilds its tree per evaluation, as the reference does (methods.cpp:377-401), makes a new plan per
 * evaluation: 0.5-1.2 ms at N = 2^20.  NBODY_HIP_LEAF_PLANNER=host|device in the environment forces one planner.
 * Arguments as for nbx_leaf_pair_forces.  A plan is bound to its device, dim and body count; not thread-safe. */
typedef struct nbx_leaf_plan nbx_leaf_plan;
int nbx_leaf_plan_create(nbx_leaf_plan** out, int device, int dim, size_t n_bodies,
                         const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves,
                         const uint32_t* list_offsets, const uint32_t* list_sources);
int nbx_leaf_plan_destroy(nbx_leaf_plan* plan);
/* Host bodies in, host forces out (n_bodies x Vector<dim>; zero for bodies in no leaf): one H2D copy of the Body<dim> array,
 * gather, pair kernel, one D2H copy.  kernel_ms (optional): the pair kernel's duration. */
int nbx_leaf_plan_forces(nbx_leaf_plan* plan, const void* bodies, size_t body_stride_bytes, int law, double G,
                         double* forces_out, float* kernel_ms);
/* Bodies RESIDENT on the device: positions and masses are gathered from `ctx` (a single-shard context of n_bodies bodies on
 * the plan's device, dim as the plan's; its fp32 source copy as it stands after the last upload / kick-drift), everything is
 * ordered on the context's stream.  forces_out may be NULL: the sums then stay on the device for nbx_leaf_plan_get_forces /
 * nbx_leaf_plan_kick_drift and the call does not wait for the device (kernel_ms must be NULL too).  With forces_out or
 * kernel_ms the call synchronises the stream. */
int nbx_leaf_plan_forces_ctx(nbx_leaf_plan* plan, nbx_ctx* ctx, int law, double G, double* forces_out, float* kernel_ms);
/* Forces of the last evaluation (n_bodies x Vector<dim>) -- F = +-(G m_i) x the leaf sums, the law's sign.  Synchronises.  The masses
 * are read where that evaluation found them: after nbx_leaf_plan_forces_ctx the context must still exist. */
int nbx_leaf_plan_get_forces(nbx_leaf_plan* plan, double* forces_out);
/* update_body_velocities + update_body_positions (methods.cpp:425-450) of `ctx`'s bodies from the LAST evaluation's leaf sums
 * (law and G as given there), fp64, the arithmetic of nbx_ctx_kick_drift; bodies in no leaf drift with their velocity.
 * Asynchronous on the context's stream.  This is the stepping loop of a tree code whose far field is zero: what `nbody_sim
 * -m p --steps k` runs. */
int nbx_leaf_plan_kick_drift(nbx_leaf_plan* plan, nbx_ctx* ctx, double dt);
/* nsteps x { nbx_leaf_plan_forces_ctx(plan, ctx, law, G, NULL, NULL); nbx_leaf_plan_kick_drift(plan, ctx, dt) } with the structure
 * standing (the loop of methods.cpp:425-450 around the leaf sums; same kernels, same arithmetic, same results), in one call:
 * the steps are queued ahead of the device without the two calls' event bookkeeping.  Asynchronous; afterwards the plan holds the
 * LAST step's sums (of the positions before that step's drift), as after the two calls.  nsteps >= 0. */
int nbx_leaf_plan_step(nbx_leaf_plan* plan, nbx_ctx* ctx, int law, double G, double dt, int nsteps);
/* MEASUREMENT entry (tools/, bench.py): the pair kernel launched `reps` times back to back on the bodies of the last
 * evaluation (same sums every time); mean_ms = mean duration of the second half of the launches -- the kernel with the clocks
 * up, which a single launch from idle does not see.  1 <= reps <= 1000.  Synchronises. */
int nbx_leaf_plan_time_kernel(nbx_leaf_plan* plan, int law, int reps, float* mean_ms);
/* Counts of the layout: padded slots, copy runs, workgroups, wave64 per workgroup (any pointer may be NULL). */
int nbx_leaf_plan_info(const nbx_leaf_plan* plan, size_t* slots, size_t* runs, size_t* workgroups, int* waves_per_workgroup);

/* ---- device-resident context ------------------------------------------------------------------
 * A context owns the targets of ONE shard of an N-body system on ONE device and a full-length
 * fp32 copy of all N sources.  n_shards = 1, shard = 0 is the single-GPU case.  With n_shards = G
 * the bodies are split into G contiguous shards of shard_len = ceil(N/G) bodies (the last one
 * short); rank g creates a context with shard = g and exchanges fp32 positions with the others
 * once per step (RCCL all-gather of the buffer described by nbx_ctx_gather_layout).
 */
int nbx_ctx_create(nbx_ctx** out, int device, int dim, size_t n_total, int n_shards, int shard);
int nbx_ctx_destroy(nbx_ctx* ctx);

/* Use a caller-owned HIP stream (hipStream_t as void*) for every launch and copy of this context;
 * NULL restores the context's own stream.  Lets a torch.cuda.Stream order the kernels. */
int nbx_ctx_set_stream(nbx_ctx* ctx, void* hip_stream);

/* Use caller-owned device memory for the two all-gather buffers (see nbx_ctx_gather_layout for the
 * required sizes).  Must be called before nbx_ctx_upload_bodies.  This is how the one-process-per-
 * GPU host layer hands torch-allocated tensors to RCCL and to the kernels without a copy. */
int nbx_ctx_set_gather_buffers(nbx_ctx* ctx, void* pos_all_f32, void* mass_all_f32);

/* Layout of the exchange buffers:
 *   pos_all : float[n_shards][dim][shard_pad]   -- chunk g = shard g's x[], y[], (z[]) arrays
 *   mass_all: float[n_shards][shard_pad]
 * shard_len = bodies per shard, shard_pad = shard_len rounded up to a multiple of 4096 bodies; pad
 * entries are massless bodies at the origin (they contribute exactly zero).
 * Any out pointer may be NULL. */
int nbx_ctx_gather_layout(const nbx_ctx* ctx, size_t* shard_len, size_t* shard_pad,
                          void** pos_all_f32, void** mass_all_f32);

/* Upload ALL n_total bodies (Body<dim> array as above).  Fills the fp32 source copy for every
 * shard and the fp64 position/velocity/mass state of this context's own shard. */
int nbx_ctx_upload_bodies(nbx_ctx* ctx, const void* bodies, size_t body_stride_bytes);

/* The same upload for a sharded context that moves only ITS OWN bodies over the host link (one process per GPU: every rank
 * uploading the whole array costs G copies of it):
 *   1. nbx_ctx_upload_shard: shard_bodies = the shard_len-or-fewer bodies this context owns (Body<dim> array); packs the own
 *      chunk of the exchange buffers and the fp64 state; returns the largest |mass| and |coordinate| of the shard;
 *   2. the caller fills the OTHER chunks of pos_all / mass_all device to device (one all-gather of each; layout above) and
 *      combines the two maxima over all ranks;
 *   3. nbx_ctx_upload_finish with the combined maxima: the fast path's preconditions are decided from them (they are properties
 *      of all bodies), and -- for small-coordinate systems -- the close-set probe runs against every chunk.
 * nbx_ctx_upload_bodies = the three steps on the whole array in one call. */
int nbx_ctx_upload_shard(nbx_ctx* ctx, const void* shard_bodies, size_t body_stride_bytes, double* max_abs_mass, double* max_abs_coord);
int nbx_ctx_upload_finish(nbx_ctx* ctx, double max_abs_mass_all, double max_abs_coord_all);

/* Accelerations of this shard's targets: a_i = sum_j m_j (p_j - p_i)/r^4 over the selected sources
 * (the reference force without the -(G m_i) factor).  which: 0 = all shards' sources,
 * 1 = only this shard's own chunk (needs no remote data), 2 = every other shard's chunk, added to
 * what a preceding which=1 call produced.  Asynchronous on the context's stream. */
enum { NBX_SRC_ALL = 0, NBX_SRC_LOCAL = 1, NBX_SRC_REMOTE = 2 };
int nbx_ctx_compute_accel(nbx_ctx* ctx, int which);

/* Fused kick + drift for this shard (methods.cpp:425-438 then :440-450, fp64):
 *   F = -(G m) a;  v += (F / m) * dt;  x += v * dt;
 * and refresh of this shard's fp32 chunk of pos_all.  Asynchronous on the context's stream. */
int nbx_ctx_kick_drift(nbx_ctx* ctx, double G, double dt);
/* The same kernel with separate steps for the two helpers:  v += (F / m) * dt_kick;  x += v * dt_drift.
 * dt_drift = 0 is a pure kick (update_body_velocities alone): positions, accelerations and close-set lists stay valid. */
int nbx_ctx_kick_drift2(nbx_ctx* ctx, double G, double dt_kick, double dt_drift);

/* nsteps x { compute_accel(ALL); kick_drift } -- single-shard contexts only (n_shards == 1).  From 4 steps
 * on, one step is captured into a hipGraph and replayed (launch-bound regime at small N); nbx_ctx_kernel_time
 * then reports those steps by their whole-step time (one event pair around the replay sequence).
 * NBODY_HIP_NO_GRAPHS=1 disables the capture. */
int nbx_ctx_step(nbx_ctx* ctx, double G, double dt, int nsteps);
/* EXTENSION: the same two helpers composed as a synchronised kick-drift-kick leapfrog (second order in dt):
 *   v += (F/m) dt/2;  x += v dt;  F = forces(x);  v += (F/m) dt/2      per step,
 * with adjacent half-kicks merged: nsteps + 1 force evaluations for nsteps steps.  Single-shard contexts only; the
 * sharded form is nbx_node_step_kdk.  (nbx_ctx_step is the reference helpers' plain order, kick then drift: first order.) */
int nbx_ctx_step_kdk(nbx_ctx* ctx, double G, double dt, int nsteps);

/* Forces of this shard's targets as Vector<dim>[shard_len] doubles: F_i = -(G m_i) a_i.
 * Synchronises the stream. */
int nbx_ctx_get_forces(nbx_ctx* ctx, double G, double* forces_out);
/* The reference's accuracy metric (utils.h:170-219, ACCURACY_PCT_THRESHOLD 1 %, ACCURACY_FORCE_THRESHOLD 1e-20)
 * evaluated on the device against reference_forces (host, Vector<dim>[shard_len] of this shard): percent of
 * bodies whose every force component is within 1 % (absolute 1e-9 where |ref| < 1e-20).  The device forces are
 * never copied back; 4 bytes return.  Synchronises the stream. */
int nbx_ctx_accuracy(nbx_ctx* ctx, double G, const double* reference_forces, double* percent);
/* Raw fp32 accelerations, SoA float[dim][shard_len].  Synchronises the stream. */
int nbx_ctx_get_accel(nbx_ctx* ctx, float* accel_out);
/* Write this shard's positions and velocities (fp64 state) back into the caller's full-length
 * Body<dim> array (only entries [shard*shard_len, ...) are touched).  Synchronises the stream. */
int nbx_ctx_download_bodies(nbx_ctx* ctx, void* bodies, size_t body_stride_bytes);

/* Energy diagnostic (for energy-drift checks; the reference has none -- SURVEY 5): this shard's share of
 *   kinetic   = sum_i m_i |v_i|^2 / 2                       (fp64 state)
 *   potential = sum_i (G m_i / 4) sum_{j != i} m_j / r_ij^2   (fp32 pair terms, fp64 sums, pairs with
 *               r^2 < 1e-10 skipped) -- summed over all shards this is U = sum_{i<j} G m_i m_j / (2 r^2),
 * the potential whose gradient is the reference's force law (methods.cpp:21-37), so kinetic + potential is
 * what a symplectic kick/drift conserves.  Uses the positions currently in the exchange buffer for the
 * sources.  One O(N^2/G) kernel; synchronises the stream. */
int nbx_ctx_energy(nbx_ctx* ctx, double G, double* kinetic, double* potential);

int nbx_ctx_synchronize(nbx_ctx* ctx);

/* Tuning knobs.  source_splits: number of slices the source loop is cut into (0 = automatic, else a
 * lower bound in [1,256]; more workgroups for small shards; partial sums are combined in a fixed
 * order).  variant: force-kernel variant id in [0, nbx_num_variants()), -1 = library default (see
 * DESIGN.md for the table).  "fast" variants drop the per-pair r^2 guard, find the targets that own a
 * pair closer than 1e-3 exactly (candidates = a coordinate below 16384 in magnitude, checked against each
 * other once per position update) and evaluate those with the guarded kernel inside the same launch, so the
 * result keeps the reference's skip semantics; small-coordinate systems (more than 1/8 of the shard in the candidate set)
 * find their close pairs through sorted cells instead; when a mass exceeds 1e10, or more than 1/8 of the shard really
 * owns a pair closer than 1e-3, the library substitutes the guarded default. */
int nbx_ctx_set_tuning(nbx_ctx* ctx, int source_splits, int variant);
/* How the fast path currently keeps the reference's r^2 < 1e-10 skip rule (diagnostic):
 *   NBX_CLOSE_CANDIDATE_PAIRS  candidate targets x candidate sources (few bodies have a coordinate below 16384)
 *   NBX_CLOSE_SORTED_CELLS     most bodies are candidates: close pairs found through sorted cells
 *   NBX_CLOSE_GUARDED_KERNEL   the per-pair guarded kernel runs instead (mass above 1e10, > 1/8 of the shard owning a close
 *                              pair, or an exact variant selected)
 *   NBX_CLOSE_NONE             softened law: nothing to guard
 * Decided at upload and re-evaluated during long runs from an asynchronous read-back of the device counters every 16
 * steps (never a wait) -- also inside one long nbx_ctx_step call, whose graph replays go out in blocks of 16 steps (the look takes effect
 * within the call only while the device keeps up with the host: at large N the host has queued every block before the first
 * copy lands, and the next call acts on it; a change of mode inside the call drains the stream before the step is re-captured).  candidates_seen / bad_seen: the most recent counts the host has seen (0 before the first). */
enum { NBX_CLOSE_CANDIDATE_PAIRS = 0, NBX_CLOSE_SORTED_CELLS = 1, NBX_CLOSE_GUARDED_KERNEL = 2, NBX_CLOSE_NONE = 3 };
int nbx_ctx_close_set_mode(nbx_ctx* ctx, int* mode, unsigned* candidates_seen, unsigned* bad_seen);

/* EXTENSION (not in the reference: its brute force is unsoftened): Plummer softening of the pair law,
 *   a_i = sum_{j != i} m_j (p_j - p_i) / (r^2 + epsilon^2)^2 ,   U = sum_{i<j} G m_i m_j / (2 (r^2 + epsilon^2)),
 * every pair counted (no r^2 < 1e-10 skip; a body never acts on itself).  epsilon = 0 (default) is the reference law.
 * Same kernel, same speed: epsilon^2 replaces the fast kernel's r^2 bias and no close-set bookkeeping is needed.
 * epsilon must be 0 or in [1e-6, 1e15], and max|m| / epsilon^4 (Newtonian law: / epsilon^3) must be a finite, normal fp32
 * number -- neither overflow nor underflow to an all-zero field (checked at the next force evaluation: NBX_ERR_INVALID; with
 * the reference's masses up to 1e8 that is epsilon <= ~3e9).  Applies to compute_accel / step / energy of this context. */
int nbx_ctx_set_softening(nbx_ctx* ctx, double epsilon);
/* EXTENSION (SURVEY 5/7 `--law {reference,newton}`; the reference has one law): NBX_FORCE_LAW_REFERENCE (default) is the
 * reference's repulsive m_j d / r^4 form; NBX_FORCE_LAW_NEWTON is the attractive, Plummer-softened Newtonian law
 *   F_i = +G m_i sum_{j != i} m_j (p_j - p_i) / (r^2 + epsilon^2)^(3/2),   U = -sum_{i<j} G m_i m_j / sqrt(r^2 + epsilon^2),
 * which needs a softening length > 0 (NBX_ERR_STATE at the next evaluation otherwise).  Same kernel with v_rsq_f32 in
 * place of v_rcp_f32 and one more multiply.  Forces, kick/drift, energy and the accuracy metric all follow the law. */
enum { NBX_FORCE_LAW_REFERENCE = 0, NBX_FORCE_LAW_NEWTON = 1 };
int nbx_ctx_set_law(nbx_ctx* ctx, int law);
/* PRECISION.  The reference's arithmetic is fp64 throughout (vector.h:9-12, methods.cpp:21-37); the device computes pair terms
 * in fp32 (stated tolerance: DESIGN.md section 4).  Two ways to go beyond plain fp32, both on the reference law -- the second is
 * what every context starts with (1e-5):
 *  - the variant "strict_f64_t4" (nbx_ctx_set_tuning): EVERY pair term, the skip-rule comparison and every sum in fp64 on
 *    the device's fp32-representable positions and masses -- agrees with brute_force_seq_n_body on the same inputs to
 *    ~1e-13 relative; about 2.5x the default kernel's time;
 *  - MIXED MODE, nbx_ctx_set_refine(ctx, rel_tolerance, sigma_factor): fp32 for every target, then the targets whose fp32
 *    sum cannot be trusted to rel_tolerance are re-evaluated by the strict kernel.  The criterion (force_kernel.hip,
 *    refine_select_kernel): the rounding error of a target's fp32 sum has standard deviation ~ c u sqrt(Q_i), u = 2^-24,
 *    Q_i = sum over the 64-source blocks of |block partial sum|^2 (accumulated by the default three-level kernel at no measurable
 *    cost; the two-level variant "fastpk_t8_w3_u4" sums over its 256-source tiles);
 *    target i is re-evaluated when  rel_tolerance |a_i| < sigma_factor u sqrt(Q_i)  -- a chance cancellation: the blocks'
 *    pulls add up to far less than they are -- or when it is a close-set target.  sigma_factor = 0 takes the library's
 *    calibrated default (nbx_refine_sigma_default).  rel_tolerance = 0 switches the mode off; a new context starts with the
 *    process default (nbx_set_default_refine: 1e-5 unless changed).  Ignored with a softening length, the
 *    Newtonian law, or a non-fast variant.  EVERY selected target is re-evaluated: the fp64 pass has room for the whole
 *    shard (its source slices shrink as the list grows, on the device) -- nbx_ctx_refine_stats reports the count. */
int nbx_ctx_set_refine(nbx_ctx* ctx, double rel_tolerance, double sigma_factor);
/* After a mixed-mode force evaluation: selected = targets the rule listed, refined = those re-evaluated in fp64
 * (the same number: there is no capacity to overflow).  Either pointer may be NULL.  Synchronises the stream. */
int nbx_ctx_refine_stats(nbx_ctx* ctx, unsigned* selected, unsigned* refined);
/* The per-target statistic the last force evaluation wrote beside the accelerations, double[shard_len]:
 *   mixed mode:                  Q_i = sum over the source blocks of |block partial sum|^2 (what the selection rule thresholds);
 *   variant "strict_f64_t4_mag": S_i = sum_j |a_ij|, the sum of the pair terms' magnitudes (the yardstick of a cancelling
 *                                sum's error: backward error = |da_i| / S_i, condition number kappa_i = S_i / |a_i|).
 * NBX_ERR_STATE after any other evaluation.  Synchronises the stream. */
int nbx_ctx_get_aux(nbx_ctx* ctx, double* out);
/* The variant and slice count the next force evaluation will use (after upload). */
int nbx_ctx_effective_tuning(nbx_ctx* ctx, int* variant, int* source_splits);
int nbx_num_variants(void);
const char* nbx_variant_name(int variant);
int nbx_default_variant(void);

/* Mean duration in ms of the force-kernel launches since the last call (hipEvent pairs recorded on
 * the context's stream around each launch), and how many launches that covers.  Synchronises. */
int nbx_ctx_kernel_time(nbx_ctx* ctx, float* mean_ms, int* launches);
/* Device time in ms (total, not mean) of the mixed mode's select / fp64 re-evaluation / fold kernels that followed the force-kernel
 * launches covered by the LAST nbx_ctx_kernel_time call (graph-replayed steps carry theirs inside the whole-step time). */
int nbx_ctx_refine_time(nbx_ctx* ctx, float* total_ms);

/* ---- MEASUREMENT entries (bench.py, tools/; nothing on the product path calls them) ----------------------------------
 * The symbol of the force kernel that an evaluation with `variant` (nbx_ctx_effective_tuning) launches for `dim`, in its plain
 * (mixed_mode = 0) or mixed-mode (1: also writes the spread sums) reference-law build, demangled as rocprofv3 prints it (e.g.
 * "void nbx::(anonymous namespace)::accel_fast3l_kernel<3, 4, 3, 4, 64, 1, 0, 1>(nbx::KArgs)"): what ties a live run to the
 * committed profiles/ of the same kernel.  Needs no device.  len >= 16 (NBX_ERR_INVALID otherwise; the name is truncated to fit). */
int nbx_variant_kernel_symbol(int variant, int dim, int mixed_mode, char* buf, size_t len);
/* In-kernel clock stamps: while on, every workgroup of the default (three-level) force kernel reads the shader clock
 * (s_memtime) and the constant 100 MHz clock (s_memrealtime) around its work -- two scalar loads and one 16-byte store per
 * workgroup of ~40 ms; no measurable cost (profiles/r5) -- and nbx_ctx_shader_clock reports, for the LAST stamped launch, the
 * median / smallest / largest of  d s_memtime / d s_memrealtime x 100 MHz  over its workgroups: the clock the chip actually
 * held under THIS kernel on THIS box, which is what a roofline fraction quoted at the 2.4 GHz peak cannot say.
 * NBX_ERR_STATE when the kernel that would run carries no stamps (non-default variants) or nothing stamped has run.
 * nbx_ctx_shader_clock synchronises the stream.  Any out pointer may be NULL. */
int nbx_ctx_enable_clock_stamps(nbx_ctx* ctx, int on);
int nbx_ctx_shader_clock(nbx_ctx* ctx, double* median_mhz, double* min_mhz, double* max_mhz, int* workgroups);
/* The same box's ceiling: a pure v_pk_fma_f32 stream (16 independent chains per lane, 24 workgroups of 256 lanes per CU, as many resident as fit)
 * for about target_ms (1..2000) on `device`; *tflops = what it achieved (4 flop per lane-instruction), *shader_mhz = the
 * clock it held.  The guide's 157.3 TFLOP/s assumes 2.4 GHz and one packed FMA per cycle per lane pair. */
int nbx_measure_valu_ceiling(int device, double target_ms, double* tflops, double* shader_mhz);

/* ---- single-process multi-GPU node ---------------------------------------------------------------
 * The reference is single-process, single-device; this is new.  One context per rank
 * (n_shards = n_ranks, shard = rank) on devices[rank] (NULL: 0..n_ranks-1; a device may appear more than
 * once -- "virtual ranks", used to rehearse the sharded path on one GPU), driven by the calling thread.
 * Per step every rank's freshly drifted fp32 position chunk is all-gathered on a second HIP stream while
 * the rank's compute stream evaluates the forces from its own chunk; the pass over the other ranks'
 * chunks waits on the exchange's event.  exchange: RCCL = ncclAllGather over one communicator per node
 * (librccl is dlopen'ed on demand; needs distinct devices), PEER_COPY = every rank pushes its chunk into
 * each peer's buffer with hipMemcpyPeerAsync, AUTO = RCCL when n_ranks > 1 distinct devices, else PEER_COPY.
 * The one-process-per-GPU twin of this layer is nbody-simulation-parallel_amd/dist.py. */
typedef struct nbx_node nbx_node;
enum { NBX_EXCHANGE_AUTO = 0, NBX_EXCHANGE_PEER_COPY = 1, NBX_EXCHANGE_RCCL = 2 };
int nbx_node_create(nbx_node** out, int n_ranks, const int* devices, int dim, size_t n_total, int exchange);
int nbx_node_destroy(nbx_node* node);
int nbx_node_exchange_mode(const nbx_node* node, int* mode);
/* `bodies` is the whole Body<D> array; each rank copies only its own shard of it to its device, the other chunks of its
 * source copy arrive device to device (masses once, positions through the step's exchange).  With the RCCL exchange and
 * more than one rank the first upload over a communicator set also runs nbx_node_verify_exchange and fails (NBX_ERR_HIP,
 * detail text) if the all-gather did not deliver every chunk. */
int nbx_node_upload_bodies(nbx_node* node, const void* bodies, size_t body_stride_bytes);
/* Self-check of the position exchange: every rank's copies of the chunks it does not own are overwritten with
 * NaN, one exchange runs (RCCL all-gather or peer copies, as configured), and each copy is compared bit for bit
 * with the owner's chunk on the host.  *mismatches = number of differing fp32 values over all ranks (0 = the
 * exchange delivered every chunk everywhere).  Synchronises; leaves the buffers whole when it passes. */
int nbx_node_verify_exchange(nbx_node* node, size_t* mismatches);
int nbx_node_set_tuning(nbx_node* node, int source_splits, int variant);
int nbx_node_set_softening(nbx_node* node, double epsilon);   /* nbx_ctx_set_softening on every rank */
int nbx_node_set_law(nbx_node* node, int law);                /* nbx_ctx_set_law on every rank */
int nbx_node_set_refine(nbx_node* node, double rel_tolerance, double sigma_factor);   /* nbx_ctx_set_refine on every rank */
/* nbx_ctx_refine_stats summed over the ranks (after nbx_node_compute_forces / a step); NBX_ERR_STATE when no rank ran the mixed mode. */
int nbx_node_refine_stats(nbx_node* node, unsigned* selected, unsigned* refined);
/* Forces on all n_total bodies (Vector<dim>[n_total]); same contract as nbx_brute_force_forces. */
int nbx_node_compute_forces(nbx_node* node, double G, double* forces_out);
/* nsteps x { exchange || local forces; remote forces; kick+drift } on every rank.  Asynchronous. */
int nbx_node_step(nbx_node* node, double G, double dt, int nsteps);
int nbx_node_step_kdk(nbx_node* node, double G, double dt, int nsteps);   /* kick-drift-kick form, see nbx_ctx_step_kdk */
int nbx_node_synchronize(nbx_node* node);
int nbx_node_download_bodies(nbx_node* node, void* bodies, size_t body_stride_bytes);
int nbx_node_energy(nbx_node* node, double G, double* kinetic, double* potential);
int nbx_node_kernel_time(nbx_node* node, float* mean_ms, int* launches);
/* Self-description of a sharded evaluation (what a first run on a multi-GPU node prints): with timing enabled every evaluation
 * records events around each rank's LOCAL and REMOTE pass (compute stream) and around its part of the exchange (comm stream);
 * nbx_node_pass_times reports the LAST evaluation's figures for `rank` -- device, targets, the three durations in ms, and whether
 * the exchange had finished before the LOCAL pass did (exchange_hidden).  Any out pointer may be NULL.  Synchronises. */
int nbx_node_enable_timing(nbx_node* node, int on);
int nbx_node_pass_times(nbx_node* node, int rank, int* device, size_t* targets, float* local_ms, float* remote_ms,
                        float* exchange_ms, int* exchange_hidden);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_H */
