// methods_hip.h -- solver entry points with the reference's methods.h shape, backed by the MI355X
// HIP library through its C ABI (include/nbody_hip.h).  Drop-in for nbody-sim-new/methods.h:29-37
// and :85-91: include it next to (or instead of) methods.h, compile methods_hip.cpp with the
// reference's flags, link -lnbody_hip, and add one harness block.
//
// It only needs `Body<D>` / `Vector<D>` with the reference's memory layout: this repository's
// body.h / vector.h, or the reference's own when built inside the reference tree.
#ifndef NBODY_AMD_METHODS_HIP_H
#define NBODY_AMD_METHODS_HIP_H

#include <vector>

#if __has_include("nbody_types.h")
#include "nbody_types.h"  // this repository's layout-locked Vector<D> / Body<D>
#else
#include "body.h"  // inside the reference tree: the reference's own types (same memory)
#include "vector.h"
#endif

// Same contract as brute_force_seq_n_body<D> / brute_force_omp_n_body_{1,2}<D>: forces (not
// accelerations) on every body under the reference's law, F_i = -G m_i sum_j m_j (p_j-p_i)/r^4 with
// pairs of r^2 < 1e-10 skipped.  Throws std::runtime_error on any device failure (the harness's
// safely_execute, utils.h:95-103, then logs it and skips the row); there is no CPU fallback.
// Precision: the reference computes in fp64 (vector.h:9-12); the device runs the library's MIXED MODE by default --
// fp32 pair terms for every body, an fp64 re-evaluation of the bodies whose fp32 sum cannot be trusted -- so that every
// body's force is within 1e-5 relative of the sequential reference's on the same inputs.  set_hip_refine() changes that.
template <int D>
std::vector<Vector<D>> brute_force_hip_n_body(const std::vector<Body<D>>& bodies);

// The same evaluation on ONE GPU whatever set_hip_devices() says: the yardstick the harness holds a sharded row against
// (sampled rows of BruteForce_HIP_x<G> vs this), so that a first run on a multi-GPU node proves more than delivery.
template <int D>
std::vector<Vector<D>> brute_force_hip_single_gpu(const std::vector<Body<D>>& bodies, int device);

// Self-description of the sharded path for a first run on a multi-GPU node (the same facts bench.py --gpus N puts in its JSON
// line): over the devices of set_hip_devices(), one poisoned-buffer exchange check and ONE timed evaluation -- transport,
// mismatching values, per-rank LOCAL / REMOTE / exchange times, whether the exchange hid behind the LOCAL pass, mixed-mode counts.
struct HipNodeReport {
    struct Rank { int rank, device; std::size_t targets; float local_ms, remote_ms, exchange_ms; bool exchange_hidden; };
    const char* transport = "";          // "rccl" | "peer copies" | "none (one rank)"
    std::size_t mismatching_values = 0;  // of the exchange self-check (0 = every chunk arrived on every rank)
    std::size_t checked_values_per_rank = 0;
    std::vector<Rank> ranks;
    double refine_tolerance = 0.0;
    unsigned refine_selected = 0, refine_refined = 0;
};
template <int D>
HipNodeReport describe_hip_node(const std::vector<Body<D>>& bodies);

// nsteps x { forces; update_body_velocities(bodies, forces, dt); update_body_positions(bodies, dt); }
// (methods.cpp:425-450) with the state resident on the device between steps.
template <int D>
void leapfrog_hip_n_body(std::vector<Body<D>>& bodies, double dt, int nsteps);

// Device-resident simulation for callers that want to look at the system between steps (energy-drift
// logging, BASELINE config 5): bodies are uploaded once, stepped on the device(s), read back on demand.
// Uses the devices of set_hip_devices() (one GPU by default).  Every method throws std::runtime_error on
// a device failure.
template <int D>
class HipSimulation {
public:
    // softening > 0: Plummer-softened pair law (an extension, nbx_ctx_set_softening); 0 = the reference's law
    // newton = true: the attractive softened Newtonian law (nbx_ctx_set_law; needs softening > 0)
    HipSimulation(const std::vector<Body<D>>& bodies, double G, double softening = 0.0, bool newton = false);
    ~HipSimulation();
    HipSimulation(const HipSimulation&) = delete;
    HipSimulation& operator=(const HipSimulation&) = delete;
    void step(double dt, int nsteps);            // asynchronous; the reference helpers' order: kick, then drift
    void step_kdk(double dt, int nsteps);        // extension: synchronised kick-drift-kick leapfrog (second order)
    void energy(double& kinetic, double& potential);  // of the whole system, under the reference law's potential
    void download(std::vector<Body<D>>& bodies);
    double force_kernel_seconds();               // per-rank force-kernel time since the last call
private:
    struct nbx_node* node_ = nullptr;
    double G_;
    int ranks_ = 1;
};

// The reference's accuracy metric (compute_accuracy, utils.h:170-219) evaluated ON THE DEVICE against `reference`
// (nbx_ctx_accuracy): the device forces are never copied back -- what `-a 1` needs at sizes where the force array is
// large.  Runs its own force evaluation on one GPU; returns the percentage.  Throws like the solvers.
template <int D>
double brute_force_hip_accuracy(const std::vector<Body<D>>& bodies, const std::vector<Vector<D>>& reference);

// Timing of the most recent call on this thread, for pair-interactions/s and roofline reporting.
struct HipRunInfo {
    float kernel_ms = 0.0f;   // device time of the force evaluation(s) (hipEvent), summed over the call's launches
    int device = 0;
    double refine_tolerance = 0.0;     // mixed mode's relative tolerance in force during the call (0: plain fp32)
    unsigned refine_selected = 0;      // targets the mixed mode listed in the call's (last) force evaluation ...
    unsigned refine_refined = 0;       // ... and re-evaluated in fp64 (always the same number: nothing overflows)
};
const HipRunInfo& last_hip_run_info();

// Bring the device(s) up before anything is timed (HIP runtime, code-object load): nbx_warmup.
// Returns false (and changes nothing) if no device is usable -- the solver calls will then throw.
bool warm_up_hip();
// Give back what the library keeps between calls (idle streams, RCCL communicators): nbx_release_cached.  Before exit.
void release_hip_caches();

// Precision of every HIP entry point from now on (nbx_set_default_refine): rel_tolerance = 0 is plain fp32, otherwise the
// mixed mode's per-body relative tolerance.  The library's default is 1e-5.  Throws on an out-of-range tolerance.
void set_hip_refine(double rel_tolerance);
double hip_refine_tolerance();

// Number of HIP devices visible (0 when there is none or the runtime fails).
int hip_device_count();

// Device ordinal used by the two entry points above (default 0; NBODY_HIP_DEVICE overrides).
void set_hip_device(int device);

// Shard the bodies over several GPUs of the node inside this process (nbx_node_* of nbody_hip.h: one
// context per entry, RCCL all-gather of positions per step, or peer copies when entries share a device).
// An empty list or a single entry selects the single-GPU path.
void set_hip_devices(const std::vector<int>& devices);

#endif  // NBODY_AMD_METHODS_HIP_H
