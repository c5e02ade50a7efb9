// leaf_pairs_hip.h -- host-side mirror of the tree codes' direct (near-field) sums, backed by nbx_leaf_pair_forces
// (include/nbody_hip.h).  In the reference this step is FMM_Parlay<D>::p2p_phase(forces, bodies)
// (nbody-sim-new/fmm_parlay.cpp:916-1022, tree held by the object), the BVH leaf loop (bvh.cpp:150-176) and the
// octree leaf term (octree.cpp:105-125); here the tree is handed over as CSR leaf lists so that any tree can feed it.
#ifndef NBODY_AMD_LEAF_PAIRS_HIP_H
#define NBODY_AMD_LEAF_PAIRS_HIP_H

#include <cstdint>
#include <vector>

#if __has_include("nbody_types.h")
#include "nbody_types.h"
#else
#include "body.h"
#include "vector.h"
#endif

// leaf l owns bodies leaf_bodies[leaf_offsets[l] .. leaf_offsets[l+1]); target leaf t sums over the source leaves
// list_sources[list_offsets[t] .. list_offsets[t+1]) (own leaf included explicitly, fmm_parlay.cpp:973-974).
struct LeafLists {
    std::vector<std::uint32_t> leaf_offsets{0}, leaf_bodies, list_offsets{0}, list_sources;
    std::size_t leaves() const { return leaf_offsets.size() - 1; }
};

enum class LeafLaw : int {
    Brute = 0,     // methods.cpp:21-37   repulsive, r^2 < 1e-10 skipped
    TreeLeaf = 1,  // octree.cpp:105-125, bvh.cpp:150-176   attractive, r^2 < 1e-9 skipped
    FmmP2P = 2     // fmm_parlay.cpp:992-1020   attractive, identical positions skipped, r^2 < 1e-10 smoothed by (1e-5)^2
};

// Forces from the listed leaf pairs only (zero for bodies in no leaf).  Throws std::runtime_error on invalid lists or
// any device failure, like the solver wrappers of methods_hip.h; there is no CPU fallback.
template <int D>
std::vector<Vector<D>> leaf_pair_direct_forces_hip(const std::vector<Body<D>>& bodies, const LeafLists& lists, LeafLaw law);

// The same sums for a tree that STANDS while the bodies move -- the reference's call pattern: bvh.cpp:143-176 evaluates the
// leaf sums of a built BVH in every force evaluation, fmm_parlay.cpp:916-1022 once per step.  The structure is validated, laid
// out and uploaded once (nbx_leaf_plan_*), the bodies live on the device (a context), and every evaluation only re-gathers
// positions and runs the pair kernel.  Every method throws std::runtime_error on failure; no CPU fallback.
template <int D>
class LeafPairSimulationHip {
public:
    LeafPairSimulationHip(const std::vector<Body<D>>& bodies, const LeafLists& lists);
    ~LeafPairSimulationHip();
    LeafPairSimulationHip(const LeafPairSimulationHip&) = delete;
    LeafPairSimulationHip& operator=(const LeafPairSimulationHip&) = delete;
    // leaf sums of the bodies as they stand on the device, brought to the host
    std::vector<Vector<D>> forces(LeafLaw law, double G);
    // the same evaluation with the sums left on the device; returns after the device has finished (wall-clock friendly)
    void evaluate(LeafLaw law, double G);
    // nsteps x { leaf sums; update_body_velocities; update_body_positions } (methods.cpp:425-450) on the device, asynchronous
    void step(LeafLaw law, double G, double dt, int nsteps);
    void synchronize();   // waits for the steps queued so far
    void download(std::vector<Body<D>>& bodies);
    float single_launch_ms(LeafLaw law, double G);      // pair kernel of one evaluation
    float back_to_back_ms(LeafLaw law, int reps);      // measurement: mean of the second half of `reps` launches in a row
private:
    struct nbx_ctx* ctx_ = nullptr;
    struct nbx_leaf_plan* plan_ = nullptr;
    std::size_t n_ = 0;
};

// Fixed-depth subdivision of the bodies' bounding box (2^depth cells per axis, box padded like fmm.cpp:386-387):
// non-empty cells are the leaves, each leaf's list is itself followed by its non-empty adjacent cells -- the simplest
// tree that produces the reference's neighbour-list structure (fmm.cpp:455-476).
template <int D>
LeafLists build_uniform_leaves(const std::vector<Body<D>>& bodies, int depth);

// kernel time of the most recent leaf_pair_direct_forces_hip call on this thread (ms)
float last_leaf_pair_kernel_ms();

#endif
