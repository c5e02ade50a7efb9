// ref_driver.cpp -- thin extern "C" shim over the REFERENCE's own object code.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/build_ref.sh (only where /root/reference exists)
// into oracle/_ref/libnbody_ref.so together with the reference's methods.cpp compiled from where
// it lies.  Nothing from the reference is copied into this repository: this file only #includes
// the reference headers at build time and forwards plain pointers to the reference's templates.
//
// Used to (1) pin oracle/nbody_oracle.c against the true reference, (2) generate the golden
// vectors under tests/golden/, (3) optionally serve as bench.py's cpu_baseline kind "reference".
#include "methods.h"   // /root/reference/nbody-sim-new/methods.h (via -I)

#include <chrono>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

namespace {
template <int D>
std::vector<Body<D>> wrap(const double* raw, size_t n) {
    static_assert(sizeof(Body<D>) == sizeof(double) * (2 * D + 1), "Body<D> layout");
    std::vector<Body<D>> b(n);
    if (n) std::memcpy(static_cast<void*>(b.data()), raw, n * sizeof(Body<D>));
    return b;
}
template <int D>
void unwrap(const std::vector<Vector<D>>& f, double* out) {
    static_assert(sizeof(Vector<D>) == sizeof(double) * D, "Vector<D> layout");
    if (!f.empty()) std::memcpy(out, static_cast<const void*>(f.data()), f.size() * sizeof(Vector<D>));
}
template <int D>
int forces(int variant, const double* raw, size_t n, double* out) {
    auto b = wrap<D>(raw, n);
    std::vector<Vector<D>> f;
    switch (variant) {
        case 0: f = brute_force_seq_n_body<D>(b); break;
        case 1: f = brute_force_omp_n_body_1<D>(b); break;
        case 2: f = brute_force_omp_n_body_2<D>(b); break;
        case 3:
        case 4: {  // the ParlayLib twins (methods.cpp:139-186, :189-224) take and return parlay::sequence
            parlay::sequence<Body<D>> pb(b.begin(), b.end());   // as the harness does, main.cpp:95
            const parlay::sequence<Vector<D>> pf = variant == 3 ? brute_force_parlay_n_body_1<D>(pb) : brute_force_parlay_n_body_2<D>(pb);
            f.assign(pf.begin(), pf.end());
            break;
        }
        default: return -1;
    }
    unwrap<D>(f, out);
    return 0;
}
// Same call, timed around the solver only (the wrap/unwrap copies above are the shim's, not the reference's).
template <int D>
int timed(int variant, const double* raw, size_t n, double* seconds) {
    auto b = wrap<D>(raw, n);
    parlay::sequence<Body<D>> pb;
    if (variant >= 3) pb = parlay::sequence<Body<D>>(b.begin(), b.end());
    const auto t0 = std::chrono::steady_clock::now();
    size_t got = 0;
    switch (variant) {
        case 0: got = brute_force_seq_n_body<D>(b).size(); break;
        case 1: got = brute_force_omp_n_body_1<D>(b).size(); break;
        case 2: got = brute_force_omp_n_body_2<D>(b).size(); break;
        case 3: got = brute_force_parlay_n_body_1<D>(pb).size(); break;
        case 4: got = brute_force_parlay_n_body_2<D>(pb).size(); break;
        default: return -1;
    }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return got == n ? 0 : -2;
}
// The reference's BVH (bvh.h:77-103, bvh.cpp:16-126): built by its own constructor, its leaves read through the public
// `root` in left-to-right order, a leaf's bodies as indices into the caller's array (the tree holds Body<D>* into the vector
// it was built from, bvh.cpp:24-27).
template <int D>
void bvh_collect(const BVHNode<D>* node, std::vector<const BVHNode<D>*>& out) {
    if (!node) return;
    if (node->is_leaf) { out.push_back(node); return; }
    bvh_collect<D>(node->left.get(), out);
    bvh_collect<D>(node->right.get(), out);
}
template <int D>
int bvh_leaves(const double* raw, size_t n, int max_bodies, uint32_t* leaf_offsets, uint32_t* leaf_bodies, size_t* n_leaves) {
    auto b = wrap<D>(raw, n);
    BVH<D> tree(b, max_bodies);
    std::vector<const BVHNode<D>*> leaves;
    bvh_collect<D>(tree.root.get(), leaves);
    size_t k = 0;
    leaf_offsets[0] = 0;
    for (size_t l = 0; l < leaves.size(); ++l) {
        for (const Body<D>* p : leaves[l]->bodies) leaf_bodies[k++] = (uint32_t)(p - b.data());
        leaf_offsets[l + 1] = (uint32_t)k;
    }
    *n_leaves = leaves.size();
    return k == n ? 0 : -2;
}
// For every body the sum over ALL leaves of BVH<D>::calculate_force(body, leaf) (bvh.cpp:143-176: the leaf branch, a direct
// loop over the leaf's bodies): the near-field term of the BVH method with every leaf on every list, in the reference's own
// arithmetic.  No internal node is ever passed, so the approximation branch (bvh.cpp:178-246) is not involved.
template <int D>
int bvh_leaf_forces(const double* raw, size_t n, int max_bodies, double* out) {
    auto b = wrap<D>(raw, n);
    BVH<D> tree(b, max_bodies);
    std::vector<const BVHNode<D>*> leaves;
    bvh_collect<D>(tree.root.get(), leaves);
    std::vector<Vector<D>> f(n);
    for (size_t i = 0; i < n; ++i) {
        Vector<D> sum;
        for (const BVHNode<D>* leaf : leaves) sum += tree.calculate_force(b[i], leaf);
        f[i] = sum;
    }
    unwrap<D>(f, out);
    return 0;
}
}  // namespace

extern "C" {

int ref_bvh_leaves(const double* bodies, size_t n, int D, int max_bodies, uint32_t* leaf_offsets, uint32_t* leaf_bodies, size_t* n_leaves) {
    if (D == 2) return bvh_leaves<2>(bodies, n, max_bodies, leaf_offsets, leaf_bodies, n_leaves);
    if (D == 3) return bvh_leaves<3>(bodies, n, max_bodies, leaf_offsets, leaf_bodies, n_leaves);
    return -1;
}
int ref_bvh_leaf_forces(const double* bodies, size_t n, int D, int max_bodies, double* out) {
    if (D == 2) return bvh_leaf_forces<2>(bodies, n, max_bodies, out);
    if (D == 3) return bvh_leaf_forces<3>(bodies, n, max_bodies, out);
    return -1;
}

int ref_sizeof_body(int D) { return D == 2 ? (int)sizeof(Body<2>) : (int)sizeof(Body<3>); }
double ref_G() { return G; }

// variant: 0 = brute_force_seq_n_body, 1 = brute_force_omp_n_body_1, 2 = brute_force_omp_n_body_2,
//          3 = brute_force_parlay_n_body_1, 4 = brute_force_parlay_n_body_2 (thread count: PARLAY_NUM_THREADS)
int ref_brute_force(int variant, const double* bodies, size_t n, int D, double* out) {
    if (D == 2) return forces<2>(variant, bodies, n, out);
    if (D == 3) return forces<3>(variant, bodies, n, out);
    return -1;
}

// Wall time of the reference solver call alone (what the reference's safely_execute times, utils.h:87-104).
int ref_time_brute_force(int variant, const double* bodies, size_t n, int D, double* seconds) {
    if (D == 2) return timed<2>(variant, bodies, n, seconds);
    if (D == 3) return timed<3>(variant, bodies, n, seconds);
    return -1;
}

int ref_parlay_num_workers() { return (int)parlay::num_workers(); }

// The reference's Barnes-Hut octree (octree.cpp) evaluated with theta = 0: the acceptance test
// `2*half_size/dist < theta` (octree.cpp:146) never holds, so every body is reached through its single-body
// leaf (octree.cpp:105-125) -- an all-pairs sum under the tree codes' attractive leaf law, in the tree's visiting
// order.  Pins law 1 of oracle_leaf_pair_forces.  (The methods.cpp wrappers ignore their theta argument and use the
// global BARNES_HUT_THETA, methods.cpp:228-233, hence the direct call on the public root.)
int ref_octree_direct_forces(const double* bodies, size_t n, int D, double* out) {
    if (D == 2) {
        auto b = wrap<2>(bodies, n);
        Octree<2> tree(b);
        std::vector<Vector<2>> f(n);
        for (size_t i = 0; i < n; ++i) f[i] = tree.root->calculate_force(b[i], 0.0);
        unwrap<2>(f, out);
        return 0;
    }
    if (D == 3) {
        auto b = wrap<3>(bodies, n);
        Octree<3> tree(b);
        std::vector<Vector<3>> f(n);
        for (size_t i = 0; i < n; ++i) f[i] = tree.root->calculate_force(b[i], 0.0);
        unwrap<3>(f, out);
        return 0;
    }
    return -1;
}

int ref_update_body_velocities(double* bodies, const double* f, size_t n, int D, double dt) {
    if (D == 2) {
        auto b = wrap<2>(bodies, n);
        std::vector<Vector<2>> fv(n);
        if (n) std::memcpy(static_cast<void*>(fv.data()), f, n * sizeof(Vector<2>));
        update_body_velocities<2>(b, fv, dt);
        if (n) std::memcpy(bodies, static_cast<const void*>(b.data()), n * sizeof(Body<2>));
        return 0;
    }
    if (D == 3) {
        auto b = wrap<3>(bodies, n);
        std::vector<Vector<3>> fv(n);
        if (n) std::memcpy(static_cast<void*>(fv.data()), f, n * sizeof(Vector<3>));
        update_body_velocities<3>(b, fv, dt);
        if (n) std::memcpy(bodies, static_cast<const void*>(b.data()), n * sizeof(Body<3>));
        return 0;
    }
    return -1;
}

int ref_update_body_positions(double* bodies, size_t n, int D, double dt) {
    if (D == 2) {
        auto b = wrap<2>(bodies, n);
        update_body_positions<2>(b, dt);
        if (n) std::memcpy(bodies, static_cast<const void*>(b.data()), n * sizeof(Body<2>));
        return 0;
    }
    if (D == 3) {
        auto b = wrap<3>(bodies, n);
        update_body_positions<3>(b, dt);
        if (n) std::memcpy(bodies, static_cast<const void*>(b.data()), n * sizeof(Body<3>));
        return 0;
    }
    return -1;
}

// Same distributions and draw order as the reference generator (utils.h:107-135), which cannot
// be called for fixtures because it seeds from std::random_device.  libstdc++'s own mt19937 and
// uniform_real_distribution are used here, so this also pins the oracle's restated generator.
int ref_generate_random_bodies_seeded(uint32_t seed, size_t n, int D, double* out) {
    std::mt19937 gen(seed);
    std::uniform_real_distribution<double> position_dist(1, 10000000.0);
    std::uniform_real_distribution<double> velocity_dist(-10.0, 10.0);
    std::uniform_real_distribution<double> mass_dist(1, 100000000.0);
    const size_t stride = 2 * (size_t)D + 1;
    for (size_t i = 0; i < n; ++i) {
        for (int d = 0; d < D; ++d) {
            out[i * stride + d] = position_dist(gen);
            out[i * stride + D + d] = velocity_dist(gen);
        }
        out[i * stride + 2 * D] = mass_dist(gen);
    }
    return 0;
}

double ref_compute_accuracy(const double* f, const double* r, size_t n, int D) {
    if (D == 2) {
        std::vector<Vector<2>> a(n), b(n);
        if (n) { std::memcpy(static_cast<void*>(a.data()), f, n * sizeof(Vector<2>)); std::memcpy(static_cast<void*>(b.data()), r, n * sizeof(Vector<2>)); }
        return compute_accuracy_omp<2>(a, b);
    }
    std::vector<Vector<3>> a(n), b(n);
    if (n) { std::memcpy(static_cast<void*>(a.data()), f, n * sizeof(Vector<3>)); std::memcpy(static_cast<void*>(b.data()), r, n * sizeof(Vector<3>)); }
    return compute_accuracy_omp<3>(a, b);
}

}  // extern "C"
