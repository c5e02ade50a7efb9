// leaf_pairs_hip.cpp -- C++ shim of the leaf-pair direct sums onto the C ABI + the uniform leaf builder.
#include "leaf_pairs_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <unordered_map>

#include "nbody_hip.h"

namespace {
thread_local float g_leaf_ms = 0.0f;
}

float last_leaf_pair_kernel_ms() { return g_leaf_ms; }

template <int D>
std::vector<Vector<D>> leaf_pair_direct_forces_hip(const std::vector<Body<D>>& bodies, const LeafLists& L, LeafLaw law) {
    if (L.leaf_offsets.empty() || L.list_offsets.size() != L.leaf_offsets.size())
        throw std::runtime_error("leaf_pair_direct_forces_hip: leaf_offsets and list_offsets need n_leaves + 1 entries");
    std::vector<Vector<D>> forces(bodies.size());
    int device = 0;
    if (const char* e = std::getenv("NBODY_HIP_DEVICE")) device = std::atoi(e);
    g_leaf_ms = 0.0f;
    const int rc = nbx_leaf_pair_forces(bodies.data(), bodies.size(), D, sizeof(Body<D>), L.leaf_offsets.data(), L.leaf_bodies.data(),
                                        L.leaves(), L.list_offsets.data(), L.list_sources.data(), static_cast<int>(law), NBX_REFERENCE_G,
                                        device, reinterpret_cast<double*>(forces.data()), &g_leaf_ms);
    if (rc != NBX_OK) {
        std::string msg = std::string("leaf_pair_direct_forces_hip: ") + nbx_strerror(rc);
        const char* detail = nbx_last_error_detail();
        if (detail && *detail) msg += std::string(" -- ") + detail;
        throw std::runtime_error(msg);
    }
    return forces;
}

namespace {
[[noreturn]] void raise_leaf(const char* where, int rc) {
    std::string msg = std::string(where) + ": " + nbx_strerror(rc);
    const char* detail = nbx_last_error_detail();
    if (detail && *detail) msg += std::string(" -- ") + detail;
    throw std::runtime_error(msg);
}
int leaf_device() {
    if (const char* e = std::getenv("NBODY_HIP_DEVICE")) return std::atoi(e);
    return 0;
}
}  // namespace

template <int D>
LeafPairSimulationHip<D>::LeafPairSimulationHip(const std::vector<Body<D>>& bodies, const LeafLists& L) : n_(bodies.size()) {
    if (L.leaf_offsets.empty() || L.list_offsets.size() != L.leaf_offsets.size())
        throw std::runtime_error("LeafPairSimulationHip: leaf_offsets and list_offsets need n_leaves + 1 entries");
    const int device = leaf_device();
    int rc = nbx_leaf_plan_create(&plan_, device, D, n_, L.leaf_offsets.data(), L.leaf_bodies.data(), L.leaves(), L.list_offsets.data(),
                                  L.list_sources.data());
    if (!rc) rc = nbx_ctx_create(&ctx_, device, D, n_, 1, 0);
    if (!rc) rc = nbx_ctx_upload_bodies(ctx_, bodies.data(), sizeof(Body<D>));
    if (rc != NBX_OK) {
        nbx_leaf_plan_destroy(plan_);
        nbx_ctx_destroy(ctx_);
        plan_ = nullptr; ctx_ = nullptr;
        raise_leaf("LeafPairSimulationHip", rc);
    }
}
template <int D>
LeafPairSimulationHip<D>::~LeafPairSimulationHip() {
    nbx_leaf_plan_destroy(plan_);   // before the context: its last evaluation may still be queued on the context's stream
    nbx_ctx_destroy(ctx_);
}
template <int D>
std::vector<Vector<D>> LeafPairSimulationHip<D>::forces(LeafLaw law, double G) {
    std::vector<Vector<D>> f(n_);
    const int rc = nbx_leaf_plan_forces_ctx(plan_, ctx_, static_cast<int>(law), G, reinterpret_cast<double*>(f.data()), nullptr);
    if (rc != NBX_OK) raise_leaf("LeafPairSimulationHip::forces", rc);
    return f;
}
template <int D>
void LeafPairSimulationHip<D>::evaluate(LeafLaw law, double G) {
    int rc = nbx_leaf_plan_forces_ctx(plan_, ctx_, static_cast<int>(law), G, nullptr, nullptr);
    if (!rc) rc = nbx_ctx_synchronize(ctx_);
    if (rc != NBX_OK) raise_leaf("LeafPairSimulationHip::evaluate", rc);
}
template <int D>
void LeafPairSimulationHip<D>::step(LeafLaw law, double G, double dt, int nsteps) {
    const int rc = nbx_leaf_plan_step(plan_, ctx_, static_cast<int>(law), G, dt, nsteps);
    if (rc != NBX_OK) raise_leaf("LeafPairSimulationHip::step", rc);
}
template <int D>
void LeafPairSimulationHip<D>::synchronize() {
    const int rc = nbx_ctx_synchronize(ctx_);
    if (rc != NBX_OK) raise_leaf("LeafPairSimulationHip::synchronize", rc);
}
template <int D>
void LeafPairSimulationHip<D>::download(std::vector<Body<D>>& bodies) {
    if (bodies.size() != n_) throw std::runtime_error("LeafPairSimulationHip::download: body count differs");
    const int rc = nbx_ctx_download_bodies(ctx_, bodies.data(), sizeof(Body<D>));
    if (rc != NBX_OK) raise_leaf("LeafPairSimulationHip::download", rc);
}
template <int D>
float LeafPairSimulationHip<D>::single_launch_ms(LeafLaw law, double G) {
    float ms = 0.0f;
    const int rc = nbx_leaf_plan_forces_ctx(plan_, ctx_, static_cast<int>(law), G, nullptr, &ms);
    if (rc != NBX_OK) raise_leaf("LeafPairSimulationHip::single_launch_ms", rc);
    return ms;
}
template <int D>
float LeafPairSimulationHip<D>::back_to_back_ms(LeafLaw law, int reps) {
    float ms = 0.0f;
    const int rc = nbx_leaf_plan_time_kernel(plan_, static_cast<int>(law), reps, &ms);
    if (rc != NBX_OK) raise_leaf("LeafPairSimulationHip::back_to_back_ms", rc);
    return ms;
}
template class LeafPairSimulationHip<2>;
template class LeafPairSimulationHip<3>;

template <int D>
LeafLists build_uniform_leaves(const std::vector<Body<D>>& bodies, int depth) {
    LeafLists L;
    const std::size_t n = bodies.size();
    if (n == 0) return L;
    const long long g = 1LL << depth;
    Vector<D> lo = bodies[0].position, hi = bodies[0].position;
    for (const auto& b : bodies)
        for (int d = 0; d < D; ++d) { lo[d] = std::min(lo[d], b.position[d]); hi[d] = std::max(hi[d], b.position[d]); }
    double half = 0.0;
    Vector<D> centre;
    for (int d = 0; d < D; ++d) { centre[d] = (lo[d] + hi[d]) / 2.0; half = std::max(half, (hi[d] - lo[d]) / 2.0); }
    half = std::max(half * 1.01, 1e-300);
    auto cell_of = [&](const Body<D>& b, long long* c) {
        for (int d = 0; d < D; ++d) {
            long long k = (long long)std::floor((b.position[d] - (centre[d] - half)) / (2.0 * half) * (double)g);
            c[d] = std::min(std::max(k, 0LL), g - 1);
        }
    };
    auto key_of = [&](const long long* c) { long long k = 0; for (int d = 0; d < D; ++d) k = k * g + c[d]; return k; };
    std::vector<long long> key(n);
    for (std::size_t i = 0; i < n; ++i) { long long c[3]; cell_of(bodies[i], c); key[i] = key_of(c); }
    std::vector<std::uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](std::uint32_t a, std::uint32_t b) { return key[a] < key[b]; });
    L.leaf_bodies = order;
    L.leaf_offsets.clear();
    std::vector<long long> leaf_key;
    std::unordered_map<long long, std::uint32_t> index_of;
    for (std::size_t s = 0; s < n; ++s)
        if (s == 0 || key[order[s]] != key[order[s - 1]]) {
            index_of[key[order[s]]] = (std::uint32_t)leaf_key.size();
            leaf_key.push_back(key[order[s]]);
            L.leaf_offsets.push_back((std::uint32_t)s);
        }
    L.leaf_offsets.push_back((std::uint32_t)n);
    L.list_offsets.assign(1, 0u);
    int n_off = 1;
    for (int d = 0; d < D; ++d) n_off *= 3;
    for (std::size_t l = 0; l < leaf_key.size(); ++l) {
        long long c[3] = {0, 0, 0}, rest = leaf_key[l];
        for (int d = D - 1; d >= 0; --d) { c[d] = rest % g; rest /= g; }
        L.list_sources.push_back((std::uint32_t)l);  // own bodies first
        for (int o = 0; o < n_off; ++o) {            // offsets in the same order as leaves.py (first axis slowest)
            long long q[3], t = o;
            bool self = true, inside = true;
            for (int d = D - 1; d >= 0; --d) { const long long off = t % 3 - 1; t /= 3; q[d] = c[d] + off; self = self && off == 0; inside = inside && q[d] >= 0 && q[d] < g; }
            if (self || !inside) continue;
            const auto it = index_of.find(key_of(q));
            if (it != index_of.end()) L.list_sources.push_back(it->second);
        }
        L.list_offsets.push_back((std::uint32_t)L.list_sources.size());
    }
    return L;
}

template std::vector<Vector<2>> leaf_pair_direct_forces_hip<2>(const std::vector<Body<2>>&, const LeafLists&, LeafLaw);
template std::vector<Vector<3>> leaf_pair_direct_forces_hip<3>(const std::vector<Body<3>>&, const LeafLists&, LeafLaw);
template LeafLists build_uniform_leaves<2>(const std::vector<Body<2>>&, int);
template LeafLists build_uniform_leaves<3>(const std::vector<Body<3>>&, int);
