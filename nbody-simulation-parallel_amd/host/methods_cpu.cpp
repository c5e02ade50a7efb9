// methods_cpu.cpp -- fp64 CPU brute force + kick/drift for the harness (see methods_cpu.h).
// Arithmetic follows nbody-sim-new/methods.cpp:21-37 pair by pair so the rows are comparable with
// the reference's; structure (one shared pair kernel, three sweep drivers) is this repository's own.
#include "methods_cpu.h"

#include <omp.h>

#include <cstddef>

#include "utils_hip.h"

namespace {

// Force that body j exerts along (p_j - p_i) in the reference's convention; false if the pair is
// skipped (r^2 < 1e-10, methods.cpp:24).  magnitude = ((G*mi)*mj) / (r^2 * r); direction by division.
template <int D>
inline bool pair_term(const Body<D>& bi, const Body<D>& bj, Vector<D>& out) {
    const Vector<D> sep = bj.position - bi.position;
    const double r2 = sep.magnitude_squared();
    if (r2 < 1e-10) return false;
    const double r = std::sqrt(r2);
    const double scale = G * bi.mass * bj.mass / (r2 * r);
    out = sep.normalized() * scale;
    return true;
}

}  // namespace

// methods.cpp:7-42: each unordered pair once, +f on j and -f on i
template <int D>
std::vector<Vector<D>> brute_force_seq_n_body(const std::vector<Body<D>>& bodies) {
    const std::size_t n = bodies.size();
    std::vector<Vector<D>> total(n);
    Vector<D> f;
    for (std::size_t i = 0; i < n; ++i)
        for (std::size_t j = i + 1; j < n; ++j)
            if (pair_term<D>(bodies[i], bodies[j], f)) {
                total[j] += f;
                total[i] -= f;
            }
    return total;
}

// methods.cpp:45-95: symmetric sweep under `omp for`, one private n-vector per thread, serial merge
template <int D>
std::vector<Vector<D>> brute_force_omp_n_body_1(const std::vector<Body<D>>& bodies) {
    const std::size_t n = bodies.size();
    const int threads = omp_get_max_threads();
    std::vector<std::vector<Vector<D>>> scratch(static_cast<std::size_t>(threads), std::vector<Vector<D>>(n));
#pragma omp parallel
    {
        std::vector<Vector<D>>& mine = scratch[static_cast<std::size_t>(omp_get_thread_num())];
        Vector<D> f;
#pragma omp for
        for (std::size_t i = 0; i < n; ++i)
            for (std::size_t j = i + 1; j < n; ++j)
                if (pair_term<D>(bodies[i], bodies[j], f)) {
                    mine[j] += f;
                    mine[i] -= f;
                }
    }
    std::vector<Vector<D>> total(n);
    for (const auto& part : scratch)
        for (std::size_t i = 0; i < n; ++i) total[i] += part[i];
    return total;
}

// methods.cpp:98-136: every ordered pair, each target owns its sum
template <int D>
std::vector<Vector<D>> brute_force_omp_n_body_2(const std::vector<Body<D>>& bodies) {
    const std::size_t n = bodies.size();
    std::vector<Vector<D>> total(n);
#pragma omp parallel for
    for (std::size_t i = 0; i < n; ++i) {
        Vector<D> f, sum;
        for (std::size_t j = 0; j < n; ++j)
            if (j != i && pair_term<D>(bodies[i], bodies[j], f)) sum -= f;
        total[i] = sum;
    }
    return total;
}

// methods.cpp:425-438: v += (F / m) * dt
template <int D>
void update_body_velocities(std::vector<Body<D>>& bodies, const std::vector<Vector<D>>& forces, double dt) {
    const std::size_t n = bodies.size();
#pragma omp parallel for
    for (std::size_t i = 0; i < n; ++i) bodies[i].velocity += forces[i] / bodies[i].mass * dt;
}

// methods.cpp:440-450: x += v * dt
template <int D>
void update_body_positions(std::vector<Body<D>>& bodies, double dt) {
    const std::size_t n = bodies.size();
#pragma omp parallel for
    for (std::size_t i = 0; i < n; ++i) bodies[i].position += bodies[i].velocity * dt;
}

#define NBODY_INSTANTIATE(D)                                                                              \
    template std::vector<Vector<D>> brute_force_seq_n_body<D>(const std::vector<Body<D>>&);               \
    template std::vector<Vector<D>> brute_force_omp_n_body_1<D>(const std::vector<Body<D>>&);             \
    template std::vector<Vector<D>> brute_force_omp_n_body_2<D>(const std::vector<Body<D>>&);             \
    template void update_body_velocities<D>(std::vector<Body<D>>&, const std::vector<Vector<D>>&, double); \
    template void update_body_positions<D>(std::vector<Body<D>>&, double);
NBODY_INSTANTIATE(2)
NBODY_INSTANTIATE(3)
