// methods_hip.cpp -- thin C++ shim from the reference-shaped templates onto the C ABI.
#include "methods_hip.h"

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "nbody_hip.h"

namespace {
thread_local HipRunInfo g_info;
int g_device = -1;
std::vector<int> g_devices;  // more than one entry: sharded over a node

// RAII over the node handle so an exception cannot leak device memory
struct NodeHandle {
    nbx_node* h = nullptr;
    ~NodeHandle() { nbx_node_destroy(h); }
};

int device_ordinal() {
    if (g_device >= 0) return g_device;
    if (const char* e = std::getenv("NBODY_HIP_DEVICE")) return std::atoi(e);
    return 0;
}

[[noreturn]] void raise(const char* where, int status) {
    std::string msg = std::string(where) + ": " + nbx_strerror(status);
    const char* detail = nbx_last_error_detail();
    if (detail && *detail) msg += std::string(" -- ") + detail;
    throw std::runtime_error(msg);
}
}  // namespace

const HipRunInfo& last_hip_run_info() { return g_info; }
bool warm_up_hip() {
    bool ok = true;
    if (g_devices.size() > 1) {
        for (int d : g_devices) ok = (nbx_warmup(d) == NBX_OK) && ok;
        // and the node layer: the first node over these devices builds the RCCL communicators (by far the most expensive
        // call of the library) and proves the exchange on a poisoned buffer; the library keeps the set for the nodes the
        // timed calls build
        constexpr size_t n = 8192;
        std::vector<double> b(n * 7), f(n * 3);
        for (size_t i = 0; i < n; ++i) {
            double* r = &b[i * 7];
            r[0] = 1.0e5 + 37.0 * (double)(i % 97); r[1] = 2.0e5 + 11.0 * (double)(i / 97); r[2] = 3.0e5 + (double)i;
            r[3] = r[4] = r[5] = 0.0; r[6] = 1.0;
        }
        NodeHandle node;
        int rc = nbx_node_create(&node.h, (int)g_devices.size(), g_devices.data(), 3, n, NBX_EXCHANGE_AUTO);
        if (!rc) rc = nbx_node_upload_bodies(node.h, b.data(), 7 * sizeof(double));
        if (!rc) rc = nbx_node_compute_forces(node.h, NBX_REFERENCE_G, f.data());
        ok = ok && rc == NBX_OK;
    } else {
        ok = nbx_warmup(device_ordinal()) == NBX_OK;
    }
    return ok;
}
void release_hip_caches() { (void)nbx_release_cached(); }
int hip_device_count() {
    int n = 0;
    return nbx_device_count(&n) == NBX_OK ? n : 0;
}
void set_hip_refine(double rel_tolerance) {
    const int rc = nbx_set_default_refine(rel_tolerance, 0.0);
    if (rc != NBX_OK) raise("set_hip_refine", rc);
}
double hip_refine_tolerance() {
    double tol = 0.0;
    (void)nbx_get_default_refine(&tol, nullptr);
    return tol;
}
void set_hip_device(int device) { g_device = device; }
void set_hip_devices(const std::vector<int>& devices) { g_devices = devices; }

template <int D>
std::vector<Vector<D>> brute_force_hip_n_body(const std::vector<Body<D>>& bodies) {
    std::vector<Vector<D>> forces(bodies.size());
    g_info = HipRunInfo{};
    if (g_devices.size() > 1) {
        NodeHandle node;
        int rc = nbx_node_create(&node.h, (int)g_devices.size(), g_devices.data(), D, bodies.size(), NBX_EXCHANGE_AUTO);
        if (!rc) rc = nbx_node_upload_bodies(node.h, bodies.data(), sizeof(Body<D>));
        if (!rc) rc = nbx_node_compute_forces(node.h, NBX_REFERENCE_G, reinterpret_cast<double*>(forces.data()));
        int launches = 0;
        float mean = 0.f;
        if (!rc) rc = nbx_node_kernel_time(node.h, &mean, &launches);
        if (rc != NBX_OK) raise("brute_force_hip_n_body (node)", rc);
        g_info.kernel_ms = mean * (float)launches / (float)g_devices.size();  // per-rank share: ranks run concurrently
        g_info.device = g_devices[0];
        if (nbx_node_refine_stats(node.h, &g_info.refine_selected, &g_info.refine_refined) == NBX_OK)
            g_info.refine_tolerance = hip_refine_tolerance();
        return forces;
    }
    g_info.device = device_ordinal();
    nbx_eval_info info;
    const int rc = nbx_brute_force_forces_ex(bodies.data(), bodies.size(), D, sizeof(Body<D>), NBX_REFERENCE_G,
                                             g_info.device, -1.0, reinterpret_cast<double*>(forces.data()), &info);
    if (rc != NBX_OK) raise("brute_force_hip_n_body", rc);
    g_info.kernel_ms = info.kernel_ms + info.refine_ms;   // the whole evaluation: the harness derives pairs/s from it
    g_info.refine_tolerance = info.refine_tolerance;
    g_info.refine_selected = info.refine_selected;
    g_info.refine_refined = info.refine_refined;
    return forces;
}

template <int D>
HipNodeReport describe_hip_node(const std::vector<Body<D>>& bodies) {
    HipNodeReport rep;
    std::vector<int> devs = g_devices.empty() ? std::vector<int>{device_ordinal()} : g_devices;
    NodeHandle node;
    int rc = nbx_node_create(&node.h, (int)devs.size(), devs.data(), D, bodies.size(), NBX_EXCHANGE_AUTO);
    if (!rc) rc = nbx_node_upload_bodies(node.h, bodies.data(), sizeof(Body<D>));
    int mode = NBX_EXCHANGE_PEER_COPY;
    if (!rc) rc = nbx_node_exchange_mode(node.h, &mode);
    rep.transport = devs.size() == 1 ? "none (one rank)" : mode == NBX_EXCHANGE_RCCL ? "rccl" : "peer copies";
    if (!rc) rc = nbx_node_verify_exchange(node.h, &rep.mismatching_values);
    if (!rc) rc = nbx_node_enable_timing(node.h, 1);
    std::vector<Vector<D>> forces(bodies.size());
    if (!rc) rc = nbx_node_compute_forces(node.h, NBX_REFERENCE_G, reinterpret_cast<double*>(forces.data()));
    for (int r = 0; r < (int)devs.size() && !rc; ++r) {
        HipNodeReport::Rank k{};
        k.rank = r;
        int hidden = 0;
        rc = nbx_node_pass_times(node.h, r, &k.device, &k.targets, &k.local_ms, &k.remote_ms, &k.exchange_ms, &hidden);
        k.exchange_hidden = hidden != 0;
        rep.ranks.push_back(k);
    }
    if (rc != NBX_OK) raise("describe_hip_node", rc);
    if (!rep.ranks.empty()) rep.checked_values_per_rank = (std::size_t)D * (bodies.size() - rep.ranks[0].targets);
    if (nbx_node_refine_stats(node.h, &rep.refine_selected, &rep.refine_refined) == NBX_OK) rep.refine_tolerance = hip_refine_tolerance();
    return rep;
}

template <int D>
std::vector<Vector<D>> brute_force_hip_single_gpu(const std::vector<Body<D>>& bodies, int device) {
    std::vector<Vector<D>> forces(bodies.size());
    const int rc = nbx_brute_force_forces(bodies.data(), bodies.size(), D, sizeof(Body<D>), NBX_REFERENCE_G, device,
                                          reinterpret_cast<double*>(forces.data()), nullptr);
    if (rc != NBX_OK) raise("brute_force_hip_single_gpu", rc);
    return forces;
}

template <int D>
double brute_force_hip_accuracy(const std::vector<Body<D>>& bodies, const std::vector<Vector<D>>& reference) {
    if (reference.size() != bodies.size()) throw std::runtime_error("brute_force_hip_accuracy: reference size differs from the bodies");
    struct Ctx { nbx_ctx* h = nullptr; ~Ctx() { nbx_ctx_destroy(h); } } c;
    double percent = 0.0;
    int rc = nbx_ctx_create(&c.h, device_ordinal(), D, bodies.size(), 1, 0);
    if (!rc) rc = nbx_ctx_upload_bodies(c.h, bodies.data(), sizeof(Body<D>));
    if (!rc) rc = nbx_ctx_compute_accel(c.h, NBX_SRC_ALL);
    if (!rc) rc = nbx_ctx_accuracy(c.h, NBX_REFERENCE_G, reinterpret_cast<const double*>(reference.data()), &percent);
    if (rc != NBX_OK) raise("brute_force_hip_accuracy", rc);
    return percent;
}

template <int D>
void leapfrog_hip_n_body(std::vector<Body<D>>& bodies, double dt, int nsteps) {
    g_info = HipRunInfo{};
    if (g_devices.size() > 1) {
        NodeHandle node;
        int rc = nbx_node_create(&node.h, (int)g_devices.size(), g_devices.data(), D, bodies.size(), NBX_EXCHANGE_AUTO);
        if (!rc) rc = nbx_node_upload_bodies(node.h, bodies.data(), sizeof(Body<D>));
        if (!rc) rc = nbx_node_step(node.h, NBX_REFERENCE_G, dt, nsteps);
        if (!rc) rc = nbx_node_synchronize(node.h);
        if (!rc) rc = nbx_node_download_bodies(node.h, bodies.data(), sizeof(Body<D>));
        int launches = 0;
        float mean = 0.f;
        if (!rc) rc = nbx_node_kernel_time(node.h, &mean, &launches);
        if (rc != NBX_OK) raise("leapfrog_hip_n_body (node)", rc);
        g_info.kernel_ms = mean * (float)launches / (float)g_devices.size();
        g_info.device = g_devices[0];
        return;
    }
    g_info.device = device_ordinal();
    const int rc = nbx_leapfrog(bodies.data(), bodies.size(), D, sizeof(Body<D>), NBX_REFERENCE_G, dt, nsteps,
                                g_info.device, &g_info.kernel_ms);
    if (rc != NBX_OK) raise("leapfrog_hip_n_body", rc);
}

template <int D>
HipSimulation<D>::HipSimulation(const std::vector<Body<D>>& bodies, double G, double softening, bool newton) : G_(G) {
    std::vector<int> devs = g_devices.empty() ? std::vector<int>{device_ordinal()} : g_devices;
    ranks_ = (int)devs.size();
    int rc = nbx_node_create(&node_, ranks_, devs.data(), D, bodies.size(), NBX_EXCHANGE_AUTO);
    if (!rc) rc = nbx_node_upload_bodies(node_, bodies.data(), sizeof(Body<D>));
    if (!rc && softening != 0.0) rc = nbx_node_set_softening(node_, softening);
    if (!rc && newton) rc = nbx_node_set_law(node_, NBX_FORCE_LAW_NEWTON);
    if (rc != NBX_OK) {
        nbx_node_destroy(node_);
        node_ = nullptr;
        raise("HipSimulation", rc);
    }
}
template <int D>
HipSimulation<D>::~HipSimulation() { nbx_node_destroy(node_); }
template <int D>
void HipSimulation<D>::step(double dt, int nsteps) {
    const int rc = nbx_node_step(node_, G_, dt, nsteps);
    if (rc != NBX_OK) raise("HipSimulation::step", rc);
}
template <int D>
void HipSimulation<D>::step_kdk(double dt, int nsteps) {
    const int rc = nbx_node_step_kdk(node_, G_, dt, nsteps);
    if (rc != NBX_OK) raise("HipSimulation::step_kdk", rc);
}
template <int D>
void HipSimulation<D>::energy(double& kinetic, double& potential) {
    const int rc = nbx_node_energy(node_, G_, &kinetic, &potential);
    if (rc != NBX_OK) raise("HipSimulation::energy", rc);
}
template <int D>
void HipSimulation<D>::download(std::vector<Body<D>>& bodies) {
    int rc = nbx_node_synchronize(node_);
    if (!rc) rc = nbx_node_download_bodies(node_, bodies.data(), sizeof(Body<D>));
    if (rc != NBX_OK) raise("HipSimulation::download", rc);
}
template <int D>
double HipSimulation<D>::force_kernel_seconds() {
    float mean = 0.f;
    int launches = 0;
    const int rc = nbx_node_kernel_time(node_, &mean, &launches);
    if (rc != NBX_OK) raise("HipSimulation::force_kernel_seconds", rc);
    return (double)mean * launches / ranks_ * 1e-3;
}
template class HipSimulation<2>;
template class HipSimulation<3>;

// explicit instantiations, like nbody-sim-new/methods.cpp:452-499 does for the CPU solvers
template std::vector<Vector<2>> brute_force_hip_n_body<2>(const std::vector<Body<2>>&);
template std::vector<Vector<3>> brute_force_hip_n_body<3>(const std::vector<Body<3>>&);
template HipNodeReport describe_hip_node<2>(const std::vector<Body<2>>&);
template HipNodeReport describe_hip_node<3>(const std::vector<Body<3>>&);
template std::vector<Vector<2>> brute_force_hip_single_gpu<2>(const std::vector<Body<2>>&, int);
template std::vector<Vector<3>> brute_force_hip_single_gpu<3>(const std::vector<Body<3>>&, int);
template double brute_force_hip_accuracy<2>(const std::vector<Body<2>>&, const std::vector<Vector<2>>&);
template double brute_force_hip_accuracy<3>(const std::vector<Body<3>>&, const std::vector<Vector<3>>&);
template void leapfrog_hip_n_body<2>(std::vector<Body<2>>&, double, int);
template void leapfrog_hip_n_body<3>(std::vector<Body<3>>&, double, int);
