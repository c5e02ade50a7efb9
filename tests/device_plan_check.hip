// Test driver for csrc/leaf_plan_device.h (built with hipcc and run on the GPU box by tests/test_gpu_leaf_plan_device.py): reads the
// caller's CSR arrays from raw uint32 files <dir>/{leaf_offsets,leaf_bodies,list_offsets,list_sources}.u32, lays the launch out on
// the host (csrc/leaf_plan.h plan_leaves) and on the device (enqueue_device_plan), and compares every array of the two plans word
// for word.  Prints "identical ..." and returns 0, or says where they differ.  With a second argument N it also times N device
// layouts (stream time between two events, copies included); with "refuse" only the device planner runs and its verdict is printed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../nbody-simulation-parallel_amd/csrc/leaf_plan_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)

static std::vector<uint32_t> load(const std::string& path) {
    std::vector<uint32_t> v;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)bytes / 4);
    if (bytes && fread(v.data(), 4, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

template <class T>
static bool same(const char* what, const std::vector<T>& host, const void* dev, size_t count_dev) {
    if (host.size() != count_dev) { printf("DIFFERENT %s: host %zu entries, device %zu\n", what, host.size(), count_dev); return false; }
    std::vector<T> got(host.size());
    if (!host.empty()) CK(hipMemcpy(got.data(), dev, host.size() * sizeof(T), hipMemcpyDeviceToHost));
    const uint32_t* a = reinterpret_cast<const uint32_t*>(host.data());
    const uint32_t* b = reinterpret_cast<const uint32_t*>(got.data());
    const size_t words = host.size() * sizeof(T) / 4;
    for (size_t i = 0; i < words; ++i)
        if (a[i] != b[i]) {
            printf("DIFFERENT %s: word %zu of entry %zu (of %zu): host %u, device %u\n", what, i % (sizeof(T) / 4), i / (sizeof(T) / 4), host.size(), a[i], b[i]);
            return false;
        }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    using namespace nbx_leaf_dev;
    const std::string d = std::string(argv[1]) + "/";
    const bool device_only = argc > 2 && !strcmp(argv[2], "refuse");   // a structure the host planner must not be shown (it follows every index unchecked)
    const int reps = argc > 2 && !device_only ? atoi(argv[2]) : 0;
    const std::vector<uint32_t> lo = load(d + "leaf_offsets.u32"), lb = load(d + "leaf_bodies.u32"), so = load(d + "list_offsets.u32"),
                                ss = load(d + "list_sources.u32");
    const size_t n_leaves = lo.size() - 1;
    uint32_t n = 0;
    for (uint32_t b : lb) n = b + 1 > n ? b + 1 : n;
    if (FILE* f = fopen((d + "n_bodies.txt").c_str(), "r")) { unsigned v = 0; if (fscanf(f, "%u", &v) == 1) n = v; fclose(f); }
    nbx_leaf::LeafPlan host;
    const char* why = device_only ? nullptr : nbx_leaf::plan_leaves(lo.data(), lb.data(), n_leaves, so.data(), ss.data(), host);
    Bounds b{n, n_leaves, lo[n_leaves], so[n_leaves]};
    const Layout L = make_layout(b, 3);
    char* arena = nullptr;
    CK(hipMalloc((void**)&arena, L.total));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    Summary sum{};
    CK(enqueue_device_plan(b, 3, lo.data(), lb.data(), so.data(), ss.data(), true, arena, L, s, &sum));
    CK(hipStreamSynchronize(s));
    if (device_only) {
        printf("device: %s\n", sum.err != kErrNone ? error_text(sum.err) : "accepted");
        return 0;
    }
    if (why || sum.err != kErrNone) {
        printf("host: %s; device: %s\n", why ? why : "accepted", sum.err != kErrNone ? error_text(sum.err) : "accepted");
        return (why != nullptr) == (sum.err != kErrNone) ? 0 : 1;
    }
    const DevicePlan dp = plan_pointers(arena, L);
    bool ok = true;
    if (sum.waves != (uint32_t)host.waves) { printf("DIFFERENT waves: host %d device %u\n", host.waves, sum.waves); ok = false; }
    if (sum.pslots != host.pslots()) { printf("DIFFERENT pslots: host %zu device %u\n", host.pslots(), sum.pslots); ok = false; }
    ok = ok && same("unit_off", host.unit_off, arena + L.unit_off, n_leaves + 1);
    ok = ok && same("pslot_body", host.pslot_body, dp.pslot_body, sum.pslots);
    ok = ok && same("op_off", host.op_off, arena + L.op_off, n_leaves + 1);
    ok = ok && same("stream_units", host.stream_units, arena + L.stream_units, n_leaves);
    ok = ok && same("ops", host.ops, dp.ops, sum.n_ops);
    ok = ok && same("pack_subs", host.pack_subs, dp.subs, sum.n_subs);
    ok = ok && same("pack_blocks", host.pack_blocks, dp.packs, sum.n_packs);
    ok = ok && same("blocks", host.blocks, dp.blocks, sum.n_blocks);
    {   // body_slot: the inverse of pslot_body
        std::vector<uint32_t> body_slot(n, 0xffffffffu);
        for (size_t p = 0; p < host.pslot_body.size(); ++p) if (host.pslot_body[p] != 0xffffffffu) body_slot[host.pslot_body[p]] = (uint32_t)p;
        ok = ok && same("body_slot", body_slot, dp.body_slot, n);
    }
    if (!ok) return 1;
    printf("identical: %zu leaves, %u slots, %u runs, %u one-leaf workgroups (%u waves each), %u packed waves of %u leaves\n", n_leaves, sum.pslots, sum.n_ops, sum.n_blocks,
           sum.waves, sum.n_packs, sum.n_subs);
    if (reps > 0) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f, best_k = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, s));
            CK(enqueue_device_plan(b, 3, lo.data(), lb.data(), so.data(), ss.data(), true, arena, L, s, &sum));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // the kernels alone: the arrays are on the device already -- time from behind the copies (a second layout queued right behind)
        (void)best_k;
        printf("device layout, copies of the four arrays (%.1f MB) included: best of %d = %.3f ms of stream time\n",
               (double)(2 * (n_leaves + 1) + lo[n_leaves] + so[n_leaves]) * 4e-6, reps, best);
    }
    return 0;
}
