// Host-side cost of a leaf-pair launch's layout (csrc/leaf_plan.h plan_leaves), section by section and by thread count, on CSR
// arrays dumped by `python tools/time_leaf_pairs.py <N> <shape> --dump <dir>` (raw uint32 files).  No GPU involved.
//   g++ -O2 -std=c++17 -pthread tools/time_leaf_layout.cpp -o /tmp/time_leaf_layout && /tmp/time_leaf_layout <dir>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "../nbody-simulation-parallel_amd/csrc/leaf_plan.h"

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

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    const std::string d = std::string(argv[1]) + "/";
    const std::vector<uint32_t> lo = load(d + "leaf_offsets.u32"), lb = load(d + "leaf_bodies.u32"), so = load(d + "list_offsets.u32"),
                                ss = load(d + "list_sources.u32");
    printf("%zu leaves, %zu bodies in leaves, %zu list entries\n", lo.size() - 1, lb.size(), ss.size());
    nbx_leaf::LeafPlan plan;
    for (unsigned threads : {1u, 2u, 4u, 8u, 8u}) {
        double best = 1e30, sec[4] = {0, 0, 0, 0}, best_sec[4] = {0, 0, 0, 0};
        for (int r = 0; r < 3; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            if (const char* why = nbx_leaf::plan_leaves(lo.data(), lb.data(), lo.size() - 1, so.data(), ss.data(), plan, true, threads, sec)) { printf("refused: %s\n", why); return 1; }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) { best = ms; for (int k = 0; k < 4; ++k) best_sec[k] = sec[k]; }
        }
        printf("plan_leaves, %u thread(s): %7.3f ms  (padded slots %.3f, copy runs %.3f, workgroups %.3f, launch order %.3f)   %zu workgroups, %zu packed waves, %zu runs\n",
               threads, best, best_sec[0], best_sec[1], best_sec[2], best_sec[3], plan.blocks.size(), plan.pack_blocks.size(), plan.ops.size());
    }
    return 0;
}
