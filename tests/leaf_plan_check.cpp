// Test driver for csrc/leaf_plan.h (built and run by tests/test_leaf_plan_cpu.py; no GPU): reads the caller's CSR arrays from raw
// uint32 files <dir>/{leaf_offsets,leaf_bodies,list_offsets,list_sources}.u32, lays the launch out, writes the plan's arrays back.
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
static void store(const std::string& path, const void* p, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || (bytes && fwrite(p, 1, bytes, f) != bytes)) exit(2);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    const std::string d = std::string(argv[1]) + "/";
    const std::vector<uint32_t> lo = load(d + "leaf_offsets.u32"), lb = load(d + "leaf_bodies.u32"), so = load(d + "list_offsets.u32"),
                                ss = load(d + "list_sources.u32");
    nbx_leaf::LeafPlan plan;
    const char* why = nbx_leaf::plan_leaves(lo.data(), lb.data(), lo.size() - 1, so.data(), ss.data(), plan);
    if (why) { printf("refused: %s\n", why); return 1; }
    static_assert(sizeof(nbx_leaf::LeafBlock) == 32 && sizeof(nbx_leaf::CopyOp) == 8, "written as raw words");
    store(d + "unit_off.u32", plan.unit_off.data(), plan.unit_off.size() * 4);
    store(d + "pslot_body.u32", plan.pslot_body.data(), plan.pslot_body.size() * 4);
    store(d + "ops.u32", plan.ops.data(), plan.ops.size() * 8);
    store(d + "op_off.u32", plan.op_off.data(), plan.op_off.size() * 4);
    store(d + "blocks.u32", plan.blocks.data(), plan.blocks.size() * 32);
    static_assert(sizeof(nbx_leaf::PackSub) == 16 && sizeof(nbx_leaf::PackBlock) == 32, "written as raw words");
    store(d + "pack_subs.u32", plan.pack_subs.data(), plan.pack_subs.size() * 16);
    store(d + "pack_blocks.u32", plan.pack_blocks.data(), plan.pack_blocks.size() * 32);
    printf("waves %d blocks %zu ops %zu pslots %zu packs %zu\n", plan.waves, plan.blocks.size(), plan.ops.size(), plan.pslots(), plan.pack_blocks.size());
    return 0;
}
