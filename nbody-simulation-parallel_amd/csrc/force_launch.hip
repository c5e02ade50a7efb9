// force_launch.hip -- host-side launcher of the force kernel: merges the variant tables of the two
// code-generation flavours of force_kernel.hip, checks launch shapes on the host, sizes the grid.
#include "nbx_internal.h"

#include <cstring>
#include <vector>

namespace nbx {
namespace {

struct Table {
    std::vector<KernelVariant> v;
    int def = 0;
    Table() {
        int n = 0;
        const KernelVariant* a = variants_scalar(&n);
        for (int i = 0; i < n; ++i) v.push_back(a[i]);
        const KernelVariant* b = variants_slp(&n);
        for (int i = 0; i < n; ++i) v.push_back(b[i]);
        for (size_t i = 0; i < v.size(); ++i)
            if (std::strcmp(v[i].name, NBX_DEFAULT_VARIANT) == 0) def = (int)i;
    }
};
const Table& table() {
    static const Table t;
    return t;
}

}  // namespace

int num_variants() { return (int)table().v.size(); }
const char* variant_name(int v) { return (v >= 0 && v < num_variants()) ? table().v[v].name : "?"; }
int variant_tpl(int v) { return (v >= 0 && v < num_variants()) ? table().v[v].tpl : 1; }
int default_variant() { return table().def; }

hipError_t launch_accel(int dim, const AccelLaunch& L, hipStream_t stream) {
    if ((dim != 2 && dim != 3) || L.pad == 0 || L.pad % kPadQuantum != 0 || L.splits < 1 || L.vchunks < 1)
        return hipErrorInvalidValue;
    const int v = (L.variant >= 0 && L.variant < num_variants()) ? L.variant : default_variant();
    const KernelVariant& V = table().v[v];
    KArgs a;
    a.pos_all = L.pos_all;
    a.mass_all = L.mass_all;
    a.acc = L.acc;
    a.pad = L.pad;
    a.tiles_per_chunk = L.pad / kTile;
    a.total_tiles = (unsigned)L.vchunks * a.tiles_per_chunk;
    a.tiles_per_split = (a.total_tiles + (unsigned)L.splits - 1) / (unsigned)L.splits;
    a.tgt_chunk = L.tgt_chunk;
    a.chunk_first = L.chunk_first;
    a.chunk_skip = L.chunk_skip;
    a.accumulate = L.accumulate;
    // host-side shape check: every target block and every source tile lies inside its chunk
    const unsigned tgt_per_block = 256u * (unsigned)V.tpl;
    if (L.pad % tgt_per_block != 0) return hipErrorInvalidValue;
    dim3 grid(L.pad / tgt_per_block, (unsigned)L.splits, 1), block(256, 1, 1);
    void (*k)(KArgs) = (dim == 3) ? V.k3 : V.k2;
    hipLaunchKernelGGL(k, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace nbx
