// force_launch.hip -- host-side launcher of one force evaluation: looks up the variant table of
// force_kernel.hip, checks launch shapes on the host, sizes the grids,
// and for fast variants runs the close-set pipeline (classify when positions changed -> fast kernel
// whose extra workgroups evaluate the close set -> scatter), all asynchronous on the caller's stream with no host read-back.
#include "nbx_internal.h"

#include <cxxabi.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace nbx {
namespace {

struct Table {
    std::vector<KernelVariant> v;
    int def = 0, def_exact = 0, def_two_rcp = 0;
    CloseKernels ck;
    Table() {
        int n = 0;
        const KernelVariant* a = kernel_variants(&n);
        for (int i = 0; i < n; ++i) v.push_back(a[i]);
        for (size_t i = 0; i < v.size(); ++i) {
            if (std::strcmp(v[i].name, NBX_DEFAULT_VARIANT) == 0) def = (int)i;
            if (std::strcmp(v[i].name, NBX_DEFAULT_EXACT_VARIANT) == 0) def_exact = (int)i;
        }
        for (size_t i = v.size(); i-- > 0;)
            if (v[i].fast && !v[i].needs_extent) def_two_rcp = (int)i;
        ck = close_kernels();
    }
};
const Table& table() {
    static const Table t;
    return t;
}
bool valid(int v) { return v >= 0 && v < (int)table().v.size(); }

// the build of variant V that one force launch runs: Newtonian / softened / mixed-mode (spread sums) / plain
void (*pick_kernel(const KernelVariant& V, int dim, int law, bool soft, bool qsum))(KArgs) {
    return law ? ((dim == 3) ? V.newton3 : V.newton2)
               : soft ? ((dim == 3) ? V.soft3 : V.soft2)
               : (qsum && !V.aux) ? ((dim == 3) ? V.qs3 : V.qs2) : ((dim == 3) ? V.k3 : V.k2);
}

}  // namespace

int num_variants() { return (int)table().v.size(); }
const char* variant_name(int v) { return valid(v) ? table().v[v].name : "?"; }
int variant_tpl(int v) { return valid(v) ? table().v[v].tpl : 1; }
int variant_is_fast(int v) { return valid(v) ? table().v[v].fast : 0; }
int variant_max_tiles_per_slice(int v) { return valid(v) ? table().v[v].max_tiles_per_slice : 0; }
int variant_needs_extent(int v) { return valid(v) ? table().v[v].needs_extent : 0; }
int variant_has_law_builds(int v) { return valid(v) && table().v[v].soft2 && table().v[v].soft3 && table().v[v].newton2 && table().v[v].newton3; }
int variant_planes(int v) { return valid(v) ? table().v[v].planes : 1; }
int variant_has_qsum(int v) { return valid(v) && table().v[v].qs2 && table().v[v].qs3; }
int variant_writes_aux(int v) { return valid(v) ? table().v[v].aux : 0; }
int variant_has_clock_stamps(int v) { return valid(v) ? table().v[v].stamps : 0; }
int variant_kernel_symbol(int v, int dim, int law, int soft, int qsum, char* buf, size_t len) {
    if (!valid(v) || (dim != 2 && dim != 3) || !buf || len == 0) return -1;
    void (*k)(KArgs) = pick_kernel(table().v[v], dim, law, soft != 0, qsum != 0);
    if (!k) return -1;
    // the name the code object was registered under (no device needed), demangled the way rocprofv3 prints it
    const char* mangled = hipKernelNameRefByPtr(reinterpret_cast<const void*>(k), nullptr);
    (void)hipGetLastError();
    if (!mangled) return -1;
    int status = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    std::snprintf(buf, len, "%s", (status == 0 && dem) ? dem : mangled);
    std::free(dem);
    return 0;
}
int default_fast_two_rcp_variant() { return table().def_two_rcp; }
int variant_by_name(const char* name) {
    for (int i = 0; i < num_variants(); ++i)
        if (std::strcmp(table().v[i].name, name) == 0) return i;
    return -1;
}
int default_variant() { return table().def; }
int default_exact_variant() { return table().def_exact; }

hipError_t launch_accel(int dim, const AccelLaunch& L, hipStream_t stream) {
    if ((dim != 2 && dim != 3) || L.pad == 0 || L.pad % kPadQuantum != 0 || L.splits < 1 || L.vchunks < 1 ||
        L.count > L.pad)
        return hipErrorInvalidValue;
    const int v = valid(L.variant) ? L.variant : default_variant();
    const KernelVariant& V = table().v[v];
    KArgs a;
    a.pos_all = L.pos_all;
    a.mass_all = L.mass_all;
    a.acc = L.acc;
    a.pad = L.pad;
    a.count = L.count;
    a.tiles_per_chunk = L.pad / kTile;
    a.total_tiles = (unsigned)L.vchunks * a.tiles_per_chunk;
    if (L.splits % V.planes != 0) return hipErrorInvalidValue;   // acc planes = source slices x planes per slice
    const unsigned slices = (unsigned)(L.splits / V.planes);
    a.tiles_per_split = (a.total_tiles + slices - 1) / slices;
    a.tgt_chunk = L.tgt_chunk;
    a.chunk_first = L.chunk_first;
    a.chunk_skip = L.chunk_skip;
    a.accumulate = L.accumulate;
    a.splits = L.splits;
    a.grid_slices = (int)slices;
    a.qsum = L.qsum;
    a.strict_list = nullptr; a.strict_acc = nullptr; a.strict_cap = 0; a.strict_slices = 0; a.strict_budget = 0; a.refine_c2 = 0.0;
    a.clk = nullptr;
    a.cand_list = L.cand_list;
    a.cand_pos = L.cand_pos;
    a.bad_list = L.bad_list;
    a.bad_flag = L.bad_flag;
    a.counters = L.counters;
    a.close_acc = L.close_acc;
    a.src_cand_pos = L.src_cand_pos;
    a.src_stride = (unsigned)L.n_chunks * L.pad;
    a.n_total = (unsigned)L.n_total;
    a.shard_len = (unsigned)L.shard_len;
    a.eps2 = L.eps2;
    a.clk = V.stamps ? L.clk : nullptr;
    const bool soft = L.eps2 > 0.0f;
    if (soft && !(V.fast && V.soft2 && V.soft3)) return hipErrorInvalidValue;   // softened law: fast kernels only
    if (L.law != 0 && !(soft && V.newton2 && V.newton3)) return hipErrorInvalidValue;   // Newtonian law: softened only
    if (L.qsum && !V.aux && (soft || !(V.qs2 && V.qs3))) return hipErrorInvalidValue;   // mixed mode: reference law, fast kernels
    if (V.aux && !L.qsum) return hipErrorInvalidValue;
    // host-side shape checks: every target block and every source tile lies inside its chunk
    const unsigned tgt_per_block = 256u * (unsigned)V.tpl;
    if (L.pad % tgt_per_block != 0) return hipErrorInvalidValue;
    if (V.fast && !soft && (!L.cand_list || !L.cand_pos || !L.bad_list || !L.bad_flag || !L.counters || !L.close_acc || !L.src_cand_pos)) return hipErrorInvalidValue;
    if (V.fast && !soft && (L.n_chunks < 1 || L.n_total > ((size_t)1 << 31) || (size_t)L.n_chunks * L.pad > 0xffffffffull ||
                   L.chunk_first < 0 || L.chunk_first + L.vchunks + (L.chunk_skip != INT_MAX ? 1 : 0) > L.n_chunks))
        return hipErrorInvalidValue;
    if (V.max_tiles_per_slice > 0 && a.tiles_per_split > (unsigned)V.max_tiles_per_slice) return hipErrorInvalidValue;

    a.close_blocks = (V.fast && !soft) ? (unsigned)kCloseBlocksX : 0u;
    hipError_t e = hipSuccess;
    dim3 block(256, 1, 1);
    const int di = dim - 2;
    if (V.fast && !soft) {
        // Candidate targets: a property of the own chunk's positions (rebuilt after every position update).
        // Bad targets: a property of the own chunk AND of the pass's source chunks, which for a sharded ALL /
        // REMOTE pass are rewritten behind the library's back by the exchange -- rebuilt for every such launch.
        const bool build_targets = !(L.tgt_cand_valid && *L.tgt_cand_valid);
        const bool build_bad = build_targets || !(L.cacheable && L.bad_list_pass && *L.bad_list_pass == L.pass);
        if (build_targets) {   // one memset for all three counters; classify_close_kernel clears its targets' flags itself
            if ((e = hipMemsetAsync(L.counters, 0, 3 * sizeof(unsigned), stream)) != hipSuccess) return e;
            if (L.count) {
                hipLaunchKernelGGL(table().ck.classify[di], dim3((L.count + 255u) / 256u, 1, 1), block, 0, stream, a);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            }
            if (L.tgt_cand_valid) *L.tgt_cand_valid = 1;
            if (L.bad_list_pass) *L.bad_list_pass = -1;
        } else if (build_bad) {
            if ((e = hipMemsetAsync(L.counters + 1, 0, 2 * sizeof(unsigned), stream)) != hipSuccess) return e;
            if ((e = hipMemsetAsync(L.bad_flag, 0, (size_t)L.pad * sizeof(unsigned), stream)) != hipSuccess) return e;
        }
        if (build_bad) {
            if (L.count) {
                hipLaunchKernelGGL(table().ck.classify_src[di], dim3(L.pad / 256u, (unsigned)L.vchunks, 1), block, 0, stream, a);
                if ((e = hipGetLastError()) != hipSuccess) return e;
                if (L.hash.keys) {   // most bodies are candidates: sorted cells instead of candidates x candidates
                    if ((e = hash_refine(dim, a, L.hash, stream)) != hipSuccess) return e;
                } else {
                    hipLaunchKernelGGL(table().ck.refine[di], dim3(1024, 1, 1), block, 0, stream, a);
                    if ((e = hipGetLastError()) != hipSuccess) return e;
                }
            }
            if (L.bad_list_pass) *L.bad_list_pass = L.cacheable ? L.pass : -1;
        }
    }
    if (L.lists_only) return hipSuccess;
    dim3 grid(L.pad / tgt_per_block + a.close_blocks, slices, 1);
    if (a.clk) {   // close-set workgroups leave their slots zero: the reader skips them
        if ((e = hipMemsetAsync(a.clk, 0, (size_t)grid.x * grid.y * 2 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
        if (L.clk_slots) *L.clk_slots = grid.x * grid.y;
    }
    if (L.ev_start && (e = hipEventRecord(L.ev_start, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(pick_kernel(V, dim, L.law, soft, L.qsum != nullptr), grid, block, 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (L.ev_stop && (e = hipEventRecord(L.ev_stop, stream)) != hipSuccess) return e;
    if (!V.fast || soft) return hipSuccess;
    hipLaunchKernelGGL(table().ck.scatter[di], dim3(64, 1, 1), block, 0, stream, a);
    return hipGetLastError();
}

namespace {
// Kernel-side view of the refinement: the evaluation's planes and spread sums, and ALL chunks as the strict pass's sources.
hipError_t refine_args(int dim, const RefineLaunch& R, KArgs& a) {
    const AccelLaunch& L = R.base;
    if ((dim != 2 && dim != 3) || L.pad == 0 || L.pad % kPadQuantum != 0 || L.splits < 1 || L.n_chunks < 1 || L.count > L.pad ||
        !L.acc || !L.qsum || !L.counters || !L.bad_flag || R.grid_slices < 1 || L.splits % R.grid_slices != 0)
        return hipErrorInvalidValue;
    a = KArgs{};
    a.pos_all = L.pos_all; a.mass_all = L.mass_all; a.acc = L.acc; a.pad = L.pad; a.count = L.count;
    a.tiles_per_chunk = L.pad / kTile;
    a.total_tiles = (unsigned)L.n_chunks * a.tiles_per_chunk;
    a.tgt_chunk = L.tgt_chunk; a.chunk_first = 0; a.chunk_skip = INT_MAX; a.accumulate = 0;
    a.splits = L.splits;
    a.grid_slices = R.grid_slices;
    a.bad_flag = L.bad_flag; a.counters = L.counters; a.qsum = L.qsum;
    a.strict_list = R.strict_list; a.strict_acc = R.strict_acc; a.strict_cap = R.strict_cap; a.strict_slices = R.strict_slices;
    a.strict_budget = R.strict_budget;
    a.refine_c2 = R.c2;
    return hipSuccess;
}
}  // namespace

hipError_t launch_refine(int dim, const RefineLaunch& R, hipStream_t stream) {
    KArgs a;
    hipError_t e = refine_args(dim, R, a);
    if (e != hipSuccess) return e;
    // the list has room for every target of the shard, and the sums for the whole shard in one slice: whatever the selection
    // lists is re-evaluated (the device picks slices x stride within the budget, force_kernel.hip strict_layout)
    if (!R.strict_list || !R.strict_acc || R.strict_cap < R.base.pad || R.strict_slices < 1 || R.strict_slices > 256 ||
        (unsigned)R.strict_slices > a.total_tiles || R.strict_budget < (unsigned long long)dim * R.base.pad)
        return hipErrorInvalidValue;
    a.tiles_per_split = 0;   // the fp64 pass derives its own from the list's length
    if (a.count == 0) return hipSuccess;
    const int di = dim - 2;
    const dim3 block(256, 1, 1);
    if ((e = hipMemsetAsync(a.counters + 3, 0, sizeof(unsigned), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(table().ck.refine_select[di], dim3((a.count + 63u) / 64u, 1, 1), block, 0, stream, a);   // four lanes per target
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // listed targets x strict slices; a workgroup whose list block does not exist returns at once
    // grid = (source slices, rows of list blocks): a row's workgroups take list blocks row, row + 32, ...; a short list (the usual
    // case) needs one to four rows, the idle rows cost a scalar load per workgroup; 256 x 32 workgroups of one target per lane fill the
    // chip for a long one.  Slices along x: see accel_f64_kernel
    const unsigned list_blocks = (a.count + 255u) / 256u;
    hipLaunchKernelGGL(table().ck.strict_list[di], dim3((unsigned)R.strict_slices, list_blocks < 32u ? list_blocks : 32u, 1), block, 0, stream, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(table().ck.refine_fold[di], dim3(512, 1, 1), block, 0, stream, a);   // 2,048 waves, one (target, component) each per turn
    return hipGetLastError();
}

hipError_t launch_potential(int dim, const AccelLaunch& L, hipStream_t stream) {
    if ((dim != 2 && dim != 3) || L.pad == 0 || L.pad % kPadQuantum != 0 || L.splits < 1 || L.vchunks < 1)
        return hipErrorInvalidValue;
    KArgs a = {};
    a.pos_all = L.pos_all;
    a.mass_all = L.mass_all;
    a.acc = L.acc;
    a.pad = L.pad;
    a.count = L.count;
    a.tiles_per_chunk = L.pad / kTile;
    a.total_tiles = (unsigned)L.vchunks * a.tiles_per_chunk;
    a.tiles_per_split = (a.total_tiles + (unsigned)L.splits - 1) / (unsigned)L.splits;
    a.tgt_chunk = L.tgt_chunk;
    a.chunk_first = L.chunk_first;
    a.chunk_skip = L.chunk_skip;
    a.splits = L.splits;
    a.eps2 = L.eps2;
    if (L.law != 0 && !(L.eps2 > 0.0f)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(L.law ? table().ck.potential_newton[dim - 2] : L.eps2 > 0.0f ? table().ck.potential_soft[dim - 2] : table().ck.potential[dim - 2], dim3(L.pad / 512u, (unsigned)L.splits, 1), dim3(256, 1, 1), 0, stream, a);
    return hipGetLastError();
}

}  // namespace nbx
