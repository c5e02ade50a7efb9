// leaf_plan_device.h -- the leaf plan of leaf_plan.h laid out ON THE DEVICE (round 5).
//
// Why: the reference builds a new tree inside every force evaluation (nbody-sim-new/methods.cpp:377-401: BVH<D> bvh(bodies, ...) per
// call; bvh.cpp:34-73), so a caller that mirrors it makes a new plan per evaluation.  plan_leaves() on eight host threads cost
// 3-9 ms at N = 2^20 against pair kernels of 0.1-0.3 ms (profiles/r4/leaf_host_costs.txt).  Here the caller's four CSR arrays go
// to the device as they are and everything the pair kernels follow is built there, by ~35 small kernels on one stream with no
// host read-back in between (every kernel is launched for an upper bound and reads its actual count from device memory); one
// 64-byte summary comes back at the end (status, counts).
//
// The result is the SAME plan, array for array, as plan_leaves() makes on the host -- same padded slots, same copy runs, same
// packed waves (windows sorted by the same keys), same workgroup pieces, same launch order -- so the forces are bit-identical
// whichever planner ran, and tests/device_plan_check.hip compares the arrays word for word.  The host planner stays: it is the
// comparator, the sanitizer builds' subject (tests/test_leaf_plan_cpu.py), and what small structures use (a few microseconds of
// host work beat 35 launches).
//
//   step                         kernels                                       host planner's counterpart (leaf_plan.h)
//   padded slots                 dp_sizes, scan                                unit_off
//   slot <-> body maps + checks  dp_slots (one lane per leaf body)             pslot_body, validate_csr's body checks
//   copy runs                    dp_runs<count>, scan, dp_runs<write>          ops / op_off / stream_units (one wave64 per leaf:
//                                                                              neighbours via ballots and shuffles)
//   classes, one-leaf workgroups dp_classify, radix pass (stable partition),   packable[k], blocks
//                                scan, dp_blocks
//   packed waves                 dp_pack (one 128-lane workgroup per window:   pack_windows
//                                bitonic sort of the window's keys in LDS)
//   launch order                 dp_item_keys, 2 radix passes, dp_deal         order_launch (longest first, a class's eighths to the XCDs)
#pragma once
#include <hip/hip_runtime.h>

#include "device_sort.h"
#include "leaf_plan.h"

namespace nbx_leaf_dev {
using namespace nbx_leaf;

constexpr uint32_t kNoSlot = 0xffffffffu;
// why a structure was refused (Summary::err keeps the smallest code seen: the order of validate_csr's checks)
enum : uint32_t { kErrNone = 0xffffffffu, kErrBodyRange = 1, kErrBodyTwice = 2, kErrListRange = 3, kErrStreamTooLong = 4, kErrTooManyRuns = 5 };
inline const char* error_text(uint32_t e) {
    switch (e) {
        case kErrBodyRange: return "leaf_bodies entry out of range";
        case kErrBodyTwice: return "a body may belong to at most one leaf";
        case kErrListRange: return "list_sources entry out of range";
        case kErrStreamTooLong: return "a leaf's source list names more than 2^32 bodies";
        case kErrTooManyRuns: return "source lists too long";
        default: return "the structure could not be laid out";
    }
}

struct Summary {            // device words; the host sets n_leaves and err before the first kernel and reads all of it back once
    uint32_t err;           // kErrNone, or the smallest kErr* seen
    uint32_t n_leaves;
    uint32_t nonempty;
    uint32_t pslots;
    uint32_t n_ops;
    uint32_t n_blocks;
    uint32_t n_packs;
    uint32_t n_subs;
    uint32_t waves;
    uint32_t pack_small;
    uint32_t longest_pack;  // longest trips among the packed waves
    uint32_t longest_block; // longest duration key among the one-leaf workgroups
    uint32_t n_items;       // n_packs + n_blocks
    uint32_t pad_[3];
};
static_assert(sizeof(Summary) == 64, "one cache line, copied back whole");

// Upper bounds known to the host before anything runs (from n, n_leaves and the two totals of the offset arrays)
struct Bounds {
    size_t n, n_leaves, slots, n_list;
    size_t pslots_max() const { return (slots + n_leaves + 1) & ~(size_t)1; }   // a leaf pads by at most one slot
    size_t blocks_max() const { return n_leaves + slots / 64 + 1; }
    size_t packs_max() const { return n_leaves / 4 + 8; }                       // >= 4 leaves to a wave in every class, + one short wave per class
    size_t items_max() const { return blocks_max() + packs_max(); }
};

// Everything the device planner touches, as offsets into one device allocation (the plan's arena).  The first group is the plan
// itself (what the pair kernels and the gather / scatter kernels read); the second is the caller's arrays and scratch.
struct Layout {
    size_t xp, sums, pslot_body, body_slot, ops, blocks, subs, packs, max_mass, summary;
    size_t leaf_offsets, leaf_bodies, list_offsets, list_sources;
    size_t unit_off, op_off, stream_units, groups, blk_base, cls_key, cls_key2, leaf_id, order, hist, tile_sums;
    size_t item_key, item_key2, item_val, item_val2, blocks_tmp, packs_tmp;
    size_t total;
};
inline Layout make_layout(const Bounds& b, int dim) {
    Layout L{};
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) / 256 * 256 + 256; return o; };
    const size_t P = b.pslots_max(), nl = b.n_leaves, items = b.items_max();
    L.xp = take((P + 2) * 16);
    L.sums = take((size_t)dim * P * sizeof(double));
    L.pslot_body = take(P * 4);
    L.body_slot = take(b.n * 4);
    L.ops = take(b.n_list * sizeof(CopyOp));
    L.blocks = take(b.blocks_max() * sizeof(LeafBlock));
    L.subs = take(nl * sizeof(PackSub));
    L.packs = take(b.packs_max() * sizeof(PackBlock));
    L.max_mass = take(4);
    L.summary = take(sizeof(Summary));
    L.leaf_offsets = take((nl + 1) * 4);
    L.leaf_bodies = take(b.slots * 4);
    L.list_offsets = take((nl + 1) * 4);
    L.list_sources = take(b.n_list * 4);
    L.unit_off = take((nl + 2) * 4);
    L.op_off = take((nl + 2) * 4);
    L.stream_units = take(nl * 4);
    L.groups = take(nl * 4);
    L.blk_base = take((nl + 2) * 4);
    L.cls_key = take(nl * 4);
    L.cls_key2 = take(nl * 4);
    L.leaf_id = take(nl * 4);
    L.order = take(nl * 4);
    const size_t sort_cap = nl > items ? nl : items;
    L.hist = take(nbx_sort::radix_temp_bytes((unsigned)sort_cap));
    L.tile_sums = take(((size_t)nbx_sort::scan_tiles((unsigned)(nl + 1)) + 2) * 4);
    L.item_key = take(items * 4);
    L.item_key2 = take(items * 4);
    L.item_val = take(items * 4);
    L.item_val2 = take(items * 4);
    L.blocks_tmp = take(b.blocks_max() * sizeof(LeafBlock));
    L.packs_tmp = take(b.packs_max() * sizeof(PackBlock));
    L.total = at;
    return L;
}

namespace {

__device__ __forceinline__ void report(Summary* S, uint32_t code) { atomicMin(&S->err, code); }

__device__ __forceinline__ uint32_t pack_lanes_of(int k) { return k == 0 ? kPackLanes[0] : k == 1 ? kPackLanes[1] : k == 2 ? kPackLanes[2] : k == 3 ? kPackLanes[3] : kPackLanes[4]; }
__device__ __forceinline__ uint32_t pack_groups_of(int k) { return k == 0 ? kPackGroups[0] : k == 1 ? kPackGroups[1] : k == 2 ? kPackGroups[2] : k == 3 ? kPackGroups[3] : kPackGroups[4]; }
__device__ __forceinline__ int size_class_of(uint32_t c) { return c <= 2u ? 0 : c <= 4u ? 1 : c <= 6u ? 2 : c <= 8u ? 3 : 4; }

// padded[l] = leaf l's slots rounded up to whole pairs; the number of non-empty leaves
__global__ __launch_bounds__(256) void dp_sizes_kernel(const uint32_t* __restrict__ lo, uint32_t nl, uint32_t* __restrict__ padded, Summary* S) {
    const uint32_t l = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c = l < nl ? lo[l + 1] - lo[l] : 0u;
    if (l < nl) padded[l] = (c + 1u) & ~1u;
    const unsigned long long some = __ballot(c != 0u);
    if ((threadIdx.x & 63u) == 0u && some) atomicAdd(&S->nonempty, (uint32_t)__popcll(some));
}

// one lane per leaf body: its padded slot, both maps, the range and uniqueness checks
__global__ __launch_bounds__(256) void dp_slots_kernel(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ lb, const uint32_t* __restrict__ unit_off,
                                                       uint32_t nl, uint32_t slots, uint32_t n, uint32_t* __restrict__ pslot_body,
                                                       uint32_t* __restrict__ body_slot, Summary* S) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= slots) return;
    uint32_t a = 0, b = nl;                 // the last leaf whose first slot is <= s (lo[nl] = slots > s; empty leaves in between are passed over)
    while (a < b) {
        const uint32_t mid = a + (b - a + 1u) / 2u;
        if (lo[mid] <= s) a = mid; else b = mid - 1u;
    }
    const uint32_t l = a, first = lo[l], c = lo[l + 1] - first, k = s - first;
    const uint32_t p = unit_off[l] + k;
    const uint32_t body = lb[s];
    pslot_body[p] = body < n ? body : kNoSlot;
    if (k == c - 1u && (c & 1u)) pslot_body[p + 1u] = kNoSlot;
    if (body >= n) { report(S, kErrBodyRange); return; }
    if (atomicExch(&body_slot[body], p) != kNoSlot) report(S, kErrBodyTwice);
}

__device__ __forceinline__ unsigned long long wave_excl_scan_u64(unsigned long long v, unsigned lane, unsigned long long* total) {
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d);
        if (lane >= (unsigned)d) incl += up;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

// Copy runs of one leaf per wave64 (leaf_plan.h lay_out): list entry e starts a run unless its leaf's units follow the previous
// non-empty entry's.  WRITE = false: op_cnt[l] = runs, stream_units[l]; WRITE = true: the runs themselves at ops[op_off[l] ...].
template <bool WRITE>
__global__ __launch_bounds__(256) void dp_runs_kernel(const uint32_t* __restrict__ list_off, const uint32_t* __restrict__ list_src,
                                                      const uint32_t* __restrict__ unit_off, uint32_t nl, uint32_t* __restrict__ op_cnt_or_off,
                                                      uint32_t* __restrict__ stream_units, CopyOp* __restrict__ ops, Summary* S) {
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t leaf = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (leaf >= nl) return;
    if (WRITE && S->err != kErrNone) return;          // the counts of a refused structure are not to be trusted as offsets
    const uint32_t e0 = list_off[leaf], e1 = list_off[leaf + 1];
    const uint32_t op_base = WRITE ? op_cnt_or_off[leaf] : 0u;
    unsigned long long stream = 0;                    // wave-uniform: units before this chunk
    uint32_t starts = 0, prev_tail = 0;               // runs before this chunk; unit behind the last non-empty entry before this chunk
    bool have_prev = false;
    for (uint32_t c = e0; c < e1; c += 64u) {
        const uint32_t e = c + lane;
        const bool active = e < e1;
        uint32_t s = active ? list_src[e] : 0u;
        bool ok = active;
        if (active && s >= nl) { report(S, kErrListRange); ok = false; s = 0u; }
        const uint32_t first = unit_off[s];
        const uint32_t len = ok ? unit_off[s + 1] - first : 0u;
        const bool nonempty = len != 0u;
        const unsigned long long mask = __ballot(nonempty);
        const unsigned long long below = mask & ((1ull << lane) - 1ull);
        const uint32_t tail = first + len;
        const int pl = below ? 63 - __builtin_clzll(below) : 0;
        const uint32_t tail_below = (uint32_t)__shfl((int)tail, pl);
        const bool hasprev = below ? true : have_prev;
        const uint32_t ptail = below ? tail_below : prev_tail;
        const bool start = nonempty && (!hasprev || ptail != first);
        const unsigned long long smask = __ballot(start);
        unsigned long long chunk_total = 0;
        const unsigned long long pos = stream + wave_excl_scan_u64((unsigned long long)len, lane, &chunk_total);
        if (WRITE && start) {
            const uint32_t r = starts + (uint32_t)__popcll(smask & ((1ull << lane) - 1ull));
            ops[op_base + r].base = first - (uint32_t)pos;
            if (r > 0u) ops[op_base + r - 1u].end = (uint32_t)pos;
        }
        if (mask) {
            const int hl = 63 - __builtin_clzll(mask);
            prev_tail = (uint32_t)__shfl((int)tail, hl);
            have_prev = true;
        }
        stream += chunk_total;
        starts += (uint32_t)__popcll(smask);
    }
    if (stream > 0xfffff000ull) { if (lane == 0u) report(S, kErrStreamTooLong); stream = 0; }
    if (lane == 0u) {
        if (WRITE) { if (starts) ops[op_base + starts - 1u].end = (uint32_t)stream; }
        else { op_cnt_or_off[leaf] = starts; stream_units[leaf] = (uint32_t)stream; }
    }
}

// size class of every leaf (sort key of the stable partition) and the one-leaf workgroups it needs; the launch-wide decisions
__global__ __launch_bounds__(256) void dp_classify_kernel(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ op_off, const uint32_t* __restrict__ unit_off,
                                                          uint32_t nl, uint32_t slots, int allow_pack, Summary* S, uint32_t* __restrict__ cls_key,
                                                          uint32_t* __restrict__ leaf_id, uint32_t* __restrict__ groups) {
    const uint32_t l = blockIdx.x * 256u + threadIdx.x;
    if (l >= nl) return;
    const uint32_t nonempty = S->nonempty, units = unit_off[nl];
    const bool pack_small = allow_pack && nonempty && (unsigned long long)slots <= (unsigned long long)kPackMeanLeaf * nonempty && units < (1u << 28) - 2u;
    const uint32_t waves = (nonempty && slots / nonempty <= (uint32_t)kSmallLeaf) ? 1u : (uint32_t)kMaxWaves;
    const uint32_t per_group = 64u * waves;
    const uint32_t c = lo[l + 1] - lo[l], opn = op_off[l + 1] - op_off[l];
    uint32_t cls = 6u, g = 0u;
    if (c) {
        if (pack_small && c <= (uint32_t)kPackMaxTargets && opn <= (uint32_t)kPackMaxOps) cls = (uint32_t)size_class_of(c);
        else { cls = 5u; g = (c + per_group - 1u) / per_group; }
    }
    cls_key[l] = cls;
    leaf_id[l] = l;
    groups[l] = g;
    if (l == 0u) {
        S->waves = waves; S->pack_small = pack_small ? 1u : 0u; S->pslots = units; S->n_ops = op_off[nl];
        if (op_off[nl] > 0xfffffff0u) report(S, kErrTooManyRuns);
    }
}

struct PackBases { uint32_t cls[kPackClasses + 1], sub[kPackClasses + 1], pack[kPackClasses + 1], win[kPackClasses + 1]; };
// where each class's leaves, waves and windows go: known from the class sizes alone (hist[digit * tiles] after the partition's scan)
__device__ __forceinline__ PackBases pack_bases(const uint32_t* __restrict__ hist, uint32_t tiles) {
    PackBases B;
    B.sub[0] = B.pack[0] = B.win[0] = 0u;
#pragma unroll
    for (int k = 0; k <= kPackClasses; ++k) B.cls[k] = hist[(size_t)k * tiles];
#pragma unroll
    for (int k = 0; k < kPackClasses; ++k) {
        const uint32_t cnt = B.cls[k + 1] - B.cls[k], per_wave = 64u / pack_lanes_of(k), per_win = (uint32_t)kPackWindowWaves * per_wave;
        B.sub[k + 1] = B.sub[k] + cnt;
        B.pack[k + 1] = B.pack[k] + (cnt + per_wave - 1u) / per_wave;
        B.win[k + 1] = B.win[k] + (cnt + per_win - 1u) / per_win;
    }
    return B;
}

static_assert(kPackWindowWaves * (size_t)kPackMaxSubs <= 128, "dp_pack_kernel sorts a window's keys with 128 lanes");
// One workgroup of 128 lanes per window of a class (leaf_plan.h pack_windows): the window's leaves sorted longest stream first (leaf
// order among equals), cut into waves.
__global__ __launch_bounds__(128) void dp_pack_kernel(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ op_off, const uint32_t* __restrict__ unit_off,
                                                      const uint32_t* __restrict__ stream_units, const uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ hist, uint32_t hist_tiles, PackSub* __restrict__ subs,
                                                      PackBlock* __restrict__ packs_tmp, Summary* S) {
    __shared__ unsigned long long key[128];
    const PackBases B = pack_bases(hist, hist_tiles);
    if (blockIdx.x == 0u && threadIdx.x == 0u) { S->n_packs = B.pack[kPackClasses]; S->n_subs = B.sub[kPackClasses]; }
    const uint32_t wi = blockIdx.x;
    if (wi >= B.win[kPackClasses] || S->err != kErrNone) return;
    int k = 0;
    while (wi >= B.win[k + 1]) ++k;
    const uint32_t w = pack_lanes_of(k), per_wave = 64u / w, P = pack_groups_of(k), per_win = (uint32_t)kPackWindowWaves * per_wave;
    const uint32_t cnt = B.cls[k + 1] - B.cls[k];
    const uint32_t i = (wi - B.win[k]) * per_win;
    const uint32_t m = (i + per_win < cnt ? i + per_win : cnt) - i;       // leaves of this window
    const uint32_t j = threadIdx.x;
    unsigned long long mine = ~0ull;
    if (j < m) {
        const uint32_t l = order[B.cls[k] + i + j];
        mine = ((unsigned long long)(0xffffffffu - stream_units[l]) << 32) | l;
    }
    key[j] = mine;
    __syncthreads();
    for (unsigned size = 2; size <= 128u; size <<= 1) {                   // bitonic sort, ascending
        for (unsigned stride = size >> 1; stride > 0u; stride >>= 1) {
            const unsigned partner = j ^ stride;
            if (partner > j) {
                const unsigned long long a = key[j], b = key[partner];
                const bool up = (j & size) == 0u;
                if ((a > b) == up) { key[j] = b; key[partner] = a; }
            }
            __syncthreads();
        }
    }
    if (j >= m) return;
    const uint32_t l = (uint32_t)key[j];
    subs[B.sub[k] + i + j] = PackSub{op_off[l], op_off[l + 1] - op_off[l], unit_off[l], lo[l + 1] - lo[l]};
    if (j % per_wave == 0u) {
        PackBlock b{};
        b.sub_lo = B.sub[k] + i + j;
        b.n_sub = m - j < per_wave ? m - j : per_wave;
        b.w = w; b.inv_w = (65536u + w - 1u) / w; b.P = P; b.shape = (uint32_t)k;
        const uint32_t longest = 0xffffffffu - (uint32_t)(key[j] >> 32);   // sorted longest first: the wave's first leaf
        const uint32_t per_group = ((longest >> 1) + P - 1u) / P;
        b.trips = (per_group + (uint32_t)kPackPairsPerTrip - 1u) / (uint32_t)kPackPairsPerTrip * (uint32_t)kPackPairsPerTrip;
        b.longest = longest;
        packs_tmp[B.pack[k] + (i + j) / per_wave] = b;
        atomicMax(&S->longest_pack, b.trips);
    }
}

// the one-leaf workgroups of every leaf that is not packed, in leaf order (leaf_plan.h, the loop behind pack_windows)
__global__ __launch_bounds__(256) void dp_blocks_kernel(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ op_off, const uint32_t* __restrict__ unit_off,
                                                        const uint32_t* __restrict__ stream_units, const uint32_t* __restrict__ groups,
                                                        const uint32_t* __restrict__ blk_base, uint32_t nl, PieceCut cuts, LeafBlock* __restrict__ blocks_tmp, Summary* S) {
    const uint32_t l = blockIdx.x * 256u + threadIdx.x;
    if (l == 0u) { S->n_blocks = blk_base[nl]; S->n_items = S->n_packs + blk_base[nl]; }
    if (l >= nl || S->err != kErrNone) return;
    const uint32_t g = groups[l];
    if (!g) return;
    const uint32_t waves = S->waves, c = lo[l + 1] - lo[l];
    uint32_t f = unit_off[l];
    auto lanes = [](uint32_t cc) -> uint32_t { const uint32_t p = cc ? 64u / cc : (uint32_t)kMaxLanesPerTarget; return p > (uint32_t)kMaxLanesPerTarget ? (uint32_t)kMaxLanesPerTarget : p; };
    uint32_t longest = 0;
    for (uint32_t gi = 0; gi < g; ++gi) {
        const uint32_t share = c / g + (gi < c % g ? 1u : 0u);
        const uint32_t c1 = waves == 2u ? cuts.c1[share] : share;
        LeafBlock b;
        b.op_lo = op_off[l];
        b.op_n = op_off[l + 1] - op_off[l];
        b.pad_[0] = b.pad_[1] = 0;
        b.piece[0] = Piece{f, c1};
        b.piece[1] = Piece{f + c1, share - c1};
        const uint32_t slower = lanes(c1) < lanes(share - c1) ? lanes(c1) : lanes(share - c1);
        b.pad_[0] = (b.op_n ? stream_units[l] : 0u) / slower;
        if (b.pad_[0] > longest) longest = b.pad_[0];
        blocks_tmp[blk_base[l] + gi] = b;
        f += share;
    }
    atomicMax(&S->longest_block, longest);
}

// sort key of every launch item (packed waves first, then one-leaf workgroups): leaf_plan.h order_class, with the item's kind on top
__global__ __launch_bounds__(256) void dp_item_keys_kernel(const PackBlock* __restrict__ packs_tmp, const LeafBlock* __restrict__ blocks_tmp, const Summary* S,
                                                           uint32_t* __restrict__ item_key, uint32_t* __restrict__ item_val) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_packs = S->n_packs, n_items = S->n_items;
    if (i >= n_items || S->err != kErrNone) return;
    uint32_t key;
    if (i < n_packs) {
        const uint32_t longest = S->longest_pack > 1u ? S->longest_pack : 1u;
        key = order_class(packs_tmp[i].trips, longest, packs_tmp[i].shape);
    } else {
        const uint32_t longest = S->longest_block > 1u ? S->longest_block : 1u;
        key = (kOrderLevels * kOrderShapes) | order_class(blocks_tmp[i - n_packs].pad_[0], longest, 0u);
    }
    item_key[i] = key;
    item_val[i] = i;
}
static_assert(kOrderLevels * kOrderShapes * 2u <= 65536u, "the item keys are sorted in two 8-bit passes");

// the sorted items into launch order: a class's blocks dealt to the XCDs in eighths (leaf_plan.h order_launch)
__global__ __launch_bounds__(256) void dp_deal_kernel(const uint32_t* __restrict__ item_key, const uint32_t* __restrict__ item_val, const PackBlock* __restrict__ packs_tmp,
                                                      const LeafBlock* __restrict__ blocks_tmp, const Summary* S, PackBlock* __restrict__ packs,
                                                      LeafBlock* __restrict__ blocks) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_packs = S->n_packs, n_items = S->n_items;
    if (i >= n_items || S->err != kErrNone) return;
    const bool is_pack = i < n_packs;
    const uint32_t r0 = is_pack ? 0u : n_packs, rn = is_pack ? n_packs : n_items - n_packs, li = i - r0;
    uint32_t src = li;
    if (rn >= 2u && rn >= (uint32_t)kXcdOrderFrom) {
        const uint32_t k = item_key[i];
        uint32_t a = 0, b = rn;                                  // [b0, b1): the class of this index within its region
        while (a < b) { const uint32_t mid = a + (b - a) / 2u; if (item_key[r0 + mid] < k) a = mid + 1u; else b = mid; }
        const uint32_t b0 = a;
        b = rn;
        while (a < b) { const uint32_t mid = a + (b - a) / 2u; if (item_key[r0 + mid] <= k) a = mid + 1u; else b = mid; }
        const uint32_t b1 = a;
        if (b1 - b0 >= 2u * kXcds) {
            const uint32_t x = li % kXcds;
            uint32_t run_lo = b0, first_x = 0;
            for (uint32_t y = 0; y <= x; ++y) {
                const uint32_t first = b0 + (y + kXcds - b0 % kXcds) % kXcds;
                if (y == x) { first_x = first; break; }
                run_lo += first < b1 ? (b1 - first + kXcds - 1u) / kXcds : 0u;
            }
            src = run_lo + (li - first_x) / kXcds;
        }
    }
    const uint32_t v = item_val[r0 + src];
    if (is_pack) packs[li] = packs_tmp[v];
    else blocks[li] = blocks_tmp[v - n_packs];
}

}  // namespace

struct DevicePlan {      // device pointers into the arena
    float4* xp; double* sums; uint32_t* pslot_body; uint32_t* body_slot; CopyOp* ops; LeafBlock* blocks; PackSub* subs; PackBlock* packs;
    uint32_t* max_mass; Summary* summary;
};
inline DevicePlan plan_pointers(char* arena, const Layout& L) {
    DevicePlan d;
    d.xp = reinterpret_cast<float4*>(arena + L.xp); d.sums = reinterpret_cast<double*>(arena + L.sums);
    d.pslot_body = reinterpret_cast<uint32_t*>(arena + L.pslot_body); d.body_slot = reinterpret_cast<uint32_t*>(arena + L.body_slot);
    d.ops = reinterpret_cast<CopyOp*>(arena + L.ops); d.blocks = reinterpret_cast<LeafBlock*>(arena + L.blocks);
    d.subs = reinterpret_cast<PackSub*>(arena + L.subs); d.packs = reinterpret_cast<PackBlock*>(arena + L.packs);
    d.max_mass = reinterpret_cast<uint32_t*>(arena + L.max_mass); d.summary = reinterpret_cast<Summary*>(arena + L.summary);
    return d;
}

// Queue the whole layout on `stream`: the four host arrays in (offsets already checked by the caller: start at 0, non-decreasing --
// the copies' lengths come from them), every kernel, the summary out into *summary_host (pinned or pageable).  Returns without
// waiting; the caller synchronises the stream before it reads the summary.  n_leaves >= 1.
inline hipError_t enqueue_device_plan(const Bounds& b, int dim, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, const uint32_t* list_offsets,
                                      const uint32_t* list_sources, bool allow_pack, char* arena, const Layout& L, hipStream_t stream, Summary* summary_host) {
    using namespace nbx_sort;
    auto at = [&](size_t off) { return reinterpret_cast<uint32_t*>(arena + off); };
    const uint32_t nl = (uint32_t)b.n_leaves, slots = (uint32_t)b.slots, n = (uint32_t)b.n;
    const DevicePlan d = plan_pointers(arena, L);
    hipError_t e;
#define DP_TRY(expr) do { if ((e = (expr)) != hipSuccess) return e; } while (0)
    Summary init{};
    init.err = kErrNone; init.n_leaves = nl;
    *summary_host = init;   // staged from here: the copy below may be asynchronous
    DP_TRY(hipMemcpyAsync(d.summary, summary_host, sizeof(Summary), hipMemcpyHostToDevice, stream));
    DP_TRY(hipMemcpyAsync(at(L.leaf_offsets), leaf_offsets, (b.n_leaves + 1) * 4, hipMemcpyHostToDevice, stream));
    DP_TRY(hipMemcpyAsync(at(L.list_offsets), list_offsets, (b.n_leaves + 1) * 4, hipMemcpyHostToDevice, stream));
    if (b.slots) DP_TRY(hipMemcpyAsync(at(L.leaf_bodies), leaf_bodies, b.slots * 4, hipMemcpyHostToDevice, stream));
    if (b.n_list) DP_TRY(hipMemcpyAsync(at(L.list_sources), list_sources, b.n_list * 4, hipMemcpyHostToDevice, stream));
    if (b.n) DP_TRY(hipMemsetAsync(d.body_slot, 0xff, b.n * 4, stream));
    const uint32_t* count_leaves = &d.summary->n_leaves;
    const dim3 blk(256);
    const unsigned leaf_grid = (nl + 255u) / 256u;
    // padded slots
    hipLaunchKernelGGL(dp_sizes_kernel, dim3(leaf_grid), blk, 0, stream, at(L.leaf_offsets), nl, at(L.groups) /* scratch */, d.summary);
    DP_TRY(exclusive_scan(at(L.groups), at(L.unit_off), count_leaves, nl, at(L.tile_sums), stream));
    if (slots)
        hipLaunchKernelGGL(dp_slots_kernel, dim3((slots + 255u) / 256u), blk, 0, stream, at(L.leaf_offsets), at(L.leaf_bodies), at(L.unit_off), nl, slots, n,
                           d.pslot_body, d.body_slot, d.summary);
    // copy runs: count, offsets, write
    hipLaunchKernelGGL(dp_runs_kernel<false>, dim3((nl + 3u) / 4u), blk, 0, stream, at(L.list_offsets), at(L.list_sources), at(L.unit_off), nl, at(L.cls_key) /* scratch: counts */,
                       at(L.stream_units), d.ops, d.summary);
    DP_TRY(exclusive_scan(at(L.cls_key), at(L.op_off), count_leaves, nl, at(L.tile_sums), stream));
    hipLaunchKernelGGL(dp_runs_kernel<true>, dim3((nl + 3u) / 4u), blk, 0, stream, at(L.list_offsets), at(L.list_sources), at(L.unit_off), nl, at(L.op_off),
                       at(L.stream_units), d.ops, d.summary);
    // classes (stable partition by one radix pass), one-leaf workgroups
    hipLaunchKernelGGL(dp_classify_kernel, dim3(leaf_grid), blk, 0, stream, at(L.leaf_offsets), at(L.op_off), at(L.unit_off), nl, slots, allow_pack ? 1 : 0, d.summary,
                       at(L.cls_key), at(L.leaf_id), at(L.groups));
    DP_TRY(radix_pass(at(L.cls_key), at(L.leaf_id), at(L.cls_key2), at(L.order), count_leaves, nl, 0, at(L.hist), stream));
    const uint32_t hist_tiles = sort_tiles(nl);
    DP_TRY(exclusive_scan(at(L.groups), at(L.blk_base), count_leaves, nl, at(L.tile_sums), stream));
    const unsigned windows_max = (unsigned)(b.n_leaves / 32 + kPackClasses + 1);   // >= 32 leaves to a full window in every class
    hipLaunchKernelGGL(dp_pack_kernel, dim3(windows_max), dim3(128), 0, stream, at(L.leaf_offsets), at(L.op_off), at(L.unit_off), at(L.stream_units), at(L.order),
                       at(L.hist), hist_tiles, d.subs, reinterpret_cast<PackBlock*>(arena + L.packs_tmp), d.summary);
    static const PieceCut cuts = best_cuts();
    hipLaunchKernelGGL(dp_blocks_kernel, dim3(leaf_grid), blk, 0, stream, at(L.leaf_offsets), at(L.op_off), at(L.unit_off), at(L.stream_units), at(L.groups),
                       at(L.blk_base), nl, cuts, reinterpret_cast<LeafBlock*>(arena + L.blocks_tmp), d.summary);
    // launch order
    const unsigned items_max = (unsigned)b.items_max();
    const unsigned item_grid = (items_max + 255u) / 256u;
    hipLaunchKernelGGL(dp_item_keys_kernel, dim3(item_grid), blk, 0, stream, reinterpret_cast<const PackBlock*>(arena + L.packs_tmp),
                       reinterpret_cast<const LeafBlock*>(arena + L.blocks_tmp), d.summary, at(L.item_key), at(L.item_val));
    const uint32_t* count_items = &d.summary->n_items;
    DP_TRY(radix_pass(at(L.item_key), at(L.item_val), at(L.item_key2), at(L.item_val2), count_items, items_max, 0, at(L.hist), stream));
    DP_TRY(radix_pass(at(L.item_key2), at(L.item_val2), at(L.item_key), at(L.item_val), count_items, items_max, 8, at(L.hist), stream));
    hipLaunchKernelGGL(dp_deal_kernel, dim3(item_grid), blk, 0, stream, at(L.item_key), at(L.item_val), reinterpret_cast<const PackBlock*>(arena + L.packs_tmp),
                       reinterpret_cast<const LeafBlock*>(arena + L.blocks_tmp), d.summary, d.packs, d.blocks);
    DP_TRY(hipGetLastError());
    DP_TRY(hipMemcpyAsync(summary_host, d.summary, sizeof(Summary), hipMemcpyDeviceToHost, stream));
#undef DP_TRY
    return hipSuccess;
}

}  // namespace nbx_leaf_dev
