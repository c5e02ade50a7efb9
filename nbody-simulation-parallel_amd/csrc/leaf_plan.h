// leaf_plan.h -- host side of the leaf-pair path (csrc/leaf_pair_kernel.hip): everything the pair kernel follows, laid out once per
// call from the caller's CSR arrays.  Plain C++ (no HIP): tests/test_leaf_plan_cpu.py compiles it with g++ and checks its
// invariants without a GPU.  The comment at the top of leaf_pair_kernel.hip says what each piece is for.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#ifdef __HIPCC__
#define NBX_LEAF_HD __host__ __device__
#else
#define NBX_LEAF_HD
#endif

namespace nbx_leaf {

constexpr int kMaxWaves = 2;                   // wave64 per workgroup: each owns one piece of the leaf's targets, all stage the tiles
constexpr int kSmallLeaf = 20;                 // mean bodies per leaf up to which a workgroup is one wave (and one leaf piece)
constexpr int kMaxLanesPerTarget = 8;          // a piece of few targets gives each up to this many lanes (they split the sources)

struct Piece {
    uint32_t first;    // first target (padded slot)
    uint32_t count;    // <= 64; 0: this wave only helps staging
};
struct LeafBlock {     // one workgroup; read with scalar loads
    uint32_t op_lo, op_n;   // the leaf's copy ops
    uint32_t pad_[2];       // [0]: host-side sort key
    Piece piece[2];
};
static_assert(sizeof(LeafBlock) == 32, "LeafBlock is read with scalar loads");
// ---- packed small leaves (round 4) ----
// A leaf of a few bodies cannot fill a wave: at 4 bodies per leaf a workgroup's start-up, staging and closing reduction cost ten
// times its pair arithmetic (0.06 of the fp32 peak in round 3).  Leaves of up to kPackMaxTargets bodies whose list is at most
// kPackMaxOps copy runs are therefore PACKED: one wave64 takes K = 64 / w consecutive leaves, each on its own w = 4, 6, 8 or 16 lanes --
// its own copy runs, P lane groups (two targets per lane; the same P for every leaf of the wave -- leaves are packed with leaves of their size class
// -- so that every lane runs the same trip count), each lane group streaming its share of its leaf's source pairs straight from
// memory -- and all of them run the pair loop together.  Everything else (larger leaves, longer lists) keeps the one-leaf
// workgroups above.
constexpr int kPackMaxTargets = 16;
#ifndef NBX_PACK_MAX_OPS
#define NBX_PACK_MAX_OPS 32
#endif
constexpr int kPackMaxOps = NBX_PACK_MAX_OPS;
// Packing is used for structures whose leaves hold at most this many bodies ON AVERAGE.  Measured at N = 2^20
// (profiles/r4/leaf_pack_ab.txt, leaf_direct_ab.txt; same box, back-to-back launches): 4-body grid leaves 0.226 ms one workgroup
// per leaf, 0.213 ms packed and staged through LDS (this round's first packed kernel), 0.140 ms packed and streamed; median-split
// leaves of 8 bodies 0.146 -> 0.194 ms and of 16 bodies 0.209 -> 0.373 ms (a leaf of 8-16 bodies fills a wave of its own at
// 4-8 lanes per target and shares its staged sources among them; side by side every lane loads its own).
#ifndef NBX_PACK_MEAN_LEAF
#define NBX_PACK_MEAN_LEAF 8   /* A/B builds raise it to pack larger leaves too */
#endif
constexpr int kPackMeanLeaf = NBX_PACK_MEAN_LEAF;
#ifndef NBX_PACK_WINDOW_WAVES
#define NBX_PACK_WINDOW_WAVES 8
#endif
constexpr size_t kPackWindowWaves = NBX_PACK_WINDOW_WAVES;   // waves whose leaves are picked from one window of neighbouring leaves (A/B builds: 32)
// lanes per leaf and lane groups per leaf of the four size classes (1-2, 3-4, 5-8, 9-16 bodies; a lane holds two targets)
#ifndef NBX_PACK_TINY_LANES
#define NBX_PACK_TINY_LANES 4   /* A/B: 8 = eight leaves of 1-4 bodies to a wave, at 8 and 4 lane groups (this round's earlier layout) */
#endif
#ifndef NBX_PACK_SMALL_LANES
#define NBX_PACK_SMALL_LANES NBX_PACK_TINY_LANES
#endif
#ifndef NBX_PACK_SIX_LANES
#define NBX_PACK_SIX_LANES 6    /* A/B: 8 = leaves of 5-6 bodies share the 7-8-body class (8 lanes, two of them idle) */
#endif
#ifndef NBX_PACK_EIGHT_LANES
#define NBX_PACK_EIGHT_LANES 8   /* A/B: 4 = leaves of 7-8 bodies on 4 lanes, ONE lane group (16 leaves to a wave, twice the trips) */
#endif
constexpr int kPackClasses = 5;                 // 1-2, 3-4, 5-6, 7-8, 9-16 bodies
constexpr uint32_t kPackLanes[kPackClasses] = {NBX_PACK_TINY_LANES, NBX_PACK_SMALL_LANES, NBX_PACK_SIX_LANES, NBX_PACK_EIGHT_LANES, 16u};
constexpr uint32_t kPackGroups[kPackClasses] = {NBX_PACK_TINY_LANES, NBX_PACK_SMALL_LANES / 2u, 2u, NBX_PACK_EIGHT_LANES / 4u, 2u};
constexpr int kPackMaxSubs = 16;                // leaves to a wave at most: the packed kernel's LDS run tables and the layout's window keys are sized for it
constexpr bool pack_classes_fit() {
    for (int k = 0; k < kPackClasses; ++k) {
        const uint32_t w = kPackLanes[k], P = kPackGroups[k];
        const uint32_t max_c = k == 0 ? 2u : k == 1 ? 4u : k == 2 ? 6u : k == 3 ? 8u : 16u;   // largest leaf of the class
        if (w < 1u || 64u / w > (uint32_t)kPackMaxSubs) return false;       // leaves to a wave
        if (P < 1u || P * ((max_c + 1u) / 2u) > w) return false;            // every lane group's ceil(c / 2) lanes fit the leaf's w lanes
    }
    return true;
}
static_assert(pack_classes_fit(), "NBX_PACK_*_LANES: a class's lanes per leaf must hold its lane groups (P x ceil(c / 2) <= w) and give at most kPackMaxSubs leaves to a wave");
static_assert(kPackClasses <= 8, "order_launch keeps one class per shape: kOrderShapes");
constexpr int kPackPairsPerTrip = 2;            // the packed kernel computes two pairs while the next two are in flight
struct PackSub {                               // one packed leaf
    uint32_t op_lo, op_n;                      // its copy runs
    uint32_t first, count;                     // its targets (padded slots)
};
struct PackBlock {                             // one wave64; read with scalar loads
    uint32_t sub_lo, n_sub;                    // its leaves: subs[sub_lo .. sub_lo + n_sub)
    uint32_t w, P;                             // lanes per leaf (4 | 6 | 8 | 16), lane groups per leaf (2 | 4; 8 in A/B builds): each walks 1 / P of the stream
    uint32_t trips;                            // source pairs every lane group walks: ceil(longest stream's pairs / P), rounded up to kPackPairsPerTrip
    uint32_t inv_w;                            // ceil(65536 / w): lane / w = (lane * inv_w) >> 16 for lane < 64 (w need not be a power of two)
    uint32_t longest;                          // longest stream among the wave's leaves, in 16-byte units
    uint32_t shape;                            // size class of the wave's leaves (host-side sort key: waves of different shapes never share a duration class)
};
static_assert(sizeof(PackSub) == 16 && sizeof(PackBlock) == 32, "read with vector / scalar loads");

struct CopyOp {
    uint32_t end;      // length of the leaf's source stream up to and including this run, in 16-byte units
    uint32_t base;     // unit of the run's first body minus the stream position it lands on (mod 2^32): source = base + position
};


// How a run of c targets (<= 128) is cut into the two pieces of a workgroup: the cut that keeps the most lanes busy
// (a piece of c runs min(64 / c, 8) lanes per target and takes 1 / that of the source pairs' trips), the more even one among equals.
struct PieceCut { uint8_t c1[129]; };
inline PieceCut best_cuts() {
    PieceCut r;
    auto trips = [](int c) -> double {
        if (c == 0) return 0.0;
        int lanes = 64 / c;
        if (lanes > kMaxLanesPerTarget) lanes = kMaxLanesPerTarget;
        return 1.0 / lanes;
    };
    r.c1[0] = 0;
    for (int c = 1; c <= 128; ++c) {
        double best = 1e30, best_max = 1e30;
        int arg = c <= 64 ? c : 64;
        for (int a = (c > 64 ? c - 64 : 0); a <= c && a <= 64; ++a) {
            const double ta = trips(a), tb = trips(c - a);
            const double sum = ta + tb, mx = ta > tb ? ta : tb;
            if (sum < best - 1e-12 || (sum < best + 1e-12 && mx < best_max - 1e-12)) { best = sum; best_max = mx; arg = a; }
        }
        r.c1[c] = (uint8_t)arg;
    }
    return r;
}


struct LeafPlan {
    std::vector<uint32_t> unit_off;     // [n_leaves + 1] first padded slot (= 16-byte unit) of each leaf; a leaf of odd size gets one more slot
    std::vector<uint32_t> pslot_body;   // [pslots] body of each padded slot, 0xffffffff for a leaf's pad
    std::vector<CopyOp> ops;            // every leaf's source list as runs of consecutive units
    std::vector<uint32_t> op_off;       // [n_leaves + 1] leaf l's runs: ops[op_off[l] .. op_off[l+1])
    std::vector<uint32_t> stream_units; // [n_leaves] length of leaf l's source stream, in 16-byte units
    std::vector<std::vector<CopyOp>> part_ops;   // scratch of the layout's threads (kept for their capacity)
    std::vector<LeafBlock> blocks;      // the one-leaf workgroups, longest first
    int waves = kMaxWaves;              // wave64 per one-leaf workgroup of this launch
    std::vector<PackSub> pack_subs;     // packed small leaves ...
    std::vector<PackBlock> pack_blocks; // ... and the waves that take them, longest first
    size_t pslots() const { return unit_off.empty() ? 0 : unit_off.back(); }
};

// The order of a launch's workgroups.
//  * Longest first: the launch ends when its last workgroup does, and workgroups are dispatched in index order -- with the
//    short ones last the machine drains in a fraction of a mean workgroup's time (leaf order: 0.272 ms, sorted: 0.264 ms).
//    A counting sort over 1024 duration levels x 8 shapes, leaf order kept within a class.
//  * One eighth of the structure per XCD: workgroup i runs on XCD i % 8 and every XCD has an L2 of its own, so with a class in
//    plain leaf order every one of the eight L2s ends up fetching every body of the launch.  From kXcdOrderFrom workgroups on,
//    the blocks of a duration class (in leaf order) are cut into eight runs, and run x takes the indices of the class that are
//    x mod 8: XCD x sees the x-th eighth of every class -- for a structure whose durations do not follow position, the same
//    eighth of space throughout -- while every stretch of the index space still holds blocks of one duration (dispatch is in
//    order: eight runs of unequal durations side by side were measured 40 % SLOWER at 4-body leaves, the XCDs with the short
//    blocks waiting for the dispatcher to get past the ones with the long blocks).
constexpr uint32_t kXcds = 8;
constexpr const char* kPlanAllocFailed = "host allocation failed while the launch was laid out";
constexpr unsigned kPlanThreads = 8;            // host threads that lay out a launch's copy runs ...
#ifndef NBX_PLAN_THREADS_FROM
#define NBX_PLAN_THREADS_FROM 200000
#endif
constexpr size_t kPlanThreadsFrom = NBX_PLAN_THREADS_FROM;   // ... from this many list entries on (thread start-up: ~0.1 ms)
#ifndef NBX_XCD_ORDER_FROM
#define NBX_XCD_ORDER_FROM 4096
#endif
constexpr size_t kXcdOrderFrom = NBX_XCD_ORDER_FROM;
// f(0) on the caller's thread, f(1) .. f(n - 1) on threads of their own
template <class F>
inline void run_threads(unsigned n, F f) {
    std::thread helpers[kPlanThreads];
    if (n > kPlanThreads) n = kPlanThreads;
    unsigned started = 1;                               // pieces [1, started) run on threads of their own, the others here
    for (; started < n; ++started) {
        try { helpers[started] = std::thread(f, started); } catch (...) { break; }   // no thread to be had: that piece and the rest run here
    }
    f(0u);
    for (unsigned t = started; t < n; ++t) f(t);
    for (unsigned t = 1; t < started; ++t) helpers[t].join();
}

// The sort key is lexicographic: (duration level, shape).  Durations are quantised into kOrderLevels levels of the longest one;
// blocks of different SHAPES (the packed waves' size classes: equal trips are not equal durations) never share a class, whatever
// the trip counts -- a class is dealt to the XCDs side by side, which was measured 40 % slower for mixed durations.
constexpr uint32_t kOrderLevels = 1024, kOrderShapes = 8;
NBX_LEAF_HD inline uint32_t order_class(uint32_t dur, uint32_t longest, uint32_t shape) {
    return ((kOrderLevels - 1u) - (uint32_t)((uint64_t)dur * (kOrderLevels - 1u) / longest)) * kOrderShapes + shape;
}
template <class Block, class Dur, class Shape>
inline void order_launch(std::vector<Block>& v, Dur dur, Shape shape) {
    const size_t n = v.size();
    if (n < 2) return;
    uint32_t longest = 1;
    for (const Block& b : v) if (dur(b) > longest) longest = dur(b);
    constexpr uint32_t kClasses = kOrderLevels * kOrderShapes;
    auto cls = [&](const Block& b) -> uint32_t { return order_class(dur(b), longest, shape(b)); };
    std::vector<uint32_t> start(kClasses + 2, 0u);
    for (const Block& b : v) ++start[cls(b) + 2u];
    for (uint32_t k = 0; k <= kClasses; ++k) start[k + 1] += start[k];      // start[k + 1]: first index of class k
    std::vector<Block> sorted(n);
    for (const Block& b : v) sorted[start[cls(b) + 1u]++] = b;              // afterwards start[k + 1] = end of class k, start[k] = its begin
    if (n < kXcdOrderFrom) { v.swap(sorted); return; }
    for (uint32_t k = 0; k < kClasses; ++k) {
        const size_t b0 = start[k], b1 = start[k + 1];
        if (b1 - b0 < 2u * kXcds) { std::copy(sorted.begin() + (ptrdiff_t)b0, sorted.begin() + (ptrdiff_t)b1, v.begin() + (ptrdiff_t)b0); continue; }
        size_t first[kXcds], run_lo[kXcds], at = b0;                       // first index of the class that is x mod 8; where run x starts
        for (uint32_t x = 0; x < kXcds; ++x) {
            first[x] = b0 + (x + kXcds - (uint32_t)(b0 % kXcds)) % kXcds;
            run_lo[x] = at;
            at += first[x] < b1 ? (b1 - first[x] + kXcds - 1u) / kXcds : 0u;
        }
        for (size_t i = b0; i < b1; ++i) {
            const uint32_t x = (uint32_t)(i % kXcds);
            v[i] = sorted[run_lo[x] + (i - first[x]) / kXcds];
        }
    }
}

// The CSR arrays must have been validated (offsets non-decreasing from 0, every index in range).  Returns nullptr, or why the
// structure cannot be laid out (more than 2^32 units).
inline const char* plan_leaves(const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets,
                               const uint32_t* list_sources, LeafPlan& plan, bool pack_small_leaves = true, unsigned max_threads = kPlanThreads,
                               double* section_ms = nullptr /* [4] for tools/time_leaf_layout.cpp: slots, copy runs, workgroups, order */) {
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_last = t_begin;
    auto section = [&](int k) {
        const auto now = std::chrono::steady_clock::now();
        if (section_ms) section_ms[k] = std::chrono::duration<double, std::milli>(now - t_last).count();
        t_last = now;
    };
    const size_t slots = n_leaves ? leaf_offsets[n_leaves] : 0;
    const size_t n_list = n_leaves ? list_offsets[n_leaves] : 0;
    // padded slots: a leaf of odd size gets one more slot, so that every leaf is a run of whole source pairs
    std::vector<uint32_t>& unit_off = plan.unit_off;
    unit_off.resize(n_leaves + 1);              // every entry is written below
    {
        uint64_t u = 0;
        for (size_t l = 0; l < n_leaves; ++l) {
            unit_off[l] = (uint32_t)u;
            u += (uint64_t)((leaf_offsets[l + 1] - leaf_offsets[l] + 1u) & ~1u);
        }
        if (u > 0xfffffff0ull) return "too many bodies / leaves";
        unit_off[n_leaves] = (uint32_t)u;
    }
    const size_t pslots = unit_off[n_leaves];
    std::vector<uint32_t>& pslot_body = plan.pslot_body;
    pslot_body.resize(pslots);                  // every slot is written below: bodies by the copy, an odd leaf's pad explicitly              // body of each padded slot, 0xffffffff for a leaf's pad
    const unsigned n_threads = n_list >= kPlanThreadsFrom && max_threads > 1u ? (max_threads < kPlanThreads ? max_threads : kPlanThreads) : 1u;
    run_threads(n_threads, [&](unsigned t) {
        for (size_t l = n_leaves * t / n_threads; l < n_leaves * (t + 1u) / n_threads; ++l) {
            const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
            if (c) memcpy(&pslot_body[unit_off[l]], leaf_bodies + leaf_offsets[l], (size_t)c * sizeof(uint32_t));
            if (c & 1u) pslot_body[unit_off[l] + c] = 0xffffffffu;
        }
    });
    section(0);
    // copy ops: the source list of each leaf as runs of consecutive units (neighbours in leaf order merged), empty leaves dropped.
    // 5 ns per list entry on one core (the merge is a branch no predictor learns) and 6.7 million entries for 65,536 BVH leaves
    // at N = 2^20: the leaves are cut into kPlanThreads ranges of equal list length, every range is laid out by a thread of its
    // own into its own array, and the arrays are joined in leaf order -- the result does not depend on the number of threads.
    std::vector<CopyOp>& ops = plan.ops;
    std::vector<uint32_t>& op_off = plan.op_off;
    std::vector<uint32_t>& stream_units = plan.stream_units;
    op_off.resize(n_leaves + 1);
    stream_units.resize(n_leaves);
    if (plan.part_ops.size() < n_threads) plan.part_ops.resize(n_threads);
    size_t cut[kPlanThreads + 1];
    cut[0] = 0;
    for (unsigned t = 1; t < n_threads; ++t)
        cut[t] = (size_t)(std::lower_bound(list_offsets, list_offsets + n_leaves, (uint32_t)((uint64_t)n_list * t / n_threads)) - list_offsets);
    cut[n_threads] = n_leaves;
    const char* part_err[kPlanThreads] = {nullptr};
    auto lay_out = [&](unsigned t) {
      try {                                               // nothing may leave a thread (or the C ABI above) as an exception
        std::vector<CopyOp> mine;                       // this thread's own header: the scratch arrays' headers sit side by side in
        mine.swap(plan.part_ops[t]);                    // one cache line, and push_back writes the header every time
        struct PutBack { std::vector<CopyOp>& a; std::vector<CopyOp>& b; ~PutBack() { a.swap(b); } } put_back{mine, plan.part_ops[t]};
        mine.clear();
        mine.reserve((list_offsets[cut[t + 1]] - list_offsets[cut[t]]) / 2 + 16);
        for (size_t l = cut[t]; l < cut[t + 1]; ++l) {
            op_off[l] = (uint32_t)mine.size();          // within the range; the range's base is added below
            uint64_t stream = 0;        // units so far
            uint32_t run_first = 0, run_len = 0;
            auto close_run = [&]() {
                if (!run_len) return;
                stream += run_len;
                mine.push_back(CopyOp{(uint32_t)stream, run_first - (uint32_t)(stream - run_len)});
                run_len = 0;
            };
            for (uint32_t e = list_offsets[l]; e < list_offsets[l + 1]; ++e) {
                const uint32_t s = list_sources[e];
                const uint32_t first = unit_off[s], len = unit_off[s + 1] - unit_off[s];
                if (!len) continue;
                if (run_len && first == run_first + run_len) run_len += len;
                else { close_run(); run_first = first; run_len = len; }
                if (stream + run_len > 0xfffff000ull) { part_err[t] = "a leaf's source list names more than 2^32 bodies"; return; }
            }
            close_run();
            stream_units[l] = (uint32_t)stream;
        }
      } catch (...) { part_err[t] = kPlanAllocFailed; }
    };
    run_threads(n_threads, lay_out);
    size_t base[kPlanThreads + 1];
    base[0] = 0;
    for (unsigned t = 0; t < n_threads; ++t) {
        if (part_err[t]) return part_err[t];
        base[t + 1] = base[t] + plan.part_ops[t].size();
    }
    if (base[n_threads] > 0xfffffff0ull) return "source lists too long";
    ops.resize(base[n_threads]);
    auto join_range = [&](unsigned t) {
        if (!plan.part_ops[t].empty()) memcpy(ops.data() + base[t], plan.part_ops[t].data(), plan.part_ops[t].size() * sizeof(CopyOp));
        if (base[t]) for (size_t l = cut[t]; l < cut[t + 1]; ++l) op_off[l] += (uint32_t)base[t];
    };
    run_threads(n_threads, join_range);
    op_off[n_leaves] = (uint32_t)ops.size();
    section(1);
    // workgroups.  Leaves of the size the reference's FMM keeps (tens of bodies, methods.h:26): two waves, 128 targets of a leaf at
    // most, cut into the two waves' pieces.  Small leaves (the BVH's 16 bodies and below): one wave per workgroup and no cut -- a
    // leaf of 16 fills a wave at 4 lanes per target, and a workgroup barrier costs more than two waves sharing ~100 staged bodies save.
    size_t nonempty = 0;
    for (size_t l = 0; l < n_leaves; ++l) nonempty += leaf_offsets[l + 1] > leaf_offsets[l];
    const int waves = plan.waves = (nonempty && slots / nonempty <= (size_t)kSmallLeaf) ? 1 : kMaxWaves;
    const uint32_t per_group = 64u * (uint32_t)waves;
    static const PieceCut cuts = best_cuts();
    std::vector<LeafBlock>& blocks = plan.blocks;
    blocks.clear();
    blocks.reserve(n_leaves);
    {
        size_t nonempty_leaves = 0;
        for (size_t l = 0; l < n_leaves; ++l) nonempty_leaves += leaf_offsets[l + 1] > leaf_offsets[l];
        pack_small_leaves = pack_small_leaves && nonempty_leaves && slots <= (size_t)kPackMeanLeaf * nonempty_leaves &&
                            unit_off[n_leaves] < (1u << 28) - 2u;   // the packed kernel addresses units by 32-bit byte offsets
    }
    // packed waves.  A lane holds two targets, so a leaf of c bodies takes ceil(c / 2) lanes per lane group.  Leaves are packed
    // with leaves of their own SIZE CLASS -- 1-2 and 3-4 bodies on 4 lanes each (4 and 2 lane groups: 16 leaves to a wave, so that
    // a wave's start-up and closing reduction, ~400 instructions, are shared by 16 leaves and its loop is twice as long), 5-6
    // bodies on 6 lanes (10 leaves to a wave; on 8 lanes two of them idle), 7-8 bodies on 8 lanes, 9-16 bodies on 16 lanes (2
    // groups each); kPackLanes / kPackGroups -- in leaf order within the class, so that every wave runs the most lane groups its
    // leaves allow (mixed, one 8-body leaf would hold seven smaller ones at two groups).
    std::vector<PackSub>& subs = plan.pack_subs;
    std::vector<PackBlock>& packs = plan.pack_blocks;
    subs.clear();
    packs.clear();
    auto size_class = [](uint32_t c) -> int { return c <= 2u ? 0 : c <= 4u ? 1 : c <= 6u ? 2 : c <= 8u ? 3 : 4; };
    std::vector<uint32_t> packable[kPackClasses];
    for (size_t l = 0; l < n_leaves; ++l) {
        const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
        if (pack_small_leaves && c >= 1u && c <= (uint32_t)kPackMaxTargets && op_off[l + 1] - op_off[l] <= (uint32_t)kPackMaxOps)
            packable[size_class(c)].push_back((uint32_t)l);
    }
    // Where a class's waves and leaves go is known from the class sizes alone, so the classes' windows (eight waves' leaves each) are
    // laid out by the layout's threads, every window on its own.
    // A wave runs as long as its longest leaf: within every window (neighbouring leaves: the locality stays) the leaves are taken
    // longest stream first, so that a wave's leaves are of nearly one length (a 27-cell neighbourhood of ~4.5-body cells: the longest
    // of 8 random ones is 12 % above the mean, of 8 consecutive ones of a sorted 64 2 %).
    size_t sub_base[kPackClasses + 1] = {0}, pack_base[kPackClasses + 1] = {0}, win_base[kPackClasses + 1] = {0};
    for (int k = 0; k < kPackClasses; ++k) {
        const size_t per_wave = 64u / kPackLanes[k];
        sub_base[k + 1] = sub_base[k] + packable[k].size();
        pack_base[k + 1] = pack_base[k] + (packable[k].size() + per_wave - 1u) / per_wave;
        win_base[k + 1] = win_base[k] + (packable[k].size() + kPackWindowWaves * per_wave - 1u) / (kPackWindowWaves * per_wave);
    }
    subs.resize(sub_base[kPackClasses]);
    packs.resize(pack_base[kPackClasses]);
    const size_t n_windows = win_base[kPackClasses];
    const unsigned pack_threads = n_windows >= 64u ? n_threads : 1u;
    auto pack_windows = [&](unsigned t) {
        const size_t w_lo = n_windows * t / pack_threads, w_hi = n_windows * (t + 1u) / pack_threads;
        for (size_t wi = w_lo; wi < w_hi; ++wi) {
            int k = 0;
            while (wi >= win_base[k + 1]) ++k;
            const uint32_t w = kPackLanes[k], per_wave = 64u / w;
            const uint32_t P = kPackGroups[k];
            std::vector<uint32_t>& leaves_k = packable[k];
            const size_t i = (wi - win_base[k]) * kPackWindowWaves * per_wave;
            const size_t e = i + kPackWindowWaves * per_wave < leaves_k.size() ? i + kPackWindowWaves * per_wave : leaves_k.size();
            uint64_t key[kPackWindowWaves * (size_t)kPackMaxSubs];                          // longest first, leaf order among equals: one integer sort per window
            for (size_t j = i; j < e; ++j) key[j - i] = ((uint64_t)(0xffffffffu - stream_units[leaves_k[j]]) << 32) | leaves_k[j];
            std::sort(key, key + (e - i));
            for (size_t j = i; j < e; ++j) leaves_k[j] = (uint32_t)key[j - i];
            for (size_t i0 = i; i0 < e; i0 += per_wave) {
                PackBlock b{};
                b.sub_lo = (uint32_t)(sub_base[k] + i0);
                b.w = w; b.inv_w = (65536u + w - 1u) / w; b.P = P; b.shape = (uint32_t)k;
                uint32_t longest = 0;
                for (size_t j = i0; j < e && j < i0 + per_wave; ++j) {
                    const uint32_t l = leaves_k[j];
                    subs[sub_base[k] + j] = PackSub{op_off[l], op_off[l + 1] - op_off[l], unit_off[l], leaf_offsets[l + 1] - leaf_offsets[l]};
                    ++b.n_sub;
                    if (stream_units[l] > longest) longest = stream_units[l];
                }
                const uint32_t per_group = ((longest >> 1) + P - 1u) / P;   // streams are whole pairs (every leaf is padded to an even size)
                b.trips = (per_group + (uint32_t)kPackPairsPerTrip - 1u) / (uint32_t)kPackPairsPerTrip * (uint32_t)kPackPairsPerTrip;
                b.longest = longest;
                packs[pack_base[k] + i0 / per_wave] = b;
            }
        }
    };
    run_threads(pack_threads, pack_windows);
    for (size_t l = 0; l < n_leaves; ++l) {
        const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
        if (!c) continue;
        if (pack_small_leaves && c <= (uint32_t)kPackMaxTargets && op_off[l + 1] - op_off[l] <= (uint32_t)kPackMaxOps) continue;   // packed above
        const uint32_t groups = (c + per_group - 1u) / per_group;
        uint32_t f = unit_off[l];
        for (uint32_t gi = 0; gi < groups; ++gi) {
            const uint32_t share = c / groups + (gi < c % groups ? 1u : 0u);   // <= 64 x waves
            const uint32_t c1 = waves == 2 ? cuts.c1[share] : share;
            LeafBlock b;
            b.op_lo = op_off[l];
            b.op_n = op_off[l + 1] - op_off[l];
            b.pad_[0] = b.pad_[1] = 0;
            b.piece[0] = Piece{f, c1};
            b.piece[1] = Piece{f + c1, share - c1};
            {   // how long the workgroup will run: the slower wave's trips over the leaf's stream (sort key below)
                auto lanes = [](uint32_t c) -> uint32_t { const uint32_t p = c ? 64u / c : (uint32_t)kMaxLanesPerTarget; return p > (uint32_t)kMaxLanesPerTarget ? (uint32_t)kMaxLanesPerTarget : p; };
                const uint32_t stream_units = b.op_n ? ops[b.op_lo + b.op_n - 1].end : 0u;
                const uint32_t slower = lanes(c1) < lanes(share - c1) ? lanes(c1) : lanes(share - c1);
                b.pad_[0] = stream_units / slower;
            }
            blocks.push_back(b);
            f += share;
        }
    }
    section(2);
    // waves of different shapes (lanes per leaf, lane groups) never share a duration class: equal trips are not equal durations, and
    // the blocks of a class are dealt to the XCDs side by side
    order_launch(packs, [](const PackBlock& b) -> uint32_t { return b.trips; }, [](const PackBlock& b) -> uint32_t { return b.shape; });
    order_launch(blocks, [](const LeafBlock& b) -> uint32_t { return b.pad_[0]; }, [](const LeafBlock&) -> uint32_t { return 0u; });
    section(3);

    return nullptr;
}

}  // namespace nbx_leaf
