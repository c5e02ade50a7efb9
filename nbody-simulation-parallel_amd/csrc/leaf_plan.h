// leaf_plan.h -- host side of the leaf-pair path (csrc/leaf_pair_kernel.hip): everything the pair kernel follows, laid out once per
// call from the caller's CSR arrays.  Plain C++ (no HIP): tests/test_leaf_plan_cpu.py compiles it with g++ and checks its
// invariants without a GPU.  The comment at the top of leaf_pair_kernel.hip says what each piece is for.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

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
// kPackMaxOps copy runs are therefore PACKED: one wave64 takes K = 64 / w consecutive leaves, each on its own w = 8 or 16 lanes --
// its own copy runs, its own region of the LDS tile, P lanes per target (the same for every leaf of the wave -- leaves are packed
// with leaves of their size class -- so that every lane runs the same trip count) -- and all of them run the pair loop together.  Everything else (larger leaves,
// longer lists) keeps the one-leaf workgroups above.
constexpr int kPackMaxTargets = 16;
#ifndef NBX_PACK_MAX_OPS
#define NBX_PACK_MAX_OPS 16
#endif
constexpr int kPackMaxOps = NBX_PACK_MAX_OPS;
// Packing is used for structures whose leaves hold at most this many bodies ON AVERAGE.  Measured at N = 2^20
// (profiles/r4/leaf_pack_ab.txt, same box, back-to-back launches): 4-body grid leaves 0.226 -> 0.213 ms packed; median-split
// leaves of 8 bodies 0.146 -> 0.203 ms and of 16 bodies 0.200 -> 0.273 ms (a leaf of 8-16 bodies fills a wave of its own at
// 4-8 lanes per target; side by side on 8-16 lanes each they run one lane per target and a quarter of the waves).
#ifndef NBX_PACK_MEAN_LEAF
#define NBX_PACK_MEAN_LEAF 6   /* A/B builds raise it to pack larger leaves too */
#endif
constexpr int kPackMeanLeaf = NBX_PACK_MEAN_LEAF;
constexpr int kPackUnitsPerLane = 8;           // a sub-leaf's tile is 8 w units: 64 (w = 8) or 128 (w = 16) bodies
struct PackSub {                               // one packed leaf
    uint32_t op_lo, op_n;                      // its copy runs
    uint32_t first, count;                     // its targets (padded slots)
};
struct PackBlock {                             // one wave64; read with scalar loads
    uint32_t sub_lo, n_sub;                    // its leaves: subs[sub_lo .. sub_lo + n_sub)
    uint32_t w, P;                             // lanes per leaf (8 | 16), lanes per target (1 .. 8)
    uint32_t tiles;                            // tile iterations: ceil(longest stream / (kPackUnitsPerLane * w))
    uint32_t w_log2;                           // 3 | 4
    uint32_t longest;                          // longest stream among the wave's leaves, in 16-byte units
    uint32_t pad_;
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
    std::vector<LeafBlock> blocks;      // the one-leaf workgroups, longest first
    int waves = kMaxWaves;              // wave64 per one-leaf workgroup of this launch
    std::vector<PackSub> pack_subs;     // packed small leaves ...
    std::vector<PackBlock> pack_blocks; // ... and the waves that take them, longest first
    size_t pslots() const { return unit_off.empty() ? 0 : unit_off.back(); }
};

// The CSR arrays must have been validated (offsets non-decreasing from 0, every index in range).  Returns nullptr, or why the
// structure cannot be laid out (more than 2^32 units).
inline const char* plan_leaves(const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets,
                               const uint32_t* list_sources, LeafPlan& plan, bool pack_small_leaves = true) {
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
    for (size_t l = 0; l < n_leaves; ++l) {
        const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
        if (c) memcpy(&pslot_body[unit_off[l]], leaf_bodies + leaf_offsets[l], (size_t)c * sizeof(uint32_t));
        if (c & 1u) pslot_body[unit_off[l] + c] = 0xffffffffu;
    }
    // copy ops: the source list of each leaf as runs of consecutive units (neighbours in leaf order merged), empty leaves dropped
    std::vector<CopyOp>& ops = plan.ops;
    ops.clear();
    std::vector<uint32_t>& op_off = plan.op_off;
    op_off.resize(n_leaves + 1);
    ops.reserve(n_list / 2 + 16);
    for (size_t l = 0; l < n_leaves; ++l) {
        op_off[l] = (uint32_t)ops.size();
        uint64_t stream = 0;        // units so far
        uint32_t run_first = 0, run_len = 0;
        auto close_run = [&]() {
            if (!run_len) return;
            stream += run_len;
            ops.push_back(CopyOp{(uint32_t)stream, run_first - (uint32_t)(stream - run_len)});
            run_len = 0;
        };
        for (uint32_t e = list_offsets[l]; e < list_offsets[l + 1]; ++e) {
            const uint32_t s = list_sources[e];
            const uint32_t first = unit_off[s], len = unit_off[s + 1] - unit_off[s];
            if (!len) continue;
            if (run_len && first == run_first + run_len) run_len += len;
            else { close_run(); run_first = first; run_len = len; }
            if (stream + run_len > 0xfffff000ull) return "a leaf's source list names more than 2^32 bodies";
        }
        close_run();
        if (ops.size() > 0xfffffff0ull) return "source lists too long";
    }
    op_off[n_leaves] = (uint32_t)ops.size();
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
        pack_small_leaves = pack_small_leaves && nonempty_leaves && slots <= (size_t)kPackMeanLeaf * nonempty_leaves;
    }
    // packed waves.  Leaves are packed with leaves of their own SIZE CLASS -- 1, 2, 3-4, 5-8 bodies on 8 lanes each (8, 4, 2, 1
    // lanes per target), 9-16 bodies on 16 lanes -- in leaf order within the class, so that every wave runs the most lanes per
    // target its leaves allow (mixed, one 8-body leaf would hold seven smaller ones at one lane per target).
    std::vector<PackSub>& subs = plan.pack_subs;
    std::vector<PackBlock>& packs = plan.pack_blocks;
    subs.clear();
    packs.clear();
    auto size_class = [](uint32_t c) -> int { return c <= 1u ? 0 : c <= 2u ? 1 : c <= 4u ? 2 : c <= 8u ? 3 : 4; };
    std::vector<uint32_t> packable[5];
    for (size_t l = 0; l < n_leaves; ++l) {
        const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
        if (pack_small_leaves && c >= 1u && c <= (uint32_t)kPackMaxTargets && op_off[l + 1] - op_off[l] <= (uint32_t)kPackMaxOps)
            packable[size_class(c)].push_back((uint32_t)l);
    }
    for (int k = 0; k < 5; ++k) {
        const uint32_t w = k == 4 ? 16u : 8u, per_wave = 64u / w;
        const uint32_t P = k == 0 ? 8u : k == 1 ? 4u : k == 2 ? 2u : 1u;
        for (size_t i = 0; i < packable[k].size(); i += per_wave) {
            PackBlock b{};
            b.sub_lo = (uint32_t)subs.size();
            b.w = w; b.w_log2 = k == 4 ? 4u : 3u; b.P = P;
            uint32_t longest = 0;
            for (size_t j = i; j < packable[k].size() && j < i + per_wave; ++j) {
                const uint32_t l = packable[k][j];
                const uint32_t n_ops = op_off[l + 1] - op_off[l];
                const uint32_t stream = n_ops ? ops[op_off[l] + n_ops - 1].end : 0u;
                subs.push_back(PackSub{op_off[l], n_ops, unit_off[l], leaf_offsets[l + 1] - leaf_offsets[l]});
                ++b.n_sub;
                if (stream > longest) longest = stream;
            }
            const uint32_t tile_units = (uint32_t)kPackUnitsPerLane * w;
            b.tiles = (longest + tile_units - 1u) / tile_units;
            b.longest = longest;
            packs.push_back(b);
        }
    }
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
    if (packs.size() > 1) {   // longest first, stable (leaf order kept among waves of equal length)
        std::vector<PackBlock> sorted(packs);
        uint32_t longest = 0;
        for (const PackBlock& b : packs) if (b.tiles > longest) longest = b.tiles;
        std::vector<uint32_t> start(longest + 2, 0u);
        for (const PackBlock& b : packs) ++start[longest - b.tiles + 1u];
        for (uint32_t k = 0; k <= longest; ++k) start[k + 1] += start[k];
        for (const PackBlock& b : packs) sorted[start[longest - b.tiles]++] = b;
        packs.swap(sorted);
    }
    // Longest first: the launch ends when its last workgroup does, and workgroups are dispatched in index order -- with the
    // short ones last the machine drains in a fraction of a mean workgroup's time (leaf order: 0.272 ms, sorted: 0.264 ms).
    // A counting sort over 1024 duration classes, leaf order kept within a class (neighbours share their sources in L2).
    if (blocks.size() > 1) {
        uint32_t longest = 1;
        for (const LeafBlock& b : blocks) if (b.pad_[0] > longest) longest = b.pad_[0];
        constexpr uint32_t kClasses = 1024;
        auto cls = [&](const LeafBlock& b) -> uint32_t { return (kClasses - 1u) - (uint32_t)((uint64_t)b.pad_[0] * (kClasses - 1u) / longest); };
        std::vector<uint32_t> start(kClasses + 1, 0u);
        for (const LeafBlock& b : blocks) ++start[cls(b) + 1u];
        for (uint32_t k = 0; k < kClasses; ++k) start[k + 1] += start[k];
        std::vector<LeafBlock> sorted(blocks.size());
        for (const LeafBlock& b : blocks) sorted[start[cls(b)]++] = b;
        blocks.swap(sorted);
    }

    return nullptr;
}

}  // namespace nbx_leaf
