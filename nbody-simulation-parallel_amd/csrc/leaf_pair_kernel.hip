// leaf_pair_kernel.hip -- SURVEY 8(f-4): batched (target leaf, source leaf) direct sums for the reference's tree
// codes -- the near-field step either side of the brute-force path:
//   FMM_Parlay<D>::p2p_phase   nbody-sim-new/fmm_parlay.cpp:916-1022   (law NBX_LAW_FMM_P2P)
//   BVH leaf loop              nbody-sim-new/bvh.cpp:150-176           (law NBX_LAW_TREE_LEAF)
//   octree leaf term           nbody-sim-new/octree.cpp:105-125        (law NBX_LAW_TREE_LEAF)
// and, for completeness, the brute-force law itself over leaf lists (NBX_LAW_BRUTE, methods.cpp:21-37).
// All three are m_j d / r^4 sums; they differ in sign and in what happens below ~1e-5 separation.
//
// Mapping to CDNA4.  The reference walks pointer lists per (body, neighbour leaf) work item and adds into
// forces[body] from several work items at once (fmm_parlay.cpp:986-1020).  Here the bodies are gathered once into
// leaf order as fp32 {x,y,z,m} (one 16-byte load per body), and the work is target-leaf-major: ONE WAVE64 owns up to 128
// targets of one leaf -- two per lane, held as packed fp32 pairs like the brute-force kernel's -- and that leaf's whole
// source-leaf list, staged through LDS leaf by leaf; fp32 sums over at most 256 terms, flushed into fp64 accumulators.
// No atomics, a fixed summation order, every output written once.  The comment at the kernel says how the lanes share
// the work.  Leaves are small (the reference caps them at 100 bodies, methods.h:26), so the launch is tens of thousands
// of short single-wave workgroups; HBM traffic is 16 B per (target block, source body), served mostly from L2.
// VALU-issue-bound: 14 VALU per source and lane (= per two pair terms), ~12 % on top for staging, flushes and the
// prologue (round 2's kernel: 16 per two terms, 24 % on top, 4-way LDS bank conflicts on its tile writes).
#include "../../include/nbody_hip.h"
#include "nbx_ctx.h"

#include <cstdio>
#include <type_traits>
#include <vector>

using namespace nbx;

namespace {

constexpr int kWave = 64;               // lanes per block of targets: one wave64
constexpr int kWavesPerGroup = 4;       // waves per workgroup, each with a target block of its own (no workgroup barrier)
constexpr int kMaxTargetsPerLane = 4;    // a block holds up to 64 x 4 targets
constexpr int kMaxGroups = 16;          // lane groups that split the sources of a block with few targets
constexpr int kLeafTile = 64;               // source bodies per LDS tile

// smallest fp32 thresholds that are >= the reference's fp64 ones, so (r2 < T_f32) == ((double)r2 < T) for fp32 r2
constexpr float kTreeSkipF = 0x1.12e0c0p-30f;   // 1.00000008e-9  (octree.cpp:119, bvh.cpp:167: dist_sq < 1e-9)
constexpr float kSmoothF = 0x1.b7cdfep-34f;     // 1.00000001e-10 (fmm_parlay.cpp:1010: dist_sq < 1e-10)
constexpr float kNormZeroF = 0x1.79ca12p-67f;   // 1.00000005e-20 (vector.h:93-97: |diff| < 1e-10 -> zero vector)
constexpr float kSameF = 1.0e-14f;              // largest fp32 <= 1e-14 (fmm_parlay.cpp:995-1000: |d_k| > 1e-14 -> distinct)
static_assert((double)kTreeSkipF >= 1e-9 && (double)kSmoothF >= 1e-10 && (double)kNormZeroF >= 1e-20 && (double)kSameF <= 1e-14,
              "fp32 thresholds must sit on the right side of the fp64 ones");
constexpr float kFar = 1.0e18f;                 // pad bodies: sources at +kFar, pad targets at -kFar (r^2 ~ 1e37, weight underflows to 0)

struct TargetBlock {   // built on the host: a runtime integer division costs the GPU ~30 VALU, the kernel would need six per block
    uint32_t leaf;     // target leaf
    uint32_t first;    // first target slot (leaf order)
    uint32_t count;    // <= 256: four targets per lane
    uint32_t L;        // lanes per group = ceil(count / 4)
    uint32_t G;        // lane groups: the largest of {1..10, 12, 16} that fits 64 / L
    uint32_t inv_L;    // q / L = (q * inv_L) >> 16 for q < 128  (inv = 65536 / L + 1)
    uint32_t pad_[2];
};
static_assert(sizeof(TargetBlock) == 32, "TargetBlock is read with scalar loads");

struct LeafArgs {
    const float4* __restrict__ xm;     // [slots] leaf-ordered {x, y, z (0 in 2D), m}: one 16-byte load stages a body
    uint32_t slots;
    const uint32_t* __restrict__ leaf_offsets;
    const uint32_t* __restrict__ list_offsets;
    const uint32_t* __restrict__ list_sources;
    const TargetBlock* __restrict__ blocks;
    uint32_t n_blocks;
    double* __restrict__ acc;          // [dim][slots]
};

// Weight of d = p_j - p_i in the law's sum for ONE pair, every special case included: m_j / r^4 for an ordinary pair.
template <int D, int LAW>
__device__ __forceinline__ float leaf_weight(float r2, float mj, float dx, float dy, float dz) {
    if (LAW == NBX_LAW_BRUTE) {
        const float g = (r2 < kR2SkipF) ? __builtin_inff() : r2;           // methods.cpp:24
        const float ri = __builtin_amdgcn_rcpf(g);
        return mj * ri * ri;
    } else if (LAW == NBX_LAW_TREE_LEAF) {
        // "same position" (every |d_k| <= 1e-9) implies r2 <= 3e-18 < 1e-9: one test covers both skips
        const float g = (r2 < kTreeSkipF) ? __builtin_inff() : r2;
        const float ri = __builtin_amdgcn_rcpf(g);
        return mj * ri * ri;
    } else {
        if (r2 < kSmoothF) {   // rare: smoothed magnitude, unsmoothed direction (fmm_parlay.cpp:1010-1020, vector.h:93-97)
            const bool same = __builtin_fabsf(dx) <= kSameF && __builtin_fabsf(dy) <= kSameF && (D == 2 || __builtin_fabsf(dz) <= kSameF);
            const float r2s = r2 + 1.0e-10f;                                                   // epsilon^2, epsilon = 1e-5
            const float mag = mj * __builtin_amdgcn_rcpf(r2s) * __builtin_amdgcn_rsqf(r2s);    // m / (r2s * sqrt(r2s))
            const float inv = (r2 < kNormZeroF) ? 0.0f : __builtin_amdgcn_rsqf(r2);               // normalized(): 0 below 1e-10
            return same ? 0.0f : mag * inv;
        }
        const float ri = __builtin_amdgcn_rcpf(r2);
        return mj * ri * ri;
    }
}

typedef float f2 __attribute__((ext_vector_type(2)));

// Ordering of a wave's own LDS traffic (its tile and its sums are private to it): the LDS executes one wave's accesses in
// program order, so all that is needed is that the compiler keeps that order and waits for outstanding accesses.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Below this r^2 a pair leaves the plain m_j d / r^4 form under the law (skip or smoothing); leaf_weight decides how.
template <int LAW>
__device__ __forceinline__ constexpr float law_special_below() {
    return LAW == NBX_LAW_BRUTE ? kR2SkipF : LAW == NBX_LAW_TREE_LEAF ? kTreeSkipF : kSmoothF;
}

// One wave64 = up to 256 targets of one leaf against that leaf's source list.
//  * FOUR TARGETS PER LANE as two packed fp32 pairs (lane p holds targets p, p + L, p + 2L, p + 3L of the block, L = ceil(count /
//    4)), every source a broadcast: per source and target pair 3 v_pk_add (d), v_pk_mul + 2 v_pk_fma (r^2), 2 v_rcp,
//    2 v_pk_mul (w^2, .m with the op_sel form of the brute-force kernel), 3 v_pk_fma (accumulate) and one v_min3 = 14 VALU
//    per two pair terms.
//  * A block runs G = floor(64 / L) <= 16 LANE GROUPS that split the sources G ways (group g takes bodies g, g + G, ... of
//    the stream); their sums meet in LDS at the end, in group order (deterministic).  G IS A TEMPLATE PARAMETER (the kernel
//    switches on the block's value once): the stride between a group's sources, the tile length T = 2 G floor(64 / 2G) and
//    every LDS offset of the pair loop are compile-time constants -- no address arithmetic in the loop, one loop form
//    for full and partial tiles (the stream's end is padded to a whole trip of 2 G massless bodies), no remainder loop.
//  * The source list is staged LEAF BY LEAF into a body STREAM in natural order: lane e holds list entry e (source leaf ->
//    slot range), the wave walks the entries with v_readlane, one 16-byte global load per lane (issued two pieces ahead)
//    and one ds_write_b128 put up to T bodies of a leaf behind the bodies already staged; T staged bodies are one tile.
//    No prefix sum, no per-body search of the list.
//  * fp32 sums per lane (a lane sees sources / G terms: 108 at 864 sources and 8 groups), flushed into fp64 sums that live
//    in LDS (the reduction buffer itself) whenever 256 terms are reached -- no fp64 accumulator occupies a register during
//    the pair loop.
//  * The law's special cases (skip / smoothing below ~1e-5 separation) cost one v_min3 per two pair terms in the common
//    path: it keeps the smallest r^2 a lane has seen.  Tiles that hold bodies of the target's own leaf (every body meets
//    itself there) run the GUARDED loop -- a compare and select per pair term, and for the FMM law the smoothing branch
//    behind a wave vote.  If at the end some lane's minimum lies below the law's threshold -- two distinct bodies of
//    different leaves closer than 3e-5: next to never -- the whole block is redone with the guarded loop throughout.
template <int D, int LAW, int G>
__device__ __forceinline__ void leaf_block(const LeafArgs& a, const TargetBlock tb, float4 (&tile)[2][kLeafTile], double (&red)[12][kWave]) {
    constexpr int PAIRS = 2, NS = 2;                           // target pairs per lane, sources per loop trip
    constexpr unsigned T = (unsigned)(NS * G * (kLeafTile / (NS * G)));   // bodies per tile: whole trips of NS * G
    const unsigned lane = threadIdx.x & (unsigned)(kWave - 1);
    const unsigned c = tb.count, L = tb.L;
    const unsigned g_raw = (lane * tb.inv_L) >> 16, p = lane - g_raw * L;   // lane / L, lane % L
    const unsigned g = g_raw < (unsigned)G ? g_raw : 0u;        // lanes left over compute along with group 0, unused
    f2 ix[PAIRS], iy[PAIRS], iz[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const unsigned t0 = p + (unsigned)(2 * q) * L, t1 = t0 + L;
        const float4 far = make_float4(-kFar, -kFar, (D == 3) ? -kFar : 0.0f, 0.0f);   // pad target: every weight underflows to 0
        const float4 m0 = t0 < c ? a.xm[tb.first + t0] : far, m1 = t1 < c ? a.xm[tb.first + t1] : far;
        ix[q] = f2{m0.x, m1.x}; iy[q] = f2{m0.y, m1.y}; iz[q] = f2{(D == 3) ? m0.z : 0.0f, (D == 3) ? m1.z : 0.0f};
    }
    const float4 pad = make_float4(kFar, kFar, (D == 3) ? kFar : 0.0f, 0.0f);
    f2 ax[PAIRS], ay[PAIRS], az[PAIRS];                         // fp32 sums since the last flush
    unsigned pending;                                           // terms in them
    float rmin;                                                 // smallest r^2 seen by the unguarded loop
    double* const mine = &red[0][lane];                         // this lane's fp64 sums: [component * 4 + target][lane]

    auto flush = [&]() {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            mine[(0 * 4 + 2 * q) * kWave] += (double)ax[q].x; mine[(0 * 4 + 2 * q + 1) * kWave] += (double)ax[q].y;
            mine[(1 * 4 + 2 * q) * kWave] += (double)ay[q].x; mine[(1 * 4 + 2 * q + 1) * kWave] += (double)ay[q].y;
            if (D == 3) { mine[(2 * 4 + 2 * q) * kWave] += (double)az[q].x; mine[(2 * 4 + 2 * q + 1) * kWave] += (double)az[q].y; }
            ax[q] = ay[q] = az[q] = f2{0.f, 0.f};
        }
        pending = 0;
    };
    // NS sources (this group's: G slots apart) against the lane's PAIRS target pairs, stage by stage: NS * PAIRS chains
    auto fast = [&](const float4* __restrict__ src) {
        f2 dx[NS][PAIRS], dy[NS][PAIRS], dz[NS][PAIRS], r2[NS][PAIRS], w[NS][PAIRS], szm[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float4 s = src[k * G];
            szm[k] = f2{s.z, s.w};
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
                dx[k][q] = f2{s.x, s.x} - ix[q];
                dy[k][q] = f2{s.y, s.y} - iy[q];
                dz[k][q] = (D == 3) ? f2{s.z, s.z} - iz[q] : f2{0.f, 0.f};
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) r2[k][q] = dx[k][q] * dx[k][q];
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) r2[k][q] = __builtin_elementwise_fma(dy[k][q], dy[k][q], r2[k][q]);
        if (D == 3) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int q = 0; q < PAIRS; ++q) r2[k][q] = __builtin_elementwise_fma(dz[k][q], dz[k][q], r2[k][q]);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) { w[k][q].x = __builtin_amdgcn_rcpf(r2[k][q].x); w[k][q].y = __builtin_amdgcn_rcpf(r2[k][q].y); }
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) rmin = __builtin_fminf(__builtin_fminf(rmin, r2[k][q].x), r2[k][q].y);   // v_min3_f32
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) r2[k][q] = w[k][q] * w[k][q];
        // the mass is the HIGH half of the source's {z, m} register pair: op_sel spelled out (force_kernel.hip), applied to
        // w^2 (plain code, hazards handled by the compiler), never directly to a v_rcp result
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(w[k][q]) : "v"(szm[k]), "v"(r2[k][q]));
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) ax[q] = __builtin_elementwise_fma(w[k][q], dx[k][q], ax[q]);
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) ay[q] = __builtin_elementwise_fma(w[k][q], dy[k][q], ay[q]);
        if (D == 3) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int q = 0; q < PAIRS; ++q) az[q] = __builtin_elementwise_fma(w[k][q], dz[k][q], az[q]);
        }
    };
    auto guarded1 = [&](const float4 s) {   // exact law per pair term: compare and select; FMM smoothing behind a wave vote
        constexpr float TH = law_special_below<LAW>();
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            const f2 dx = f2{s.x, s.x} - ix[q], dy = f2{s.y, s.y} - iy[q], dz = (D == 3) ? f2{s.z, s.z} - iz[q] : f2{0.f, 0.f};
            f2 r2 = dx * dx;
            r2 = __builtin_elementwise_fma(dy, dy, r2);
            if (D == 3) r2 = __builtin_elementwise_fma(dz, dz, r2);
            f2 w;
            if (LAW == NBX_LAW_FMM_P2P &&
                (__builtin_amdgcn_ballot_w64(r2.x < TH && r2.x > 0.0f) | __builtin_amdgcn_ballot_w64(r2.y < TH && r2.y > 0.0f)) != 0ull) {
                w = f2{leaf_weight<D, LAW>(r2.x, s.w, dx.x, dy.x, dz.x), leaf_weight<D, LAW>(r2.y, s.w, dx.y, dy.y, dz.y)};
            } else {   // below the threshold: skipped (brute force, tree leaf), or the same position (FMM: r^2 = 0) -- weight 0
                const f2 r2g = {(r2.x < TH) ? __builtin_inff() : r2.x, (r2.y < TH) ? __builtin_inff() : r2.y};
                w = f2{__builtin_amdgcn_rcpf(r2g.x), __builtin_amdgcn_rcpf(r2g.y)};
                w = w * w;
                w = w * f2{s.w, s.w};
            }
            ax[q] = __builtin_elementwise_fma(w, dx, ax[q]);
            ay[q] = __builtin_elementwise_fma(w, dy, ay[q]);
            if (D == 3) az[q] = __builtin_elementwise_fma(w, dz, az[q]);
        }
    };

    // ---- the source list, leaf by leaf ----
    const uint32_t e_begin = a.list_offsets[tb.leaf], e_end = a.list_offsets[tb.leaf + 1];
    uint32_t e_base = 0;                                       // first list entry of the chunk held in the lanes
    unsigned n_ent = 0, e_next = 0;                            // entries in the chunk, next one to issue
    uint32_t v_first = 0, v_len = 0, v_src = 0, off_next = 0;  // lane e: entry e_base + e
    auto load_chunk = [&]() {
        n_ent = (e_end - e_base < (uint32_t)kWave) ? (unsigned)(e_end - e_base) : (unsigned)kWave;
        v_first = v_len = 0; v_src = 0xffffffffu;
        if (lane < n_ent) {
            v_src = a.list_sources[e_base + lane];
            v_first = a.leaf_offsets[v_src];
            v_len = a.leaf_offsets[v_src + 1] - v_first;
        }
        e_next = 0; off_next = 0;
    };
    // a piece = up to T bodies of one source leaf: (v, n, own) = this lane's body, the piece's size, "it is the target's leaf".
    // Lanes past the piece's end load its last body again (an address inside the leaf, no branch) and do not write it.
    auto issue = [&](float4& v, unsigned& n, bool& own) {      // wave-uniform control; the load stays in flight
        n = 0u; own = false;
        for (;;) {
            if (e_next == n_ent) {
                e_base += n_ent;
                n_ent = e_next = 0;
                if (e_base >= e_end) return;                   // the list is exhausted: an empty piece, again and again
                load_chunk();
                continue;
            }
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)v_len, (int)e_next);
            if (off_next >= len) { ++e_next; off_next = 0; continue; }   // empty leaves are stepped over here as well
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)v_first, (int)e_next);
            const uint32_t src_leaf = (uint32_t)__builtin_amdgcn_readlane((int)v_src, (int)e_next);
            n = (len - off_next < T) ? (unsigned)(len - off_next) : T;
            own = src_leaf == tb.leaf;
            const float4* __restrict__ base = a.xm + (first + off_next);   // uniform: scalar base + per-lane 32-bit offset
            v = base[lane < n ? lane : n - 1u];
            off_next += n;
            return;
        }
    };
    bool guard_all = false;
    for (int pass = 0; pass < 2; ++pass) {                     // pass 1 only after a sub-threshold pair outside the own leaf
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 12; ++k) mine[k * kWave] = 0.0;
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) ax[q] = ay[q] = az[q] = f2{0.f, 0.f};
        pending = 0;
        rmin = __builtin_inff();
        e_base = e_begin; n_ent = e_next = 0; off_next = 0;
        int cur = 0;
        unsigned fill = 0;                                     // bodies staged in tile[cur], the overflow in tile[cur ^ 1]
        bool own_cur = false, own_nxt = false;                 // the tile holds bodies of the target's own leaf
        auto consume = [&](const int buf, const unsigned trips, const bool guard) {   // trips of NS * G bodies each
            wave_lds_sync();                                   // the tile's writes have landed
            if (pending + trips * (unsigned)NS > 256u) flush();
            const float4* __restrict__ src = &tile[buf][g];
            if (guard) {
#pragma unroll 1
                for (unsigned k = 0; k < trips * (unsigned)NS; ++k) guarded1(src[k * (unsigned)G]);
            } else {
#pragma unroll 1
                for (unsigned k = 0; k < trips; ++k) fast(src + k * (unsigned)(NS * G));   // one loop form for full and partial tiles
            }
            pending += trips * (unsigned)NS;
            wave_lds_sync();                                   // the tile is free again
        };
        auto stage = [&](const float4 pv, const unsigned pn, const bool pown) {
            const unsigned q = fill + lane;
            if (lane < pn) {
                if (q < T) tile[cur][q] = pv;
                else tile[cur ^ 1][q - T] = pv;
            }
            if (pown) { own_cur = own_cur || fill < T; own_nxt = own_nxt || fill + pn > T; }
            fill += pn;
            if (fill >= T) {
                consume(cur, T / (unsigned)(NS * G), guard_all || own_cur);
                cur ^= 1; fill -= T;
                own_cur = own_nxt; own_nxt = false;
            }
        };
        float4 v0, v1;
        unsigned n0, n1;
        bool own0, own1;
        issue(v0, n0, own0);
        issue(v1, n1, own1);
        while (n0) {                                           // two pieces per trip: their loads are issued two pieces ahead
            stage(v0, n0, own0);
            issue(v0, n0, own0);
            if (!n1) break;
            stage(v1, n1, own1);
            issue(v1, n1, own1);
        }
        if (fill) {                                            // the stream's end: padded to whole trips with massless bodies far away
            const unsigned trips = (fill + (unsigned)(NS * G) - 1u) / (unsigned)(NS * G);
            if (lane < trips * (unsigned)(NS * G) - fill) tile[cur][fill + lane] = pad;
            consume(cur, trips, guard_all || own_cur);
        }
        flush();
        if (guard_all || __builtin_amdgcn_ballot_w64(!(rmin >= law_special_below<LAW>())) == 0ull) break;   // a NaN r^2 also redoes
        guard_all = true;
    }

    // ---- the lane groups' sums meet, group order; every output written once ----
    wave_lds_sync();
    if (g_raw == 0u) {
        double o[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = mine[k * kWave];
#pragma unroll 1
        for (unsigned q = 1; q < (unsigned)G; ++q) {       // (unrolled, the 12 x (G - 1) loads cost 170 registers)
            const double* const theirs = &red[0][q * L + p];
#pragma unroll
            for (int k = 0; k < 12; ++k) o[k] += theirs[k * kWave];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned tt = p + (unsigned)t * L;
            if (tt < c) {
                const uint32_t s0 = tb.first + tt;
                a.acc[s0] = o[t];
                a.acc[(size_t)a.slots + s0] = o[4 + t];
                if (D == 3) a.acc[2 * (size_t)a.slots + s0] = o[8 + t];
            }
        }
    }
}

// Waves per SIMD the register allocation is held to: left alone (one wave64 per workgroup) the allocator spends 300+ VGPRs on
// hoisted loads; at 5 waves (96 VGPRs) it spills inside the pair loop, at 4 waves (120 used) it does not.
#ifndef NBX_LEAF_WAVES
#define NBX_LEAF_WAVES 4
#endif
template <int D, int LAW>
__global__ __launch_bounds__(kWave * kWavesPerGroup, NBX_LEAF_WAVES) void leaf_pair_kernel(LeafArgs a) {
    __shared__ float4 tiles[kWavesPerGroup][2][kLeafTile];
    __shared__ double reds[kWavesPerGroup][12][kWave];
    const unsigned wave = threadIdx.x / (unsigned)kWave;
    const unsigned block = blockIdx.x * (unsigned)kWavesPerGroup + wave;   // wave-uniform
    if (block >= a.n_blocks) return;
    float4 (&tile)[2][kLeafTile] = tiles[wave];
    double (&red)[12][kWave] = reds[wave];
    const TargetBlock tb = a.blocks[__builtin_amdgcn_readfirstlane((int)block)];
    switch (tb.G) {   // wave-uniform; the lane-group count is a compile-time constant of the loop it selects
        case 1: leaf_block<D, LAW, 1>(a, tb, tile, red); break;
        case 2: leaf_block<D, LAW, 2>(a, tb, tile, red); break;
        case 3: leaf_block<D, LAW, 3>(a, tb, tile, red); break;
        case 4: leaf_block<D, LAW, 4>(a, tb, tile, red); break;
        case 5: leaf_block<D, LAW, 5>(a, tb, tile, red); break;
        case 6: leaf_block<D, LAW, 6>(a, tb, tile, red); break;
        case 7: leaf_block<D, LAW, 7>(a, tb, tile, red); break;
        case 8: leaf_block<D, LAW, 8>(a, tb, tile, red); break;
        case 9: leaf_block<D, LAW, 9>(a, tb, tile, red); break;
        case 10: leaf_block<D, LAW, 10>(a, tb, tile, red); break;
        case 12: leaf_block<D, LAW, 12>(a, tb, tile, red); break;
        default: leaf_block<D, LAW, 16>(a, tb, tile, red); break;
    }
}

// staged Body<D> AoS fp64 (host order) -> leaf-ordered {x, y, z, m} fp32
__global__ __launch_bounds__(256) void leaf_gather_kernel(const double* __restrict__ raw, size_t stride_d, int dim,
                                                          const uint32_t* __restrict__ leaf_bodies, uint32_t slots,
                                                          float4* __restrict__ xm) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= slots) return;
    const double* __restrict__ b = raw + (size_t)leaf_bodies[s] * stride_d;
    xm[s] = make_float4((float)b[0], (float)b[1], dim == 3 ? (float)b[2] : 0.0f, (float)b[2 * dim]);
}

// forces_out[body] = sign * (G m_body) * acc[slot]   (fp64; every body belongs to at most one leaf)
__global__ __launch_bounds__(256) void leaf_scatter_kernel(const double* __restrict__ acc, const double* __restrict__ raw, size_t stride_d,
                                                           int dim, const uint32_t* __restrict__ leaf_bodies, uint32_t slots, double signedG,
                                                           double* __restrict__ forces) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= slots) return;
    const uint32_t body = leaf_bodies[s];
    const double gm = signedG * raw[(size_t)body * stride_d + 2 * dim];
    for (int k = 0; k < dim; ++k) forces[(size_t)body * dim + k] = gm * acc[(size_t)k * slots + s];
}

typedef void (*LeafKernel)(LeafArgs);
LeafKernel pick(int dim, int law) {
    static const LeafKernel table[2][3] = {
        {leaf_pair_kernel<2, NBX_LAW_BRUTE>, leaf_pair_kernel<2, NBX_LAW_TREE_LEAF>, leaf_pair_kernel<2, NBX_LAW_FMM_P2P>},
        {leaf_pair_kernel<3, NBX_LAW_BRUTE>, leaf_pair_kernel<3, NBX_LAW_TREE_LEAF>, leaf_pair_kernel<3, NBX_LAW_FMM_P2P>}};
    return table[dim - 2][law];
}

struct DeviceBuffers {   // frees whatever was allocated when the call leaves, on every path
    std::vector<void*> ptrs;
    hipStream_t stream = nullptr;
    int device = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~DeviceBuffers() {
        const bool idle = stream && hipStreamSynchronize(stream) == hipSuccess;
        for (void* p : ptrs) (void)hipFree(p);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (idle) nbx::park_stream(device, stream);   // back to the pool (nbx_api.hip): a stream costs more than this call's kernels
        else if (stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace

extern "C" int nbx_leaf_pair_forces(const void* bodies, size_t n, int dim, size_t stride_bytes, const uint32_t* leaf_offsets,
                                    const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets,
                                    const uint32_t* list_sources, int law, double G, int device, double* forces_out,
                                    float* kernel_ms) {
    if (kernel_ms) *kernel_ms = 0.0f;
    if (dim != 2 && dim != 3) return fail(NBX_ERR_INVALID, "dim must be 2 or 3");
    if (law < NBX_LAW_BRUTE || law > NBX_LAW_FMM_P2P) return fail(NBX_ERR_INVALID, "unknown law");
    if ((!bodies || !forces_out) && n) return fail(NBX_ERR_INVALID, "null argument");
    if (n > ((size_t)1 << 31) || n_leaves > ((size_t)1 << 31)) return fail(NBX_ERR_INVALID, "too many bodies / leaves");
    const size_t min_stride = (size_t)(2 * dim + 1) * sizeof(double);
    if (stride_bytes < min_stride || stride_bytes % sizeof(double) != 0)
        return fail(NBX_ERR_INVALID, "body stride must be a multiple of 8 and >= sizeof(Body<dim>)");
    if (n_leaves && (!leaf_offsets || !list_offsets)) return fail(NBX_ERR_INVALID, "null leaf arrays");
    // ---- host-side validation of the CSR structure: every index the kernel will follow is checked here ----
    const size_t slots = n_leaves ? leaf_offsets[n_leaves] : 0;
    const size_t n_list = n_leaves ? list_offsets[n_leaves] : 0;
    if (n_leaves && (leaf_offsets[0] != 0 || list_offsets[0] != 0)) return fail(NBX_ERR_INVALID, "CSR offsets must start at 0");
    for (size_t l = 0; l < n_leaves; ++l)
        if (leaf_offsets[l + 1] < leaf_offsets[l] || list_offsets[l + 1] < list_offsets[l]) return fail(NBX_ERR_INVALID, "CSR offsets must be non-decreasing");
    if ((slots && !leaf_bodies) || (n_list && !list_sources)) return fail(NBX_ERR_INVALID, "null leaf arrays");
    {
        std::vector<unsigned char> seen(n, 0);
        for (size_t s = 0; s < slots; ++s) {
            const uint32_t b = leaf_bodies[s];
            if (b >= n) return fail(NBX_ERR_INVALID, "leaf_bodies entry out of range");
            if (seen[b]) return fail(NBX_ERR_INVALID, "a body may belong to at most one leaf");
            seen[b] = 1;
        }
    }
    for (size_t e = 0; e < n_list; ++e)
        if (list_sources[e] >= n_leaves) return fail(NBX_ERR_INVALID, "list_sources entry out of range");
    int ndev = 0;
    int rc = nbx_device_count(&ndev);
    if (rc != NBX_OK) return rc;
    if (device < 0 || device >= ndev) return fail(NBX_ERR_NO_DEVICE, "device ordinal out of range");
    if (slots == 0) {   // no leaf holds a body: every force is zero (otherwise the device array, zeroed there, is copied out whole)
        for (size_t i = 0; i < n * (size_t)dim; ++i) forces_out[i] = 0.0;
        return NBX_OK;
    }

    // Target blocks: a leaf of c targets is cut into k pieces of ceil(c / k), four targets per lane.  A piece of t targets runs
    // G lane groups (the largest of {1..10, 12, 16} within 64 / ceil(t / 4)), every lane making (sources / G) trips of 28 VALU,
    // and stages the leaf's source list once (about half an instruction per body, plus the block's fixed part): k minimises
    // k x (sources x (28 / G + 0.5) + 300).  32 targets: one block at 8 groups; 34: one block at 7.
    auto groups_of = [&](uint32_t piece) -> uint32_t {
        const uint32_t lanes = (piece + 3) / 4;
        if (lanes > (uint32_t)kWave) return 0;
        uint32_t g = (uint32_t)kWave / lanes;
        if (g > (uint32_t)kMaxGroups) g = (uint32_t)kMaxGroups;
        if (g == 11) g = 10;
        if (g > 12 && g < 16) g = 12;
        return g;
    };
    std::vector<TargetBlock> blocks;
    for (size_t l = 0; l < n_leaves; ++l) {
        const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
        if (!c) continue;
        double sources = 0.0;
        for (uint32_t e = list_offsets[l]; e < list_offsets[l + 1]; ++e)
            sources += (double)(leaf_offsets[list_sources[e] + 1] - leaf_offsets[list_sources[e]]);
        const uint32_t k_min = (c + (uint32_t)(kWave * kMaxTargetsPerLane) - 1) / (uint32_t)(kWave * kMaxTargetsPerLane);
        uint32_t best_k = k_min;
        double best = 1e300;
        for (uint32_t k = k_min; k <= k_min + 3 && k <= c; ++k) {
            const uint32_t g = groups_of((c + k - 1) / k);
            if (!g) continue;
            const double v = (double)k * (sources * (28.0 / g + 0.5) + 300.0);
            if (v < best * (1.0 - 1e-9)) { best = v; best_k = k; }
        }
        const uint32_t piece = (c + best_k - 1) / best_k;
        for (uint32_t f = leaf_offsets[l]; f < leaf_offsets[l + 1]; f += piece) {
            TargetBlock tb = {};
            tb.leaf = (uint32_t)l; tb.first = f; tb.count = (leaf_offsets[l + 1] - f < piece) ? leaf_offsets[l + 1] - f : piece;
            tb.L = (tb.count + 3) / 4;
            tb.G = groups_of(tb.count);
            tb.inv_L = 65536u / tb.L + 1u;
            blocks.push_back(tb);
        }
    }

    NBX_HIP_TRY(hipSetDevice(device));
    DeviceBuffers d;
    d.device = device;
    NBX_HIP_TRY(nbx::take_stream(device, &d.stream));
    NBX_HIP_TRY(hipEventCreate(&d.ev0));
    NBX_HIP_TRY(hipEventCreate(&d.ev1));
    // one allocation for the call's ten device arrays (each hipFree of a large buffer costs 0.2 ms on this runtime)
    const size_t sizes[10] = {n * stride_bytes, slots * sizeof(float4), 0, (size_t)dim * slots * sizeof(double),
                              n * (size_t)dim * sizeof(double), (n_leaves + 1) * sizeof(uint32_t), slots * sizeof(uint32_t),
                              (n_leaves + 1) * sizeof(uint32_t), n_list * sizeof(uint32_t), blocks.size() * sizeof(TargetBlock)};
    size_t offs[10], total_bytes = 0;
    for (int i = 0; i < 10; ++i) { offs[i] = total_bytes; total_bytes += (sizes[i] + 255) / 256 * 256 + 256; }
    char* arena = nullptr;
    NBX_HIP_TRY(hipMalloc((void**)&arena, total_bytes));
    d.ptrs.push_back(arena);
    double* raw = reinterpret_cast<double*>(arena + offs[0]);
    float4* xm = reinterpret_cast<float4*>(arena + offs[1]);
    double* acc = reinterpret_cast<double*>(arena + offs[3]);
    double* dforces = reinterpret_cast<double*>(arena + offs[4]);
    uint32_t* d_lo = reinterpret_cast<uint32_t*>(arena + offs[5]);
    uint32_t* d_lb = reinterpret_cast<uint32_t*>(arena + offs[6]);
    uint32_t* d_so = reinterpret_cast<uint32_t*>(arena + offs[7]);
    uint32_t* d_ss = reinterpret_cast<uint32_t*>(arena + offs[8]);
    TargetBlock* d_blocks = reinterpret_cast<TargetBlock*>(arena + offs[9]);
    NBX_HIP_TRY(hipMemcpyAsync(raw, bodies, n * stride_bytes, hipMemcpyHostToDevice, d.stream));
    NBX_HIP_TRY(hipMemcpyAsync(d_lo, leaf_offsets, (n_leaves + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    NBX_HIP_TRY(hipMemcpyAsync(d_lb, leaf_bodies, slots * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    NBX_HIP_TRY(hipMemcpyAsync(d_so, list_offsets, (n_leaves + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    if (n_list) NBX_HIP_TRY(hipMemcpyAsync(d_ss, list_sources, n_list * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    NBX_HIP_TRY(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(TargetBlock), hipMemcpyHostToDevice, d.stream));
    NBX_HIP_TRY(hipMemsetAsync(dforces, 0, n * (size_t)dim * sizeof(double), d.stream));
    (void)hipGetLastError();
    const unsigned gs = (unsigned)((slots + 255) / 256);
    hipLaunchKernelGGL(leaf_gather_kernel, dim3(gs), dim3(256), 0, d.stream, raw, stride_bytes / sizeof(double), dim, d_lb, (uint32_t)slots, xm);
    NBX_HIP_TRY(hipGetLastError());
    LeafArgs a;
    a.xm = xm; a.slots = (uint32_t)slots; a.leaf_offsets = d_lo; a.list_offsets = d_so; a.list_sources = d_ss;
    a.blocks = d_blocks; a.n_blocks = (uint32_t)blocks.size(); a.acc = acc;
    NBX_HIP_TRY(hipEventRecord(d.ev0, d.stream));
    hipLaunchKernelGGL(pick(dim, law), dim3((unsigned)((blocks.size() + kWavesPerGroup - 1) / kWavesPerGroup)), dim3(kWave * kWavesPerGroup), 0, d.stream, a);
    NBX_HIP_TRY(hipGetLastError());
    NBX_HIP_TRY(hipEventRecord(d.ev1, d.stream));
    const double signedG = (law == NBX_LAW_BRUTE) ? -G : G;   // brute force: forces[i] -= f (methods.cpp:131); tree codes: += (attractive)
    hipLaunchKernelGGL(leaf_scatter_kernel, dim3(gs), dim3(256), 0, d.stream, acc, raw, stride_bytes / sizeof(double), dim, d_lb, (uint32_t)slots,
                       signedG, dforces);
    NBX_HIP_TRY(hipGetLastError());
    NBX_HIP_TRY(hipMemcpyAsync(forces_out, dforces, n * (size_t)dim * sizeof(double), hipMemcpyDeviceToHost, d.stream));
    NBX_HIP_TRY(hipStreamSynchronize(d.stream));
    if (kernel_ms) NBX_HIP_TRY(hipEventElapsedTime(kernel_ms, d.ev0, d.ev1));
    return NBX_OK;
}
