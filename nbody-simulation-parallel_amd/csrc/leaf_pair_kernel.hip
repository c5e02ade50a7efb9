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

constexpr int kWave = 64;               // lanes per workgroup: one wave64
constexpr int kMaxTargetsPerLane = 4;    // a block holds up to 64 x 4 targets
constexpr int kMaxGroups = 16;          // lane groups that split the sources of a block with few targets
constexpr int kLeafTile = 64;               // source bodies per LDS tile
constexpr int kLeafTileSlots = 96;          // float4 slots per tile buffer: groups x (padded) bodies per group <= 81

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
    uint32_t count;    // <= 64 * tpl
    uint32_t tpl;      // targets per lane: 2 or 4
    uint32_t L;        // lanes per group = ceil(count / tpl)
    uint32_t G;        // lane groups = min(64 / L, kMaxGroups)
    uint32_t TGp;      // float4 slots per group in a tile: ceil(64 / G), made odd
    uint32_t inv_L;    // q / L = (q * inv_L) >> 16 for q < 128  (inv = 65536 / d + 1)
    uint32_t inv_G;
    uint32_t pad_[3];
};
static_assert(sizeof(TargetBlock) == 48, "TargetBlock is read with scalar loads");

struct LeafArgs {
    const float4* __restrict__ xm;     // [slots] leaf-ordered {x, y, z (0 in 2D), m}: one 16-byte load stages a body
    uint32_t slots;
    const uint32_t* __restrict__ leaf_offsets;
    const uint32_t* __restrict__ list_offsets;
    const uint32_t* __restrict__ list_sources;
    const TargetBlock* __restrict__ blocks;
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

// Below this r^2 a pair leaves the plain m_j d / r^4 form under the law (skip or smoothing); leaf_weight decides how.
template <int LAW>
__device__ __forceinline__ constexpr float law_special_below() {
    return LAW == NBX_LAW_BRUTE ? kR2SkipF : LAW == NBX_LAW_TREE_LEAF ? kTreeSkipF : kSmoothF;
}

// One wave64 = up to 64 x TPL targets of one leaf against that leaf's source list.
//  * TPL = 2 or 4 TARGETS PER LANE as packed fp32 pairs (lane p holds targets p, p + L, ... of the block, L = ceil(count /
//    TPL)), every source a broadcast: per source and target pair 3 v_pk_add (d), v_pk_mul + 2 v_pk_fma (r^2), 2 v_rcp,
//    2 v_pk_mul (w^2, .m with the op_sel form of the brute-force kernel), 3 v_pk_fma (accumulate) and one v_min3 = 14 VALU
//    per two pair terms.
//  * A block of few targets runs G = floor(64 / L) <= 16 LANE GROUPS that split the sources of every tile G ways; their
//    fp64 sums meet in LDS at the end, in group order (deterministic).  The host picks TPL and the number of blocks a leaf
//    is cut into so that (blocks) x (sources) x (7 TPL / G + staging) is smallest.
//  * The source list is staged LEAF BY LEAF: lane e holds list entry e (source leaf -> slot range), the wave walks the
//    entries with v_readlane, and ONE SOURCE LEAF (a piece of up to 64 bodies of it) IS ONE TILE: one 16-byte global load
//    per lane, issued two pieces ahead, one conflict-free ds_write_b128 into the lane's own slot.  No body stream, no
//    prefix sum, no per-body search of the list (24 % of the instructions of round 2's kernel).
//  * The tile is laid out GROUP-MAJOR in LDS (body q at slot (q mod G) * TGp + q / G, TGp odd): lane group g reads
//    consecutive 16-byte slots with immediate offsets -- no address arithmetic in the pair loop, distinct groups on
//    distinct banks -- one ds_read_b128 per source and lane.
//  * The law's special cases (skip / smoothing below ~1e-5 separation) cost one v_min3 per two pair terms in the common
//    path: it keeps the smallest r^2 a lane has seen.  Only the piece that is the target's own leaf (every body meets
//    itself there) runs the GUARDED loop -- a compare and select per pair term, and for the FMM law the smoothing branch
//    behind a wave vote.  If at the end some lane's minimum lies below the law's threshold -- two distinct bodies of
//    different leaves closer than 3e-5: next to never -- the whole block is redone with the guarded loop throughout.
template <int D, int LAW, int TPL>
__device__ __forceinline__ void leaf_block(const LeafArgs& a, const TargetBlock tb, float4 (&tile)[kLeafTileSlots], double (&red)[12][kWave]) {
    constexpr int PAIRS = TPL / 2;
    const unsigned lane = threadIdx.x;
    // ---- block geometry (wave-uniform, from the host's block table) ----
    const unsigned c = tb.count, L = tb.L, G = tb.G, TGp = tb.TGp, inv_L = tb.inv_L, inv_G = tb.inv_G;
    const unsigned g_raw = (lane * inv_L) >> 16, p = lane - g_raw * L;       // lane / L, lane % L
    const unsigned g = g_raw < G ? g_raw : 0u;                  // lanes left over compute along with group 0, unused
    const unsigned qa = (lane * inv_G) >> 16, qb = ((lane + 64u) * inv_G) >> 16;
    const unsigned slot_a = (lane - qa * G) * TGp + qa;         // tile slot of body `lane` of a piece ...
    const unsigned slot_b = (lane + 64u - qb * G) * TGp + qb;   // ... and of the pad bodies 64 .. 64 + G - 1
    f2 ix[PAIRS], iy[PAIRS], iz[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const unsigned t0 = p + (unsigned)(2 * q) * L, t1 = t0 + L;
        const float4 far = make_float4(-kFar, -kFar, (D == 3) ? -kFar : 0.0f, 0.0f);   // pad target: every weight underflows to 0
        const float4 m0 = t0 < c ? a.xm[tb.first + t0] : far, m1 = t1 < c ? a.xm[tb.first + t1] : far;
        ix[q] = f2{m0.x, m1.x}; iy[q] = f2{m0.y, m1.y}; iz[q] = f2{(D == 3) ? m0.z : 0.0f, (D == 3) ? m1.z : 0.0f};
    }
    const float4 pad = make_float4(kFar, kFar, (D == 3) ? kFar : 0.0f, 0.0f);
    double o[3][TPL];                                           // fp64 sums [component][target of this lane]
    f2 ax[PAIRS], ay[PAIRS], az[PAIRS];                         // fp32 sums since the last flush
    unsigned pending;
    float rmin;                                                 // smallest r^2 seen by the unguarded loop

    auto flush = [&]() {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            o[0][2 * q] += (double)ax[q].x; o[0][2 * q + 1] += (double)ax[q].y;
            o[1][2 * q] += (double)ay[q].x; o[1][2 * q + 1] += (double)ay[q].y;
            if (D == 3) { o[2][2 * q] += (double)az[q].x; o[2][2 * q + 1] += (double)az[q].y; }
            ax[q] = ay[q] = az[q] = f2{0.f, 0.f};
        }
        pending = 0;
    };
    // NS sources against the lane's PAIRS target pairs, stage by stage (NS * PAIRS independent chains)
    auto fast = [&](const float4* __restrict__ src, auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
        f2 dx[NS][PAIRS], dy[NS][PAIRS], dz[NS][PAIRS], r2[NS][PAIRS], w[NS][PAIRS], szm[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float4 s = src[k];
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
        constexpr float T = law_special_below<LAW>();
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            const f2 dx = f2{s.x, s.x} - ix[q], dy = f2{s.y, s.y} - iy[q], dz = (D == 3) ? f2{s.z, s.z} - iz[q] : f2{0.f, 0.f};
            f2 r2 = dx * dx;
            r2 = __builtin_elementwise_fma(dy, dy, r2);
            if (D == 3) r2 = __builtin_elementwise_fma(dz, dz, r2);
            f2 w;
            if (LAW == NBX_LAW_FMM_P2P &&
                (__builtin_amdgcn_ballot_w64(r2.x < T && r2.x > 0.0f) | __builtin_amdgcn_ballot_w64(r2.y < T && r2.y > 0.0f)) != 0ull) {
                w = f2{leaf_weight<D, LAW>(r2.x, s.w, dx.x, dy.x, dz.x), leaf_weight<D, LAW>(r2.y, s.w, dx.y, dy.y, dz.y)};
            } else {   // below the threshold: skipped (brute force, tree leaf), or the same position (FMM: r^2 = 0) -- weight 0
                const f2 r2g = {(r2.x < T) ? __builtin_inff() : r2.x, (r2.y < T) ? __builtin_inff() : r2.y};
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
    // a piece = up to 64 bodies of one source leaf: (v, n, own) = this lane's body, the piece's size, "it is the target's leaf".
    // Lanes past the piece's end load its last body again (an address inside the leaf, no branch); the tile write gives them
    // mass 0, so they pad the last trip: a massless copy of a real source contributes exactly 0.
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
            n = (len - off_next < (uint32_t)kWave) ? (unsigned)(len - off_next) : (unsigned)kWave;
            own = src_leaf == tb.leaf;
            const float4* __restrict__ base = a.xm + (first + off_next);   // uniform: scalar base + per-lane 32-bit offset
            v = base[lane < n ? lane : n - 1u];
            off_next += n;
            return;
        }
    };
    bool guard_all = false;
    for (int pass = 0; pass < 2; ++pass) {                     // pass 1 only after a sub-threshold pair outside the own leaf
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < TPL; ++t) o[k][t] = 0.0;
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) ax[q] = ay[q] = az[q] = f2{0.f, 0.f};
        pending = 0;
        rmin = __builtin_inff();
        e_base = e_begin; n_ent = e_next = 0; off_next = 0;
        float4 v0, v1;
        unsigned n0, n1;
        bool own0, own1;
        issue(v0, n0, own0);
        issue(v1, n1, own1);
        const float4* __restrict__ src = &tile[g * TGp];      // this lane group's sources: consecutive slots
        auto one_piece = [&](const float4 pv, const unsigned pn, const bool pown) {   // the piece is the tile: stage, then the pair loop
            const unsigned trips = ((pn + G - 1u) * inv_G) >> 16;
            __syncthreads();                                   // the previous tile has been consumed
            tile[slot_a] = make_float4(pv.x, pv.y, pv.z, lane < pn ? pv.w : 0.0f);
            if (trips * G > (unsigned)kWave && lane < trips * G - (unsigned)kWave) tile[slot_b] = pad;
            __syncthreads();
            if (pending + trips > 256u) flush();
            if (guard_all || pown) {
                for (unsigned k = 0; k < trips; ++k) guarded1(src[k]);
            } else {
                constexpr int NS = (TPL == 2) ? 4 : 2;
                unsigned k = 0;
                for (; k + (unsigned)NS <= trips; k += (unsigned)NS) fast(src + k, std::integral_constant<int, NS>{});
                for (; k < trips; ++k) fast(src + k, std::integral_constant<int, 1>{});
            }
            pending += trips;
        };
        while (n0) {                                           // two pieces per trip: their loads are issued two pieces ahead
            one_piece(v0, n0, own0);
            issue(v0, n0, own0);
            if (!n1) break;
            one_piece(v1, n1, own1);
            issue(v1, n1, own1);
        }
        flush();
        if (guard_all || __builtin_amdgcn_ballot_w64(!(rmin >= law_special_below<LAW>())) == 0ull) break;   // a NaN r^2 also redoes
        guard_all = true;
    }

    // ---- the lane groups' sums meet, group order ----
    if (G > 1u) {
        __syncthreads();
        double* const mine = &red[0][lane];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < TPL; ++t) mine[(k * TPL + t) * kWave] = o[k][t];
        __syncthreads();
        if (g_raw == 0u)
            for (unsigned q = 1; q < G; ++q) {
                const double* const theirs = &red[0][q * L + p];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int t = 0; t < TPL; ++t) o[k][t] += theirs[(k * TPL + t) * kWave];
            }
    }
    if (g_raw == 0u) {
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const unsigned tt = p + (unsigned)t * L;
            if (tt < c) {
                const uint32_t s0 = tb.first + tt;
                a.acc[s0] = o[0][t];
                a.acc[(size_t)a.slots + s0] = o[1][t];
                if (D == 3) a.acc[2 * (size_t)a.slots + s0] = o[2][t];
            }
        }
    }
}

template <int D, int LAW>
__global__ __launch_bounds__(kWave) void leaf_pair_kernel(LeafArgs a) {
    __shared__ float4 tile[kLeafTileSlots];
    __shared__ double red[12][kWave];
    const TargetBlock tb = a.blocks[blockIdx.x];
    if (tb.tpl == 4u) leaf_block<D, LAW, 4>(a, tb, tile, red);   // wave-uniform
    else leaf_block<D, LAW, 2>(a, tb, tile, red);
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

    // Target blocks: a leaf of c targets is cut into k pieces of ceil(c / k), each run at 2 or 4 targets per lane.  A piece of
    // t targets at TPL per lane runs G = min(64 / ceil(t / TPL), 16) lane groups, every lane making (sources / G) trips of
    // 7 TPL VALU, and stages the leaf's source list once (about one instruction per two bodies, plus the block's fixed
    // part): (k, TPL) minimise k x (sources x (7 TPL / G + 0.5) + 200).  32 targets: one block, 4 per lane, 8 groups;
    // 34 targets: one block, 4 per lane, 7 groups (2 per lane would leave 3 groups).
    auto groups_of = [&](uint32_t piece, uint32_t tpl) -> uint32_t {
        const uint32_t lanes = (piece + tpl - 1) / tpl;
        if (lanes > (uint32_t)kWave) return 0;
        const uint32_t g = (uint32_t)kWave / lanes;
        return g > (uint32_t)kMaxGroups ? (uint32_t)kMaxGroups : g;
    };
    std::vector<TargetBlock> blocks;
    for (size_t l = 0; l < n_leaves; ++l) {
        const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
        if (!c) continue;
        double sources = 0.0;
        for (uint32_t e = list_offsets[l]; e < list_offsets[l + 1]; ++e)
            sources += (double)(leaf_offsets[list_sources[e] + 1] - leaf_offsets[list_sources[e]]);
        const uint32_t k_min = (c + (uint32_t)(kWave * kMaxTargetsPerLane) - 1) / (uint32_t)(kWave * kMaxTargetsPerLane);
        uint32_t best_k = k_min, best_tpl = 4;
        double best = 1e300;
        for (uint32_t k = k_min; k <= k_min + 3 && k <= c; ++k)
            for (uint32_t tpl = 4; tpl >= 2; tpl -= 2) {       // ties go to 4 per lane: half the trips
                const uint32_t g = groups_of((c + k - 1) / k, tpl);
                if (!g) continue;
                const double v = (double)k * (sources * (7.0 * tpl / g + 0.5) + 200.0);
                if (v < best * (1.0 - 1e-9)) { best = v; best_k = k; best_tpl = tpl; }
            }
        const uint32_t piece = (c + best_k - 1) / best_k;
        for (uint32_t f = leaf_offsets[l]; f < leaf_offsets[l + 1]; f += piece) {
            TargetBlock tb = {};
            tb.leaf = (uint32_t)l; tb.first = f; tb.count = (leaf_offsets[l + 1] - f < piece) ? leaf_offsets[l + 1] - f : piece;
            tb.tpl = best_tpl;
            tb.L = (tb.count + best_tpl - 1) / best_tpl;
            tb.G = groups_of(tb.count, best_tpl);
            const uint32_t tg = ((uint32_t)kLeafTile + tb.G - 1) / tb.G;
            tb.TGp = tg | 1u;
            tb.inv_L = 65536u / tb.L + 1u;
            tb.inv_G = 65536u / tb.G + 1u;
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
    a.blocks = d_blocks; a.acc = acc;
    NBX_HIP_TRY(hipEventRecord(d.ev0, d.stream));
    hipLaunchKernelGGL(pick(dim, law), dim3((unsigned)blocks.size()), dim3(kWave), 0, d.stream, a);
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
