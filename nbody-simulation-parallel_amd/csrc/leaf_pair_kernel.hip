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
#include <vector>

using namespace nbx;

namespace {

constexpr int kWave = 64;               // lanes per workgroup: one wave64
constexpr int kTargetsPerBlock = 128;   // two targets per lane
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

struct TargetBlock {
    uint32_t leaf;     // target leaf
    uint32_t first;    // first target slot (leaf order)
    uint32_t count;    // <= kTargetsPerBlock
};

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

// One wave64 = up to 128 targets of one leaf against that leaf's source list.
//  * TWO TARGETS PER LANE as packed fp32 pairs (lane p holds targets p and p + L of the block, L = ceil(count / 2)), every
//    source a broadcast: per source and lane 3 v_pk_add (d), v_pk_mul + 2 v_pk_fma (r^2), 2 v_rcp, 2 v_pk_mul (w^2, .m with
//    the op_sel form of the brute-force kernel), 3 v_pk_fma (accumulate) and one v_min3 = 14 VALU per two pair terms.
//  * A block of few targets (L <= 32 lanes) runs G = floor(64 / L) <= 16 LANE GROUPS that split the sources of every tile
//    G ways; their fp64 sums meet in LDS at the end, in group order (deterministic).  The host cuts a leaf into the number
//    of blocks that minimises (blocks) x (sources) x (14 / G + staging) -- e.g. 34 targets: two blocks of 17 at 7 groups
//    instead of one at 3.
//  * The tile is laid out GROUP-MAJOR in LDS (body q of the tile at slot (q mod G) * TGp + q / G, TGp odd): lane group g
//    reads consecutive 16-byte slots with immediate offsets -- no address arithmetic in the pair loop, distinct groups on
//    distinct banks -- one ds_read_b128 per source and lane, one conflict-free ds_write_b128 per staged body.
//  * The source list is staged LEAF BY LEAF, not body by body: lane e holds list entry e (source leaf -> slot range), the
//    wave walks the entries with v_readlane, and a leaf's bodies go from one 16-byte global load per lane straight to their
//    tile slots; two leaves are in flight while a tile is consumed.  (Round 2 turned the list into a body stream by a
//    prefix sum in LDS and searched the entry of every staged body: 24 % of its instructions.)
//  * The law's special cases (skip / smoothing below ~1e-5 separation) and a body meeting itself cost nothing in the common
//    path: the v_min3 keeps the smallest r^2 a lane saw in the tile; tiles that hold bodies of the target's own leaf, and
//    any tile after which some lane's minimum lies below the law's threshold (the sums are then put back to what they
//    were before the tile), go through the GUARDED loop -- a compare and select per pair term, and for the FMM law the
//    smoothing branch behind a wave vote.
template <int D, int LAW>
__global__ __launch_bounds__(kWave) void leaf_pair_kernel(LeafArgs a) {
    __shared__ float4 tile[2][kLeafTileSlots];
    __shared__ double red[6][kWave];
    const unsigned lane = threadIdx.x;
    const TargetBlock tb = a.blocks[blockIdx.x];
    // ---- block geometry (wave-uniform) ----
    const unsigned c = tb.count;
    const unsigned L = (c + 1u) >> 1;
    const unsigned fit = (unsigned)kWave / (L ? L : 1u);
    const unsigned G = fit < (unsigned)kMaxGroups ? fit : (unsigned)kMaxGroups;
    const unsigned TG = ((unsigned)kLeafTile + G - 1u) / G;
    const unsigned TGp = TG | 1u;                               // odd: lane groups land on distinct LDS banks
    const unsigned inv_g = 65536u / G + 1u;                     // q / G = (q * inv_g) >> 16 for q < 128
    const unsigned p = lane % L, g_raw = lane / L;
    const bool valid = g_raw < G;                               // lanes left over compute along with group 0, unused
    const unsigned g = valid ? g_raw : 0u;
    const bool has1 = p + L < c;
    const float4 me0 = a.xm[tb.first + p];
    const float4 me1 = has1 ? a.xm[tb.first + p + L] : make_float4(-kFar, -kFar, (D == 3) ? -kFar : 0.0f, 0.0f);
    const f2 ix = {me0.x, me1.x}, iy = {me0.y, me1.y}, iz = {(D == 3) ? me0.z : 0.0f, (D == 3) ? me1.z : 0.0f};
    double o[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};               // fp64 sums: x0 x1 y0 y1 z0 z1
    f2 ax = {0.f, 0.f}, ay = {0.f, 0.f}, az = {0.f, 0.f};       // fp32 sums since the last flush
    unsigned pending = 0;                                       // terms in them
    const float4 pad = make_float4(kFar, kFar, (D == 3) ? kFar : 0.0f, 0.0f);
    auto slot_of = [&](unsigned q) -> unsigned {                // body q of a tile -> its float4 slot
        const unsigned qd = (q * inv_g) >> 16;
        return (q - qd * G) * TGp + qd;
    };

    // ---- the pair loops over one staged tile: `trips` sources per lane group ----
    auto flush = [&]() {
        o[0] += (double)ax.x; o[1] += (double)ax.y; o[2] += (double)ay.x; o[3] += (double)ay.y;
        if (D == 3) { o[4] += (double)az.x; o[5] += (double)az.y; }
        ax = ay = az = f2{0.f, 0.f};
        pending = 0;
    };
    auto fast4 = [&](const float4* __restrict__ src, float& rmin) {   // four sources, stage by stage (four independent chains)
        f2 dx[4], dy[4], dz[4], r2[4], w[4], szm[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 s = src[q];
            szm[q] = f2{s.z, s.w};
            dx[q] = f2{s.x, s.x} - ix;
            dy[q] = f2{s.y, s.y} - iy;
            dz[q] = (D == 3) ? f2{s.z, s.z} - iz : f2{0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) r2[q] = dx[q] * dx[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) r2[q] = __builtin_elementwise_fma(dy[q], dy[q], r2[q]);
        if (D == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r2[q] = __builtin_elementwise_fma(dz[q], dz[q], r2[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { w[q].x = __builtin_amdgcn_rcpf(r2[q].x); w[q].y = __builtin_amdgcn_rcpf(r2[q].y); }
#pragma unroll
        for (int q = 0; q < 4; ++q) rmin = __builtin_fminf(__builtin_fminf(rmin, r2[q].x), r2[q].y);   // v_min3_f32
#pragma unroll
        for (int q = 0; q < 4; ++q) r2[q] = w[q] * w[q];
        // the mass is the HIGH half of the source's {z, m} register pair: op_sel spelled out (force_kernel.hip), applied to
        // w^2 (plain code, hazards handled by the compiler), never directly to a v_rcp result
#pragma unroll
        for (int q = 0; q < 4; ++q) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(w[q]) : "v"(szm[q]), "v"(r2[q]));
#pragma unroll
        for (int q = 0; q < 4; ++q) ax = __builtin_elementwise_fma(w[q], dx[q], ax);
#pragma unroll
        for (int q = 0; q < 4; ++q) ay = __builtin_elementwise_fma(w[q], dy[q], ay);
        if (D == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) az = __builtin_elementwise_fma(w[q], dz[q], az);
        }
    };
    auto fast1 = [&](const float4 s, float& rmin) {
        const f2 dx = f2{s.x, s.x} - ix, dy = f2{s.y, s.y} - iy, dz = (D == 3) ? f2{s.z, s.z} - iz : f2{0.f, 0.f};
        f2 r2 = dx * dx;
        r2 = __builtin_elementwise_fma(dy, dy, r2);
        if (D == 3) r2 = __builtin_elementwise_fma(dz, dz, r2);
        rmin = __builtin_fminf(__builtin_fminf(rmin, r2.x), r2.y);
        f2 w = {__builtin_amdgcn_rcpf(r2.x), __builtin_amdgcn_rcpf(r2.y)};
        w = w * w;
        w = w * f2{s.w, s.w};
        ax = __builtin_elementwise_fma(w, dx, ax);
        ay = __builtin_elementwise_fma(w, dy, ay);
        if (D == 3) az = __builtin_elementwise_fma(w, dz, az);
    };
    auto guarded1 = [&](const float4 s) {   // exact law per pair term: compare and select; FMM smoothing behind a wave vote
        const f2 dx = f2{s.x, s.x} - ix, dy = f2{s.y, s.y} - iy, dz = (D == 3) ? f2{s.z, s.z} - iz : f2{0.f, 0.f};
        f2 r2 = dx * dx;
        r2 = __builtin_elementwise_fma(dy, dy, r2);
        if (D == 3) r2 = __builtin_elementwise_fma(dz, dz, r2);
        f2 w;
        constexpr float T = law_special_below<LAW>();
        if (LAW == NBX_LAW_FMM_P2P &&
            (__builtin_amdgcn_ballot_w64(r2.x < T && r2.x > 0.0f) | __builtin_amdgcn_ballot_w64(r2.y < T && r2.y > 0.0f)) != 0ull) {
            w = f2{leaf_weight<D, LAW>(r2.x, s.w, dx.x, dy.x, dz.x), leaf_weight<D, LAW>(r2.y, s.w, dx.y, dy.y, dz.y)};
        } else {   // below the threshold: skipped (brute force, tree leaf), or the same position (FMM: r^2 = 0) -- weight 0
            const f2 r2g = {(r2.x < T) ? __builtin_inff() : r2.x, (r2.y < T) ? __builtin_inff() : r2.y};
            w = f2{__builtin_amdgcn_rcpf(r2g.x), __builtin_amdgcn_rcpf(r2g.y)};
            w = w * w;
            w = w * f2{s.w, s.w};
        }
        ax = __builtin_elementwise_fma(w, dx, ax);
        ay = __builtin_elementwise_fma(w, dy, ay);
        if (D == 3) az = __builtin_elementwise_fma(w, dz, az);
    };
    auto consume = [&](int buf, unsigned cnt, bool own_leaf_inside) {   // all arguments wave-uniform
        const unsigned trips = (cnt + G - 1u) / G;
        if (lane < trips * G - cnt) tile[buf][slot_of(cnt + lane)] = pad;   // fill the last trip: massless bodies far away
        __syncthreads();
        if (pending + trips > 256u) flush();
        const float4* __restrict__ src = &tile[buf][g * TGp];
        bool guard = own_leaf_inside;
        if (!guard) {
            const f2 sx = ax, sy = ay, sz = az;
            float rmin = __builtin_inff();
            unsigned k = 0;
            for (; k + 4u <= trips; k += 4u) fast4(src + k, rmin);
            for (; k < trips; ++k) fast1(src[k], rmin);
            guard = __builtin_amdgcn_ballot_w64(!(rmin >= law_special_below<LAW>())) != 0ull;   // a NaN r^2 also lands here
            if (guard) { ax = sx; ay = sy; az = sz; }          // rare: this tile's terms are taken back and redone below
        }
        if (guard)
            for (unsigned k = 0; k < trips; ++k) guarded1(src[k]);
        pending += trips;
        __syncthreads();                                       // the tile is free again
    };

    // ---- the source list, leaf by leaf ----
    // (sub-)entries in flight: a leaf of more than 64 bodies is staged in pieces of 64
    const uint32_t e_end = a.list_offsets[tb.leaf + 1];
    uint32_t e_base = a.list_offsets[tb.leaf];                 // first list entry of the chunk held in the lanes
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
    struct Piece { float4 v; unsigned n; bool own; };
    auto issue = [&]() -> Piece {                              // wave-uniform control; the load stays in flight
        Piece pc{pad, 0u, false};
        for (;;) {
            if (e_next == n_ent) {
                e_base += n_ent;
                n_ent = e_next = 0;
                if (e_base >= e_end) return pc;                // the list is exhausted: an empty piece, again and again
                load_chunk();
                continue;
            }
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)v_len, (int)e_next);
            if (off_next >= len) { ++e_next; off_next = 0; continue; }   // empty leaves are stepped over here as well
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)v_first, (int)e_next);
            const uint32_t src_leaf = (uint32_t)__builtin_amdgcn_readlane((int)v_src, (int)e_next);
            pc.n = (len - off_next < (uint32_t)kWave) ? (unsigned)(len - off_next) : (unsigned)kWave;
            pc.own = src_leaf == tb.leaf;
            if (lane < pc.n) pc.v = a.xm[first + off_next + lane];
            off_next += pc.n;
            return pc;
        }
    };
    int cur = 0;
    unsigned fill = 0;                                         // bodies staged in tile[cur] (+ overflow into tile[cur ^ 1])
    bool own_cur = false, own_nxt = false;                     // the tile holds bodies of the target's own leaf
    Piece p0 = issue(), p1 = issue();
    while (p0.n) {
        // piece p0 -> its tile slots (the part past the tile's end goes to the other buffer)
        if (lane < p0.n) {
            const unsigned q = fill + lane;
            if (q < (unsigned)kLeafTile) tile[cur][slot_of(q)] = p0.v;
            else tile[cur ^ 1][slot_of(q - (unsigned)kLeafTile)] = p0.v;
        }
        if (p0.own) { own_cur = own_cur || fill < (unsigned)kLeafTile; own_nxt = own_nxt || fill + p0.n > (unsigned)kLeafTile; }
        fill += p0.n;
        p0 = p1;
        p1 = issue();
        if (fill >= (unsigned)kLeafTile) {
            consume(cur, (unsigned)kLeafTile, own_cur);
            cur ^= 1; fill -= (unsigned)kLeafTile;
            own_cur = own_nxt; own_nxt = false;
        }
    }
    if (fill) consume(cur, fill, own_cur);
    flush();

    // ---- the lane groups' sums meet, group order ----
    if (G > 1u) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k) red[k][lane] = o[k];
        __syncthreads();
        if (g_raw == 0u)
            for (unsigned q = 1; q < G; ++q) {
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] += red[k][q * L + p];
            }
    }
    if (g_raw == 0u) {
        const uint32_t s0 = tb.first + p;
        a.acc[s0] = o[0];
        a.acc[(size_t)a.slots + s0] = o[2];
        if (D == 3) a.acc[2 * (size_t)a.slots + s0] = o[4];
        if (has1) {
            const uint32_t s1 = s0 + L;
            a.acc[s1] = o[1];
            a.acc[(size_t)a.slots + s1] = o[3];
            if (D == 3) a.acc[2 * (size_t)a.slots + s1] = o[5];
        }
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

    // Target blocks: a leaf of c targets is cut into k pieces of ceil(c / k).  A piece of t targets runs G = min(64 / ceil(t/2),
    // 16) lane groups, every lane making (sources / G) trips of 14 VALU, and stages the leaf's source list once (~0.3
    // instructions per body): k is the one that minimises k x (14 / G + 0.3), e.g. 32 targets: one block at 4 groups;
    // 34: two blocks of 17 at 7 groups instead of one at 3.
    auto cost = [&](uint32_t piece) -> double {
        const uint32_t lanes = (piece + 1) / 2;
        uint32_t groups = (uint32_t)kWave / lanes;
        if (groups > (uint32_t)kMaxGroups) groups = (uint32_t)kMaxGroups;
        return 14.0 / (double)groups + 0.3;
    };
    std::vector<TargetBlock> blocks;
    for (size_t l = 0; l < n_leaves; ++l) {
        const uint32_t c = leaf_offsets[l + 1] - leaf_offsets[l];
        if (!c) continue;
        const uint32_t k_min = (c + (uint32_t)kTargetsPerBlock - 1) / (uint32_t)kTargetsPerBlock;
        uint32_t best_k = k_min;
        double best = 1e300;
        for (uint32_t k = k_min; k <= k_min + 7 && k <= c; ++k) {
            const double v = (double)k * cost((c + k - 1) / k);
            if (v < best - 1e-12) { best = v; best_k = k; }
        }
        const uint32_t piece = (c + best_k - 1) / best_k;
        for (uint32_t f = leaf_offsets[l]; f < leaf_offsets[l + 1]; f += piece)
            blocks.push_back(TargetBlock{(uint32_t)l, f, (leaf_offsets[l + 1] - f < piece) ? leaf_offsets[l + 1] - f : piece});
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
