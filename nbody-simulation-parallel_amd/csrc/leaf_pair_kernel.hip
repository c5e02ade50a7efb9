// leaf_pair_kernel.hip -- SURVEY 8(f-4): batched (target leaf, source leaf) direct sums for the reference's tree
// codes -- the near-field step either side of the brute-force path:
//   FMM_Parlay<D>::p2p_phase   nbody-sim-new/fmm_parlay.cpp:916-1022   (law NBX_LAW_FMM_P2P)
//   BVH leaf loop              nbody-sim-new/bvh.cpp:150-176           (law NBX_LAW_TREE_LEAF)
//   octree leaf term           nbody-sim-new/octree.cpp:105-125        (law NBX_LAW_TREE_LEAF)
// and, for completeness, the brute-force law itself over leaf lists (NBX_LAW_BRUTE, methods.cpp:21-37).
// All three are m_j d / r^4 sums; they differ in sign and in what happens below ~1e-5 separation.
//
// Mapping to CDNA4.  The reference walks pointer lists per (body, neighbour leaf) work item and adds into
// forces[body] from several work items at once (fmm_parlay.cpp:986-1020).  Here the work is target-leaf-major and
// everything the kernel follows is laid out for it once per call:
//  * bodies are gathered into leaf order as fp32 SOURCE PAIRS {xa,xb,ya,yb},{za,zb,ma,mb} (32 B; a leaf of odd size
//    ends in a massless body far away), so a source leaf is a run of whole pairs that is copied into LDS as it lies
//    and read back with ds_read_b128 straight into the aligned register pairs of v_pk_*_f32;
//  * a leaf's source list becomes a few COPY OPS on the host: consecutive list entries whose leaves are neighbours
//    in leaf order are merged (the 27 cells of a grid neighbourhood are 9 to 11 runs), with the running length of the
//    stream they form -- no list walk, no leaf-offset lookup and no prefix sum on the device;
//  * one workgroup = two wave64 = ONE leaf: the waves share the staged tiles (512 bodies) and each owns one PIECE of
//    the leaf's targets.  A piece of c targets runs floor(64 / c) lanes per target (at most 8), which split the
//    tile's pairs between them; the host cuts the leaf where the two pieces together waste the fewest lanes
//    (43 targets: 11 x 5 lanes + 32 x 2 lanes = 96 % of the lanes busy; one wave64 with one lane each: 67 %).  Small leaves
//    (a mean of <= 20 bodies) get one wave per workgroup and no cut;
//  * structures of SMALL leaves (a mean of <= 8 bodies) are PACKED instead: 4, 8, 10 or 16 leaves share a wave64, every lane holds two
//    targets and streams its share of its leaf's source pairs straight from memory (leaf_pack_kernel below; leaf_plan.h);
//  * the workgroups are launched longest first, so that the launch drains in a fraction of a mean workgroup's time, and the
//    blocks of a duration class are dealt to the eight XCDs in consecutive runs (leaf_plan.h order_launch);
//  * the law's special cases sit below r^2 = 1e-9.  A target whose fp32 coordinates are all >= 2^14 in magnitude
//    cannot own such a pair other than with an identical position (the brute-force kernel's argument,
//    nbx_internal.h: distinct fp32 numbers that large differ by >= 2^-10), so a wave of such targets runs the pair
//    loop with no compare at all: r^2 biased by 2^-47, an identical position contributes m 2^94 x 0 = 0 as every law
//    asks.  Each wave decides for itself from its lanes' own coordinates (and the call's largest mass); the other waves
//    take the guarded loop (the smallest r^2 of a trip's four pairs, one compare, a wave vote, and the law's exact
//    weights for the wave that saw a pair below the threshold).
// fp32 sums over at most 256 terms per lane, flushed into fp64 accumulators.  No atomics, a fixed summation order,
// every output written once.  Leaves are small (the reference caps them at 100 bodies, methods.h:26), so the launch
// is tens of thousands of short workgroups; HBM traffic is 16 B per (workgroup, source body), served mostly from L2.
#include "../../include/nbody_hip.h"
#include "nbx_ctx.h"
#include "leaf_plan.h"
#include "leaf_plan_device.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <vector>

using namespace nbx;
using namespace nbx_leaf;

#ifndef NBX_LEAF_PACK
#define NBX_LEAF_PACK 1   /* 0: A/B build without packed small leaves (make LEAF_DEFS=-DNBX_LEAF_PACK=0 ...) */
#endif

namespace {

constexpr int kUnitsPerLane = 4;               // 16-byte units (= bodies) a lane stages per tile (the kernel names that many registers)
// a tile is 64 x WAVES x kUnitsPerLane bodies: 512 (256 source pairs, 8 KB of LDS) with two waves per workgroup
constexpr int kPadPairs = 16;                  // massless pairs behind the tile's last one: the lane groups' last trips reach up to 2 P - 1 past it
constexpr int kMaxOps = 64;                    // copy ops held in LDS at a time (longer lists go in chunks)
constexpr unsigned kFlushTerms = 248;          // fp32 terms per lane between flushes into the fp64 sums (+ 2 for a closing single pair)

// smallest fp32 thresholds that are >= the reference's fp64 ones, so (r2 < T_f32) == ((double)r2 < T) for fp32 r2
constexpr float kTreeSkipF = 0x1.12e0c0p-30f;   // 1.00000008e-9  (octree.cpp:119, bvh.cpp:167: dist_sq < 1e-9)
constexpr float kSmoothF = 0x1.b7cdfep-34f;     // 1.00000001e-10 (fmm_parlay.cpp:1010: dist_sq < 1e-10)
constexpr float kNormZeroF = 0x1.79ca12p-67f;   // 1.00000005e-20 (vector.h:93-97: |diff| < 1e-10 -> zero vector)
constexpr float kSameF = 1.0e-14f;              // largest fp32 <= 1e-14 (fmm_parlay.cpp:995-1000: |d_k| > 1e-14 -> distinct)
static_assert((double)kTreeSkipF >= 1e-9 && (double)kSmoothF >= 1e-10 && (double)kNormZeroF >= 1e-20 && (double)kSameF <= 1e-14,
              "fp32 thresholds must sit on the right side of the fp64 ones");
constexpr float kFar = 1.0e18f;                 // pad bodies: massless, r^2 ~ 1e36 is finite in fp32 and the weight underflows to 0
static_assert(kTreeSkipF < 9.0e-7f, "a target outside the close set (nbx_internal.h) has no non-zero r^2 below 9.5e-7: no law's special case can apply to it");

struct LeafArgs {
    const float4* __restrict__ xp;     // [pslots] units: pair p = units 2p {xa,xb,ya,yb} and 2p+1 {za,zb,ma,mb}
    uint32_t pslots;                   // padded slots = 2 x pairs
    const CopyOp* __restrict__ ops;
    const LeafBlock* __restrict__ blocks;
    double* __restrict__ acc;          // [dim][pslots]
    const uint32_t* __restrict__ max_mass_bits;   // bit pattern of the largest |mass| as fp32 (leaf_gather_kernel); a NaN compares above every number
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

// One source pair {A, B} (two float4 of the tile) against this lane's target.
template <int D>
struct PairTerm {
    f2 dx, dy, dz, r2, sm;
    __device__ __forceinline__ PairTerm(const float4 A, const float4 B, const f2 ix2, const f2 iy2, const f2 iz2, const f2 bias) {
        sm = f2{B.z, B.w};
        dx = f2{A.x, A.y} - ix2;
        dy = f2{A.z, A.w} - iy2;
        dz = (D == 3) ? f2{B.x, B.y} - iz2 : f2{0.f, 0.f};
        r2 = __builtin_elementwise_fma(dx, dx, bias);
        r2 = __builtin_elementwise_fma(dy, dy, r2);
        if (D == 3) r2 = __builtin_elementwise_fma(dz, dz, r2);
    }
    __device__ __forceinline__ f2 plain() const {
        f2 w = {__builtin_amdgcn_rcpf(r2.x), __builtin_amdgcn_rcpf(r2.y)};
        w = w * w;
        return w * sm;
    }
    template <int LAW>
    __device__ __forceinline__ unsigned long long special() const {
        return __builtin_amdgcn_ballot_w64(r2.x < law_special_below<LAW>()) | __builtin_amdgcn_ballot_w64(r2.y < law_special_below<LAW>());
    }
    template <int LAW>
    __device__ __forceinline__ f2 guarded() const {
        float ra = r2.x, rb = r2.y;
        asm volatile("" : "+v"(ra), "+v"(rb));   // keeps the guarded form's compares in this (rare) branch: hipcc hoists them otherwise
        return f2{leaf_weight<D, LAW>(ra, sm.x, dx.x, dy.x, dz.x), leaf_weight<D, LAW>(rb, sm.y, dx.y, dy.y, dz.y)};
    }
};

// Does any lane of the wave hold a pair (of the four of a trip) below the law's threshold?  The smallest of the four r^2 by two
// v_min3_f32, one compare, one vote.
template <int LAW, int D>
__device__ __forceinline__ bool any_special(const PairTerm<D>& q0, const PairTerm<D>& q1) {
    const float m = __builtin_fminf(__builtin_fminf(q0.r2.x, q0.r2.y), __builtin_fminf(q1.r2.x, q1.r2.y));
    return __builtin_amdgcn_ballot_w64(m < law_special_below<LAW>()) != 0ull;
}

// The sums of one lane: fp32 per-tile level, fp64 second level.
template <int D>
struct Sums {
    f2 ax = {0.f, 0.f}, ay = {0.f, 0.f}, az = {0.f, 0.f};
    double* o;              // this lane's fp64 sums, kept in LDS (they are touched once per ~250 terms; 6 VGPRs less): o[0], o[stride], o[2 stride]
    unsigned stride;        // lanes of the workgroup
    unsigned pending = 0;   // terms in the fp32 sums since the last flush (wave-uniform)
    __device__ __forceinline__ void add(const PairTerm<D>& q, const f2 w) {
        ax = __builtin_elementwise_fma(w, q.dx, ax);
        ay = __builtin_elementwise_fma(w, q.dy, ay);
        if (D == 3) az = __builtin_elementwise_fma(w, q.dz, az);
    }
    __device__ __forceinline__ void store() {   // flush() into sums that hold nothing yet: no read, no zeroing beforehand
        o[0] = (double)ax.x + (double)ax.y;
        o[stride] = (double)ay.x + (double)ay.y;
        if (D == 3) o[2u * stride] = (double)az.x + (double)az.y;
        ax = ay = az = f2{0.f, 0.f};
        pending = 0;
    }
    __device__ __forceinline__ void flush() {
        o[0] += (double)ax.x + (double)ax.y;
        o[stride] += (double)ay.x + (double)ay.y;
        if (D == 3) o[2u * stride] += (double)az.x + (double)az.y;
        ax = ay = az = f2{0.f, 0.f};
        pending = 0;
    }
};

// T source pairs (T odd) at s, for this lane's target: (T - 1) / 2 trips of two pairs -- two independent dependency chains --
// and a closing single pair.  GUARD: the law's special cases are possible for this wave's targets (one wave vote per trip);
// without it a trip is 26 VALU (6 v_pk_add, 6 + 6 v_pk_fma, 4 v_pk_mul, 4 v_rcp_f32) and half an address increment, and the
// LDS reads of the trip after next are in flight while a trip is computed.
template <int D, int LAW, bool GUARD>
__device__ __forceinline__ void consume(const float4* __restrict__ s, const unsigned T, const f2 ix2, const f2 iy2, const f2 iz2, Sums<D>& S) {
    const f2 bias = GUARD ? f2{0.f, 0.f} : f2{kTiny, kTiny};
    const unsigned ndt = T >> 1;
    unsigned done = 0;
    while (done < ndt) {                                    // wave-uniform bookkeeping
        if (S.pending + 4u > kFlushTerms) S.flush();
        const unsigned room = (kFlushTerms - S.pending) >> 2;
        const unsigned n = __builtin_amdgcn_readfirstlane((ndt - done < room) ? ndt - done : room);
        if (!GUARD) {
            // the trip after next is read while this one is computed: two register sets, taking turns (no copies)
            auto trip = [&](const float4 A0, const float4 B0, const float4 A1, const float4 B1) {
                const PairTerm<D> q0(A0, B0, ix2, iy2, iz2, bias), q1(A1, B1, ix2, iy2, iz2, bias);
                const f2 w0 = q0.plain(), w1 = q1.plain();
                S.add(q0, w0);
                S.add(q1, w1);
            };
            float4 a0 = s[0], a1 = s[1], a2 = s[2], a3 = s[3];
            unsigned i = n;
            for (; i >= 2u; i -= 2u, s += 8) {
                const float4 b0 = s[4], b1 = s[5], b2 = s[6], b3 = s[7];
                __builtin_amdgcn_sched_barrier(0);
                trip(a0, a1, a2, a3);
                __builtin_amdgcn_sched_barrier(0);
                a0 = s[8]; a1 = s[9]; a2 = s[10]; a3 = s[11];     // may lie past the segment: read, not used (the tile has room)
                __builtin_amdgcn_sched_barrier(0);
                trip(b0, b1, b2, b3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i) { trip(a0, a1, a2, a3); s += 4; }
        } else {
            // the guarded form keeps one register set (with two, the rare branch's live values push the kernel past 96 VGPRs)
            for (unsigned i = n; i != 0u; --i, s += 4) {
                const PairTerm<D> q0(s[0], s[1], ix2, iy2, iz2, bias), q1(s[2], s[3], ix2, iy2, iz2, bias);
                f2 w0, w1;
                if (__builtin_expect(any_special<LAW>(q0, q1), 0)) {   // wave-uniform, rare
                    w0 = q0.template guarded<LAW>();
                    w1 = q1.template guarded<LAW>();
                } else {
                    w0 = q0.plain();
                    w1 = q1.plain();
                }
                S.add(q0, w0);
                S.add(q1, w1);
            }
        }
        S.pending += 4u * n;
        done += n;
    }
    const PairTerm<D> q0(s[0], s[1], ix2, iy2, iz2, bias);
    const f2 w0 = (GUARD && __builtin_expect(q0.template special<LAW>() != 0ull, 0)) ? q0.template guarded<LAW>() : q0.plain();
    S.add(q0, w0);
    S.pending += 2u;
}

// One workgroup = one target leaf (or 128 targets of a larger one) against the stream of source pairs its copy ops describe.
//  * Staging: a tile is kTileUnits consecutive stream positions; lane l copies positions l, l + 128, ... (one 16-byte
//    load and one ds_write_b128 each, conflict-free); the run a position falls in is found by stepping a cursor through the
//    ops' running lengths in LDS -- it only ever moves forward and a run holds ~100 bodies.  The next tile's loads
//    are issued before the current tile is consumed.
//  * The pair loop: lane group g of a piece takes the T consecutive pairs from g T on, T = ceil(pairs / P) made odd:
//    consecutive pairs sit at immediate offsets of one address register, and the groups' addresses differ by odd
//    multiples of 32 B, which no two of <= 8 groups share a bank on.  Pairs past the tile's end are the pad pairs.
//  * A body meets itself in its own leaf: r^2 = 0 falls under every law's skip rule (methods.cpp:113 skips i == j by
//    index, which only differs from the r^2 rule for r^2 >= 1e-10 -- impossible for a body and itself).
// LDS of one one-leaf workgroup: the tile (+ what the pipelined pair loop reads ahead of its last trip), the run tables, the closing sums
template <int WAVES>
constexpr unsigned pair_smem_bytes() {
    return (unsigned)((64 * WAVES * kUnitsPerLane + 2 * kPadPairs + 16) * sizeof(float4) + 2 * kMaxOps * sizeof(uint32_t) + 3 * 64 * WAVES * sizeof(double));
}
template <int D, int LAW, int WAVES>
__device__ __forceinline__ void leaf_pair_body(const LeafArgs& a, const unsigned block, char* __restrict__ smem) {
    constexpr int kThreads = 64 * WAVES;
    constexpr int kTileUnits = kThreads * kUnitsPerLane;
    constexpr int kTileSlots = kTileUnits + 2 * kPadPairs + 16;
    float4* const tile = reinterpret_cast<float4*>(smem);
    uint32_t* const op_end = reinterpret_cast<uint32_t*>(smem + kTileSlots * sizeof(float4));
    uint32_t* const op_base = op_end + kMaxOps;
    double (*const osum)[kThreads] = reinterpret_cast<double (*)[kThreads]>(smem + kTileSlots * sizeof(float4) + 2 * kMaxOps * sizeof(uint32_t));
    static_assert(kUnitsPerLane == 2 || kUnitsPerLane == 4, "the kernel names two or four staging registers");
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LeafBlock* __restrict__ bp = a.blocks + block;
    const uint32_t op_lo = bp->op_lo, op_n = bp->op_n;
    const uint32_t max_mass_bits = *a.max_mass_bits;
    const uint32_t p_first = bp->piece[wave].first, p_count = bp->piece[wave].count;
    // lanes per target: as many whole groups of `count` lanes as the wave holds (21 targets: 3 lanes each)
    const unsigned W = p_count ? p_count : 1u;
    const unsigned fit = 64u / W;
    const unsigned P = fit < (unsigned)kMaxLanesPerTarget ? fit : (unsigned)kMaxLanesPerTarget;
    const unsigned inv_P = (65536u + P - 1u) / P;                 // q / P = (q * inv_P) >> 16 for q < 2^13
    // lane / W through fp32: (lane + 0.5) / W is at least 1/128 away from an integer, the arithmetic is good to 2e-5
    const unsigned g_raw = (unsigned)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)W));
    const unsigned t = lane - g_raw * W;
    const bool valid = p_count != 0u && g_raw < P;             // lanes left over compute along with group 0, unused
    const unsigned g = valid ? g_raw : 0u;
    const uint32_t pslot = p_first + (valid ? t : 0u);
    const float* __restrict__ xf = reinterpret_cast<const float*>(a.xp) + (size_t)(pslot >> 1) * 8u + (pslot & 1u);
    float ix = 0.f, iy = 0.f, iz = 0.f;
    if (p_count) { ix = xf[0]; iy = xf[2]; if (D == 3) iz = xf[4]; }
    const f2 ix2 = {ix, ix}, iy2 = {iy, iy}, iz2 = {iz, iz};
    // No special case of any law can apply to a pair of this wave (comment at the top of the file): every target of the piece lies
    // outside the close set, and m 2^94 stays finite for every mass.  The lanes left over hold the piece's first target.
    const bool in_close_set = !(__builtin_fabsf(ix) >= kCloseCoord && __builtin_fabsf(iy) >= kCloseCoord && (D == 2 || __builtin_fabsf(iz) >= kCloseCoord));
    const bool safe = __builtin_amdgcn_ballot_w64(in_close_set) == 0ull && max_mass_bits <= __builtin_bit_cast(uint32_t, (float)kFastMaxMass);
    Sums<D> S;
    osum[0][tid] = 0.0; osum[1][tid] = 0.0; osum[2][tid] = 0.0;   // only this lane touches them until the closing barrier
    S.o = &osum[0][tid];
    S.stride = (unsigned)kThreads;
    // a pad pair: unit {x,x,y,y} = far, unit {z,z,m,m} = {far (0 in 2D), 0}; selected by component (an indexed pair of constants ends up in scratch)
    const bool odd_unit = (tid & 1u) != 0u;
    const float pad_xy = odd_unit ? ((D == 3) ? kFar : 0.0f) : kFar, pad_zm = odd_unit ? 0.0f : kFar;
    const float4 pad_unit = make_float4(pad_xy, pad_xy, pad_zm, pad_zm);
    // the ops in chunks of kMaxOps (one chunk for every list the reference's trees produce); all workgroup-uniform
    for (uint32_t c0 = 0; c0 < op_n; c0 += (uint32_t)kMaxOps) {
        const unsigned n_ops = (op_n - c0 < (uint32_t)kMaxOps) ? (unsigned)(op_n - c0) : (unsigned)kMaxOps;
        __syncthreads();                                       // the previous chunk's last tile and tables are done with
        if (tid < n_ops) {
            const CopyOp o = a.ops[op_lo + c0 + tid];
            op_end[tid] = o.end;
            op_base[tid] = o.base;
        }
        const uint32_t u_begin = c0 ? a.ops[op_lo + c0 - 1u].end : 0u;
        const uint32_t u_end = a.ops[op_lo + c0 + n_ops - 1u].end;
        __syncthreads();
        unsigned k = 0;                                        // this lane's cursor in the chunk's ops; only ever moves forward
        uint32_t run_end = 0u, run_base = 0u;                  // the run it is in, in registers (every unit read them from LDS before: two reads and their waits)
        // kUnitsPerLane named registers (an indexed array ends up in scratch)
        float4 nxt0 = pad_unit, nxt1 = pad_unit, nxt2 = pad_unit, nxt3 = pad_unit;
        auto load1 = [&](const uint32_t u, float4& v) {
            if (u < u_end) {
                while (u >= run_end) { run_end = op_end[k]; run_base = op_base[k]; ++k; }   // u < u_end = the last run's end: k stays inside the chunk
                v = a.xp[run_base + u];
            }
        };
        auto load = [&](const uint32_t u0) {
            load1(u0 + tid, nxt0);
            load1(u0 + (uint32_t)kThreads + tid, nxt1);
            if (kUnitsPerLane > 2) {
                load1(u0 + 2u * (uint32_t)kThreads + tid, nxt2);
                load1(u0 + 3u * (uint32_t)kThreads + tid, nxt3);
            }
        };
        load(u_begin);
        for (uint32_t u0 = u_begin; u0 < u_end; u0 += (uint32_t)kTileUnits) {
            const unsigned cur = (u_end - u0 < (uint32_t)kTileUnits) ? (unsigned)(u_end - u0) : (unsigned)kTileUnits;   // even
            __syncthreads();                                   // previous tile fully consumed by both waves
            if (tid < cur) tile[tid] = nxt0;
            if ((unsigned)kThreads + tid < cur) tile[(unsigned)kThreads + tid] = nxt1;
            if (kUnitsPerLane > 2) {
                if (2u * (unsigned)kThreads + tid < cur) tile[2u * (unsigned)kThreads + tid] = nxt2;
                if (3u * (unsigned)kThreads + tid < cur) tile[3u * (unsigned)kThreads + tid] = nxt3;
            }
            if (tid < 2u * (unsigned)kPadPairs) tile[cur + tid] = pad_unit;
            __syncthreads();
            if (u0 + (uint32_t)kTileUnits < u_end) load(u0 + (uint32_t)kTileUnits);   // in flight while this tile is consumed
            if (p_count) {                                     // wave-uniform
                const unsigned pairs = cur >> 1;
                const unsigned T = (((pairs + P - 1u) * inv_P) >> 16) | 1u;
                const float4* s = tile + 2u * g * T;
                if (safe) consume<D, LAW, false>(s, T, ix2, iy2, iz2, S);
                else consume<D, LAW, true>(s, T, ix2, iy2, iz2, S);
            }
        }
    }
    S.flush();
    // the lane groups' fp64 sums meet in LDS, in group order (deterministic)
    __syncthreads();
    if (valid && g == 0u) {
        double ox = 0.0, oy = 0.0, oz = 0.0;
        const unsigned l0 = (tid & ~63u) + t;
        for (unsigned q = 0; q < P; ++q) { ox += osum[0][l0 + q * W]; oy += osum[1][l0 + q * W]; oz += osum[2][l0 + q * W]; }
        a.acc[pslot] = ox;
        a.acc[(size_t)a.pslots + pslot] = oy;
        if (D == 3) a.acc[2 * (size_t)a.pslots + pslot] = oz;
    }
}

template <int D, int LAW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void leaf_pair_kernel(LeafArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[pair_smem_bytes<WAVES>()];
    leaf_pair_body<D, LAW, WAVES>(a, blockIdx.x, smem);
}

// Packed small leaves (leaf_plan.h PackBlock): one wave64 = K = 64 / w leaves side by side, P lanes per target.  Nothing is staged:
// lane group g of a target walks pairs [g T, (g + 1) T) of its leaf's source stream straight from memory (two 16-byte loads per
// pair; the lanes of a leaf's targets share addresses, so a wave's load touches K x P distinct 32-byte records -- L2 hits, the
// neighbouring leaves' lists name the same bodies).  The leaf's copy runs sit in LDS, the run a lane is in in registers.  Every
// lane holds TWO targets (t and t + W of its leaf, W = ceil(count / 2) lanes per group): a loaded pair feeds four terms, which
// halves the loads (the texture-address path limits a loop of 13 packed VALU per 32 bytes loaded by every lane) and the
// bookkeeping per term.  The loop computes two pairs while the next two are in flight, two register sets taking turns (no
// copies): four loads outstanding per lane, so that a wave waits for memory once per ~90 instructions rather than once per pair.  T is the block's (every lane of the
// wave runs the same trips); a lane past its leaf's stream, or without a target, reads the launch's pad pair (massless, far
// away: exact zeros) at unit `pslots`.  No barrier in the loop, one start-up per K leaves.
// Replaces this round's first packed kernel (same blocks, each leaf's stream staged through its own region of an LDS tile):
// 0.213 -> 0.140 ms at 4-body leaves, profiles/r4/leaf_direct_ab.txt.
struct LeafPackArgs {
    const float4* __restrict__ xp;     // [pslots + 2] units: the launch's pad pair behind the last leaf's
    uint32_t pslots;
    const CopyOp* __restrict__ ops;
    const PackBlock* __restrict__ blocks;
    const PackSub* __restrict__ subs;
    double* __restrict__ acc;
    const uint32_t* __restrict__ max_mass_bits;
};

#ifndef NBX_PACK_WAVES
#define NBX_PACK_WAVES 4   /* waves per SIMD the register allocation aims at: 3D 110 VGPRs (5 spills 14 registers, 6 spills 34) */
#endif
// Round 5: the pair OFFSETS of the wave's leaves are expanded into LDS once, before the loop -- entry q of a leaf's row is the byte
// offset of pair q of its stream (the pad pair's behind the stream's end) -- so that the loop reads two offsets with one ds_read_b64
// (one stage ahead of the loads they address) where it used to step a cursor through the copy runs for every pair: ~16 VALU per
// loop body of 52 (profiles/r4/pmc_pack_kernel.txt: 39 % of the wave's instructions were bookkeeping).  Waves whose rows do not
// fit kPackExpand entries (K x P x T; long streams) keep the cursor loop.
#ifndef NBX_PACK_EXPAND
#define NBX_PACK_EXPAND 1280
#endif
constexpr unsigned kPackExpand = NBX_PACK_EXPAND;           // offsets per wave (5 KB; they share their LDS with the cursor loop's run tables: 8.2 KB per wave with the closing sums)
constexpr unsigned kPackExpandSlack = 8;         // the pipelined loop reads offsets up to 6 entries past its share
constexpr unsigned kPackTableWords = 2u * kPackMaxSubs * (kPackMaxOps + 1);
// LDS of a packed wave: ONE region for the cursor loop's run tables or the expanded loop's offsets (a wave runs one of the two loops),
// and the closing sums behind it
constexpr unsigned kPackLoopWords = kPackTableWords > kPackExpand + kPackExpandSlack ? kPackTableWords : kPackExpand + kPackExpandSlack;
constexpr unsigned kPackSmemBytes = kPackLoopWords * 4u + 3u * 128u * (unsigned)sizeof(double);
template <int D, int LAW>
__device__ __forceinline__ void leaf_pack_body(const LeafPackArgs& a, const unsigned block, char* __restrict__ smem) {
    static_assert(kPackLoopWords % 4u == 0u, "the closing sums behind the loop's region are doubles");
    uint32_t* const tables = reinterpret_cast<uint32_t*>(smem);                 // the cursor loop: the leaves' copy runs
    uint32_t* const poff = tables;                                              // the expanded loop: the pair offsets (same region)
    uint32_t (*op_end)[kPackMaxOps + 1] = reinterpret_cast<uint32_t (*)[kPackMaxOps + 1]>(tables);
    uint32_t (*op_base)[kPackMaxOps + 1] = reinterpret_cast<uint32_t (*)[kPackMaxOps + 1]>(tables + kPackMaxSubs * (kPackMaxOps + 1));
    const unsigned lane = threadIdx.x;
    const PackBlock* __restrict__ bp = a.blocks + block;
    const uint32_t w = bp->w, inv_w = bp->inv_w, P = bp->P, n_sub = bp->n_sub, sub_lo = bp->sub_lo, T = bp->trips;   // wave-uniform
    const unsigned sub = (lane * inv_w) >> 16, lw = lane - sub * w;   // w is 4, 6, 8 or 16: 64 / w leaves; the lanes left over (w = 6: four) hold no leaf
    PackSub my = PackSub{0u, 0u, 0u, 0u};
    if (sub < n_sub) my = a.subs[sub_lo + sub];
    // TWO targets per lane (t and t + W): a loaded source pair is used twice, and the loop's bookkeeping is shared by four terms
    const uint32_t W = my.count ? (my.count + 1u) >> 1 : 1u;       // lanes per group
    // lw / W through fp32, as above ((lw + 0.5) / W is at least 1/32 away from an integer)
    const unsigned g_raw = (unsigned)(((float)lw + 0.5f) * __builtin_amdgcn_rcpf((float)W));
    const unsigned t = lw - g_raw * W;
    const bool valid = my.count != 0u && g_raw < P;
    const bool valid1 = valid && t + W < my.count;                 // a leaf of odd size: its last lane holds one target
    const unsigned g = valid ? g_raw : 0u;
    const uint32_t pslot0 = my.first + (valid ? t : 0u), pslot1 = valid1 ? pslot0 + W : pslot0;
    const char* __restrict__ const xp_bytes = reinterpret_cast<const char*>(a.xp);   // 32-bit byte offsets from a scalar base: the plan packs nothing beyond 2^28 units
    const float* __restrict__ xf0 = reinterpret_cast<const float*>(xp_bytes + ((pslot0 >> 1) * 32u + (pslot0 & 1u) * 4u));
    const float* __restrict__ xf1 = reinterpret_cast<const float*>(xp_bytes + ((pslot1 >> 1) * 32u + (pslot1 & 1u) * 4u));
    float ix = 0.f, iy = 0.f, iz = 0.f, jx = 0.f, jy = 0.f, jz = 0.f;
    if (my.count) { ix = xf0[0]; iy = xf0[2]; jx = xf1[0]; jy = xf1[2]; if (D == 3) { iz = xf0[4]; jz = xf1[4]; } }
    const f2 ix2 = {ix, ix}, iy2 = {iy, iy}, iz2 = {iz, iz}, jx2 = {jx, jx}, jy2 = {jy, jy}, jz2 = {jz, jz};
    // lanes of an unused leaf slot hold no target: they must not drag the wave into the guarded loop
    auto close = [](const float x, const float y, const float z) { return !(__builtin_fabsf(x) >= kCloseCoord && __builtin_fabsf(y) >= kCloseCoord && (D == 2 || __builtin_fabsf(z) >= kCloseCoord)); };
    const bool in_close_set = my.count != 0u && (close(ix, iy, iz) || close(jx, jy, jz));
    const bool safe = __builtin_amdgcn_ballot_w64(in_close_set) == 0ull && *a.max_mass_bits <= __builtin_bit_cast(uint32_t, (float)kFastMaxMass);
    // this leaf's copy runs, as {end of the run in the stream, byte offset of the run's first unit minus 16 x its stream position}
    // (32-bit byte offsets from a scalar base: the plan packs nothing beyond 2^28 units), and behind the last one the PAD RUN:
    // it never ends, and every position of it is the launch's pad pair (its positions are masked to zero)
    const uint32_t PT = P * T;                                     // entries of a leaf's row = pairs its lane groups walk between them
    const bool expanded = T != 0u && n_sub * PT <= kPackExpand;    // wave-uniform
    if (!expanded) {                                               // the cursor loop's run tables
        for (unsigned k = lw; k < my.op_n; k += w) {
            const CopyOp o = a.ops[my.op_lo + k];
            op_end[sub][k] = o.end;
            op_base[sub][k] = o.base << 4;
        }
        if (lw == 0u) { op_end[sub][my.op_n] = 0xffffffffu; op_base[sub][my.op_n] = a.pslots << 4; }
    }
    Sums<D> S0, S1;
    bool flushed = false;                                          // wave-uniform: the fp64 sums hold something (they are not zeroed: the first flush stores)
    double* const osum = reinterpret_cast<double*>(smem + kPackLoopWords * 4u);   // [component][second target? 64 : 0][lane]
    S0.o = osum + lane;
    S1.o = osum + 64u + lane;
    S0.stride = S1.stride = 128u;
    if (!expanded) __syncthreads();                                // the run tables (one wave: cheap)
    // The cursor k only ever moves forward; the run it is in sits in registers.  A lane without a leaf starts (and stays) in a pad
    // run of its own.
    unsigned k = 0;
    uint32_t run_end = my.count ? 0u : 0xffffffffu, run_off = a.pslots << 4, run_mask = 0u;
    auto offset_of = [&](const uint32_t v) -> uint32_t {           // byte offset of the pair at stream position v (even)
        while (v >= run_end) {                                     // rare (a run is ~3 leaves); the pad run ends the search
            run_end = op_end[sub][k];
            run_off = op_base[sub][k];
            run_mask = k == my.op_n ? 0u : 0xffffffffu;
            ++k;
        }
        return ((v & run_mask) << 4) + run_off;                    // a run is whole leaves, a leaf whole pairs: both units of the pair
    };
    static_assert(kPackPairsPerTrip == 2, "the loops below");
    // one form of the pair term per wave (GUARD: a special case of the law is possible for one of this wave's targets)
    auto compute_with = [&](auto guard, const float4 (&A)[2], const float4 (&B)[2]) {
        constexpr bool GUARD = decltype(guard)::value;
        const f2 bias = GUARD ? f2{0.f, 0.f} : f2{kTiny, kTiny};
        auto term = [&](const float4 A1, const float4 B1, const f2 x2, const f2 y2, const f2 z2, Sums<D>& S) {
            const PairTerm<D> q(A1, B1, x2, y2, z2, bias);
            f2 wgt;
            if (GUARD && __builtin_expect(q.template special<LAW>() != 0ull, 0)) wgt = q.template guarded<LAW>();
            else wgt = q.plain();
            S.add(q, wgt);
        };
        if (S0.pending + 4u > kFlushTerms) {
            if (flushed) { S0.flush(); S1.flush(); } else { S0.store(); S1.store(); flushed = true; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            term(A[j], B[j], ix2, iy2, iz2, S0);
            term(A[j], B[j], jx2, jy2, jz2, S1);
        }
        S0.pending += 4u;
    };
    if (expanded) {
        // every lane of a leaf fills the entries q = lw, lw + w, ... of its leaf's row; lanes without a leaf fill the slack behind the
        // last row (the loop's read-ahead must find loadable offsets there) and walk row 0 (their sums are never stored)
        // RUN by run, straight from the plan's copy runs (no tables, no search): lane lw of a leaf takes its runs lw, lw + w, ... and
        // writes their pairs' offsets (consecutive records, 32 bytes apart), then its share of the pad entries behind the stream's end
        const uint32_t row = sub < n_sub ? sub * PT : 0u;
        const uint32_t pad_off = a.pslots << 4;
        if (sub < n_sub) {
            for (uint32_t r = lw; r < my.op_n; r += w) {
                const CopyOp o = a.ops[my.op_lo + r];
                const uint32_t begin = r ? a.ops[my.op_lo + r - 1u].end : 0u;
                uint32_t off = (o.base + begin) << 4;              // the run's first unit
                for (uint32_t q = begin >> 1; q < (o.end >> 1); ++q, off += 32u) poff[row + q] = off;
            }
            const uint32_t stream_pairs = my.op_n ? a.ops[my.op_lo + my.op_n - 1u].end >> 1 : 0u;   // <= PT: T covers the wave's longest stream
            for (uint32_t q = stream_pairs + lw; q < PT; q += w) poff[row + q] = pad_off;
        }
        if (lane < kPackExpandSlack) poff[n_sub * PT + lane] = pad_off;
        __syncthreads();
        const uint2* __restrict__ const mine = reinterpret_cast<const uint2*>(poff + row + g * T);   // T and PT are even: 8-byte aligned
        auto issue = [&](const uint2 o, float4 (&A)[2], float4 (&B)[2]) {
            const float4* __restrict__ s0 = reinterpret_cast<const float4*>(xp_bytes + o.x);
            const float4* __restrict__ s1 = reinterpret_cast<const float4*>(xp_bytes + o.y);
            A[0] = s0[0]; B[0] = s0[1];
            A[1] = s1[0]; B[1] = s1[1];
        };
        auto run = [&](auto guard) {
            float4 A0[2], B0[2], A1[2], B1[2];
            uint2 o_now = mine[0], o_next = mine[1];               // the offsets run one stage ahead of the loads they address
            issue(o_now, A0, B0);
            uint32_t i = 0;
            for (; i + 4u <= T; i += 4u) {                         // loads past the share's end fetch the next share's pairs or the pad pair: not used
                issue(o_next, A1, B1);
                o_now = mine[(i >> 1) + 2u];
                compute_with(guard, A0, B0);
                issue(o_now, A0, B0);
                o_next = mine[(i >> 1) + 3u];
                compute_with(guard, A1, B1);
            }
            if (i < T) compute_with(guard, A0, B0);                // T is even: the last two pairs
        };
        // Leaves of 3-4 bodies (two lanes to a lane group, which read the same records): the loads are what limits their waves (texture
        // address unit 71 % busy, VALU 46 %: profiles/r5/pmc_pack_kernel.txt), so the two lanes of a group load ONE half of each
        // 32-byte record each -- the even lane {xa,xb,ya,yb}, the odd lane {za,zb,ma,mb} -- and hand it to each other with DPP moves
        // (quad_perm: a leaf's four lanes are one quad): half the load instructions for 16 more moves per loop body of 52.
        // Measured and lost (profiles/r5/leaf_small_leaves.txt): the same exchange for EVERY class through ds_bpermute_b32 (no VALU
        // instruction added, but its latency sits between the loads and the arithmetic): 8-body leaves 0.315 -> 0.265, 4-body cells 0.161 -> 0.152.
        auto run_shared = [&](auto guard) {
            const uint32_t half = (lane & 1u) << 4;                // this lane's half of a record
            auto issue_half = [&](const uint2 o, float4 (&H)[2]) {
                H[0] = *reinterpret_cast<const float4*>(xp_bytes + (o.x + half));
                H[1] = *reinterpret_cast<const float4*>(xp_bytes + (o.y + half));
            };
            auto from_even = [](const float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xA0, 0xF, 0xF, true)); };   // quad_perm:[0,0,2,2]
            auto from_odd = [](const float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xF5, 0xF, 0xF, true)); };    // quad_perm:[1,1,3,3]
            auto compute_halves = [&](const float4 (&H)[2]) {
                float4 A[2], B[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    A[j] = make_float4(from_even(H[j].x), from_even(H[j].y), from_even(H[j].z), from_even(H[j].w));
                    B[j] = make_float4(from_odd(H[j].x), from_odd(H[j].y), from_odd(H[j].z), from_odd(H[j].w));
                }
                compute_with(guard, A, B);
            };
            float4 H0[2], H1[2];
            uint2 o_now = mine[0], o_next = mine[1];
            issue_half(o_now, H0);
            uint32_t i = 0;
            for (; i + 4u <= T; i += 4u) {
                issue_half(o_next, H1);
                o_now = mine[(i >> 1) + 2u];
                compute_halves(H0);
                issue_half(o_now, H0);
                o_next = mine[(i >> 1) + 3u];
                compute_halves(H1);
            }
            if (i < T) compute_halves(H0);
        };
        if (bp->shape == 1u && w == 4u && P == 2u) {               // wave-uniform: every leaf of the wave has W = 2 lanes per group
            if (safe) run_shared(std::false_type{});
            else run_shared(std::true_type{});
        } else if (safe) run(std::false_type{});
        else run(std::true_type{});
    } else if (T) {
        // the cursor loop (rounds 3-4): lane group g walks stream positions [2 g T, 2 (g + 1) T) of its leaf straight from the run tables
        if (!valid) { run_end = 0xffffffffu; run_off = a.pslots << 4; run_mask = 0u; }   // no target: a pad run of its own
        const uint32_t v_begin = 2u * g * T;
        auto issue = [&](const uint32_t i, float4 (&A)[2], float4 (&B)[2]) {   // pairs i and i + 1 of this lane's share
            const uint32_t v = v_begin + 2u * i;
            const uint32_t o0 = offset_of(v), o1 = offset_of(v + 2u);
            const float4* __restrict__ s0 = reinterpret_cast<const float4*>(xp_bytes + o0);
            const float4* __restrict__ s1 = reinterpret_cast<const float4*>(xp_bytes + o1);
            A[0] = s0[0]; B[0] = s0[1];
            A[1] = s1[0]; B[1] = s1[1];
        };
        auto run = [&](auto guard) {
            float4 A0[2], B0[2], A1[2], B1[2];
            issue(0u, A0, B0);
            // One back edge, no exit in the middle: with a `break` between the halves the compiler waits for EVERY outstanding load at
            // the loop head (s_waitcnt vmcnt(0)); this way it waits for the four older ones only, the reload stays in flight.
            uint32_t i = 0;
            for (; i + 4u <= T; i += 4u) {                         // loads past the share's end fetch the next share's pairs or the pad pair: not used
                issue(i + 2u, A1, B1);
                compute_with(guard, A0, B0);
                issue(i + 4u, A0, B0);
                compute_with(guard, A1, B1);
            }
            if (i < T) compute_with(guard, A0, B0);                // T is even: the last two pairs
        };
        if (safe) run(std::false_type{});
        else run(std::true_type{});
    }
    if (flushed) { S0.flush(); S1.flush(); } else { S0.store(); S1.store(); }   // T = 0: zeros are stored
    __syncthreads();
    if (valid && g == 0u) {
        const unsigned l0 = sub * w + t;
        for (unsigned h = 0; h < (valid1 ? 2u : 1u); ++h) {
            double ox = 0.0, oy = 0.0, oz = 0.0;
            for (unsigned q = 0; q < P; ++q) { ox += osum[64u * h + l0 + q * W]; oy += osum[128u + 64u * h + l0 + q * W]; oz += osum[256u + 64u * h + l0 + q * W]; }
            const uint32_t pslot = h ? pslot1 : pslot0;
            a.acc[pslot] = ox;
            a.acc[(size_t)a.pslots + pslot] = oy;
            if (D == 3) a.acc[2 * (size_t)a.pslots + pslot] = oz;
        }
    }
}

template <int D, int LAW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NBX_PACK_WAVES, 8))) void leaf_pack_kernel(LeafPackArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[kPackSmemBytes];
    leaf_pack_body<D, LAW>(a, blockIdx.x, smem);
}

// A structure with BOTH kinds of workgroups -- packed waves and a few one-leaf workgroups (leaves whose lists have too many runs to
// pack) -- in ONE launch: the first n_one workgroups are the one-leaf ones (the long ones: dispatched first), the others the packed
// waves.  Two launches ran the 25 one-leaf workgroups of an 8-body structure alone for 8.7 us ahead of 85 us of packed waves
// (profiles/r4/leaf_kernel_stats_final_8body.csv).  Both bodies use the same 64 lanes and the same LDS (the larger of the two).
template <int D, int LAW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NBX_PACK_WAVES, 8))) void leaf_fused_kernel(LeafArgs one, LeafPackArgs pack, uint32_t n_one) {
    constexpr unsigned kBytes = pair_smem_bytes<1>() > kPackSmemBytes ? pair_smem_bytes<1>() : kPackSmemBytes;
    __shared__ __attribute__((aligned(16))) char smem[kBytes];
    if (blockIdx.x < n_one) leaf_pair_body<D, LAW, 1>(one, blockIdx.x, smem);
    else leaf_pack_body<D, LAW>(pack, blockIdx.x - n_one, smem);
}

// The call's largest |mass| (as fp32 bits) into one word: the block's maximum through LDS, then ONE lane per block, and only when an
// ordinary (cached) read of the word says the block's maximum is above it.  16,384 atomics on the word cost 0.15 ms; so did 16,384
// atomic LOADS of it (one L2 channel serves them one after the other) -- the resident gather spent 0.12 of its 0.14 ms there.  A stale
// read only costs a superfluous atomicMax; the atomic decides.
__device__ __forceinline__ void publish_max_mass(uint32_t wave_max, uint32_t* __restrict__ max_mass_bits) {
    __shared__ uint32_t block_max[4];
    if ((threadIdx.x & 63u) == 0u) block_max[threadIdx.x >> 6] = wave_max;
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t m = block_max[0];
        for (unsigned w = 1; w < (blockDim.x + 63u) / 64u; ++w) m = block_max[w] > m ? block_max[w] : m;
        if (m > *static_cast<volatile const uint32_t*>(max_mass_bits)) atomicMax(max_mass_bits, m);
    }
}

// the launch's pad pair, behind the last leaf's slots (pslots is even: every leaf is padded to whole pairs); the arrays' arena keeps
// 256 bytes of slack behind each of them
__device__ __forceinline__ void write_pad_pair(float* __restrict__ xp, uint32_t pslots, int dim) {
    float* __restrict__ o = xp + (size_t)pslots * 4u;
    const float zc = dim == 3 ? kFar : 0.0f;
    o[0] = kFar; o[1] = kFar; o[2] = kFar; o[3] = kFar; o[4] = zc; o[5] = zc; o[6] = 0.0f; o[7] = 0.0f;
}

// staged Body<D> AoS fp64 (host order) -> leaf-ordered source pairs, fp32; a padded slot without a body is massless and far away
__global__ __launch_bounds__(256) void leaf_gather_kernel(const double* __restrict__ raw, size_t stride_d, int dim,
                                                          const uint32_t* __restrict__ pslot_body, uint32_t pslots, float* __restrict__ xp,
                                                          uint32_t* __restrict__ max_mass_bits) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t body = p < pslots ? pslot_body[p] : 0xffffffffu;
    float x = kFar, y = kFar, z = dim == 3 ? kFar : 0.0f, m = 0.0f;
    if (body != 0xffffffffu) {
        const double* __restrict__ b = raw + (size_t)body * stride_d;
        x = (float)b[0]; y = (float)b[1]; z = dim == 3 ? (float)b[2] : 0.0f; m = (float)b[2 * dim];
    }
    // largest |mass| of the call: non-negative fp32 order like their bit patterns, a NaN above them all
    uint32_t mb = __builtin_bit_cast(uint32_t, __builtin_fabsf(m));
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)mb, d);
        mb = other > mb ? other : mb;
    }
    // one lane per wave, and only while the wave's maximum is above what is already there (16,384 atomics on one word cost 0.15 ms)
    publish_max_mass(mb, max_mass_bits);
    if (p == 0u) write_pad_pair(xp, pslots, dim);
    if (p >= pslots) return;
    float* __restrict__ o = xp + (size_t)(p >> 1) * 8u + (p & 1u);
    o[0] = x; o[2] = y; o[4] = z; o[6] = m;
}

// forces_out[body] = sign * (G m_body) * acc[slot]   (fp64; every body belongs to at most one leaf)
__global__ __launch_bounds__(256) void leaf_scatter_kernel(const double* __restrict__ acc, const double* __restrict__ raw, size_t stride_d,
                                                           int dim, const uint32_t* __restrict__ pslot_body, uint32_t pslots, double signedG,
                                                           double* __restrict__ forces) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= pslots) return;
    const uint32_t body = pslot_body[p];
    if (body == 0xffffffffu) return;
    const double gm = signedG * raw[(size_t)body * stride_d + 2 * dim];
    for (int k = 0; k < dim; ++k) forces[(size_t)body * dim + k] = gm * acc[(size_t)k * pslots + p];
}

// ---- kernels of the device-resident plan (nbx_leaf_plan_*) ----
// resident fp32 SoA (a single-shard context's source copy: pos[dim][pad], mass[pad]) -> leaf-ordered source pairs, BODY-major: one
// lane per body reads its coordinates and mass coalesced and writes its slot's 16 bytes (four words of one 32-byte pair record).  Slot-major, every slot's four words came from four different cache lines of the SoA arrays (0.106 ms at
// N = 2^20, a quarter of an evaluation); this way the scattered side is one 32-byte sector per body.  Pad slots are written once, at plan
// creation (leaf_init_pads_kernel): nothing here touches them.
__global__ __launch_bounds__(256) void leaf_gather_by_body_kernel(const float* __restrict__ pos, const float* __restrict__ mass, unsigned pad, int dim,
                                                                  const uint32_t* __restrict__ body_slot, size_t n, float* __restrict__ xp,
                                                                  uint32_t* __restrict__ max_mass_bits) {
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t slot = 0xffffffffu;
    float x = 0.f, y = 0.f, z = 0.f, m = 0.f;
    if (b < n) {
        slot = body_slot[b];
        if (slot != 0xffffffffu) { x = pos[b]; y = pos[(size_t)pad + b]; z = dim == 3 ? pos[2 * (size_t)pad + b] : 0.0f; m = mass[b]; }
    }
    uint32_t mb = __builtin_bit_cast(uint32_t, __builtin_fabsf(m));   // bodies in no leaf are no sources: they do not count (m = 0)
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)mb, d);
        mb = other > mb ? other : mb;
    }
    publish_max_mass(mb, max_mass_bits);
    if (slot == 0xffffffffu) return;
    float* __restrict__ o = xp + (size_t)(slot >> 1) * 8u + (slot & 1u);
    o[0] = x; o[2] = y; o[4] = z; o[6] = m;
}

__global__ __launch_bounds__(256) void leaf_init_pads_kernel(const uint32_t* __restrict__ pslot_body, uint32_t pslots, int dim, float* __restrict__ xp) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p == 0u) write_pad_pair(xp, pslots, dim);
    if (p >= pslots || pslot_body[p] != 0xffffffffu) return;
    float* __restrict__ o = xp + (size_t)(p >> 1) * 8u + (p & 1u);
    o[0] = kFar; o[2] = kFar; o[4] = dim == 3 ? kFar : 0.0f; o[6] = 0.0f;   // a leaf's pad: massless and far away
}

// forces[body] = (signedG m_body) * sums[slot of body], one lane per body (every entry written: zero for a body in no leaf);
// the mass comes from mass[body * mass_stride] -- a context's m64 (stride 1) or the staged Body<D> array (offset 2 dim, stride the body's)
__global__ __launch_bounds__(256) void leaf_forces_by_body_kernel(const double* __restrict__ sums, uint32_t pslots, const uint32_t* __restrict__ body_slot,
                                                                  size_t n, int dim, double signedG, const double* __restrict__ mass, size_t mass_stride,
                                                                  double* __restrict__ forces) {
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= n) return;
    const uint32_t slot = body_slot[b];
    const double gm = signedG * mass[b * mass_stride];
    for (int k = 0; k < dim; ++k) forces[b * dim + k] = slot == 0xffffffffu ? 0.0 : gm * sums[(size_t)k * pslots + slot];
}

typedef void (*LeafKernel)(LeafArgs);
LeafKernel pick(int dim, int law, int waves) {
    static const LeafKernel table[2][2][3] = {
        {{leaf_pair_kernel<2, NBX_LAW_BRUTE, 1>, leaf_pair_kernel<2, NBX_LAW_TREE_LEAF, 1>, leaf_pair_kernel<2, NBX_LAW_FMM_P2P, 1>},
         {leaf_pair_kernel<3, NBX_LAW_BRUTE, 1>, leaf_pair_kernel<3, NBX_LAW_TREE_LEAF, 1>, leaf_pair_kernel<3, NBX_LAW_FMM_P2P, 1>}},
        {{leaf_pair_kernel<2, NBX_LAW_BRUTE, 2>, leaf_pair_kernel<2, NBX_LAW_TREE_LEAF, 2>, leaf_pair_kernel<2, NBX_LAW_FMM_P2P, 2>},
         {leaf_pair_kernel<3, NBX_LAW_BRUTE, 2>, leaf_pair_kernel<3, NBX_LAW_TREE_LEAF, 2>, leaf_pair_kernel<3, NBX_LAW_FMM_P2P, 2>}}};
    return table[waves - 1][dim - 2][law];
}

typedef void (*PackKernel)(LeafPackArgs);
PackKernel pick_pack(int dim, int law) {
    static const PackKernel table[2][3] = {
        {leaf_pack_kernel<2, NBX_LAW_BRUTE>, leaf_pack_kernel<2, NBX_LAW_TREE_LEAF>, leaf_pack_kernel<2, NBX_LAW_FMM_P2P>},
        {leaf_pack_kernel<3, NBX_LAW_BRUTE>, leaf_pack_kernel<3, NBX_LAW_TREE_LEAF>, leaf_pack_kernel<3, NBX_LAW_FMM_P2P>}};
    return table[dim - 2][law];
}

typedef void (*FusedKernel)(LeafArgs, LeafPackArgs, uint32_t);
FusedKernel pick_fused(int dim, int law) {
    static const FusedKernel table[2][3] = {
        {leaf_fused_kernel<2, NBX_LAW_BRUTE>, leaf_fused_kernel<2, NBX_LAW_TREE_LEAF>, leaf_fused_kernel<2, NBX_LAW_FMM_P2P>},
        {leaf_fused_kernel<3, NBX_LAW_BRUTE>, leaf_fused_kernel<3, NBX_LAW_TREE_LEAF>, leaf_fused_kernel<3, NBX_LAW_FMM_P2P>}};
    return table[dim - 2][law];
}

// The call's device arrays are one allocation, and a tree code calls once per step with arrays of the same size: the allocation
// of a finished call is parked (one per device, up to kArenaParkMax bytes) and taken by the next call it is large enough for --
// hipMalloc + hipFree of ~100 MB cost 0.5 ms of a 4.4-ms call.  nbx_release_cached() frees the parked ones.
constexpr size_t kArenaParkMax = (size_t)2 << 30;
constexpr size_t kArenasPerDevice = 2;
constexpr size_t kHelperCopyBytes = (size_t)4 << 20;   // staged bodies from this size on are copied by a helper thread while the launch is laid out
struct ParkedArena { int device; char* p; size_t bytes; };
std::mutex g_arena_mu;
std::vector<ParkedArena> g_arenas;

hipError_t take_arena(int device, size_t bytes, char** out, size_t* got) {
    {
        std::lock_guard<std::mutex> lock(g_arena_mu);
        size_t best = g_arenas.size();
        for (size_t i = 0; i < g_arenas.size(); ++i)
            if (g_arenas[i].device == device && g_arenas[i].bytes >= bytes && (best == g_arenas.size() || g_arenas[i].bytes < g_arenas[best].bytes)) best = i;
        if (best != g_arenas.size()) {
            *out = g_arenas[best].p;
            *got = g_arenas[best].bytes;
            g_arenas.erase(g_arenas.begin() + (long)best);
            return hipSuccess;
        }
    }
    *got = bytes;
    hipError_t e = hipMalloc((void**)out, bytes);
    if (e == hipErrorOutOfMemory) {   // the cache itself may be what is in the way: give the parked blocks back and try once more
        (void)hipGetLastError();
        nbx::release_parked_leaf_arenas();
        e = hipMalloc((void**)out, bytes);
    }
    return e;
}

void park_arena(int device, char* p, size_t bytes) {   // nothing on the device uses p any more
    char* evicted = nullptr;
    if (bytes <= kArenaParkMax) {
        std::lock_guard<std::mutex> lock(g_arena_mu);
        g_arenas.push_back(ParkedArena{device, p, bytes});
        size_t mine = 0, smallest = g_arenas.size();
        for (size_t i = 0; i < g_arenas.size(); ++i)
            if (g_arenas[i].device == device) {
                ++mine;
                if (smallest == g_arenas.size() || g_arenas[i].bytes < g_arenas[smallest].bytes) smallest = i;
            }
        if (mine > kArenasPerDevice) {                      // a call takes two (bodies; everything else): keep the two largest
            evicted = g_arenas[smallest].p;
            g_arenas.erase(g_arenas.begin() + (long)smallest);
        }
    } else {
        evicted = p;
    }
    if (evicted) (void)hipFree(evicted);
}

struct DeviceBuffers {   // gives back whatever the call took when it leaves, on every path
    char* arena = nullptr;          // everything but the staged bodies
    size_t arena_bytes = 0;
    char* body_arena = nullptr;     // the staged Body<D> array
    size_t body_bytes = 0;
    hipStream_t stream = nullptr;
    int device = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~DeviceBuffers() {
        const bool idle = stream && hipStreamSynchronize(stream) == hipSuccess;
        if (arena) {
            if (idle) park_arena(device, arena, arena_bytes);
            else (void)hipFree(arena);
        }
        if (body_arena) {
            if (idle) park_arena(device, body_arena, body_bytes);
            else (void)hipFree(body_arena);
        }
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (idle) nbx::park_stream(device, stream);   // back to the pool (nbx_api.hip): a stream costs more than this call's kernels
        else if (stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace

namespace nbx {
void release_parked_leaf_arenas() {
    std::vector<ParkedArena> parked;
    {
        std::lock_guard<std::mutex> lock(g_arena_mu);
        parked.swap(g_arenas);
    }
    int before = 0;
    const bool have = hipGetDevice(&before) == hipSuccess;
    for (ParkedArena& a : parked)
        if (hipSetDevice(a.device) == hipSuccess) (void)hipFree(a.p);
    if (have) (void)hipSetDevice(before);
    (void)hipGetLastError();
}
}  // namespace nbx

// ---- device-resident plan (include/nbody_hip.h "device-resident leaf plan") ---------------------------------------------------
// What a tree code keeps between force evaluations while its tree stands: the validated structure laid out for the kernel
// (leaf_plan.h) and every device buffer an evaluation needs.  An evaluation is then: gather (16 B per slot from the resident
// fp32 source copy), pair kernel, and -- only if the caller wants them on the host -- forces by body and one copy out.
struct nbx_leaf_plan {
    int device = 0, dim = 3, waves = 2;
    size_t n = 0, pslots = 0, n_ops = 0, n_blocks = 0, n_subs = 0, n_packs = 0;
    char* arena = nullptr;          // xp | sums | pslot_body | body_slot | ops | blocks | max_mass | packed leaves | packed waves
    size_t arena_bytes = 0;         // what take_arena handed out (a parked block may be larger than asked for)
    bool last_wait_ok = true;       // destroy: the wait for the last evaluation succeeded (else the block is freed, not parked)
    float4* xp = nullptr;
    double* sums = nullptr;         // [dim][pslots]
    uint32_t* pslot_body = nullptr; // [pslots]
    uint32_t* body_slot = nullptr;  // [n]  inverse map, 0xffffffff for a body in no leaf
    CopyOp* ops = nullptr;
    LeafBlock* blocks = nullptr;
    uint32_t* max_mass = nullptr;
    PackSub* subs = nullptr;
    PackBlock* packs = nullptr;
    double* forces = nullptr;       // [n][dim], allocated when a caller first asks for forces on the host
    double* raw = nullptr;          // staged Body<D> array of nbx_leaf_plan_forces, allocated on first use
    size_t raw_bytes = 0;
    hipStream_t stream = nullptr;   // own stream (host-bodies path)
    hipStream_t last_stream = nullptr;   // stream the last evaluation was ordered on
    hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr;
    bool evaluated = false;
    int last_law = NBX_LAW_TREE_LEAF;
    double last_signedG = 0.0;
    // masses of the last evaluation: a context's m64 (stride 1) or the staged bodies (offset 2 dim, stride the body's)
    const double* last_mass = nullptr;
    size_t last_mass_stride = 1;
    unsigned long long last_ctx_id = 0;   // the context whose m64 last_mass points into (0: the plan's own staged bodies)
    bool forces_in_arena = false;   // `forces` is a piece of the arena (the one-shot call's plan), not an allocation of its own
    bool device_planned = false;    // laid out on the device (leaf_plan_device.h); false: on the host (leaf_plan.h)
    char* raw_arena = nullptr;      // one-shot call: the staged bodies come from the parked pool instead of hipMalloc
    size_t raw_arena_bytes = 0;
    nbx_leaf_dev::Summary summary_host;   // the device planner's 64 bytes land here
};

namespace {
// The part of the validation that stays on the host whichever planner runs: the two offset arrays (n_leaves + 1 words each; the
// lengths of every copy come from them).
int validate_offsets(size_t n, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets,
                     const uint32_t* list_sources, size_t* slots_out, size_t* n_list_out) {
    if (n > ((size_t)1 << 31) || n_leaves > ((size_t)1 << 31)) return fail(NBX_ERR_INVALID, "too many bodies / leaves");
    if (n_leaves && (!leaf_offsets || !list_offsets)) return fail(NBX_ERR_INVALID, "null leaf arrays");
    const size_t slots = n_leaves ? leaf_offsets[n_leaves] : 0;
    const size_t n_list = n_leaves ? list_offsets[n_leaves] : 0;
    if (n_leaves && (leaf_offsets[0] != 0 || list_offsets[0] != 0)) return fail(NBX_ERR_INVALID, "CSR offsets must start at 0");
    uint32_t bad = 0;                                    // no exit inside the loop: vectorised
    for (size_t l = 0; l < n_leaves; ++l) bad |= (uint32_t)(leaf_offsets[l + 1] < leaf_offsets[l]) | (uint32_t)(list_offsets[l + 1] < list_offsets[l]);
    if (bad) return fail(NBX_ERR_INVALID, "CSR offsets must be non-decreasing");
    if ((slots && !leaf_bodies) || (n_list && !list_sources)) return fail(NBX_ERR_INVALID, "null leaf arrays");
    *slots_out = slots;
    *n_list_out = n_list;
    return NBX_OK;
}

// Which planner lays a structure out.  NBODY_HIP_LEAF_PLANNER=host|device in the environment decides for every call (tests run
// every case through both); otherwise the device takes structures from kDevicePlanFrom slots + list entries on -- below that the
// host's few microseconds beat the device planner's ~35 launches.
constexpr size_t kDevicePlanFrom = 65536;
bool use_device_planner(size_t n_leaves, size_t slots, size_t n_list) {
    if (n_leaves == 0 || slots == 0) return false;                                  // nothing to lay out: the host path's early exits
    if (slots + n_leaves > 0xfffffff0ull || n_list > 0xfffffff0ull) return false;     // the host planner words the refusal
    if (kPackWindowWaves * (size_t)kPackMaxSubs > 128) return false;                  // A/B builds with larger windows
    if (const char* e = std::getenv("NBODY_HIP_LEAF_PLANNER")) {
        if (!std::strcmp(e, "host")) return false;
        if (!std::strcmp(e, "device")) return true;
    }
    return slots + n_list >= kDevicePlanFrom;
}

// Host-side validation of the CSR structure: every index the kernels will follow is checked here, before anything is launched.
int validate_csr(size_t n, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets,
                 const uint32_t* list_sources, size_t* slots_out) {
    if (n > ((size_t)1 << 31) || n_leaves > ((size_t)1 << 31)) return fail(NBX_ERR_INVALID, "too many bodies / leaves");
    if (n_leaves && (!leaf_offsets || !list_offsets)) return fail(NBX_ERR_INVALID, "null leaf arrays");
    const size_t slots = n_leaves ? leaf_offsets[n_leaves] : 0;
    const size_t n_list = n_leaves ? list_offsets[n_leaves] : 0;
    if (n_leaves && (leaf_offsets[0] != 0 || list_offsets[0] != 0)) return fail(NBX_ERR_INVALID, "CSR offsets must start at 0");
    for (size_t l = 0; l < n_leaves; ++l)
        if (leaf_offsets[l + 1] < leaf_offsets[l] || list_offsets[l + 1] < list_offsets[l]) return fail(NBX_ERR_INVALID, "CSR offsets must be non-decreasing");
    if ((slots && !leaf_bodies) || (n_list && !list_sources)) return fail(NBX_ERR_INVALID, "null leaf arrays");
    {
        std::vector<unsigned char> seen;
        try { seen.assign(n, 0); } catch (...) { return fail(NBX_ERR_ALLOC, "host allocation failed"); }
        for (size_t s = 0; s < slots; ++s) {
            const uint32_t b = leaf_bodies[s];
            if (b >= n) return fail(NBX_ERR_INVALID, "leaf_bodies entry out of range");
            if (seen[b]) return fail(NBX_ERR_INVALID, "a body may belong to at most one leaf");
            seen[b] = 1;
        }
    }
    {   // the largest entry, without an exit inside the loop (so that it is vectorised: 6.7 million entries for 65,536 BVH leaves)
        uint32_t largest = 0;
        for (size_t e = 0; e < n_list; ++e) largest = list_sources[e] > largest ? list_sources[e] : largest;
        if (n_list && largest >= n_leaves) return fail(NBX_ERR_INVALID, "list_sources entry out of range");
    }
    *slots_out = slots;
    return NBX_OK;
}

// plan_leaves behind the C ABI: no exception leaves it (its arrays are std::vectors), an allocation failure is NBX_ERR_ALLOC
int lay_out_launch(const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets, const uint32_t* list_sources,
                   LeafPlan& plan) {
    const char* why = nullptr;
    try {
        why = plan_leaves(leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, plan, NBX_LEAF_PACK != 0);
    } catch (...) {
        why = kPlanAllocFailed;
    }
    if (!why) return NBX_OK;
    return fail(why == kPlanAllocFailed ? NBX_ERR_ALLOC : NBX_ERR_INVALID, why);
}

int create_plan(nbx_leaf_plan** out, int device, int dim, size_t n, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves,
                const uint32_t* list_offsets, const uint32_t* list_sources, size_t forces_bytes);

// The caller's current HIP device is put back when an entry point of this file returns
struct DeviceScope {
    int before = -1;
    DeviceScope() { if (hipGetDevice(&before) != hipSuccess) before = -1; (void)hipGetLastError(); }
    ~DeviceScope() { if (before >= 0) (void)hipSetDevice(before); }
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
    const size_t min_stride = (size_t)(2 * dim + 1) * sizeof(double);
    if (stride_bytes < min_stride || stride_bytes % sizeof(double) != 0)
        return fail(NBX_ERR_INVALID, "body stride must be a multiple of 8 and >= sizeof(Body<dim>)");
    size_t slots = 0, n_list = 0;
    if (int vrc = validate_offsets(n, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, &slots, &n_list)) return vrc;
    if (use_device_planner(n_leaves, slots, n_list)) {
        // the structure laid out on the device (leaf_plan_device.h): a plan for this call alone, its block and the staged bodies' from the
        // parked pool, so that a tree code calling once per step allocates nothing
        nbx_leaf_plan* p = nullptr;
        int prc = create_plan(&p, device, dim, n, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, n * (size_t)dim * sizeof(double));
        if (prc) return prc;
        DeviceScope scope;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = take_arena(device, n * stride_bytes + 256, &p->raw_arena, &p->raw_arena_bytes);
        if (e != hipSuccess) { prc = nbx::fail_hip(e, "staging the bodies", __FILE__, __LINE__); nbx_leaf_plan_destroy(p); return prc; }
        p->raw = reinterpret_cast<double*>(p->raw_arena);
        p->raw_bytes = n * stride_bytes;
        prc = nbx_leaf_plan_forces(p, bodies, stride_bytes, law, G, forces_out, kernel_ms);
        nbx_leaf_plan_destroy(p);
        return prc;
    }
    if (int vrc = validate_csr(n, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, &slots)) return vrc;
    int ndev = 0;
    int rc = nbx_device_count(&ndev);
    if (rc != NBX_OK) return rc;
    if (device < 0 || device >= ndev) return fail(NBX_ERR_NO_DEVICE, "device ordinal out of range");
    if (slots == 0) {   // no leaf holds a body: every force is zero (otherwise the device array, zeroed there, is copied out whole)
        for (size_t i = 0; i < n * (size_t)dim; ++i) forces_out[i] = 0.0;
        return NBX_OK;
    }

    DeviceScope scope;   // the caller's current device is restored on every path
    NBX_HIP_TRY(hipSetDevice(device));
    DeviceBuffers d;
    d.device = device;
    NBX_HIP_TRY(nbx::take_stream(device, &d.stream));
    NBX_HIP_TRY(hipEventCreate(&d.ev0));
    NBX_HIP_TRY(hipEventCreate(&d.ev1));
    // The bodies go to the device on a helper thread (58 MB at N = 2^20; the copy from pageable memory blocks its caller for 1 ms)
    // while this thread lays out the launch.
    NBX_HIP_TRY(take_arena(device, n * stride_bytes + 256, &d.body_arena, &d.body_bytes));
    double* const raw = reinterpret_cast<double*>(d.body_arena);
    hipError_t copy_rc = hipSuccess;
    std::thread copier;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{copier};   // before d goes, on every path
    if (n * stride_bytes >= kHelperCopyBytes) {
        try {
            copier = std::thread([&]() {
                copy_rc = hipSetDevice(device);
                if (copy_rc == hipSuccess) copy_rc = hipMemcpyAsync(raw, bodies, n * stride_bytes, hipMemcpyHostToDevice, d.stream);
            });
        } catch (...) {   // no thread to be had: copy here
        }
    }
    if (!copier.joinable())   // small input (a thread costs more than the copy takes), or no thread
        NBX_HIP_TRY(hipMemcpyAsync(raw, bodies, n * stride_bytes, hipMemcpyHostToDevice, d.stream));

    // ---- the layout the kernel follows (leaf_plan.h; comment at the top of this file) ----
    static thread_local LeafPlan plan;   // a tree code calls once per step: the arrays keep their capacity (and their pages) between calls
    if (int prc = lay_out_launch(leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, plan)) return prc;
    const size_t pslots = plan.pslots();
    const std::vector<uint32_t>& pslot_body = plan.pslot_body;
    const std::vector<CopyOp>& ops = plan.ops;
    const std::vector<LeafBlock>& blocks = plan.blocks;
    const int waves = plan.waves;

    // one allocation for the call's other device arrays (each hipFree of a large buffer costs 0.2 ms on this runtime)
    const size_t sizes[9] = {pslots * sizeof(float4), (size_t)dim * pslots * sizeof(double), n * (size_t)dim * sizeof(double),
                             pslots * sizeof(uint32_t), ops.size() * sizeof(CopyOp), blocks.size() * sizeof(LeafBlock), sizeof(uint32_t),
                             plan.pack_subs.size() * sizeof(PackSub), plan.pack_blocks.size() * sizeof(PackBlock)};
    size_t offs[9], total_bytes = 0;
    for (int i = 0; i < 9; ++i) { offs[i] = total_bytes; total_bytes += (sizes[i] + 255) / 256 * 256 + 256; }
    NBX_HIP_TRY(take_arena(device, total_bytes, &d.arena, &d.arena_bytes));
    char* const arena = d.arena;
    float4* xp = reinterpret_cast<float4*>(arena + offs[0]);
    double* acc = reinterpret_cast<double*>(arena + offs[1]);
    double* dforces = reinterpret_cast<double*>(arena + offs[2]);
    uint32_t* d_pb = reinterpret_cast<uint32_t*>(arena + offs[3]);
    CopyOp* d_ops = reinterpret_cast<CopyOp*>(arena + offs[4]);
    LeafBlock* d_blocks = reinterpret_cast<LeafBlock*>(arena + offs[5]);
    uint32_t* d_max_mass = reinterpret_cast<uint32_t*>(arena + offs[6]);
    PackSub* d_subs = reinterpret_cast<PackSub*>(arena + offs[7]);
    PackBlock* d_packs = reinterpret_cast<PackBlock*>(arena + offs[8]);
    if (copier.joinable()) copier.join();
    NBX_HIP_TRY(copy_rc);
    NBX_HIP_TRY(hipMemcpyAsync(d_pb, pslot_body.data(), pslots * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    if (!ops.empty()) NBX_HIP_TRY(hipMemcpyAsync(d_ops, ops.data(), ops.size() * sizeof(CopyOp), hipMemcpyHostToDevice, d.stream));
    if (!blocks.empty()) NBX_HIP_TRY(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(LeafBlock), hipMemcpyHostToDevice, d.stream));
    if (!plan.pack_blocks.empty()) {
        NBX_HIP_TRY(hipMemcpyAsync(d_subs, plan.pack_subs.data(), sizes[7], hipMemcpyHostToDevice, d.stream));
        NBX_HIP_TRY(hipMemcpyAsync(d_packs, plan.pack_blocks.data(), sizes[8], hipMemcpyHostToDevice, d.stream));
    }
    NBX_HIP_TRY(hipMemsetAsync(dforces, 0, n * (size_t)dim * sizeof(double), d.stream));
    NBX_HIP_TRY(hipMemsetAsync(d_max_mass, 0, sizeof(uint32_t), d.stream));
    (void)hipGetLastError();
    const unsigned gs = (unsigned)((pslots + 255) / 256);
    hipLaunchKernelGGL(leaf_gather_kernel, dim3(gs), dim3(256), 0, d.stream, raw, stride_bytes / sizeof(double), dim, d_pb, (uint32_t)pslots,
                       reinterpret_cast<float*>(xp), d_max_mass);
    NBX_HIP_TRY(hipGetLastError());
    LeafArgs a;
    a.xp = xp; a.pslots = (uint32_t)pslots; a.ops = d_ops; a.blocks = d_blocks; a.acc = acc; a.max_mass_bits = d_max_mass;
    NBX_HIP_TRY(hipEventRecord(d.ev0, d.stream));
    if (!blocks.empty()) {   // one-leaf workgroups first: they are the long ones
        hipLaunchKernelGGL(pick(dim, law, waves), dim3((unsigned)blocks.size()), dim3(64u * (unsigned)waves), 0, d.stream, a);
        NBX_HIP_TRY(hipGetLastError());
    }
    if (!plan.pack_blocks.empty()) {
        LeafPackArgs pa;
        pa.xp = xp; pa.pslots = (uint32_t)pslots; pa.ops = d_ops; pa.blocks = d_packs; pa.subs = d_subs; pa.acc = acc; pa.max_mass_bits = d_max_mass;
        hipLaunchKernelGGL(pick_pack(dim, law), dim3((unsigned)plan.pack_blocks.size()), dim3(64), 0, d.stream, pa);
        NBX_HIP_TRY(hipGetLastError());
    }
    NBX_HIP_TRY(hipEventRecord(d.ev1, d.stream));
    const double signedG = (law == NBX_LAW_BRUTE) ? -G : G;   // brute force: forces[i] -= f (methods.cpp:131); tree codes: += (attractive)
    hipLaunchKernelGGL(leaf_scatter_kernel, dim3(gs), dim3(256), 0, d.stream, acc, raw, stride_bytes / sizeof(double), dim, d_pb, (uint32_t)pslots,
                       signedG, dforces);
    NBX_HIP_TRY(hipGetLastError());
    NBX_HIP_TRY(hipMemcpyAsync(forces_out, dforces, n * (size_t)dim * sizeof(double), hipMemcpyDeviceToHost, d.stream));
    NBX_HIP_TRY(hipStreamSynchronize(d.stream));
    if (kernel_ms) NBX_HIP_TRY(hipEventElapsedTime(kernel_ms, d.ev0, d.ev1));
    return NBX_OK;
}

namespace {

int plan_set_device(const nbx_leaf_plan* p) {
    (void)hipGetLastError();
    NBX_HIP_TRY(hipSetDevice(p->device));
    return NBX_OK;
}

// Work about to be queued on `s` must see the plan's buffers as the last evaluation (possibly on another stream) left them.
int plan_order_after_last(nbx_leaf_plan* p, hipStream_t s) {
    if (p->last_stream && p->last_stream != s) NBX_HIP_TRY(hipStreamWaitEvent(s, p->done, 0));
    return NBX_OK;
}

int plan_mark_done(nbx_leaf_plan* p, hipStream_t s) {
    NBX_HIP_TRY(hipEventRecord(p->done, s));
    p->last_stream = s;
    return NBX_OK;
}

int plan_launch_pairs(nbx_leaf_plan* p, int law, hipStream_t s, bool timed) {
    if (p->n_blocks == 0 && p->n_packs == 0) return NBX_OK;
    if (timed) NBX_HIP_TRY(hipEventRecord(p->ev0, s));
    LeafArgs a;
    a.xp = p->xp; a.pslots = (uint32_t)p->pslots; a.ops = p->ops; a.blocks = p->blocks; a.acc = p->sums; a.max_mass_bits = p->max_mass;
    LeafPackArgs pa;
    pa.xp = p->xp; pa.pslots = (uint32_t)p->pslots; pa.ops = p->ops; pa.blocks = p->packs; pa.subs = p->subs; pa.acc = p->sums; pa.max_mass_bits = p->max_mass;
    if (p->n_blocks && p->n_packs && p->waves == 1) {   // both kinds (packing implies one-wave workgroups): one launch, the one-leaf workgroups first
        hipLaunchKernelGGL(pick_fused(p->dim, law), dim3((unsigned)(p->n_blocks + p->n_packs)), dim3(64), 0, s, a, pa, (uint32_t)p->n_blocks);
        NBX_HIP_TRY(hipGetLastError());
    } else {
        if (p->n_blocks) {
            hipLaunchKernelGGL(pick(p->dim, law, p->waves), dim3((unsigned)p->n_blocks), dim3(64u * (unsigned)p->waves), 0, s, a);
            NBX_HIP_TRY(hipGetLastError());
        }
        if (p->n_packs) {
            hipLaunchKernelGGL(pick_pack(p->dim, law), dim3((unsigned)p->n_packs), dim3(64), 0, s, pa);
            NBX_HIP_TRY(hipGetLastError());
        }
    }
    if (timed) NBX_HIP_TRY(hipEventRecord(p->ev1, s));
    return NBX_OK;
}

int plan_forces_out(nbx_leaf_plan* p, hipStream_t s, double* forces_out) {
    if (p->last_ctx_id && !nbx::ctx_alive(p->last_ctx_id))   // the masses were read where the evaluation found them: in a context that is gone
        return fail(NBX_ERR_STATE, "the context of the last evaluation no longer exists: evaluate again before asking for forces");
    if (!p->forces && p->n) NBX_HIP_TRY(hipMalloc((void**)&p->forces, p->n * (size_t)p->dim * sizeof(double)));
    if (p->n) {
        hipLaunchKernelGGL(leaf_forces_by_body_kernel, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, s, p->sums, (uint32_t)p->pslots,
                           p->body_slot, p->n, p->dim, p->last_signedG, p->last_mass, p->last_mass_stride, p->forces);
        NBX_HIP_TRY(hipGetLastError());
        NBX_HIP_TRY(hipMemcpyAsync(forces_out, p->forces, p->n * (size_t)p->dim * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    NBX_HIP_TRY(hipStreamSynchronize(s));
    return NBX_OK;
}


#define PLAN_TRY(expr)                                                                                             \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) { const int r_ = nbx::fail_hip(e_, #expr, __FILE__, __LINE__); nbx_leaf_plan_destroy(p); return r_; } \
    } while (0)

// The plan's buffers on the device, laid out there (leaf_plan_device.h): the four CSR arrays go over as they are, ~35 kernels build
// what plan_leaves builds on the host, 64 bytes come back.  forces_bytes > 0 reserves the one-shot call's force array in the same block.
int create_plan_on_device(nbx_leaf_plan* p, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets,
                          const uint32_t* list_sources, size_t slots, size_t n_list, size_t forces_bytes) {
    using namespace nbx_leaf_dev;
    const Bounds b{p->n, n_leaves, slots, n_list};
    const Layout L = make_layout(b, p->dim);
    const size_t forces_off = L.total;
    const size_t total = L.total + (forces_bytes ? (forces_bytes + 255) / 256 * 256 + 256 : 0);
    PLAN_TRY(take_arena(p->device, total, &p->arena, &p->arena_bytes));
    const DevicePlan d = plan_pointers(p->arena, L);
    p->xp = d.xp; p->sums = d.sums; p->pslot_body = d.pslot_body; p->body_slot = d.body_slot; p->ops = d.ops; p->blocks = d.blocks;
    p->subs = d.subs; p->packs = d.packs; p->max_mass = d.max_mass;
    if (forces_bytes) { p->forces = reinterpret_cast<double*>(p->arena + forces_off); p->forces_in_arena = true; }
    PLAN_TRY(enqueue_device_plan(b, p->dim, leaf_offsets, leaf_bodies, list_offsets, list_sources, NBX_LEAF_PACK != 0, p->arena, L, p->stream, &p->summary_host));
    PLAN_TRY(hipStreamSynchronize(p->stream));
    const Summary& S = p->summary_host;
    if (S.err != kErrNone) {
        const uint32_t code = S.err;
        nbx_leaf_plan_destroy(p);
        return fail(NBX_ERR_INVALID, error_text(code));
    }
    p->device_planned = true;
    p->waves = (int)S.waves; p->pslots = S.pslots; p->n_ops = S.n_ops; p->n_blocks = S.n_blocks; p->n_subs = S.n_subs; p->n_packs = S.n_packs;
    return NBX_OK;
}

int create_plan_on_host(nbx_leaf_plan* p, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves, const uint32_t* list_offsets,
                        const uint32_t* list_sources, size_t forces_bytes) {
    const size_t n = p->n;
    const int dim = p->dim;
    LeafPlan host;
    if (int prc = lay_out_launch(leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, host)) { nbx_leaf_plan_destroy(p); return prc; }
    p->waves = host.waves;
    p->pslots = host.pslots(); p->n_ops = host.ops.size(); p->n_blocks = host.blocks.size();
    p->n_subs = host.pack_subs.size(); p->n_packs = host.pack_blocks.size();
    std::vector<uint32_t> body_slot;
    try { body_slot.assign(n, 0xffffffffu); } catch (...) { nbx_leaf_plan_destroy(p); return fail(NBX_ERR_ALLOC, "host allocation failed"); }
    for (size_t s = 0; s < p->pslots; ++s)
        if (host.pslot_body[s] != 0xffffffffu) body_slot[host.pslot_body[s]] = (uint32_t)s;
    const size_t sizes[10] = {(p->pslots + 2) * sizeof(float4), (size_t)dim * p->pslots * sizeof(double), p->pslots * sizeof(uint32_t),
                              n * sizeof(uint32_t), p->n_ops * sizeof(CopyOp), p->n_blocks * sizeof(LeafBlock), sizeof(uint32_t),
                              p->n_subs * sizeof(PackSub), p->n_packs * sizeof(PackBlock), forces_bytes};
    size_t offs[10], total = 0;
    for (int i = 0; i < 10; ++i) { offs[i] = total; total += (sizes[i] + 255) / 256 * 256 + 256; }
    PLAN_TRY(take_arena(p->device, total, &p->arena, &p->arena_bytes));   // a tree code makes a plan per step: the last plan's block, parked by its destroy
    p->xp = reinterpret_cast<float4*>(p->arena + offs[0]);
    p->sums = reinterpret_cast<double*>(p->arena + offs[1]);
    p->pslot_body = reinterpret_cast<uint32_t*>(p->arena + offs[2]);
    p->body_slot = reinterpret_cast<uint32_t*>(p->arena + offs[3]);
    p->ops = reinterpret_cast<CopyOp*>(p->arena + offs[4]);
    p->blocks = reinterpret_cast<LeafBlock*>(p->arena + offs[5]);
    p->max_mass = reinterpret_cast<uint32_t*>(p->arena + offs[6]);
    p->subs = reinterpret_cast<PackSub*>(p->arena + offs[7]);
    p->packs = reinterpret_cast<PackBlock*>(p->arena + offs[8]);
    if (forces_bytes) { p->forces = reinterpret_cast<double*>(p->arena + offs[9]); p->forces_in_arena = true; }
    if (p->n_packs) {
        PLAN_TRY(hipMemcpyAsync(p->subs, host.pack_subs.data(), sizes[7], hipMemcpyHostToDevice, p->stream));
        PLAN_TRY(hipMemcpyAsync(p->packs, host.pack_blocks.data(), sizes[8], hipMemcpyHostToDevice, p->stream));
    }
    if (p->pslots) PLAN_TRY(hipMemcpyAsync(p->pslot_body, host.pslot_body.data(), sizes[2], hipMemcpyHostToDevice, p->stream));
    if (n) PLAN_TRY(hipMemcpyAsync(p->body_slot, body_slot.data(), sizes[3], hipMemcpyHostToDevice, p->stream));
    if (p->n_ops) PLAN_TRY(hipMemcpyAsync(p->ops, host.ops.data(), sizes[4], hipMemcpyHostToDevice, p->stream));
    if (p->n_blocks) PLAN_TRY(hipMemcpyAsync(p->blocks, host.blocks.data(), sizes[5], hipMemcpyHostToDevice, p->stream));
    PLAN_TRY(hipStreamSynchronize(p->stream));   // the host arrays above go out of scope
    return NBX_OK;
}

// nbx_leaf_plan_create and the one-shot call's plan: validation, the layout (device or host planner), the buffers an evaluation needs.
int create_plan(nbx_leaf_plan** out, int device, int dim, size_t n, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies, size_t n_leaves,
                const uint32_t* list_offsets, const uint32_t* list_sources, size_t forces_bytes) {
    if (!out) return fail(NBX_ERR_INVALID, "out is null");
    *out = nullptr;
    if (dim != 2 && dim != 3) return fail(NBX_ERR_INVALID, "dim must be 2 or 3");
    size_t slots = 0, n_list = 0;
    if (int vrc = validate_offsets(n, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, &slots, &n_list)) return vrc;
    const bool on_device = use_device_planner(n_leaves, slots, n_list);
    if (!on_device)   // the host planner follows every index: all of them are checked first (the device planner checks as it goes)
        if (int vrc = validate_csr(n, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, &slots)) return vrc;
    int ndev = 0;
    int rc = nbx_device_count(&ndev);
    if (rc != NBX_OK) return rc;
    if (device < 0 || device >= ndev) return fail(NBX_ERR_NO_DEVICE, "device ordinal out of range");
    nbx_leaf_plan* p = new (std::nothrow) nbx_leaf_plan();
    if (!p) return fail(NBX_ERR_ALLOC, "host allocation failed");
    p->device = device; p->dim = dim; p->n = n;
    DeviceScope scope;
    PLAN_TRY(hipSetDevice(device));
    PLAN_TRY(nbx::take_stream(device, &p->stream));
    PLAN_TRY(hipEventCreate(&p->ev0));
    PLAN_TRY(hipEventCreate(&p->ev1));
    PLAN_TRY(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    rc = on_device ? create_plan_on_device(p, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, slots, n_list, forces_bytes)
                   : create_plan_on_host(p, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, forces_bytes);
    if (rc) return rc;                                                    // p is gone already
    // queued behind the layout, waited for by whoever evaluates first (plan_order_after_last): the sums of slots no workgroup
    // writes (a leaf's pad) stay zero, and the pads of odd leaves are massless and far away (the body-major gather never touches them)
    const size_t sum_bytes = (size_t)dim * p->pslots * sizeof(double);
    PLAN_TRY(hipMemsetAsync(p->sums, 0, sum_bytes ? sum_bytes : 8, p->stream));
    if (p->pslots) {
        hipLaunchKernelGGL(leaf_init_pads_kernel, dim3((unsigned)((p->pslots + 255) / 256)), dim3(256), 0, p->stream, p->pslot_body, (uint32_t)p->pslots,
                           dim, reinterpret_cast<float*>(p->xp));
        PLAN_TRY(hipGetLastError());
    }
    if (int mrc = plan_mark_done(p, p->stream)) { nbx_leaf_plan_destroy(p); return mrc; }
    *out = p;
    return NBX_OK;
}
#undef PLAN_TRY

}  // namespace

extern "C" {

int nbx_leaf_plan_create(nbx_leaf_plan** out, int device, int dim, size_t n, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies,
                         size_t n_leaves, const uint32_t* list_offsets, const uint32_t* list_sources) {
    return create_plan(out, device, dim, n, leaf_offsets, leaf_bodies, n_leaves, list_offsets, list_sources, 0);
}

int nbx_leaf_plan_destroy(nbx_leaf_plan* p) {
    if (!p) return NBX_OK;
    DeviceScope scope;
    (void)hipSetDevice(p->device);
    // the last evaluation may have been queued on a context's stream, and that context may be gone by now (its stream with it):
    // wait on the plan's own event, which every piece of work queued on a foreign stream is followed by
    p->last_wait_ok = !(p->last_stream && p->done) || hipEventSynchronize(p->done) == hipSuccess;
    bool idle = p->stream && hipStreamSynchronize(p->stream) == hipSuccess;
    if (p->arena) { if (idle && p->last_wait_ok) park_arena(p->device, p->arena, p->arena_bytes); else (void)hipFree(p->arena); }
    if (p->forces && !p->forces_in_arena) (void)hipFree(p->forces);
    if (p->raw_arena) { if (idle && p->last_wait_ok) park_arena(p->device, p->raw_arena, p->raw_arena_bytes); else (void)hipFree(p->raw_arena); }
    else if (p->raw) (void)hipFree(p->raw);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->done) (void)hipEventDestroy(p->done);
    if (p->stream) { if (idle) nbx::park_stream(p->device, p->stream); else (void)hipStreamDestroy(p->stream); }
    (void)hipGetLastError();
    delete p;
    return NBX_OK;
}

int nbx_leaf_plan_info(const nbx_leaf_plan* p, size_t* slots, size_t* runs, size_t* workgroups, int* waves) {
    if (!p) return fail(NBX_ERR_INVALID, "plan is null");
    if (slots) *slots = p->pslots;
    if (runs) *runs = p->n_ops;
    if (workgroups) *workgroups = p->n_blocks + p->n_packs;
    if (waves) *waves = p->waves;
    return NBX_OK;
}

int nbx_leaf_plan_forces(nbx_leaf_plan* p, const void* bodies, size_t stride_bytes, int law, double G, double* forces_out, float* kernel_ms) {
    if (kernel_ms) *kernel_ms = 0.0f;
    if (!p) return fail(NBX_ERR_INVALID, "plan is null");
    if (law < NBX_LAW_BRUTE || law > NBX_LAW_FMM_P2P) return fail(NBX_ERR_INVALID, "unknown law");
    if ((!bodies || !forces_out) && p->n) return fail(NBX_ERR_INVALID, "null argument");
    const size_t min_stride = (size_t)(2 * p->dim + 1) * sizeof(double);
    if (stride_bytes < min_stride || stride_bytes % sizeof(double) != 0)
        return fail(NBX_ERR_INVALID, "body stride must be a multiple of 8 and >= sizeof(Body<dim>)");
    DeviceScope scope;
    int rc = plan_set_device(p);
    if (rc) return rc;
    hipStream_t s = p->stream;
    if ((rc = plan_order_after_last(p, s))) return rc;
    const size_t bytes = p->n * stride_bytes;
    if (bytes > p->raw_bytes) {
        if (p->raw) { NBX_HIP_TRY(hipStreamSynchronize(s)); NBX_HIP_TRY(hipFree(p->raw)); p->raw = nullptr; p->raw_bytes = 0; }
        NBX_HIP_TRY(hipMalloc((void**)&p->raw, bytes + 256));
        p->raw_bytes = bytes;
    }
    if (bytes) NBX_HIP_TRY(hipMemcpyAsync(p->raw, bodies, bytes, hipMemcpyHostToDevice, s));
    NBX_HIP_TRY(hipMemsetAsync(p->max_mass, 0, sizeof(uint32_t), s));
    if (p->pslots) {
        hipLaunchKernelGGL(leaf_gather_kernel, dim3((unsigned)((p->pslots + 255) / 256)), dim3(256), 0, s, p->raw, stride_bytes / sizeof(double), p->dim,
                           p->pslot_body, (uint32_t)p->pslots, reinterpret_cast<float*>(p->xp), p->max_mass);
        NBX_HIP_TRY(hipGetLastError());
    }
    if ((rc = plan_launch_pairs(p, law, s, true))) return rc;
    p->evaluated = true; p->last_law = law;
    p->last_signedG = (law == NBX_LAW_BRUTE) ? -G : G;   // brute force: forces[i] -= f (methods.cpp:131); tree codes: += (attractive)
    p->last_mass = p->raw + 2 * p->dim; p->last_mass_stride = stride_bytes / sizeof(double); p->last_ctx_id = 0;
    if ((rc = plan_mark_done(p, s))) return rc;
    if ((rc = plan_forces_out(p, s, forces_out))) return rc;
    if (kernel_ms && (p->n_blocks || p->n_packs)) NBX_HIP_TRY(hipEventElapsedTime(kernel_ms, p->ev0, p->ev1));
    return NBX_OK;
}

// positions and masses of a context's resident bodies -> the plan's leaf-ordered source pairs, on stream s.  One lane per body
// (coalesced reads, four 4-byte stores into its slot's pair record): 0.048 ms at N = 2^20.  A two-kernel form (SoA -> one float4 per
// body, then one lane per slot reading its body's 16 bytes and writing whole records) was measured at 0.006 + 0.044 ms: no better.
// What had made this gather 0.14-0.17 ms was not its memory traffic but publish_max_mass's predecessor (tools/ubench_gather.hip:
// the traffic alone is 0.02 ms back to back).
static int plan_gather_resident(nbx_leaf_plan* p, nbx_ctx* c, hipStream_t s) {
    if (!p->pslots || !p->n) return NBX_OK;
    hipLaunchKernelGGL(leaf_gather_by_body_kernel, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, s, c->pos_all, c->mass_all, c->pad, p->dim,
                       p->body_slot, p->n, reinterpret_cast<float*>(p->xp), p->max_mass);
    NBX_HIP_TRY(hipGetLastError());
    return NBX_OK;
}

int nbx_leaf_plan_forces_ctx(nbx_leaf_plan* p, nbx_ctx* c, int law, double G, double* forces_out, float* kernel_ms) {
    if (kernel_ms) *kernel_ms = 0.0f;
    if (!p || !c) return fail(NBX_ERR_INVALID, "null argument");
    if (law < NBX_LAW_BRUTE || law > NBX_LAW_FMM_P2P) return fail(NBX_ERR_INVALID, "unknown law");
    if (c->device != p->device || c->dim != p->dim || c->n_total != p->n || c->n_shards != 1)
        return fail(NBX_ERR_INVALID, "the context must be a single-shard context of the plan's device, dimension and body count");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "upload bodies to the context first");
    DeviceScope scope;
    int rc = plan_set_device(p);
    if (rc) return rc;
    hipStream_t s = c->stream;
    if ((rc = plan_order_after_last(p, s))) return rc;
    NBX_HIP_TRY(hipMemsetAsync(p->max_mass, 0, sizeof(uint32_t), s));
    if ((rc = plan_gather_resident(p, c, s))) return rc;
    if ((rc = plan_launch_pairs(p, law, s, kernel_ms != nullptr))) return rc;
    p->evaluated = true; p->last_law = law;
    p->last_signedG = (law == NBX_LAW_BRUTE) ? -G : G;
    p->last_mass = c->m64; p->last_mass_stride = 1; p->last_ctx_id = c->id;
    if ((rc = plan_mark_done(p, s))) return rc;
    if (forces_out) { if ((rc = plan_forces_out(p, s, forces_out))) return rc; }
    else if (kernel_ms) NBX_HIP_TRY(hipStreamSynchronize(s));
    if (kernel_ms && (p->n_blocks || p->n_packs)) NBX_HIP_TRY(hipEventElapsedTime(kernel_ms, p->ev0, p->ev1));
    return NBX_OK;
}

int nbx_leaf_plan_get_forces(nbx_leaf_plan* p, double* forces_out) {
    if (!p || (!forces_out && p->n)) return fail(NBX_ERR_INVALID, "null argument");
    if (!p->evaluated) return fail(NBX_ERR_STATE, "no evaluation on the device");
    DeviceScope scope;
    int rc = plan_set_device(p);
    if (rc) return rc;
    // on the plan's own stream, behind the last evaluation's event (the stream that evaluation ran on may belong to a context that no longer exists)
    if ((rc = plan_order_after_last(p, p->stream))) return rc;
    if ((rc = plan_forces_out(p, p->stream, forces_out))) return rc;
    return plan_mark_done(p, p->stream);
}

int nbx_leaf_plan_kick_drift(nbx_leaf_plan* p, nbx_ctx* c, double dt) {
    if (!p || !c) return fail(NBX_ERR_INVALID, "null argument");
    if (!p->evaluated) return fail(NBX_ERR_STATE, "evaluate the leaf sums before kick_drift");
    if (c->device != p->device || c->dim != p->dim || c->n_total != p->n || c->n_shards != 1)
        return fail(NBX_ERR_INVALID, "the context must be a single-shard context of the plan's device, dimension and body count");
    if (p->last_ctx_id != c->id) return fail(NBX_ERR_STATE, "the last evaluation was not made from this context");
    DeviceScope scope;
    int rc = plan_set_device(p);
    if (rc) return rc;
    hipStream_t s = c->stream;
    if ((rc = plan_order_after_last(p, s))) return rc;
    SlotKickArgs k;
    k.sums = p->sums; k.body_slot = p->body_slot; k.pslots = (uint32_t)p->pslots; k.dim = p->dim; k.pad = c->pad; k.count = c->count;
    k.signedG = p->last_signedG; k.dt = dt; k.x64 = c->x64; k.v64 = c->v64; k.m64 = c->m64; k.pos_chunk = c->pos_all;
    NBX_HIP_TRY(launch_kick_drift_slots(k, s));
    c->have_accel = false;                          // the context's own accelerations (if any) belong to the old positions
    c->tgt_cand_valid = 0; c->bad_list_pass = -1;
    return plan_mark_done(p, s);
}

// one step's device work on stream s, nothing else (no events, no waits)
static int plan_enqueue_step(nbx_leaf_plan* p, nbx_ctx* c, int law, double signedG, double dt, hipStream_t s) {
    NBX_HIP_TRY(hipMemsetAsync(p->max_mass, 0, sizeof(uint32_t), s));
    if (int rc = plan_gather_resident(p, c, s)) return rc;
    if (int rc = plan_launch_pairs(p, law, s, false)) return rc;
    SlotKickArgs k;
    k.sums = p->sums; k.body_slot = p->body_slot; k.pslots = (uint32_t)p->pslots; k.dim = p->dim; k.pad = c->pad; k.count = c->count;
    k.signedG = signedG; k.dt = dt; k.x64 = c->x64; k.v64 = c->v64; k.m64 = c->m64; k.pos_chunk = c->pos_all;
    NBX_HIP_TRY(launch_kick_drift_slots(k, s));
    return NBX_OK;
}

int nbx_leaf_plan_step(nbx_leaf_plan* p, nbx_ctx* c, int law, double G, double dt, int nsteps) {
    if (!p || !c) return fail(NBX_ERR_INVALID, "null argument");
    if (law < NBX_LAW_BRUTE || law > NBX_LAW_FMM_P2P) return fail(NBX_ERR_INVALID, "unknown law");
    if (nsteps < 0) return fail(NBX_ERR_INVALID, "nsteps must be >= 0");
    if (c->device != p->device || c->dim != p->dim || c->n_total != p->n || c->n_shards != 1)
        return fail(NBX_ERR_INVALID, "the context must be a single-shard context of the plan's device, dimension and body count");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "upload bodies to the context first");
    if (nsteps == 0) return NBX_OK;
    DeviceScope scope;
    int rc = plan_set_device(p);
    if (rc) return rc;
    hipStream_t s = c->stream;
    if ((rc = plan_order_after_last(p, s))) return rc;
    const double signedG = (law == NBX_LAW_BRUTE) ? -G : G;
    // Plain launches, queued ahead of the device: a step is GPU-bound (0.36 ms of kernels at N = 2^20; 5 launches cost the host
    // ~25 us).  A captured HIP graph was measured: 8-15 ms to capture and instantiate, then the same 72.4 ms per 200 steps at
    // N = 2^20 and 7.97 against 8.39 ms at N = 20,000 -- it would need thousands of steps to pay for itself (tools/time_leaf_steps.py).
    for (int k = 0; k < nsteps; ++k)
        if ((rc = plan_enqueue_step(p, c, law, signedG, dt, s))) return rc;
    p->evaluated = true; p->last_law = law;
    p->last_signedG = signedG;
    p->last_mass = c->m64; p->last_mass_stride = 1; p->last_ctx_id = c->id;
    c->have_accel = false;                          // the context's own accelerations (if any) belong to the old positions
    c->tgt_cand_valid = 0; c->bad_list_pass = -1;
    return plan_mark_done(p, s);
}

int nbx_leaf_plan_time_kernel(nbx_leaf_plan* p, int law, int reps, float* mean_ms) {
    if (!p || !mean_ms) return fail(NBX_ERR_INVALID, "null argument");
    *mean_ms = 0.0f;
    if (law < NBX_LAW_BRUTE || law > NBX_LAW_FMM_P2P) return fail(NBX_ERR_INVALID, "unknown law");
    if (reps < 1 || reps > 1000) return fail(NBX_ERR_INVALID, "reps must be in [1, 1000]");
    if (!p->evaluated) return fail(NBX_ERR_STATE, "evaluate once before timing (the bodies of the last evaluation are used)");
    DeviceScope scope;
    int rc = plan_set_device(p);
    if (rc) return rc;
    hipStream_t s = p->stream;   // the plan's own stream, behind the last evaluation (see nbx_leaf_plan_get_forces)
    if ((rc = plan_order_after_last(p, s))) return rc;
    const int timed_from = reps / 2;
    for (int r = 0; r < reps; ++r) {
        if (r == timed_from) NBX_HIP_TRY(hipEventRecord(p->ev0, s));
        if ((rc = plan_launch_pairs(p, law, s, false))) return rc;
    }
    NBX_HIP_TRY(hipEventRecord(p->ev1, s));
    NBX_HIP_TRY(hipStreamSynchronize(s));
    if (p->n_blocks || p->n_packs) NBX_HIP_TRY(hipEventElapsedTime(mean_ms, p->ev0, p->ev1));
    *mean_ms /= (float)(reps - timed_from);
    // the sums now belong to `law`: keep the bookkeeping of the last evaluation consistent with them
    p->last_signedG = (law == NBX_LAW_BRUTE) ? -std::fabs(p->last_signedG) : std::fabs(p->last_signedG);
    p->last_law = law;
    return plan_mark_done(p, s);
}

}  // extern "C"
