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
// leaf order as fp32 {x,y,z,m} (one 16-byte load per body), and the work is target-leaf-major: one workgroup (one wave64 when the mean
// leaf holds <= 80 bodies, else two) owns up to 64 (128) targets of ONE leaf and walks that leaf's source-leaf list as one
// stream of bodies staged through LDS in tiles of one body per lane {x,y,z,m}; fp32 sums per tile, flushed into fp64
// second-level accumulators.  No atomics, a fixed summation order (list order, then leaf order), every output written
// once.  The comment at the kernel says how the lanes share the work.  Leaves are small (the reference caps them at 100
// bodies, methods.h:26), so the launch is tens of thousands of short workgroups; HBM traffic is 16 B per (target block,
// source body) served mostly from L2.  VALU-issue-bound (counters: profiles/r2/pmc_leaf_pair_kernel.txt): 76 % of the
// instructions are the pair loop, the rest stages tiles and flushes sums; tiles of 256 bodies instead of 64, or half the
// LDS reads, changed nothing measurable.
#include "../../include/nbody_hip.h"
#include "nbx_ctx.h"

#include <cstdio>
#include <vector>

using namespace nbx;

namespace {

// targets per workgroup = source bodies per LDS tile: 128 lanes (two wave64), or one wave64 when the leaves are small
// (the mean leaf of the reference's trees is well under 100 bodies, methods.h:26) so that fewer lanes idle
constexpr int kLeafBlock = 128;
constexpr int kLeafBlockSmall = 64;
constexpr int kMaxLanesPerTarget = 8;   // a block of few targets gives each up to this many lanes (they split the sources)

// smallest fp32 thresholds that are >= the reference's fp64 ones, so (r2 < T_f32) == ((double)r2 < T) for fp32 r2
constexpr float kTreeSkipF = 0x1.12e0c0p-30f;   // 1.00000008e-9  (octree.cpp:119, bvh.cpp:167: dist_sq < 1e-9)
constexpr float kSmoothF = 0x1.b7cdfep-34f;     // 1.00000001e-10 (fmm_parlay.cpp:1010: dist_sq < 1e-10)
constexpr float kNormZeroF = 0x1.79ca12p-67f;   // 1.00000005e-20 (vector.h:93-97: |diff| < 1e-10 -> zero vector)
constexpr float kSameF = 1.0e-14f;              // largest fp32 <= 1e-14 (fmm_parlay.cpp:995-1000: |d_k| > 1e-14 -> distinct)
static_assert((double)kTreeSkipF >= 1e-9 && (double)kSmoothF >= 1e-10 && (double)kNormZeroF >= 1e-20 && (double)kSameF <= 1e-14,
              "fp32 thresholds must sit on the right side of the fp64 ones");

struct TargetBlock {
    uint32_t leaf;     // target leaf
    uint32_t first;    // first target slot (leaf order)
    uint32_t count;    // <= the launch's block size
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

// One workgroup = up to BLOCK targets of one leaf against that leaf's source list.
//  * The list is read ONCE, by the lanes in parallel (lane k: list entry k -> that source leaf's slot range), and turned
//    into one stream of source bodies by a prefix sum over the leaf sizes in LDS; tiles are then cut from the STREAM
//    (BLOCK consecutive stream positions, whatever leaves they fall in), not from single leaves.  Walking the list leaf
//    by leaf cost a chain of three dependent loads per ~30-body tile and left half of every tile empty.
//  * Two SOURCES per lane and iteration, as packed fp32 pairs (v_pk_add/fma/mul_f32: 3 + 3 + 2 + 3 packed instructions and
//    two v_rcp_f32 for two pairs, where one source at a time took 12 scalar ones and a v_rcp per pair).  The tile is
//    staged in LDS as source PAIRS {xa,xb,ya,yb},{za,zb,ma,mb}, so each ds_read_b128 lands in aligned register pairs.
//  * The law's special cases (skip / smoothing below ~1e-5 separation, and a body meeting itself in its own leaf) are
//    rare: one v_cmp per pair and a wave-wide vote; only a wave in which some lane sees r^2 below the law's threshold
//    takes the guarded scalar weights (leaf_weight) for that source pair -- which is every pair of the target's own leaf
//    (each source there is some lane's own body) and next to nothing else.
//  * Leaves are small (the reference caps them at 100 bodies; a uniform grid at 32 per leaf leaves half of a wave64 idle
//    with one lane per target): a block whose targets fill at most half / a quarter of the lanes gives each target 2 / 4
//    lanes, which split the source pairs of every tile between them; their fp64 sums meet in LDS at the end, in lane
//    group order (deterministic).
template <int D, int LAW, int BLOCK>
__global__ __launch_bounds__(BLOCK) void leaf_pair_kernel(LeafArgs a) {
    __shared__ float4 tile[BLOCK + 2 * kMaxLanesPerTarget];   // BLOCK/2 source pairs x 2 float4, + pad pairs past the end
    __shared__ double red[3][BLOCK];
    __shared__ uint32_t seg_end[BLOCK];     // stream position one past the last body of list entry k (inclusive prefix sum)
    __shared__ uint32_t seg_first[BLOCK];   // slot of stream position 0 if entry k started there: slot = seg_first[k] + position
    const unsigned tid = threadIdx.x;
    const TargetBlock tb = a.blocks[blockIdx.x];
    // lanes per target: as many whole groups of `count` lanes as the block holds (21 targets in a wave64: 3 lanes each)
    const unsigned W = tb.count ? tb.count : 1u;
    const unsigned fit = (unsigned)BLOCK / W;
    const unsigned P = fit < (unsigned)kMaxLanesPerTarget ? fit : (unsigned)kMaxLanesPerTarget;
    const unsigned t = tid % W, g_raw = tid / W;
    const bool valid = g_raw < P;                              // lanes left over compute along with group 0, unused
    const unsigned g = valid ? g_raw : 0u;
    const uint32_t slot = tb.first + (valid ? t : 0u);
    const float4 me = a.xm[slot];
    const float ix = me.x, iy = me.y, iz = (D == 3) ? me.z : 0.0f;
    const f2 ix2 = {ix, ix}, iy2 = {iy, iy}, iz2 = {iz, iz};
    double ox = 0.0, oy = 0.0, oz = 0.0;
    float* const tf = reinterpret_cast<float*>(tile);
    const unsigned wr = (tid >> 1) * 8u + (tid & 1u);      // source tid = half (tid & 1) of pair tid / 2
    // the lane groups stride through the tile's pairs P at a time: the last trip may reach up to P - 1 pairs past the
    // tile -- massless bodies far away, staged once
    if (tid < 2u * (unsigned)kMaxLanesPerTarget)
        tile[BLOCK + tid] = (tid & 1u) ? make_float4((D == 3) ? 1.0e18f : 0.0f, (D == 3) ? 1.0e18f : 0.0f, 0.f, 0.f) : make_float4(1.0e18f, 1.0e18f, 1.0e18f, 1.0e18f);
    const uint32_t e1 = a.list_offsets[tb.leaf + 1];
    // the list in chunks of BLOCK entries (one chunk for every list the reference's trees produce); all workgroup-uniform
    for (uint32_t e0 = a.list_offsets[tb.leaf]; e0 < e1; e0 += (uint32_t)BLOCK) {
        const unsigned n_ent = (e1 - e0 < (uint32_t)BLOCK) ? (unsigned)(e1 - e0) : (unsigned)BLOCK;
        uint32_t first = 0, len = 0;
        if (tid < n_ent) {
            const uint32_t s = a.list_sources[e0 + tid];
            first = a.leaf_offsets[s];
            len = a.leaf_offsets[s + 1] - first;
        }
        __syncthreads();                                   // the previous chunk's last tile and tables are done with
        seg_end[tid] = len;
        __syncthreads();
        for (unsigned d = 1; d < (unsigned)BLOCK; d <<= 1) {   // inclusive prefix sum (Hillis-Steele)
            const uint32_t add = (tid >= d) ? seg_end[tid - d] : 0u;
            __syncthreads();
            seg_end[tid] += add;
            __syncthreads();
        }
        const uint32_t my_end = seg_end[tid];
        seg_first[tid] = first - (my_end - len);
        __syncthreads();
        const uint32_t total = seg_end[BLOCK - 1];         // bodies in this chunk's stream
        // Tiles of the stream, software-pipelined: the next tile's global loads are issued before the current tile is
        // consumed, so their latency hides behind the pair loop.
        unsigned k = 0;                                    // this lane's list entry; only ever moves forward
        auto load = [&](uint32_t pos) -> float4 {
            // positions past the stream's end stage a massless body far away: it pads the last tile to whole pairs and
            // contributes exactly 0 under every law (r^2 ~ 1e36 is finite in fp32, w = 0 * r^-4)
            float4 v = make_float4(1.0e18f, 1.0e18f, (D == 3) ? 1.0e18f : 0.0f, 0.f);
            if (pos < total) {
                while (pos >= seg_end[k]) ++k;             // empty leaves are stepped over here as well
                const uint32_t j = seg_first[k] + pos;
                v = a.xm[j];
            }
            return v;
        };
        float4 nxt = load(tid);
        // fp32 sums run over up to 256 terms per lane (as in the brute-force kernel's tiles) before they are flushed into
        // the fp64 accumulators: with P lanes per target that is several tiles -- a flush per 64-body tile was 6
        // conversions and 6 fp64 additions against as little as 8 trips of pair arithmetic
        f2 ax = {0.f, 0.f}, ay = {0.f, 0.f}, az = {0.f, 0.f};
        unsigned pending = 0;                                  // terms in the fp32 sums since the last flush
        for (uint32_t pos0 = 0; pos0 < total; pos0 += (uint32_t)BLOCK) {
            const uint32_t cur = (total - pos0 < (uint32_t)BLOCK) ? total - pos0 : (uint32_t)BLOCK;
            __syncthreads();                                   // previous tile fully consumed
            tf[wr] = nxt.x; tf[wr + 2] = nxt.y; tf[wr + 4] = nxt.z; tf[wr + 6] = nxt.w;
            __syncthreads();
            if (pos0 + (uint32_t)BLOCK < total) nxt = load(pos0 + (uint32_t)BLOCK + tid);   // in flight while this tile is consumed
            // A body meets itself in its own leaf: r^2 = 0 falls under every law's skip rule (methods.cpp:113 skips
            // i == j by index, which only differs from the r^2 rule for r^2 >= 1e-10 -- impossible for a body and itself).
            // every lane makes the same number of trips: the lanes past the stream's end staged pad bodies, so every pair of
            // the tile up to a multiple of P past the last real one is real or pad, never stale
            const unsigned trips = (((cur + 1u) >> 1) + P - 1u) / P;
            const float4* src = tile + 2u * g;
            // one source pair {A, B} against this lane's target: d, r^2, then the weights (plain form, or the guarded one
            // when the wave's vote says some lane is below the law's threshold) and the accumulation
            struct Pair { f2 dx, dy, dz, r2, sm; };
            auto geometry = [&](const float4 A, const float4 B) -> Pair {
                Pair q;
                q.sm = f2{B.z, B.w};
                q.dx = f2{A.x, A.y} - ix2;
                q.dy = f2{A.z, A.w} - iy2;
                q.dz = (D == 3) ? f2{B.x, B.y} - iz2 : f2{0.f, 0.f};
                q.r2 = q.dx * q.dx;
                q.r2 = __builtin_elementwise_fma(q.dy, q.dy, q.r2);
                if (D == 3) q.r2 = __builtin_elementwise_fma(q.dz, q.dz, q.r2);
                return q;
            };
            auto special = [&](const Pair& q) -> unsigned long long {
                return __builtin_amdgcn_ballot_w64(q.r2.x < law_special_below<LAW>()) | __builtin_amdgcn_ballot_w64(q.r2.y < law_special_below<LAW>());
            };
            auto guarded = [&](const Pair& q) -> f2 {
                float ra = q.r2.x, rb = q.r2.y;
                asm volatile("" : "+v"(ra), "+v"(rb));   // keeps the guarded form's compares in this (rare) branch: hipcc hoists them otherwise
                return f2{leaf_weight<D, LAW>(ra, q.sm.x, q.dx.x, q.dy.x, q.dz.x), leaf_weight<D, LAW>(rb, q.sm.y, q.dx.y, q.dy.y, q.dz.y)};
            };
            auto plain = [&](const Pair& q) -> f2 {
                f2 w = {__builtin_amdgcn_rcpf(q.r2.x), __builtin_amdgcn_rcpf(q.r2.y)};
                w = w * w;
                return w * q.sm;
            };
            auto add = [&](const Pair& q, const f2 w) {
                ax = __builtin_elementwise_fma(w, q.dx, ax);
                ay = __builtin_elementwise_fma(w, q.dy, ay);
                if (D == 3) az = __builtin_elementwise_fma(w, q.dz, az);
            };
            unsigned it = 0;
            for (; it + 1u < trips; it += 2u, src += 4u * P) {   // two source pairs per trip: one vote, independent chains
                const Pair q0 = geometry(src[0], src[1]), q1 = geometry(src[2u * P], src[2u * P + 1u]);
                f2 w0, w1;
                if (__builtin_expect((special(q0) | special(q1)) != 0ull, 0)) { w0 = guarded(q0); w1 = guarded(q1); }     // wave-uniform, rare
                else { w0 = plain(q0); w1 = plain(q1); }
                add(q0, w0);
                add(q1, w1);
            }
            if (it < trips) {
                const Pair q0 = geometry(src[0], src[1]);
                const f2 w0 = __builtin_expect(special(q0) != 0ull, 0) ? guarded(q0) : plain(q0);
                add(q0, w0);
            }
            pending += 2u * trips;
            if (pending + (unsigned)BLOCK > 256u || pos0 + (uint32_t)BLOCK >= total) {   // workgroup-uniform
                ox += (double)ax.x + (double)ax.y;
                oy += (double)ay.x + (double)ay.y;
                oz += (double)az.x + (double)az.y;
                ax = ay = az = f2{0.f, 0.f};
                pending = 0;
            }
        }
    }
    if (P > 1u) {                                              // block-uniform
        __syncthreads();
        red[0][tid] = ox; red[1][tid] = oy; red[2][tid] = oz;
        __syncthreads();
        if (g == 0u)
            for (unsigned q = 1; q < P; ++q) { ox += red[0][q * W + t]; oy += red[1][q * W + t]; oz += red[2][q * W + t]; }
    }
    if (valid && g == 0u) {
        a.acc[slot] = ox;
        a.acc[(size_t)a.slots + slot] = oy;
        if (D == 3) a.acc[2 * (size_t)a.slots + slot] = oz;
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
template <int BLOCK>
LeafKernel pick(int dim, int law) {
    static const LeafKernel table[2][3] = {
        {leaf_pair_kernel<2, NBX_LAW_BRUTE, BLOCK>, leaf_pair_kernel<2, NBX_LAW_TREE_LEAF, BLOCK>, leaf_pair_kernel<2, NBX_LAW_FMM_P2P, BLOCK>},
        {leaf_pair_kernel<3, NBX_LAW_BRUTE, BLOCK>, leaf_pair_kernel<3, NBX_LAW_TREE_LEAF, BLOCK>, leaf_pair_kernel<3, NBX_LAW_FMM_P2P, BLOCK>}};
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

    size_t nonempty = 0;
    for (size_t l = 0; l < n_leaves; ++l) nonempty += leaf_offsets[l + 1] > leaf_offsets[l];
    // block size by the mean leaf: up to 80 bodies per leaf one wave64 per block wastes fewer lanes than two
    const uint32_t block = (nonempty && slots / nonempty <= 80) ? (uint32_t)kLeafBlockSmall : (uint32_t)kLeafBlock;
    // Target blocks.  A block of c targets runs floor(block / c) lanes per target (kernel), so c just above block / 2 wastes
    // almost half the lanes: such a piece is cut in two when that fills the lanes better by more than the cost of staging
    // the source stream a second time (~15 %): 33..42 targets in a wave64 become two blocks at 3 lanes per target.
    auto lane_use = [&](uint32_t c) -> double {
        uint32_t lanes = block / c;
        if (lanes > (uint32_t)kMaxLanesPerTarget) lanes = (uint32_t)kMaxLanesPerTarget;
        return (double)(c * lanes) / (double)block;
    };
    std::vector<TargetBlock> blocks;
    for (size_t l = 0; l < n_leaves; ++l)
        for (uint32_t f = leaf_offsets[l]; f < leaf_offsets[l + 1]; f += block) {
            const uint32_t c = (leaf_offsets[l + 1] - f < block) ? leaf_offsets[l + 1] - f : block;
            const uint32_t half = (c + 1) / 2;
            if (c >= 2 && lane_use(half) > 1.15 * lane_use(c)) {
                blocks.push_back(TargetBlock{(uint32_t)l, f, half});
                blocks.push_back(TargetBlock{(uint32_t)l, f + half, c - half});
            } else {
                blocks.push_back(TargetBlock{(uint32_t)l, f, c});
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
    hipLaunchKernelGGL(block == (uint32_t)kLeafBlockSmall ? pick<kLeafBlockSmall>(dim, law) : pick<kLeafBlock>(dim, law),
                       dim3((unsigned)blocks.size()), dim3(block), 0, d.stream, a);
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
