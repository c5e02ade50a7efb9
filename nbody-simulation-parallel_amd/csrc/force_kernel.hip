// force_kernel.hip -- K1/K2: tiled all-pairs accumulate kernels for gfx950 (MI355X), fp32, D = 2 or 3.
//
// Computes, for every target i of one shard,
//     a_i = sum_j  m_j * (p_j - p_i) / r_ij^4 ,   pairs with r_ij^2 < 1e-10 contribute exactly 0,
// which is the reference's brute-force law (nbody-sim-new/methods.cpp:21-37 / :110-133) without the
// -(G m_i) factor (applied in fp64 by the consumers in state_kernels.hip).  The decomposition is
// the reference's omp_2 form (methods.cpp:98-136): every target owns its sum, nothing is shared.
//
// Mapping to CDNA4 (see DESIGN.md "Force kernel"):
//   * one lane = TPL targets held in VGPRs; a 256-lane workgroup = 4 wave64 = 256*TPL targets;
//   * sources stream through LDS in tiles of 256 bodies {x,y,z,m} (one ds_write_b128 per lane per
//     tile, double-buffered: one s_barrier per tile); every lane reads the same LDS address, so a
//     single ds_read_b128 broadcast feeds 64*TPL pair interactions;
//   * exact kernels: 14 VALU per pair (v_sub x3, v_mul, v_fma x2, v_cmp+v_cndmask, v_rcp, v_mul x2,
//     v_fma x3).  Fast kernels: targets are held as float2 pairs and the arithmetic is written on
//     2-vectors (v_pk_add/mul/fma_f32: 12-25 % cheaper per element than the scalar forms on gfx950,
//     tools/ubench_valu.hip), and the per-pair guard is replaced by the close-set pipeline described
//     in nbx_internal.h: 11 v_pk + 2 v_rcp per TWO pairs.  MFMA does not apply (no contraction).
//   * two-level summation: fp32 partial sums over one 256-source tile are flushed into second-level
//     accumulators -- fp64 in the exact kernels, fp32 over at most 256 tiles (one source slice) in
//     the fast kernels, whose slices are then summed in fp64 by the consumers.
//   * grid = (target blocks, source slices): slices cut the source-tile list so that small shards
//     still cover all 1024 SIMDs; slice results land in acc[slice] and are summed in a fixed order
//     by the consumer kernels (deterministic, no atomics).
#include "nbx_internal.h"

#include <climits>
#include <cmath>

// Compiled with -fno-slp-vectorize (Makefile): the packed arithmetic is written out on float2 values; letting
// hipcc's SLP vectoriser re-pair the scalar kernels' arithmetic on top of that was measured slower in round 1
// (profiles/r1b/variants_*.txt).
namespace nbx {
namespace {


typedef float f2 __attribute__((ext_vector_type(2)));

// One pair interaction.  s{xyz,m} is wave-uniform (LDS broadcast or SGPR), i{xyz} per lane.
template <int D>
__device__ __forceinline__ void interact(float sx, float sy, float sz, float sm,
                                         float ix, float iy, float iz,
                                         float& ax, float& ay, float& az) {
    const float dx = sx - ix;
    const float dy = sy - iy;
    float r2 = dx * dx;
    r2 = __builtin_fmaf(dy, dy, r2);
    float dz = 0.0f;
    if (D == 3) {
        dz = sz - iz;
        r2 = __builtin_fmaf(dz, dz, r2);
    }
    // methods.cpp:24 -- `if (dist_sq < 1e-10) continue;`  rcp(+inf) = +0 makes the pair's weight 0.
    const float r2g = (r2 < kR2SkipF) ? __builtin_inff() : r2;
    const float ri2 = __builtin_amdgcn_rcpf(r2g);  // v_rcp_f32, 1 ulp
    const float t = sm * ri2;
    const float s = t * ri2;  // m_j / r^4
    ax = __builtin_fmaf(s, dx, ax);
    ay = __builtin_fmaf(s, dy, ay);
    if (D == 3) az = __builtin_fmaf(s, dz, az);
}

// Two-vector arithmetic for PAIRS target pairs at once (the two halves of a 64-bit register = two targets),
// written stage by stage so that the PAIRS dependency chains are interleaved in program order (each v_pk
// result is consumed PAIRS instructions later).  Per source and target PAIR, D = 3:
//   ONE_RCP = 0:  v_pk_add x3, v_pk_fma x3 (r^2 + bias), v_rcp x2, v_pk_mul x2, v_pk_fma x3       = 13 VALU
//                 bias = kTiny for the reference law (see nbx_internal.h), = epsilon^2 for the softened law
//   ONE_RCP = 1:  the two reciprocals come from ONE v_rcp_f32 of the product: W = 1/(r2a*r2b),
//                 (1/r2a, 1/r2b) = W * (r2b, r2a)  -- v_mul, v_rcp, v_pk_mul(op_sel swap) instead of v_rcp x2
//                 (3.45 ns -> 3.0 ns of the 27.7 ns body, tools/ubench_valu.hip).  r2a*r2b must stay finite:
//                 launched only when every |coordinate| <= kOneRcpMaxCoord (nbx_internal.h).
// NEWTON = 1 (softened Newtonian law, an extension): the weight is m (r^2+eps^2)^-3/2 -- v_rsq_f32 in place of v_rcp_f32
// and one more v_pk_mul: 14 VALU per two pairs.
// FIRST = 1: the sums START with this source (ax = w d instead of ax += w d): a block's first source, which saves zeroing
// the 3 PAIRS accumulators per block (the three-level kernel starts a block every 64 sources).
template <int D, int PAIRS, int ONE_RCP, int NEWTON = 0, int HI_SEL = 1, int FIRST = 0>
__device__ __forceinline__ void interact2_staged(float sx, float sy, float sz, f2 szm, const f2 (&ix)[PAIRS],
                                                 const f2 (&iy)[PAIRS], const f2 (&iz)[PAIRS], f2 (&ax)[PAIRS],
                                                 f2 (&ay)[PAIRS], f2 (&az)[PAIRS], const f2 bias) {
    f2 dx[PAIRS], dy[PAIRS], dz[PAIRS], r2[PAIRS], w[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) dx[q] = f2{sx, sx} - ix[q];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) dy[q] = f2{sy, sy} - iy[q];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) dz[q] = (D == 3) ? f2{sz, sz} - iz[q] : f2{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) r2[q] = __builtin_elementwise_fma(dx[q], dx[q], bias);
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) r2[q] = __builtin_elementwise_fma(dy[q], dy[q], r2[q]);
    if (D == 3) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) r2[q] = __builtin_elementwise_fma(dz[q], dz[q], r2[q]);
    }
    if (ONE_RCP) {
        float W[PAIRS];
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) W[q] = r2[q].x * r2[q].y;
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) W[q] = __builtin_amdgcn_rcpf(W[q]);
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) w[q] = f2{W[q], W[q]} * __builtin_shufflevector(r2[q], r2[q], 1, 0);
    } else if (NEWTON) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) { w[q].x = __builtin_amdgcn_rsqf(r2[q].x); w[q].y = __builtin_amdgcn_rsqf(r2[q].y); }
    } else {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) { w[q].x = __builtin_amdgcn_rcpf(r2[q].x); w[q].y = __builtin_amdgcn_rcpf(r2[q].y); }
    }
    // weight = m * w^2 (NEWTON: m * w^3), the mass applied LAST.  The mass is the HIGH half of the source's {z, m}
    // register pair: hipcc folds a low-half or a first-pair high-half broadcast into op_sel, but copies the 4th component
    // of a ds_read_b128 with a v_mov_b32 per source (1 of 53 VALU per source), so the modifier is spelled out.  The
    // inline instruction must not read a v_rcp/v_rsq result directly: the compiler's trans->VALU hazard handling (s_nop)
    // does not see inside inline asm -- hence w^2 first (plain code, hazards handled), then the multiply by m.
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) r2[q] = w[q] * w[q];
    if (NEWTON) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) r2[q] = r2[q] * w[q];
    }
    if (HI_SEL) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(w[q]) : "v"(szm), "v"(r2[q]));
    } else {   // what the compiler makes of it: v_mov_b32 + low-half broadcast (kept for the A/B in profiles/r2)
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) w[q] = f2{szm.y, szm.y} * r2[q];
    }
    if (FIRST) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) ax[q] = w[q] * dx[q];
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) ay[q] = w[q] * dy[q];
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) az[q] = (D == 3) ? w[q] * dz[q] : f2{0.f, 0.f};
        return;
    }
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) ax[q] = __builtin_elementwise_fma(w[q], dx[q], ax[q]);
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) ay[q] = __builtin_elementwise_fma(w[q], dy[q], ay[q]);
    if (D == 3) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) az[q] = __builtin_elementwise_fma(w[q], dz[q], az[q]);
    }
}

// A target belongs to the close set if any of its coordinates is below kCloseCoord in magnitude
// (see nbx_internal.h).  Evaluated on the same fp32 values by every kernel that needs it.
template <int D>
__device__ __forceinline__ bool in_close_set(float x, float y, float z) {
    float m = __builtin_fminf(__builtin_fabsf(x), __builtin_fabsf(y));
    if (D == 3) m = __builtin_fminf(m, __builtin_fabsf(z));
    return m < kCloseCoord;
}

// Walks the virtual source-tile list of one launch: tile t of the list lives in real chunk
// c(t) at body offset k(t)*kTile.  Incremental, wave-uniform (SALU only).
struct TileWalk {
    int vc;         // virtual chunk
    unsigned k;     // tile inside the chunk
    __device__ __forceinline__ void seek(unsigned t, unsigned tiles_per_chunk) {
        vc = (int)(t / tiles_per_chunk);
        k = t - (unsigned)vc * tiles_per_chunk;
    }
    __device__ __forceinline__ void next(unsigned tiles_per_chunk) {
        if (++k == tiles_per_chunk) { k = 0; ++vc; }
    }
    __device__ __forceinline__ int chunk(int chunk_first, int chunk_skip) const {
        int c = chunk_first + vc;
        return c + (c >= chunk_skip ? 1 : 0);
    }
};

template <int D>
__device__ __forceinline__ float4 load_source(const KArgs& a, const TileWalk& tw, unsigned tid) {
    const int c = tw.chunk(a.chunk_first, a.chunk_skip);
    const float* __restrict__ sp = a.pos_all + (size_t)c * D * a.pad + tw.k * kTile + tid;
    float4 v;
    v.x = sp[0];
    v.y = sp[a.pad];
    v.z = (D == 3) ? sp[2 * (size_t)a.pad] : 0.0f;
    v.w = a.mass_all[(size_t)c * a.pad + tw.k * kTile + tid];
    return v;
}

template <int D, typename T>
__device__ __forceinline__ void store_result(const KArgs& a, float* __restrict__ out, unsigned i, T vx, T vy, T vz) {
    if (a.accumulate) {
        out[i] = (float)((double)out[i] + (double)vx);
        out[(size_t)a.pad + i] = (float)((double)out[(size_t)a.pad + i] + (double)vy);
        if (D == 3) out[2 * (size_t)a.pad + i] = (float)((double)out[2 * (size_t)a.pad + i] + (double)vz);
    } else {
        out[i] = (float)vx;
        out[(size_t)a.pad + i] = (float)vy;
        if (D == 3) out[2 * (size_t)a.pad + i] = (float)vz;
    }
}

// XCD-aware workgroup -> (target block, source slice) mapping.  The 256 CUs are 8 XCDs with a private 4 MiB L2 each, and
// the hardware deals consecutive workgroups (x fastest, then y) round-robin over the XCDs.  With the natural mapping and a
// grid width that is a multiple of 8, XCD k gets the target blocks x = k mod 8 and ALL source slices: every L2 streams the
// whole source set, which does not fit it (0.4-0.5 GB of fabric reads per launch at N = 2^20 against 29 MB algorithmic).
// Here XCD k owns source slices [k S/8, (k+1) S/8) for ALL target blocks: its share of the sources (2 MiB at N = 2^20) stays
// resident in its L2 and is fetched once.  Placement is a speed matter only -- any mapping is correct.
constexpr unsigned kXcds = 8;
__device__ __forceinline__ void xcd_tile(unsigned& bx, unsigned& by) {
    const unsigned S = gridDim.y;
    if (S % kXcds != 0) { bx = blockIdx.x; by = blockIdx.y; return; }
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;   // dispatch order; workgroup L runs on XCD L % 8
    const unsigned xcd = L % kXcds, idx = L / kXcds, per = S / kXcds;
    by = xcd * per + idx % per;
    bx = idx / per;
}

// -------------------------------------------------------------------------------------------------
// Exact LDS kernel: 256 lanes, TPL targets per lane, compare-and-select guard on every pair,
// fp64 second-level accumulators.  Self-contained (no close-set pipeline).
// -------------------------------------------------------------------------------------------------
template <int D, int TPL, int WAVES, int UNROLL>
__global__ __launch_bounds__(256, WAVES) void accel_lds_kernel(KArgs a) {
    __shared__ float4 tile[2][kTile];
    const unsigned tid = threadIdx.x;
    unsigned bx, by;
    xcd_tile(bx, by);
    const unsigned tgt0 = bx * (256u * TPL) + tid;
    const float* __restrict__ tp = a.pos_all + (size_t)a.tgt_chunk * D * a.pad;

    float ix[TPL], iy[TPL], iz[TPL];
    double ox[TPL], oy[TPL], oz[TPL];  // second-level accumulators: one fp64 add per 256 pairs
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
        const unsigned i = tgt0 + q * 256u;
        ix[q] = tp[i];
        iy[q] = tp[(size_t)a.pad + i];
        iz[q] = (D == 3) ? tp[2 * (size_t)a.pad + i] : 0.0f;
        ox[q] = oy[q] = oz[q] = 0.0;
    }

    unsigned t = by * a.tiles_per_split;
    unsigned t_end = t + a.tiles_per_split;
    if (t_end > a.total_tiles) t_end = a.total_tiles;
    TileWalk w;
    w.seek(t, a.tiles_per_chunk);

    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < t_end) nxt = load_source<D>(a, w, tid);
    int buf = 0;
    for (; t < t_end; ++t) {
        tile[buf][tid] = nxt;
        __syncthreads();
        if (t + 1 < t_end) {  // next tile's global loads fly while this tile is consumed
            w.next(a.tiles_per_chunk);
            nxt = load_source<D>(a, w, tid);
        }
        float ax[TPL], ay[TPL], az[TPL];
#pragma unroll
        for (int q = 0; q < TPL; ++q) ax[q] = ay[q] = az[q] = 0.0f;
        const float4* __restrict__ cur = tile[buf];
#pragma unroll UNROLL
        for (int j = 0; j < kTile; ++j) {
            const float4 s = cur[j];  // same address in every lane: ds_read_b128 broadcast
#pragma unroll
            for (int q = 0; q < TPL; ++q)
                interact<D>(s.x, s.y, s.z, s.w, ix[q], iy[q], iz[q], ax[q], ay[q], az[q]);
        }
#pragma unroll
        for (int q = 0; q < TPL; ++q) { ox[q] += (double)ax[q]; oy[q] += (double)ay[q]; oz[q] += (double)az[q]; }
        buf ^= 1;
    }

    float* __restrict__ out = a.acc + (size_t)by * D * a.pad;
#pragma unroll
    for (int q = 0; q < TPL; ++q) store_result<D>(a, out, tgt0 + q * 256u, ox[q], oy[q], oz[q]);
}

// -------------------------------------------------------------------------------------------------
// Strict fp64 kernel: the reference's own arithmetic type (vector.h:9-12, methods.cpp:21-37 are fp64 throughout) on the
// device's fp32-representable inputs.  Same tiling as the exact fp32 kernel -- 256 lanes, TPL targets per lane, source
// tiles of 256 bodies staged in LDS as doubles {x,y},{z,m} (two ds_read_b128 broadcasts per source) -- with
//   d = p_j - p_i (exact: both operands are fp32 values), r^2 by fma, the reference's skip rule compared in fp64
//   (r^2 < 1e-10, methods.cpp:24), 1/r^2 = v_rcp_f64 + NR Newton steps (2: full double precision), w = m/r^4, 3 fma,
// and ONE fp64 accumulator per target and component over the whole source slice (no fp32 anywhere after the loads).
// LIST = 0: every target of the chunk; the slice's fp64 sum v leaves as TWO fp32 planes of acc, hi = (float)v and
//           lo = (float)(v - hi) (planes 2*slice and 2*slice+1): the consumers already add the planes in fp64 in plane
//           order (state_kernels.hip sum_partials), which puts v back together to 2^-48 -- no consumer knows about this kernel.
// LIST = 1: the targets listed in strict_list[0 .. counters[3]) (the mixed mode's suspects, refine_select_kernel); TPL
//           targets per lane, workgroup (x, y) takes list blocks x, x + gridDim.x, ... against source slice y and writes
//           strict_acc[y][k][slot] in fp64; refine_fold_kernel adds the slices and rewrites the targets' planes.
// -------------------------------------------------------------------------------------------------
template <int D, int NR, int MAG>
__device__ __forceinline__ void interact64(double sx, double sy, double sz, double sm, double ix, double iy, double iz,
                                           double& ax, double& ay, double& az, double& mag) {
    const double dx = sx - ix, dy = sy - iy;
    double r2 = dx * dx;
    r2 = __builtin_fma(dy, dy, r2);
    double dz = 0.0;
    if (D == 3) {
        dz = sz - iz;
        r2 = __builtin_fma(dz, dz, r2);
    }
    double w = __builtin_amdgcn_rcp(r2);   // v_rcp_f64: a seed, refined below
#pragma unroll
    for (int it = 0; it < NR; ++it) {
        const double e = __builtin_fma(-r2, w, 1.0);
        w = __builtin_fma(w, e, w);
    }
    double s = (w * w) * sm;               // m_j / r^4
    s = (r2 < kR2Skip64) ? 0.0 : s;        // methods.cpp:24; also discards the NaN of r2 = 0 (rcp = inf, 0 * inf)
    ax = __builtin_fma(s, dx, ax);
    ay = __builtin_fma(s, dy, ay);
    if (D == 3) az = __builtin_fma(s, dz, az);
    // |a_ij| = (m/r^4) r: the magnitude sums that backward errors are measured against need no more than fp32's sqrt
    if (MAG) mag = __builtin_fma(s, (double)__builtin_sqrtf((float)r2), mag);
}

struct Tile64 { double2 xy, zm; };

template <int D>
__device__ __forceinline__ void stage64(Tile64& dst, const float4 v) {
    dst.xy = double2{(double)v.x, (double)v.y};
    dst.zm = double2{(double)v.z, (double)v.w};
}

__device__ __forceinline__ void store_hi_lo(const KArgs& a, float* __restrict__ hi_plane, float* __restrict__ lo_plane,
                                            size_t idx, double v) {
    if (a.accumulate) v += (double)hi_plane[idx] + (double)lo_plane[idx];
    const float hi = (float)v;
    hi_plane[idx] = hi;
    lo_plane[idx] = (float)(v - (double)hi);
}

// Layout of the mixed mode's fp64 pass, decided on the device from the length of the list the selection just wrote (the
// host never learns it: the step stays one graph).  strict_acc holds [slices][D][stride] doubles, stride = the list's length
// rounded up to whole blocks of 256; as many source slices (<= the launch's gridDim.y) as fit the buffer.  A short list --
// the usual case, a fraction of a percent of the shard -- spreads one or two list blocks over up to 256 slices; a long
// one has the list blocks to fill the chip and gets fewer slices; the whole shard still fits one slice.  Nothing overflows.
struct StrictLayout { unsigned n, stride, slices, tiles_per_split; };
template <int D>
__device__ __forceinline__ StrictLayout strict_layout(const KArgs& a) {
    StrictLayout s;
    const unsigned listed = a.counters[3];
    s.n = listed < a.strict_cap ? listed : a.strict_cap;   // the list cannot be longer than the shard; a guard, not a limit
    s.stride = s.n ? (s.n + 255u) / 256u * 256u : 256u;
    unsigned long long fit = a.strict_budget / ((unsigned long long)D * s.stride);
    if (fit > (unsigned long long)a.strict_slices) fit = (unsigned long long)a.strict_slices;
    s.slices = fit ? (unsigned)fit : 1u;
    s.tiles_per_split = (a.total_tiles + s.slices - 1u) / s.slices;
    return s;
}

// MAG = 1 (diagnostic build): also S_i = sum_j |a_ij|, the sum of the pair terms' magnitudes, into aux[slice][i] -- what the
// relative error of a cancelling sum is measured against (tests: backward error of the fp32 path for EVERY body).
template <int D, int TPL, int WAVES, int UNROLL, int NR, int LIST, int MAG = 0>
__global__ __launch_bounds__(256, WAVES) void accel_f64_kernel(KArgs a) {
    static_assert(!(LIST && MAG), "magnitude sums: whole-chunk launches only");
    __shared__ Tile64 tile[2][kTile];
    const unsigned tid = threadIdx.x;
    unsigned bx, by;
    // LIST: x = source slice, y = list-block row.  Consecutive workgroups (dealt round-robin to the 8 XCDs) are then the slices of ONE
    // list block: a short list -- one to four blocks, the usual case -- still lands on every XCD and CU.  (With x = list-block row
    // only rows 0..3 of 32 had work, i.e. XCDs 0..3: 970 targets took as long as 7,700, profiles/r4.)
    if (LIST) { by = blockIdx.x; bx = blockIdx.y; } else xcd_tile(bx, by);
    const float* __restrict__ tp = a.pos_all + (size_t)a.tgt_chunk * D * a.pad;
    unsigned n_listed = 0u, out_stride = 0u, tiles_per_split = a.tiles_per_split;
    if (LIST) {
        const StrictLayout sl = strict_layout<D>(a);
        if (by >= sl.slices) return;   // workgroup-uniform, before any barrier
        n_listed = sl.n; out_stride = sl.stride; tiles_per_split = sl.tiles_per_split;
    }
    const unsigned nblk = LIST ? (n_listed + 256u * TPL - 1u) / (256u * TPL) : bx + 1u;

    for (unsigned tb = bx; tb < nblk; tb += LIST ? gridDim.y : nblk) {
        unsigned idx[TPL];
        double ix[TPL], iy[TPL], iz[TPL], ox[TPL], oy[TPL], oz[TPL], om[MAG ? TPL : 1];
        const unsigned slot0 = tb * (256u * TPL) + tid;   // list slot (LIST) or target index of this lane's q-th target: slot0 + 256 q
#pragma unroll
        for (int q = 0; q < TPL; ++q) {
            const unsigned slot = slot0 + q * 256u;
            idx[q] = LIST ? a.strict_list[slot < n_listed ? slot : 0] : slot;
            ix[q] = (double)tp[idx[q]];
            iy[q] = (double)tp[(size_t)a.pad + idx[q]];
            iz[q] = (D == 3) ? (double)tp[2 * (size_t)a.pad + idx[q]] : 0.0;
            ox[q] = oy[q] = oz[q] = 0.0;
            if (MAG) om[q] = 0.0;
        }
        unsigned t = by * tiles_per_split;
        unsigned t_end = t + tiles_per_split;
        if (t_end > a.total_tiles) t_end = a.total_tiles;
        if (t > t_end) t = t_end;   // a slice beyond the last tile (the rounding of tiles_per_split) adds nothing
        TileWalk w;
        w.seek(t, a.tiles_per_chunk);
        float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < t_end) nxt = load_source<D>(a, w, tid);
        int buf = 0;
        for (; t < t_end; ++t) {
            stage64<D>(tile[buf][tid], nxt);
            __syncthreads();
            if (t + 1 < t_end) {
                w.next(a.tiles_per_chunk);
                nxt = load_source<D>(a, w, tid);
            }
            const Tile64* __restrict__ cur = tile[buf];
#pragma unroll UNROLL
            for (int j = 0; j < kTile; ++j) {
                const double2 sxy = cur[j].xy, szm = cur[j].zm;   // same address in every lane: two ds_read_b128 broadcasts
#pragma unroll
                for (int q = 0; q < TPL; ++q)
                    interact64<D, NR, MAG>(sxy.x, sxy.y, szm.x, szm.y, ix[q], iy[q], iz[q], ox[q], oy[q], oz[q], om[MAG ? q : 0]);
            }
            buf ^= 1;
        }
        if (LIST) {
            double* __restrict__ o = a.strict_acc + (size_t)by * D * out_stride;
#pragma unroll
            for (int q = 0; q < TPL; ++q) {
                const unsigned slot = slot0 + q * 256u;
                if (slot < n_listed) {
                    o[slot] = ox[q];
                    o[(size_t)out_stride + slot] = oy[q];
                    if (D == 3) o[2 * (size_t)out_stride + slot] = oz[q];
                }
            }
            __syncthreads();   // the next list block reuses the tile buffers
        } else {
            float* __restrict__ hi = a.acc + (size_t)(2u * by) * D * a.pad;
            float* __restrict__ lo = hi + (size_t)D * a.pad;
#pragma unroll
            for (int q = 0; q < TPL; ++q) {
                store_hi_lo(a, hi, lo, idx[q], ox[q]);
                store_hi_lo(a, hi, lo, (size_t)a.pad + idx[q], oy[q]);
                if (D == 3) store_hi_lo(a, hi, lo, 2 * (size_t)a.pad + idx[q], oz[q]);
                if (MAG) { float* __restrict__ mo = a.qsum + (size_t)by * a.pad + idx[q]; *mo = a.accumulate ? *mo + (float)om[q] : (float)om[q]; }
            }
        }
    }
}

template <int D>
__device__ __forceinline__ void close_set_path(const KArgs& a, float4 (&tile)[2][kTile], unsigned bx, unsigned by);

// -------------------------------------------------------------------------------------------------
// Fast packed LDS kernel: PAIRS float2 target pairs per lane (TPL = 2*PAIRS), no per-pair guard
// (kTiny bias), fp32 second-level accumulators (the launcher keeps a slice at <= 256 tiles).
// Flagged (bad) targets are neither stored nor trusted; the launch's extra workgroups
// (blockIdx.x < close_blocks) evaluate them with the guard (close_set_path below).
// -------------------------------------------------------------------------------------------------
// SOFT = 1: Plummer-softened law  a_i = sum_j m_j d / (r^2 + eps^2)^2  (nbx_ctx_set_softening; an extension -- the
// reference's brute force has no softening).  The bias of r^2 is then eps^2 itself, every pair is counted,
// and there is no close set: close_blocks = 0, bad_flag is not read.
// SOFT = 2: softened NEWTONIAN law  a_i = sum_j m_j d / (r^2 + eps^2)^(3/2)  (nbx_ctx_set_law; the `--law newton` of
// SURVEY 5/7; attractive -- the sign is applied with G by the consumers).  Never combined with ONE_RCP.
// QS = 1 (mixed mode, nbx_ctx_set_refine): the kernel also adds up, per target, |tile partial sum|^2 over the slice's tiles
// (3 v_pk_fma_f32 per target pair and TILE, 0.1 % of the pair loop) and writes it to qsum[slice][i]: the spread that the
// rounding error of this target's fp32 sum scales with (refine_select_kernel).
template <int D, int PAIRS, int WAVES, int UNROLL, int ONE_RCP, int SOFT = 0, int HI_SEL = 1, int XCD_MAP = 1, int QS = 0>
__global__ __launch_bounds__(256, WAVES) void accel_fast_pk_kernel(KArgs a) {
    constexpr int TPL = 2 * PAIRS;
    __shared__ float4 tile[2][kTile];
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx, by);
    if (!SOFT && bx < a.close_blocks) {  // the launch's extra workgroups: guarded evaluation of the close set
        close_set_path<D>(a, tile, bx, by);
        return;
    }
    const f2 bias = SOFT ? f2{a.eps2, a.eps2} : f2{kTiny, kTiny};
    const unsigned tid = threadIdx.x;
    const unsigned tgt0 = (bx - a.close_blocks) * (256u * TPL) + tid;
    const float* __restrict__ tp = a.pos_all + (size_t)a.tgt_chunk * D * a.pad;

    f2 ix[PAIRS], iy[PAIRS], iz[PAIRS], ox[PAIRS], oy[PAIRS], oz[PAIRS], qq[QS ? PAIRS : 1];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const unsigned i0 = tgt0 + (2 * q) * 256u, i1 = i0 + 256u;
        ix[q] = f2{tp[i0], tp[i1]};
        iy[q] = f2{tp[(size_t)a.pad + i0], tp[(size_t)a.pad + i1]};
        iz[q] = (D == 3) ? f2{tp[2 * (size_t)a.pad + i0], tp[2 * (size_t)a.pad + i1]} : f2{0.f, 0.f};
        ox[q] = oy[q] = oz[q] = f2{0.f, 0.f};
        if (QS) qq[q] = f2{0.f, 0.f};
    }

    unsigned t = by * a.tiles_per_split;
    unsigned t_end = t + a.tiles_per_split;
    if (t_end > a.total_tiles) t_end = a.total_tiles;
    TileWalk w;
    w.seek(t, a.tiles_per_chunk);

    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < t_end) nxt = load_source<D>(a, w, tid);
    int buf = 0;
    for (; t < t_end; ++t) {
        tile[buf][tid] = nxt;
        __syncthreads();
        if (t + 1 < t_end) {
            w.next(a.tiles_per_chunk);
            nxt = load_source<D>(a, w, tid);
        }
        f2 ax[PAIRS], ay[PAIRS], az[PAIRS];
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) ax[q] = ay[q] = az[q] = f2{0.f, 0.f};
        const float4* __restrict__ cur = tile[buf];
#pragma unroll UNROLL
        for (int j = 0; j < kTile; ++j) {
            const float4 s = cur[j];
            interact2_staged<D, PAIRS, (SOFT == 2) ? 0 : ONE_RCP, SOFT == 2, HI_SEL>(s.x, s.y, s.z, f2{s.z, s.w}, ix, iy, iz, ax, ay, az, bias);
        }
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) { ox[q] += ax[q]; oy[q] += ay[q]; oz[q] += az[q]; }
        if (QS) {
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
                qq[q] = __builtin_elementwise_fma(ax[q], ax[q], qq[q]);
                qq[q] = __builtin_elementwise_fma(ay[q], ay[q], qq[q]);
                if (D == 3) qq[q] = __builtin_elementwise_fma(az[q], az[q], qq[q]);
            }
        }
        buf ^= 1;
    }

    float* __restrict__ out = a.acc + (size_t)by * D * a.pad;
    float* __restrict__ qout = QS ? a.qsum + (size_t)by * a.pad : nullptr;
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned i = tgt0 + (2 * q + h) * 256u;
            if (SOFT || !a.bad_flag[i]) {  // flagged targets belong to close_set_path + scatter_close_kernel
                store_result<D>(a, out, i, h ? ox[q].y : ox[q].x, h ? oy[q].y : oy[q].x, h ? oz[q].y : oz[q].x);
                if (QS) { const float v = h ? qq[q].y : qq[q].x; qout[i] = a.accumulate ? qout[i] + v : v; }
            } else if (QS) {
                qout[i] = __builtin_inff();   // a close-set target of this pass keeps no spread sum: it is always a suspect
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Three-level fast kernel (round 4): the same packed pair arithmetic, a summation that keeps fp32 errors 2-3x smaller.
// Where the two-level kernel's error comes from: a term rides through up to 255 fp32 additions of its
// 256-source tile and then up to 256 flushes into an fp32 slice sum that has grown to the whole slice's partial sum -- the two
// levels contribute about equally, and a body whose pulls cancel to a hundredth of their size ends up 1e-5 to 3e-5 off.  Here
//   level 1: fp32 sums over LB = 64 (or 32) sources in registers (ax), flushed into
//   level 2: the fp32 sum of ONE 256-source tile (ox; 256/LB flushes), flushed once per tile into
//   level 3: fp64 slice sums that live in LDS (48 KB per workgroup: 8 targets x 3 components x 256 lanes; each lane touches only
//            its own 24 slots, 12 ds_read_b128 + 24 v_cvt_f64_f32 + 24 v_add_f64 + 12 ds_write_b128 per 13,568 VALU of pair work),
// and a slice's fp64 sum leaves as TWO fp32 planes (hi, lo), like the strict kernel's: no consumer knows.  Rounding now happens
// relative to partial sums of at most 64 terms (level 1) and 256 terms (level 2): variance ~ (32 + 2.5) a_p^2 per source against
// (128 + 128) -- errors 2.7x smaller at +12 v_pk_add per 64 sources (+0.35 %).  Register budget unchanged (the slice sums left
// the registers), 3 waves per SIMD; LDS 52 KB per workgroup x 3 workgroups per CU = 156 of 160 KB, which is why the source tile
// is SINGLE-buffered here (two barriers per tile instead of one; the three workgroups of a CU cover each other's waits).
// QS = 1: Q_i = sum over the LEVEL-1 blocks of |block sum|^2 (the running sums whose roundings dominate), one plane per slice.
// -------------------------------------------------------------------------------------------------
// REGSUMS = 1: the fp64 slice sums stay in registers instead (48 more VGPRs: two waves per SIMD -- the two-level kernel loses
// 0.4 % at that occupancy, profiles/r2/variants_shape.txt); with WAVES = 2 the source tile is double-buffered again (one barrier
// per tile), with WAVES = 3 the 52 KB of LDS per workgroup leave room for a single tile buffer only (two barriers per tile).
// Measured and lost in round 5 (profiles/r5/variants_half_tile.txt: 231.7 against 230.6 ms, bitwise-equal forces): filling and
// consuming the single tile buffer in two halves, so that none of the two barriers per tile follows another back to back.
template <int D, int PAIRS, int WAVES, int UNROLL, int LB, int QS, int REGSUMS, int PEEL = 1>
__global__ __launch_bounds__(256, WAVES) void accel_fast3l_kernel(KArgs a) {
    constexpr int TPL = 2 * PAIRS;
    constexpr int NBUF = WAVES <= 2 ? 2 : 1;
    static_assert(kTile % LB == 0 && LB % UNROLL == 0, "level-1 blocks tile the source tile");
    static_assert(!(REGSUMS && WAVES > 2), "register-resident slice sums need the register budget of two waves per SIMD");
    constexpr unsigned kTileBytes = 2u * kTile * sizeof(float4);   // room for the close-set path's two buffers in every build
    constexpr unsigned kSumBytes = REGSUMS ? 0u : (unsigned)PAIRS * D * 256u * sizeof(double2);   // [PAIRS*D][256] double2 = the two targets of a pair
    constexpr unsigned kOwnTileBytes = (unsigned)NBUF * kTile * sizeof(float4);
    constexpr unsigned kSmemBytes = (kOwnTileBytes + kSumBytes) > kTileBytes ? (kOwnTileBytes + kSumBytes) : kTileBytes;
    __shared__ __attribute__((aligned(16))) char smem[kSmemBytes];
    unsigned bx = blockIdx.x, by = blockIdx.y;
    xcd_tile(bx, by);
    if (bx < a.close_blocks) {   // the launch's extra workgroups: guarded evaluation of the close set (they keep no slice sums)
        close_set_path<D>(a, *reinterpret_cast<float4 (*)[2][kTile]>(smem), bx, by);
        return;
    }
    // measurement only (a.clk is null in every other launch; a uniform branch on a kernel argument, four SGPRs): the shader
    // clock this workgroup held = d s_memtime / d s_memrealtime x 100 MHz (nbx_ctx_shader_clock)
    unsigned long long clk_r0 = 0, clk_t0 = 0;
    if (a.clk) {
        clk_r0 = __builtin_amdgcn_s_memrealtime();
        clk_t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) here, so that the loop's LDS waits stay counted
    }
    float4* __restrict__ tile = reinterpret_cast<float4*>(smem);
    double2* __restrict__ sums = reinterpret_cast<double2*>(smem + kOwnTileBytes);
    const f2 bias = f2{kTiny, kTiny};
    const unsigned tid = threadIdx.x;
    const unsigned tgt0 = (bx - a.close_blocks) * (256u * TPL) + tid;
    const float* __restrict__ tp = a.pos_all + (size_t)a.tgt_chunk * D * a.pad;

    f2 ix[PAIRS], iy[PAIRS], iz[PAIRS], qq[QS ? PAIRS : 1];
    double2 rx[REGSUMS ? PAIRS : 1], ry[REGSUMS ? PAIRS : 1], rz[REGSUMS ? PAIRS : 1];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const unsigned i0 = tgt0 + (2 * q) * 256u, i1 = i0 + 256u;
        ix[q] = f2{tp[i0], tp[i1]};
        iy[q] = f2{tp[(size_t)a.pad + i0], tp[(size_t)a.pad + i1]};
        iz[q] = (D == 3) ? f2{tp[2 * (size_t)a.pad + i0], tp[2 * (size_t)a.pad + i1]} : f2{0.f, 0.f};
        if (QS) qq[q] = f2{0.f, 0.f};
        if (REGSUMS) rx[q] = ry[q] = rz[q] = double2{0.0, 0.0};
    }
    if (!REGSUMS) {
#pragma unroll
        for (int c = 0; c < PAIRS * D; ++c) sums[c * 256 + tid] = double2{0.0, 0.0};   // own slots only: no barrier needed
    }

    unsigned t = by * a.tiles_per_split;
    unsigned t_end = t + a.tiles_per_split;
    if (t_end > a.total_tiles) t_end = a.total_tiles;
    TileWalk w;
    w.seek(t, a.tiles_per_chunk);
    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < t_end) nxt = load_source<D>(a, w, tid);
    int buf = 0;
    for (; t < t_end; ++t) {
        tile[buf * kTile + tid] = nxt;
        __syncthreads();
        if (t + 1 < t_end) {
            w.next(a.tiles_per_chunk);
            nxt = load_source<D>(a, w, tid);
        }
        f2 ox[PAIRS], oy[PAIRS], oz[PAIRS];
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) ox[q] = oy[q] = oz[q] = f2{0.f, 0.f};
#pragma unroll 1
        for (int blk = 0; blk < kTile / LB; ++blk) {
            f2 ax[PAIRS], ay[PAIRS], az[PAIRS];
            const float4* __restrict__ cur = tile + buf * kTile + blk * LB;
            if (PEEL) {   // the block's first source starts the sums (no zeroing)
                const float4 s = cur[0];
                interact2_staged<D, PAIRS, 0, 0, 1, 1>(s.x, s.y, s.z, f2{s.z, s.w}, ix, iy, iz, ax, ay, az, bias);
            } else {
#pragma unroll
                for (int q = 0; q < PAIRS; ++q) ax[q] = ay[q] = az[q] = f2{0.f, 0.f};
            }
#pragma unroll UNROLL
            for (int j = PEEL ? 1 : 0; j < LB; ++j) {
                const float4 s = cur[j];
                interact2_staged<D, PAIRS, 0, 0, 1>(s.x, s.y, s.z, f2{s.z, s.w}, ix, iy, iz, ax, ay, az, bias);
            }
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) { ox[q] += ax[q]; oy[q] += ay[q]; if (D == 3) oz[q] += az[q]; }
            if (QS == 1) {
#pragma unroll
                for (int q = 0; q < PAIRS; ++q) {
                    qq[q] = __builtin_elementwise_fma(ax[q], ax[q], qq[q]);
                    qq[q] = __builtin_elementwise_fma(ay[q], ay[q], qq[q]);
                    if (D == 3) qq[q] = __builtin_elementwise_fma(az[q], az[q], qq[q]);
                }
            }
        }
        if (QS == 2) {   // A/B: the spread from the TILE sums, like the two-level kernel's (a quarter of the instructions)
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
                qq[q] = __builtin_elementwise_fma(ox[q], ox[q], qq[q]);
                qq[q] = __builtin_elementwise_fma(oy[q], oy[q], qq[q]);
                if (D == 3) qq[q] = __builtin_elementwise_fma(oz[q], oz[q], qq[q]);
            }
        }
        // level 3: the tile's sums into the lane's own fp64 slots
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            if (REGSUMS) {
                rx[q].x += (double)ox[q].x; rx[q].y += (double)ox[q].y;
                ry[q].x += (double)oy[q].x; ry[q].y += (double)oy[q].y;
                if (D == 3) { rz[q].x += (double)oz[q].x; rz[q].y += (double)oz[q].y; }
            } else {
                double2 v = sums[(q * D + 0) * 256 + tid];
                v.x += (double)ox[q].x; v.y += (double)ox[q].y;
                sums[(q * D + 0) * 256 + tid] = v;
                v = sums[(q * D + 1) * 256 + tid];
                v.x += (double)oy[q].x; v.y += (double)oy[q].y;
                sums[(q * D + 1) * 256 + tid] = v;
                if (D == 3) {
                    v = sums[(q * D + 2) * 256 + tid];
                    v.x += (double)oz[q].x; v.y += (double)oz[q].y;
                    sums[(q * D + 2) * 256 + tid] = v;
                }
            }
        }
        if (NBUF == 2) buf ^= 1;
        else __syncthreads();   // single buffer: everybody has read this tile before the next one overwrites it
    }

    float* __restrict__ hi = a.acc + (size_t)(2u * by) * D * a.pad;
    float* __restrict__ lo = hi + (size_t)D * a.pad;
    float* __restrict__ qout = QS ? a.qsum + (size_t)by * a.pad : nullptr;
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const double2 vx = REGSUMS ? rx[q] : sums[(q * D + 0) * 256 + tid], vy = REGSUMS ? ry[q] : sums[(q * D + 1) * 256 + tid];
        const double2 vz = (D == 3) ? (REGSUMS ? rz[q] : sums[(q * D + (D == 3 ? 2 : 0)) * 256 + tid]) : double2{0.0, 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned i = tgt0 + (2 * q + h) * 256u;
            if (!a.bad_flag[i]) {  // flagged targets belong to close_set_path + scatter_close_kernel
                store_hi_lo(a, hi, lo, i, h ? vx.y : vx.x);
                store_hi_lo(a, hi, lo, (size_t)a.pad + i, h ? vy.y : vy.x);
                if (D == 3) store_hi_lo(a, hi, lo, 2 * (size_t)a.pad + i, h ? vz.y : vz.x);
                if (QS) { const float v = h ? qq[q].y : qq[q].x; qout[i] = a.accumulate ? qout[i] + v : v; }
            } else if (QS) {
                qout[i] = __builtin_inff();   // a close-set target of this pass keeps no spread sum: it is always a suspect
            }
        }
    }
    if (a.clk) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            a.clk[2 * wg] = t1 - clk_t0;
            a.clk[2 * wg + 1] = r1 - clk_r0;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Close set (nbx_internal.h).  classify_close_kernel lists the shard's candidate targets (a coordinate
// below kCloseCoord; once per position update), classify_sources_kernel the candidate sources of the pass
// being launched (over ALL of the pass's chunks, so pairs that straddle a shard boundary are seen), and
// refine_close_kernel keeps the targets that own a pair with 0 < r^2 < kBadR2 against those sources.  close_set_path is executed by the EXTRA workgroups of a fast launch
// (blockIdx.x < close_blocks, i.e. dispatched first in every slice row): workgroup (cx, y) takes blocks cx, cx + CX, ... of 256 listed targets
// against source slice y with the exact compare-and-select guard, one target per lane, fp64 second
// level, and writes close_acc[y][k][slot]; an empty list costs one scalar load.  The guarded work thus
// rides along the main launch instead of trailing it at low occupancy.
// -------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void classify_close_kernel(KArgs a) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.count) return;
    a.bad_flag[i] = 0u;   // this launch's refinement starts from clean flags (entries >= count are never set)
    const float* __restrict__ tp = a.pos_all + (size_t)a.tgt_chunk * D * a.pad;
    const float x = tp[i], y = tp[(size_t)a.pad + i], z = (D == 3) ? tp[2 * (size_t)a.pad + i] : 0.0f;
    if (in_close_set<D>(x, y, z)) {
        const unsigned slot = atomicAdd(&a.counters[0], 1u);
        a.cand_list[slot] = i;
        a.cand_pos[slot] = x;
        a.cand_pos[(size_t)a.pad + slot] = y;
        if (D == 3) a.cand_pos[2 * (size_t)a.pad + slot] = z;
    }
}

// Candidate SOURCES of the pass being launched: every real body of the pass's chunk list with a coordinate
// below kCloseCoord, whichever shard owns it (grid = (pad/256, vchunks)).  Reads the exchange buffer as it
// stands when the pass runs, so a REMOTE pass sees the other ranks' freshly gathered chunks.
template <int D>
__global__ __launch_bounds__(256) void classify_sources_kernel(KArgs a) {
    const unsigned l = blockIdx.x * 256u + threadIdx.x;
    int c = a.chunk_first + (int)blockIdx.y;
    c += (c >= a.chunk_skip) ? 1 : 0;
    const unsigned lo = (unsigned)c * a.shard_len;   // first global body of chunk c (host checks c * shard_len < 2^32)
    const unsigned in_chunk = lo >= a.n_total ? 0u : ((a.n_total - lo < a.shard_len) ? a.n_total - lo : a.shard_len);
    if (l >= in_chunk) return;                        // pad entries (massless, at the origin) are no sources
    const float* __restrict__ sp = a.pos_all + (size_t)c * D * a.pad;
    const float x = sp[l], y = sp[(size_t)a.pad + l], z = (D == 3) ? sp[2 * (size_t)a.pad + l] : 0.0f;
    if (in_close_set<D>(x, y, z)) {
        const unsigned slot = atomicAdd(&a.counters[2], 1u);
        a.src_cand_pos[slot] = x;
        a.src_cand_pos[(size_t)a.src_stride + slot] = y;
        if (D == 3) a.src_cand_pos[2 * (size_t)a.src_stride + slot] = z;
    }
}

// Candidate targets against the pass's candidate sources: keep the targets with a partner at 0 < r^2 < kBadR2
// (see nbx_internal.h).  Fixed grid, grid-stride over (target block, source block) pairs of 256 candidates
// each, so the check of a few thousand candidates spreads over the chip instead of running as a handful of
// long workgroups; a target found bad by several workgroups is listed once (atomicExch on its flag).
template <int D>
__global__ __launch_bounds__(256) void refine_close_kernel(KArgs a) {
    constexpr unsigned kSrcBlock = 64;   // sources per work item: short items, many of them (a 256-source item ran 20 us alone on a CU)
    __shared__ float tx[kSrcBlock], ty[kSrcBlock], tz[kSrcBlock];
    const unsigned tid = threadIdx.x;
    const unsigned n = a.counters[0];    // candidate targets (own chunk)
    const unsigned ns = a.counters[2];   // candidate sources (the pass's chunks)
    const unsigned nblk = (n + 255u) / 256u, nsblk = (ns + kSrcBlock - 1u) / kSrcBlock;
    const bool keep_all = n > kRefineLimit || (unsigned long long)n * ns > kRefinePairLimit;
    const unsigned long long npairs = keep_all ? nblk : (unsigned long long)nblk * nsblk;
    for (unsigned long long p = blockIdx.x; p < npairs; p += gridDim.x) {
        const unsigned tb = keep_all ? (unsigned)p : (unsigned)(p / nsblk), sb = keep_all ? 0u : (unsigned)(p - (unsigned long long)tb * nsblk);
        const unsigned slot = tb * 256u + tid;
        const bool valid = slot < n;
        const unsigned ls = valid ? slot : 0;
        const float x = a.cand_pos[ls], y = a.cand_pos[(size_t)a.pad + ls], z = (D == 3) ? a.cand_pos[2 * (size_t)a.pad + ls] : 0.0f;
        bool bad = keep_all;
        if (!keep_all) {
            const unsigned j = sb * kSrcBlock + tid;
            __syncthreads();
            if (tid < kSrcBlock) {
                const bool in = j < ns;
                tx[tid] = in ? a.src_cand_pos[j] : 0.0f;
                ty[tid] = in ? a.src_cand_pos[(size_t)a.src_stride + j] : 0.0f;
                tz[tid] = (D == 3 && in) ? a.src_cand_pos[2 * (size_t)a.src_stride + j] : 0.0f;
            }
            __syncthreads();
            const unsigned lim = (ns - sb * kSrcBlock < kSrcBlock) ? ns - sb * kSrcBlock : kSrcBlock;   // entries beyond lim are never read
            for (unsigned k = 0; k < lim; ++k) {
                const float dx = tx[k] - x, dy = ty[k] - y;
                float r2 = __builtin_fmaf(dy, dy, dx * dx);
                if (D == 3) { const float dz = tz[k] - z; r2 = __builtin_fmaf(dz, dz, r2); }
                bad = bad || (r2 > 0.0f && r2 < kBadR2);
            }
        }
        if (valid && bad) {
            const unsigned i = a.cand_list[slot];
            if (atomicExch(&a.bad_flag[i], 1u) == 0u) a.bad_list[atomicAdd(&a.counters[1], 1u)] = i;
        }
    }
}

template <int D>
__device__ __forceinline__ void close_set_path(const KArgs& a, float4 (&tile)[2][kTile], const unsigned bx, const unsigned by) {
    const unsigned tid = threadIdx.x;
    const unsigned n = a.counters[1];
    const unsigned nblk = (n + 255u) / 256u;
    const unsigned cx = bx, cstride = a.close_blocks;
    const float* __restrict__ tp = a.pos_all + (size_t)a.tgt_chunk * D * a.pad;

    for (unsigned tb = cx; tb < nblk; tb += cstride) {
        const unsigned slot = tb * 256u + tid;
        const bool valid = slot < n;
        const unsigned i = a.bad_list[valid ? slot : 0];
        const float ix = tp[i], iy = tp[(size_t)a.pad + i], iz = (D == 3) ? tp[2 * (size_t)a.pad + i] : 0.0f;
        double ox = 0.0, oy = 0.0, oz = 0.0;

        unsigned t = by * a.tiles_per_split;
        unsigned t_end = t + a.tiles_per_split;
        if (t_end > a.total_tiles) t_end = a.total_tiles;
        TileWalk w;
        w.seek(t, a.tiles_per_chunk);
        float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < t_end) nxt = load_source<D>(a, w, tid);
        int buf = 0;
        for (; t < t_end; ++t) {
            tile[buf][tid] = nxt;
            __syncthreads();
            if (t + 1 < t_end) {
                w.next(a.tiles_per_chunk);
                nxt = load_source<D>(a, w, tid);
            }
            float ax = 0.f, ay = 0.f, az = 0.f;
            const float4* __restrict__ cur = tile[buf];
#pragma unroll 8
            for (int j = 0; j < kTile; ++j) {
                const float4 s = cur[j];
                interact<D>(s.x, s.y, s.z, s.w, ix, iy, iz, ax, ay, az);
            }
            ox += (double)ax; oy += (double)ay; oz += (double)az;
            buf ^= 1;
        }
        if (valid) {
            float* __restrict__ o = a.close_acc + (size_t)by * D * a.pad;
            o[slot] = (float)ox;
            o[(size_t)a.pad + slot] = (float)oy;
            if (D == 3) o[2 * (size_t)a.pad + slot] = (float)oz;
        }
        __syncthreads();  // the next target block reuses the tile buffers
    }
}

// Fold the close-set path's slices (fp64, slice order) into acc: slice 0 receives the sum (added to
// what a preceding LOCAL pass left there when accumulating), the other slices are zeroed.
template <int D>
__global__ __launch_bounds__(256) void scatter_close_kernel(KArgs a) {
    const unsigned n = a.counters[1];
    for (unsigned slot = blockIdx.x * 256u + threadIdx.x; slot < n; slot += gridDim.x * 256u) {
        const unsigned i = a.bad_list[slot];
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
            for (int y = 0; y < a.grid_slices; ++y) v += (double)a.close_acc[((size_t)y * D + k) * a.pad + slot];   // one per grid slice
            float* __restrict__ dst = a.acc + (size_t)k * a.pad + i;
            if (a.accumulate) {
                *dst = (float)((double)*dst + v);
            } else {
                *dst = (float)v;
                for (int s = 1; s < a.splits; ++s) a.acc[((size_t)s * D + k) * a.pad + i] = 0.0f;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Mixed mode (nbx_ctx_set_refine): fp32 for every target, fp64 for the targets whose fp32 sum cannot be trusted to the
// requested relative tolerance.  The rounding error of a target's fp32 sum does not scale with the sum but with what went
// through the adders: with unit roundoff u, the 256-term tile sums and the <= 256 tile flushes contribute a variance of
// about u^2/3 x sum over all running partial sums |p|^2, and the running sums inside a tile grow to the tile's own partial sum
// a_tile -- so the error's standard deviation is  sigma_i ~ c u sqrt(Q_i),  Q_i = sum over tiles |a_tile|^2  (written by
// the fast kernel's QS build at 3 packed fma per tile).  A target is a SUSPECT when tol |a_i| < k sigma_i, i.e.
//     |a_i|^2 < c2 Q_i ,   c2 = (sigma_factor u / tol)^2        (sigma_factor = k c, calibrated on ALL bodies against the
// strict kernel), which is a chance cancellation: the pulls of the tiles add up to far less than they
// are.  Close-set targets (evaluated by the guarded side path, which keeps no Q) are always listed.  The list is rebuilt by
// every force evaluation; its order depends on atomics, each target's result does not.
// -------------------------------------------------------------------------------------------------
// Four lanes per target: lanes 0..D-1 of a quad add one component's planes each, lane 3 the spread sums; the quad's results meet
// through shuffles.  (One lane per target walking all planes in a row -- 224 loads at 32 slices x {hi, lo} -- was latency-bound:
// 55 us at N = 65,536 as at N = 2^20, 5 % of a step at the smaller size.)
template <int D>
__global__ __launch_bounds__(256) void refine_select_kernel(KArgs a) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned i = gid >> 2, part = gid & 3u;
    const bool in = i < a.count;
    double t = 0.0;
    if (in && part < (unsigned)D) {
#pragma unroll 8
        for (int s = 0; s < a.splits; ++s) t += (double)a.acc[((size_t)s * D + part) * a.pad + i];
        t = t * t;
    } else if (in && part == 3u) {
#pragma unroll 8
        for (int s = 0; s < a.grid_slices; ++s) t += (double)a.qsum[(size_t)s * a.pad + i];   // one spread sum per grid slice
    }
    const int base = (int)(threadIdx.x & 63u) & ~3;
    const double n2 = __shfl(t, base) + __shfl(t, base + 1) + (D == 3 ? __shfl(t, base + 2) : 0.0);
    const double Q = __shfl(t, base + 3);
    if (in && part == 0u) {
        const bool suspect = a.bad_flag[i] != 0u || !(Q * a.refine_c2 <= n2);   // a NaN anywhere lists the target
        if (suspect) {
            const unsigned slot = atomicAdd(&a.counters[3], 1u);
            if (slot < a.strict_cap) a.strict_list[slot] = i;   // strict_cap = the shard's pad >= count: always true
        }
    }
}

// One wave64 per (listed target, component): lane l adds slices l, l + 64, ... in fp64, the lanes' sums meet in a butterfly of
// fixed shape (deterministic), lane 0 rewrites the target's planes: plane 0 = hi, plane 1 = lo, the rest 0.  (One lane per target
// walking 256 slices x D in a row took 0.24 ms for 970 targets -- four workgroups, latency-bound; this takes microseconds.)
template <int D>
__global__ __launch_bounds__(256) void refine_fold_kernel(KArgs a) {
    const StrictLayout sl = strict_layout<D>(a);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long items = (unsigned long long)sl.n * (unsigned)D;   // (a shard may hold up to 2^31 targets)
    for (unsigned long long item = blockIdx.x * 4u + (threadIdx.x >> 6); item < items; item += gridDim.x * 4u) {
        const unsigned slot = (unsigned)(item / (unsigned)D), k = (unsigned)(item - (unsigned long long)slot * (unsigned)D);
        double v = 0.0;
        for (unsigned y = lane; y < sl.slices; y += 64u) v += a.strict_acc[((size_t)y * D + k) * sl.stride + slot];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        const unsigned i = a.strict_list[slot];
        const float hi = (float)v;
        if (lane == 0u) {
            a.acc[(size_t)k * a.pad + i] = hi;
            if (a.splits > 1) a.acc[((size_t)D + k) * a.pad + i] = (float)(v - (double)hi);
        }
        for (unsigned sp = 2u + lane; sp < (unsigned)a.splits; sp += 64u) a.acc[((size_t)sp * D + k) * a.pad + i] = 0.0f;
    }
}

// -------------------------------------------------------------------------------------------------
// Potential kernel (diagnostic for energy-drift checks, BASELINE config 5; SOFT = 1: sum_{j != i} m_j/(r^2+eps^2),
// SOFT = 2: sum_{j != i} m_j/sqrt(r^2+eps^2) for the Newtonian law): phi_i = sum_j m_j / r_ij^2
// over the selected sources with the reference's skip rule, so that the energy matching the reference
// law is U = sum_i (G m_i / 4) phi_i  (F = -grad U for U = sum_{i<j} G m_i m_j / (2 r^2)).
// Same tiling as the exact force kernel, two targets per lane, fp64 second level; 9 VALU per pair.
// Writes phi[slice][i] (fp32), summed over slices in fp64 by export_energy_kernel.
// -------------------------------------------------------------------------------------------------
template <int D, int SOFT>
__global__ __launch_bounds__(256, 8) void potential_kernel(KArgs a) {
    constexpr int TPL = 2;
    __shared__ float4 tile[2][kTile];
    const unsigned tid = threadIdx.x;
    unsigned bx, by;
    xcd_tile(bx, by);
    const unsigned tgt0 = bx * (256u * TPL) + tid;
    const float* __restrict__ tp = a.pos_all + (size_t)a.tgt_chunk * D * a.pad;
    float ix[TPL], iy[TPL], iz[TPL];
    double o[TPL];
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
        const unsigned i = tgt0 + q * 256u;
        ix[q] = tp[i];
        iy[q] = tp[(size_t)a.pad + i];
        iz[q] = (D == 3) ? tp[2 * (size_t)a.pad + i] : 0.0f;
        o[q] = 0.0;
    }
    unsigned t = by * a.tiles_per_split;
    unsigned t_end = t + a.tiles_per_split;
    if (t_end > a.total_tiles) t_end = a.total_tiles;
    TileWalk w;
    w.seek(t, a.tiles_per_chunk);
    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < t_end) nxt = load_source<D>(a, w, tid);
    int buf = 0;
    for (; t < t_end; ++t) {
        tile[buf][tid] = nxt;
        __syncthreads();
        // softened law: a body's own entry (r^2 = 0, weight m/eps^2) is excluded by INDEX -- the tile that holds this
        // workgroup's q-th targets is tile (bx*TPL + q) of the target chunk, entry tid (workgroup-uniform test)
        const unsigned rel = w.k - bx * TPL;
        const bool own_tile = SOFT && w.chunk(a.chunk_first, a.chunk_skip) == a.tgt_chunk && rel < (unsigned)TPL;
        if (t + 1 < t_end) {
            w.next(a.tiles_per_chunk);
            nxt = load_source<D>(a, w, tid);
        }
        float p[TPL];
#pragma unroll
        for (int q = 0; q < TPL; ++q) p[q] = 0.0f;
        const float4* __restrict__ cur = tile[buf];
        if (own_tile) {
            for (int j = 0; j < kTile; ++j) {
                const float4 s = cur[j];
#pragma unroll
                for (int q = 0; q < TPL; ++q) {
                    const float dx = s.x - ix[q], dy = s.y - iy[q];
                    float r2 = __builtin_fmaf(dy, dy, dx * dx);
                    if (D == 3) { const float dz = s.z - iz[q]; r2 = __builtin_fmaf(dz, dz, r2); }
                    const float wgt = ((unsigned)q == rel && (unsigned)j == tid) ? 0.0f
                                      : (SOFT == 2 ? __builtin_amdgcn_rsqf(r2 + a.eps2) : __builtin_amdgcn_rcpf(r2 + a.eps2));
                    p[q] = __builtin_fmaf(s.w, wgt, p[q]);
                }
            }
        } else {
#pragma unroll 8
            for (int j = 0; j < kTile; ++j) {
                const float4 s = cur[j];
#pragma unroll
                for (int q = 0; q < TPL; ++q) {
                    const float dx = s.x - ix[q], dy = s.y - iy[q];
                    float r2 = __builtin_fmaf(dy, dy, dx * dx);
                    if (D == 3) { const float dz = s.z - iz[q]; r2 = __builtin_fmaf(dz, dz, r2); }
                    const float r2g = SOFT ? r2 + a.eps2 : ((r2 < kR2SkipF) ? __builtin_inff() : r2);
                    p[q] = __builtin_fmaf(s.w, SOFT == 2 ? __builtin_amdgcn_rsqf(r2g) : __builtin_amdgcn_rcpf(r2g), p[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < TPL; ++q) o[q] += (double)p[q];
        buf ^= 1;
    }
    float* __restrict__ out = a.acc + (size_t)by * a.pad;  // a.acc = phi[slice][pad] here
#pragma unroll
    for (int q = 0; q < TPL; ++q) out[tgt0 + q * 256u] = (float)o[q];
}

// ---- variant table: the default fast kernel, its one-reciprocal comparator, and the exact (guarded) kernel ------
#define NBX_LDS(TPL, WAVES, UNROLL) accel_lds_kernel<2, TPL, WAVES, UNROLL>, accel_lds_kernel<3, TPL, WAVES, UNROLL>, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0
#define NBX_F64(TPL, WAVES, UNROLL, NR, MAG) accel_f64_kernel<2, TPL, WAVES, UNROLL, NR, 0, MAG>, accel_f64_kernel<3, TPL, WAVES, UNROLL, NR, 0, MAG>, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 2, MAG
#define NBX_FAST(PAIRS, WAVES, UNROLL, ONE_RCP) \
    accel_fast_pk_kernel<2, PAIRS, WAVES, UNROLL, ONE_RCP>, accel_fast_pk_kernel<3, PAIRS, WAVES, UNROLL, ONE_RCP>, 1, 256, ONE_RCP, \
    accel_fast_pk_kernel<2, PAIRS, WAVES, UNROLL, ONE_RCP, 1>, accel_fast_pk_kernel<3, PAIRS, WAVES, UNROLL, ONE_RCP, 1>, \
    accel_fast_pk_kernel<2, PAIRS, WAVES, UNROLL, 0, 2>, accel_fast_pk_kernel<3, PAIRS, WAVES, UNROLL, 0, 2>, \
    accel_fast_pk_kernel<2, PAIRS, WAVES, UNROLL, ONE_RCP, 0, 1, 1, 1>, accel_fast_pk_kernel<3, PAIRS, WAVES, UNROLL, ONE_RCP, 0, 1, 1, 1>, 1, 0

// three-level summation: no cap on the tiles per slice (the slice sums are fp64), two planes per slice (hi, lo), no softened builds
#define NBX_FAST3L(PAIRS, WAVES, UNROLL, LB, REGSUMS) \
    accel_fast3l_kernel<2, PAIRS, WAVES, UNROLL, LB, 0, REGSUMS>, accel_fast3l_kernel<3, PAIRS, WAVES, UNROLL, LB, 0, REGSUMS>, 1, 0, 0, \
    nullptr, nullptr, nullptr, nullptr, \
    accel_fast3l_kernel<2, PAIRS, WAVES, UNROLL, LB, 1, REGSUMS>, accel_fast3l_kernel<3, PAIRS, WAVES, UNROLL, LB, 1, REGSUMS>, 2, 0, 1

const KernelVariant kVariants[] = {
    {"fastpk_t8_w3_u4", 8, NBX_FAST(4, 3, 4, 0)},        // two reciprocals per target pair
    {"fastpk3l_t8_w3_u4", 8, NBX_FAST3L(4, 3, 4, 64, 0)},   // + three-level summation (64-source blocks, tile, fp64 slice sums in LDS; single tile buffer)
    {"fastpk1r_t8_w3_u4", 8, NBX_FAST(4, 3, 4, 1)},      // one reciprocal per target pair (needs the extent precondition)
    {"lds_t1_w8_exact_u8", 1, NBX_LDS(1, 8, 8)},         // per-pair compare-and-select guard, self-contained
    {"strict_f64_t4", 4, NBX_F64(4, 2, 2, 2, 0)},        // fp64 throughout (the reference's arithmetic type); ~2.5x the fast kernel's time
    {"strict_f64_t4_mag", 4, NBX_F64(4, 2, 2, 2, 1)},    // + the magnitude sums S_i = sum_j |a_ij| (nbx_ctx_get_aux): the checker's build
};

}  // namespace

const KernelVariant* kernel_variants(int* count) {
    *count = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
    return kVariants;
}

CloseKernels close_kernels() {
    CloseKernels k;
    k.classify[0] = classify_close_kernel<2>; k.classify[1] = classify_close_kernel<3>;
    k.classify_src[0] = classify_sources_kernel<2>; k.classify_src[1] = classify_sources_kernel<3>;
    k.refine[0] = refine_close_kernel<2>;     k.refine[1] = refine_close_kernel<3>;
    k.scatter[0] = scatter_close_kernel<2>;   k.scatter[1] = scatter_close_kernel<3>;
    k.potential[0] = potential_kernel<2, 0>;  k.potential[1] = potential_kernel<3, 0>;
    k.potential_soft[0] = potential_kernel<2, 1>; k.potential_soft[1] = potential_kernel<3, 1>;
    k.potential_newton[0] = potential_kernel<2, 2>; k.potential_newton[1] = potential_kernel<3, 2>;
    k.refine_select[0] = refine_select_kernel<2>; k.refine_select[1] = refine_select_kernel<3>;
    // the mixed mode's re-evaluation: one listed target per lane, ONE Newton step (1/r^2 good to 2.2e-15 where the goal is 1e-5),
    // unroll 8.  Measured at N = 2^20, 7,703 targets (profiles/r3/mixed_mode_list_kernel_ab.txt): two Newton steps 5.66 ms, one
    // 5.23, one + unroll 8: 5.12; two targets per lane 9.26 (126 VGPRs, half as many workgroups of twice the length).
    k.strict_list[0] = accel_f64_kernel<2, 1, 4, 8, 1, 1>; k.strict_list[1] = accel_f64_kernel<3, 1, 4, 8, 1, 1>;
    k.refine_fold[0] = refine_fold_kernel<2>; k.refine_fold[1] = refine_fold_kernel<3>;
    return k;
}

}  // namespace nbx
