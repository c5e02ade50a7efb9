// force_kernel.hip -- K1/K2: tiled all-pairs accumulate kernel for gfx950 (MI355X), fp32, D = 2 or 3.
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
//     single ds_read_b128 broadcast feeds 64*TPL pair interactions -- or (SMEM variants) sources
//     arrive through the scalar cache as SGPR operands and cost no vector or LDS instruction;
//   * per pair 14 VALU instructions, all plain fp32 (v_sub x3, v_mul, v_fma x2, v_cmp+v_cndmask,
//     v_rcp, v_mul x2, v_fma x3); packed fp32 buys nothing on gfx950 (measured: v_pk_fma_f32 issues
//     at half the v_fma_f32 rate -- profiles/ubench_valu_r1.txt) and MFMA does not apply (no
//     contraction: the kernel is an element-wise map with a reciprocal in the middle);
//   * two-level summation: fp32 partial sums over one 256-source tile are flushed into an fp64
//     second-level accumulator (3 v_add_f64 per 256 pairs per target: free), which keeps the error
//     at the level of the 256-term inner sums whatever N is (a naive fp32 running sum: ~3e-4 at
//     N = 131,072; an fp32 second level: 1.6e-6 of the magnitude sum at N = 2^20, measured).
//   * grid = (target blocks, source slices): small shards are cut along the source list so the
//     launch still covers all 1024 SIMDs; slice results land in acc[slice] and are summed in a
//     fixed order by the consumer kernels (deterministic, no atomics).
#include "nbx_internal.h"

#include <climits>
#include <cmath>

// This file is compiled twice (Makefile): NBX_FLAVOUR=slp with hipcc's defaults (the SLP vectoriser
// pairs the TPL>=2 arithmetic into v_pk_*_f32) and NBX_FLAVOUR=scalar with -fno-slp-vectorize (plain
// v_*_f32 only).  Both flavours sit in one library so they can be A/B-timed in one process.
#ifndef NBX_FLAVOUR
#define NBX_FLAVOUR slp
#endif
#define NBX_STR2(x) #x
#define NBX_STR(x) NBX_STR2(x)
#define NBX_CAT2(a, b) a##b
#define NBX_CAT(a, b) NBX_CAT2(a, b)

namespace nbx {
namespace NBX_FLAVOUR {
namespace {

enum Guard { GUARD_EXACT = 0, GUARD_CLAMP = 1, GUARD_TINY = 2 };

// GUARD_TINY: r^2 is biased by kTiny so that rcp stays finite for coincident bodies (d = 0 => the
// term is exactly 0) -- no compare, no select.  Exact only if no pair has 0 < r^2 < ~1e-8.
constexpr float kTiny = 1.0e-15f;

// One pair interaction.  s{xyz,m} is wave-uniform (LDS broadcast or SGPR), i{xyz} per lane.
template <int D, int GUARD>
__device__ __forceinline__ void interact(float sx, float sy, float sz, float sm,
                                         float ix, float iy, float iz,
                                         float& ax, float& ay, float& az) {
    const float dx = sx - ix;
    const float dy = sy - iy;
    float r2 = (GUARD == GUARD_TINY) ? __builtin_fmaf(dx, dx, kTiny) : dx * dx;
    r2 = __builtin_fmaf(dy, dy, r2);
    float dz = 0.0f;
    if (D == 3) {
        dz = sz - iz;
        r2 = __builtin_fmaf(dz, dz, r2);
    }
    // methods.cpp:24 -- `if (dist_sq < 1e-10) continue;`  rcp(+inf) = +0 makes the pair's weight 0.
    float r2g;
    if (GUARD == GUARD_EXACT) r2g = (r2 < kR2SkipF) ? __builtin_inff() : r2;
    else if (GUARD == GUARD_TINY) r2g = r2;
    else r2g = __builtin_fmaxf(r2, kR2SkipF);  // experimental: exact only if no pair has 0 < r2 < 1e-10
    const float ri2 = __builtin_amdgcn_rcpf(r2g);  // v_rcp_f32, 1 ulp
    const float t = sm * ri2;
    const float s = t * ri2;  // m_j / r^4
    ax = __builtin_fmaf(s, dx, ax);
    ay = __builtin_fmaf(s, dy, ay);
    if (D == 3) az = __builtin_fmaf(s, dz, az);
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


// -------------------------------------------------------------------------------------------------
// LDS variant: 256 lanes, TPL targets per lane, sources staged in LDS as float4 {x,y,z,m}.
// -------------------------------------------------------------------------------------------------
template <int D, int TPL, int WAVES, int GUARD, int UNROLL>
__global__ __launch_bounds__(256, WAVES) void accel_lds_kernel(KArgs a) {
    __shared__ float4 tile[2][kTile];
    const unsigned tid = threadIdx.x;
    const unsigned tgt0 = blockIdx.x * (256u * TPL) + tid;
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

    unsigned t = blockIdx.y * a.tiles_per_split;
    unsigned t_end = t + a.tiles_per_split;
    if (t_end > a.total_tiles) t_end = a.total_tiles;

    TileWalk w;
    w.seek(t, a.tiles_per_chunk);
    auto load_src = [&](const TileWalk& tw) -> float4 {
        const int c = tw.chunk(a.chunk_first, a.chunk_skip);
        const float* __restrict__ sp = a.pos_all + (size_t)c * D * a.pad + tw.k * kTile + tid;
        float4 v;
        v.x = sp[0];
        v.y = sp[a.pad];
        v.z = (D == 3) ? sp[2 * (size_t)a.pad] : 0.0f;
        v.w = a.mass_all[(size_t)c * a.pad + tw.k * kTile + tid];
        return v;
    };

    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < t_end) nxt = load_src(w);
    int buf = 0;
    for (; t < t_end; ++t) {
        tile[buf][tid] = nxt;
        __syncthreads();
        if (t + 1 < t_end) {  // next tile's global loads fly while this tile is consumed
            w.next(a.tiles_per_chunk);
            nxt = load_src(w);
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
                interact<D, GUARD>(s.x, s.y, s.z, s.w, ix[q], iy[q], iz[q], ax[q], ay[q], az[q]);
        }
#pragma unroll
        for (int q = 0; q < TPL; ++q) { ox[q] += (double)ax[q]; oy[q] += (double)ay[q]; oz[q] += (double)az[q]; }
        buf ^= 1;
    }

    float* __restrict__ out = a.acc + (size_t)blockIdx.y * D * a.pad;
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
        const unsigned i = tgt0 + q * 256u;
        if (a.accumulate) {
            out[i] = (float)((double)out[i] + ox[q]);
            out[(size_t)a.pad + i] = (float)((double)out[(size_t)a.pad + i] + oy[q]);
            if (D == 3) out[2 * (size_t)a.pad + i] = (float)((double)out[2 * (size_t)a.pad + i] + oz[q]);
        } else {
            out[i] = (float)ox[q];
            out[(size_t)a.pad + i] = (float)oy[q];
            if (D == 3) out[2 * (size_t)a.pad + i] = (float)oz[q];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// SMEM variant: no LDS, no barriers.  Source arrays are read with wave-uniform addresses, which
// hipcc turns into s_load_dwordx8 through the scalar cache; x/y/z/m of a source are then SGPR
// operands of the VALU instructions (one SGPR per VOP2/VOP3 on gfx9-family encodings).
// -------------------------------------------------------------------------------------------------
template <int D, int TPL, int WAVES, int GUARD, int BATCH>
__global__ __launch_bounds__(256, WAVES) void accel_smem_kernel(KArgs a) {
    const unsigned tid = threadIdx.x;
    const unsigned tgt0 = blockIdx.x * (256u * TPL) + tid;
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

    unsigned t = blockIdx.y * a.tiles_per_split;
    unsigned t_end = t + a.tiles_per_split;
    if (t_end > a.total_tiles) t_end = a.total_tiles;
    TileWalk w;
    w.seek(t, a.tiles_per_chunk);

    for (; t < t_end; ++t) {
        const int c = w.chunk(a.chunk_first, a.chunk_skip);
        const size_t off = (size_t)c * D * a.pad + w.k * kTile;
        const float* __restrict__ sxp = a.pos_all + off;
        const float* __restrict__ syp = sxp + a.pad;
        const float* __restrict__ szp = sxp + 2 * (size_t)a.pad;
        const float* __restrict__ smp = a.mass_all + (size_t)c * a.pad + w.k * kTile;
        float ax[TPL], ay[TPL], az[TPL];
#pragma unroll
        for (int q = 0; q < TPL; ++q) ax[q] = ay[q] = az[q] = 0.0f;
#pragma unroll 1
        for (int j0 = 0; j0 < kTile; j0 += BATCH) {
            float sx[BATCH], sy[BATCH], sz[BATCH], sm[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                sx[j] = sxp[j0 + j];
                sy[j] = syp[j0 + j];
                sz[j] = (D == 3) ? szp[j0 + j] : 0.0f;
                sm[j] = smp[j0 + j];
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j)
#pragma unroll
                for (int q = 0; q < TPL; ++q)
                    interact<D, GUARD>(sx[j], sy[j], sz[j], sm[j], ix[q], iy[q], iz[q], ax[q], ay[q], az[q]);
        }
#pragma unroll
        for (int q = 0; q < TPL; ++q) { ox[q] += (double)ax[q]; oy[q] += (double)ay[q]; oz[q] += (double)az[q]; }
        w.next(a.tiles_per_chunk);
    }

    float* __restrict__ out = a.acc + (size_t)blockIdx.y * D * a.pad;
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
        const unsigned i = tgt0 + q * 256u;
        if (a.accumulate) {
            out[i] = (float)((double)out[i] + ox[q]);
            out[(size_t)a.pad + i] = (float)((double)out[(size_t)a.pad + i] + oy[q]);
            if (D == 3) out[2 * (size_t)a.pad + i] = (float)((double)out[2 * (size_t)a.pad + i] + oz[q]);
        } else {
            out[i] = (float)ox[q];
            out[(size_t)a.pad + i] = (float)oy[q];
            if (D == 3) out[2 * (size_t)a.pad + i] = (float)oz[q];
        }
    }
}

// ---- variant table --------------------------------------------------------------------------------
#define NBX_LDS(TPL, WAVES, GUARD, UNROLL) \
    accel_lds_kernel<2, TPL, WAVES, GUARD, UNROLL>, accel_lds_kernel<3, TPL, WAVES, GUARD, UNROLL>
#define NBX_SMEM(TPL, WAVES, GUARD, BATCH) \
    accel_smem_kernel<2, TPL, WAVES, GUARD, BATCH>, accel_smem_kernel<3, TPL, WAVES, GUARD, BATCH>
#define NBX_NAME(n) n "_" NBX_STR(NBX_FLAVOUR)

const KernelVariant kVariants[] = {
    {NBX_NAME("lds_t1_w8_exact_u8"), 1, NBX_LDS(1, 8, GUARD_EXACT, 8)},
    {NBX_NAME("lds_t2_w8_exact_u8"), 2, NBX_LDS(2, 8, GUARD_EXACT, 8)},
    {NBX_NAME("lds_t2_w4_exact_u8"), 2, NBX_LDS(2, 4, GUARD_EXACT, 8)},
    {NBX_NAME("lds_t4_w4_exact_u4"), 4, NBX_LDS(4, 4, GUARD_EXACT, 4)},
    {NBX_NAME("lds_t4_w2_exact_u8"), 4, NBX_LDS(4, 2, GUARD_EXACT, 8)},
    {NBX_NAME("smem_t1_w8_exact_b8"), 1, NBX_SMEM(1, 8, GUARD_EXACT, 8)},
    {NBX_NAME("smem_t2_w8_exact_b8"), 2, NBX_SMEM(2, 8, GUARD_EXACT, 8)},
    {NBX_NAME("smem_t2_w4_exact_b16"), 2, NBX_SMEM(2, 4, GUARD_EXACT, 16)},
    {NBX_NAME("smem_t4_w4_exact_b8"), 4, NBX_SMEM(4, 4, GUARD_EXACT, 8)},
    {NBX_NAME("lds_t2_w8_clamp_u8"), 2, NBX_LDS(2, 8, GUARD_CLAMP, 8)},    // experimental guard
    {NBX_NAME("lds_t1_w8_tiny_u8"), 1, NBX_LDS(1, 8, GUARD_TINY, 8)},      // experimental guard
    {NBX_NAME("lds_t2_w8_tiny_u8"), 2, NBX_LDS(2, 8, GUARD_TINY, 8)},      // experimental guard
    {NBX_NAME("lds_t4_w4_tiny_u4"), 4, NBX_LDS(4, 4, GUARD_TINY, 4)},      // experimental guard
    {NBX_NAME("lds_t4_w4_tiny_u8"), 4, NBX_LDS(4, 4, GUARD_TINY, 8)},      // experimental guard
    {NBX_NAME("smem_t2_w8_tiny_b8"), 2, NBX_SMEM(2, 8, GUARD_TINY, 8)},    // experimental guard
    {NBX_NAME("smem_t4_w4_tiny_b8"), 4, NBX_SMEM(4, 4, GUARD_TINY, 8)},    // experimental guard
    {NBX_NAME("smem_t2_w8_clamp_b8"), 2, NBX_SMEM(2, 8, GUARD_CLAMP, 8)},  // experimental guard
};

}  // namespace
}  // namespace NBX_FLAVOUR

const KernelVariant* NBX_CAT(variants_, NBX_FLAVOUR)(int* count) {
    *count = (int)(sizeof(NBX_FLAVOUR::kVariants) / sizeof(NBX_FLAVOUR::kVariants[0]));
    return NBX_FLAVOUR::kVariants;
}

}  // namespace nbx
