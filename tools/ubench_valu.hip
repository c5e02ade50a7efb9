// Micro-benchmark: VALU issue rates and sustained shader clock on gfx950 (MI355X) for the
// instruction mix of the all-pairs force kernel.  Standalone; not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// Every test runs for ~0.1-0.3 s after a 2 s warm-up so the chip sits at its sustained clock.
// Per block the kernel records s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop:
//   clock = d(memtime)/d(memrealtime) * 100 MHz;   cycles/instr/SIMD = d(memtime)/(instr * waves_per_SIMD).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum { K_FMA = 0, K_FMA_SAMEBANK, K_ADD, K_MUL, K_PKFMA, K_PKMUL, K_PKADD, K_RCP, K_RSQ, K_CMPCND, K_MAX,
       K_MIX_SCALAR, K_MIX_PK, K_MIX_NOGUARD, K_FMA_SGPR, K_NKIND };
static const char* kname[] = {"v_fma_f32", "v_fma_f32(same-bank srcs)", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                              "v_pk_add_f32", "v_rcp_f32", "v_rsq_f32", "v_cmp+v_cndmask(2)", "v_max_f32",
                              "mix scalar 14/pair", "mix pk 2tgt (11pk+2cmp+2cnd+2rcp)", "mix scalar no guard 12/pair", "v_fma_f32 (sgpr src)"};
// VALU instructions per unrolled body
static const int kinstr[] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 14 * 4, 17 * 2 + 11, 12 * 4, 16};
// pair interactions per unrolled body (0 = n/a)
static const int kpairs[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 4, 0};

struct Stamp { unsigned long long cyc, real; };

template <int KIND>
__global__ __launch_bounds__(256) void ub(float* out, int iters, Stamp* st, float sarg) {
    float a[16];
    f2 p[16];
    float b = 1.0f + threadIdx.x * 1e-7f, c = 0.5f - threadIdx.x * 1e-8f;
    f2 pb = {b, b * 1.01f}, pc = {c, c * 0.99f};
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = threadIdx.x * 0.001f + k; p[k] = f2{a[k], a[k] + 0.5f}; }
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        } else if constexpr (KIND == K_FMA_SGPR) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "s"(sarg), "v"(c));
        } else if constexpr (KIND == K_FMA_SAMEBANK) {
            // all three sources in VGPRs with equal index mod 4 (v0,v4,v8 pattern) -- explicit registers
            asm volatile(
                "v_mov_b32 v40, %0\n\tv_mov_b32 v44, %1\n\tv_mov_b32 v48, %2\n\t"
                "v_fma_f32 v52, v40, v44, v48\n\tv_fma_f32 v56, v40, v44, v48\n\tv_fma_f32 v60, v40, v44, v48\n\tv_fma_f32 v64, v40, v44, v48\n\t"
                "v_fma_f32 v52, v40, v44, v48\n\tv_fma_f32 v56, v40, v44, v48\n\tv_fma_f32 v60, v40, v44, v48\n\tv_fma_f32 v64, v40, v44, v48\n\t"
                "v_fma_f32 v52, v40, v44, v48\n\tv_fma_f32 v56, v40, v44, v48\n\tv_fma_f32 v60, v40, v44, v48\n\tv_fma_f32 v64, v40, v44, v48\n\t"
                "v_fma_f32 v52, v40, v44, v48\n\tv_fma_f32 v56, v40, v44, v48\n\tv_fma_f32 v60, v40, v44, v48\n\tv_fma_f32 v64, v40, v44, v48\n\t"
                "v_mov_b32 %0, v52"
                : "+v"(a[0]) : "v"(b), "v"(c) : "v40", "v44", "v48", "v52", "v56", "v60", "v64");
        } else if constexpr (KIND == K_ADD) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        } else if constexpr (KIND == K_MUL) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        } else if constexpr (KIND == K_MAX) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        } else if constexpr (KIND == K_PKFMA) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(pb), "v"(pc));
        } else if constexpr (KIND == K_PKMUL) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
        } else if constexpr (KIND == K_PKADD) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
        } else if constexpr (KIND == K_RCP) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
        } else if constexpr (KIND == K_RSQ) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
        } else if constexpr (KIND == K_CMPCND) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
        } else if constexpr (KIND == K_MIX_SCALAR || KIND == K_MIX_NOGUARD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dx, dy, dz, r2, s;
                if constexpr (KIND == K_MIX_SCALAR)
                    asm volatile(
                        "v_sub_f32 %0, %5, %8\n\tv_sub_f32 %1, %6, %8\n\tv_sub_f32 %2, %7, %8\n\t"
                        "v_mul_f32 %3, %0, %0\n\tv_fma_f32 %3, %1, %1, %3\n\tv_fma_f32 %3, %2, %2, %3\n\t"
                        "v_cmp_lt_f32 vcc, %3, %9\n\ts_nop 1\n\tv_cndmask_b32 %3, %3, %10, vcc\n\t"
                        "v_rcp_f32 %4, %3\n\ts_nop 0\n\tv_mul_f32 %3, %4, %11\n\tv_mul_f32 %4, %3, %4\n\t"
                        : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(r2), "=&v"(s)
                        : "v"(a[4 * k + 0]), "v"(a[4 * k + 1]), "v"(a[4 * k + 2]), "v"(b), "v"(c), "v"(pb.y), "v"(a[4 * k + 3])
                        : "vcc");
                else
                    asm volatile(
                        "v_sub_f32 %0, %5, %8\n\tv_sub_f32 %1, %6, %8\n\tv_sub_f32 %2, %7, %8\n\t"
                        "v_mul_f32 %3, %0, %0\n\tv_fma_f32 %3, %1, %1, %3\n\tv_fma_f32 %3, %2, %2, %3\n\t"
                        "v_rcp_f32 %4, %3\n\ts_nop 0\n\tv_mul_f32 %3, %4, %11\n\tv_mul_f32 %4, %3, %4\n\t"
                        : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(r2), "=&v"(s)
                        : "v"(a[4 * k + 0]), "v"(a[4 * k + 1]), "v"(a[4 * k + 2]), "v"(b), "v"(c), "v"(pb.y), "v"(a[4 * k + 3]));
                asm volatile("v_fma_f32 %0, %3, %4, %0\n\tv_fma_f32 %1, %3, %5, %1\n\tv_fma_f32 %2, %3, %6, %2"
                             : "+v"(p[k].x), "+v"(p[k].y), "+v"(p[k + 4].x)
                             : "v"(s), "v"(dx), "v"(dy), "v"(dz));
            }
        } else if constexpr (KIND == K_MIX_PK) {
            // explicit registers: v[40:41]=xi pair, v[42:43]=yi, v[44:45]=zi, sources broadcast in v[46:49] (x,y,z,m)
            // accumulators v[50:55]; temps v[56:67].  2 sources per body, 17 VALU each.
            asm volatile(
                "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %1\n\t"
                "v_mov_b32 v46, %2\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v48, %2\n\tv_mov_b32 v49, %2\n\t"
                ".rept 2\n\t"
                "v_pk_add_f32 v[56:57], v[46:47], v[40:41] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_add_f32 v[58:59], v[46:47], v[42:43] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_add_f32 v[60:61], v[48:49], v[44:45] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_mul_f32 v[62:63], v[56:57], v[56:57]\n\t"
                "v_pk_fma_f32 v[62:63], v[58:59], v[58:59], v[62:63]\n\t"
                "v_pk_fma_f32 v[62:63], v[60:61], v[60:61], v[62:63]\n\t"
                "s_nop 0\n\t"
                "v_cmp_lt_f32 vcc, v62, %3\n\ts_nop 1\n\tv_cndmask_b32 v62, v62, %4, vcc\n\t"
                "v_cmp_lt_f32 vcc, v63, %3\n\ts_nop 1\n\tv_cndmask_b32 v63, v63, %4, vcc\n\t"
                "v_rcp_f32 v64, v62\n\tv_rcp_f32 v65, v63\n\ts_nop 0\n\t"
                "v_pk_mul_f32 v[66:67], v[48:49], v[64:65] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
                "v_pk_mul_f32 v[66:67], v[66:67], v[64:65]\n\t"
                "s_nop 0\n\t"
                "v_pk_fma_f32 v[50:51], v[66:67], v[56:57], v[50:51]\n\t"
                "v_pk_fma_f32 v[52:53], v[66:67], v[58:59], v[52:53]\n\t"
                "v_pk_fma_f32 v[54:55], v[66:67], v[60:61], v[54:55]\n\t"
                ".endr\n\t"
                "v_mov_b32 %0, v50"
                : "+v"(a[0]) : "v"(b), "v"(a[1]), "v"(c), "v"(pb.y)
                : "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",
                  "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = t1 - t0; st[blockIdx.x].real = r1 - r0; }
}

template <int KIND>
static void run(int blocks_per_cu, float* d_out, Stamp* d_st, int ncu) {
    int grid = ncu * blocks_per_cu;
    int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    // calibrate iters so the timed launch runs ~150 ms
    for (int pass = 0; pass < 3; ++pass) {
        CK(hipEventRecord(e0));
        ub<KIND><<<grid, 256>>>(d_out, iters, d_st, 1.5f);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass < 2) iters = (int)std::min(4.0e7, std::max(1000.0, iters * 150.0 / std::max(ms, 0.01f)));
    }
    std::vector<Stamp> st(grid);
    CK(hipMemcpy(st.data(), d_st, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc(grid), clk(grid);
    for (int i = 0; i < grid; ++i) { cyc[i] = (double)st[i].cyc; clk[i] = (double)st[i].cyc / (double)st[i].real * 100.0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    double medcyc = cyc[grid / 2], medclk = clk[grid / 2];
    double instr_per_wave = (double)iters * kinstr[KIND];
    double cpi = medcyc / (instr_per_wave * blocks_per_cu);
    double wall_ns_per_instr_simd = (ms * 1e6) / (instr_per_wave * blocks_per_cu);
    printf("%-40s w/SIMD=%d  ms=%7.2f  clk=%5.0f MHz  cyc/instr/SIMD=%6.3f  wall ns/instr/SIMD=%6.3f", kname[KIND], blocks_per_cu, ms, medclk, cpi, wall_ns_per_instr_simd);
    if (kpairs[KIND]) {
        double pairs = (double)iters * kpairs[KIND] * 64.0 * 4.0 * grid;
        printf("  => %.3f T-pairs/s", pairs / (ms * 1e-3) * 1e-12);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("device %s  CUs=%d  clock=%d kHz  wavefront=%d\n", prop.gcnArchName, ncu, prop.clockRate, prop.warpSize);
    float* d_out; Stamp* d_st;
    CK(hipMalloc(&d_out, sizeof(float) * ncu * 8 * 256));
    CK(hipMalloc(&d_st, sizeof(Stamp) * ncu * 8));
    // warm-up ~2 s
    for (int i = 0; i < 12; ++i) { ub<K_FMA><<<ncu * 8, 256>>>(d_out, 2000000, d_st, 1.5f); }
    CK(hipDeviceSynchronize());
    for (int w : {1, 2, 4, 8}) {
        run<K_FMA>(w, d_out, d_st, ncu);
        run<K_FMA_SGPR>(w, d_out, d_st, ncu);
        run<K_FMA_SAMEBANK>(w, d_out, d_st, ncu);
        run<K_ADD>(w, d_out, d_st, ncu);
        run<K_MUL>(w, d_out, d_st, ncu);
        run<K_MAX>(w, d_out, d_st, ncu);
        run<K_PKFMA>(w, d_out, d_st, ncu);
        run<K_PKMUL>(w, d_out, d_st, ncu);
        run<K_PKADD>(w, d_out, d_st, ncu);
        run<K_RCP>(w, d_out, d_st, ncu);
        run<K_RSQ>(w, d_out, d_st, ncu);
        run<K_CMPCND>(w, d_out, d_st, ncu);
        run<K_MIX_SCALAR>(w, d_out, d_st, ncu);
        run<K_MIX_NOGUARD>(w, d_out, d_st, ncu);
        run<K_MIX_PK>(w, d_out, d_st, ncu);
        printf("\n");
    }
    return 0;
}
