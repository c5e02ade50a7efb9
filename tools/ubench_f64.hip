// Micro-benchmark: fp64 VALU issue rates on gfx950 (MI355X) for the instruction mix of the strict force kernel
// (accel_f64_kernel, csrc/force_kernel.hip), and the accuracy of v_rcp_f64 before / after Newton steps.
// Standalone; not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/ubench_f64
// Method as tools/ubench_valu.hip: s_memtime around an unrolled loop, median over workgroups, 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

enum { K_FMA64 = 0, K_ADD64, K_MUL64, K_RCP64, K_CMPCND64, K_CVT64, K_SQRT32, K_MIX64, K_NKIND };
static const char* kname[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_cmp_f64+2 v_cndmask", "v_cvt_f64_f32",
                              "v_sqrt_f32", "mix strict pair (3 add, 10 fma, 3 mul, rcp, cmp, 2 cnd)"};
static const int kinstr[] = {16, 16, 16, 16, 8 * 3, 16, 16, 4 * 20};
static const int kpairs[] = {0, 0, 0, 0, 0, 0, 0, 4};

struct Stamp { unsigned long long cyc, real; };

template <int KIND>
__global__ __launch_bounds__(256) void ub(double* out, int iters, Stamp* st) {
    double a[16];
    float f[16];
    double b = 1.0 + threadIdx.x * 1e-9, c = 0.5 - threadIdx.x * 1e-10;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = threadIdx.x * 0.001 + k + 1.0; f[k] = (float)a[k]; }
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        } else if constexpr (KIND == K_ADD64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        } else if constexpr (KIND == K_MUL64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        } else if constexpr (KIND == K_RCP64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
        } else if constexpr (KIND == K_CMPCND64) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {   // compiler-made: v_cmp_*_f64 + two v_cndmask_b32 (checked in the ISA)
                a[k] = (a[k] < b) ? c : a[k];
                asm volatile("" : "+v"(a[k]));
            }
        } else if constexpr (KIND == K_CVT64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f[k]));
        } else if constexpr (KIND == K_SQRT32) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
        } else if constexpr (KIND == K_MIX64) {
            // the strict kernel's per-pair sequence, 4 independent pairs per trip (compiler-scheduled, like the kernel)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double dx = b - a[4 * k], dy = c - a[4 * k + 1], dz = b - a[4 * k + 2];
                double r2 = dx * dx;
                r2 = __builtin_fma(dy, dy, r2);
                r2 = __builtin_fma(dz, dz, r2);
                double w = __builtin_amdgcn_rcp(r2);
                double e = __builtin_fma(-r2, w, 1.0);
                w = __builtin_fma(w, e, w);
                e = __builtin_fma(-r2, w, 1.0);
                w = __builtin_fma(w, e, w);
                double s = (w * w) * c;
                s = (r2 < 1e-10) ? 0.0 : s;
                a[4 * k] = __builtin_fma(s, dx, a[4 * k]);
                a[4 * k + 1] = __builtin_fma(s, dy, a[4 * k + 1]);
                a[4 * k + 2] = __builtin_fma(s, dz, a[4 * k + 2]);
                asm volatile("" : "+v"(a[4 * k]), "+v"(a[4 * k + 1]), "+v"(a[4 * k + 2]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += a[k] + (double)f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = t1 - t0; st[blockIdx.x].real = r1 - r0; }
}

template <int KIND>
static void run(int blocks_per_cu, double* d_out, Stamp* d_st, int ncu) {
    int grid = ncu * blocks_per_cu;
    int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int pass = 0; pass < 3; ++pass) {
        CK(hipEventRecord(e0));
        ub<KIND><<<grid, 256>>>(d_out, iters, d_st);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass < 2) iters = (int)std::min(4.0e7, std::max(1000.0, iters * 100.0 / std::max(ms, 0.01f)));
    }
    std::vector<Stamp> st(grid);
    CK(hipMemcpy(st.data(), d_st, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc(grid), clk(grid);
    for (int i = 0; i < grid; ++i) { cyc[i] = (double)st[i].cyc; clk[i] = (double)st[i].cyc / (double)st[i].real * 100.0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double instr_per_wave = (double)iters * kinstr[KIND];
    printf("%-62s w/SIMD=%d  ms=%7.2f  clk=%5.0f MHz  cyc/instr/SIMD=%6.3f  wall ns/instr/SIMD=%6.3f", kname[KIND], blocks_per_cu, ms,
           clk[grid / 2], cyc[grid / 2] / (instr_per_wave * blocks_per_cu), (ms * 1e6) / (instr_per_wave * blocks_per_cu));
    if (kpairs[KIND]) printf("  => %.3f T-pairs/s", (double)iters * kpairs[KIND] * 64.0 * 4.0 * grid / (ms * 1e-3) * 1e-12);
    printf("\n");
    fflush(stdout);
}

// accuracy of the reciprocal seed and of its Newton refinements, against 1.0/x in IEEE double
__global__ void rcp_accuracy(const double* x, double* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r2 = x[i];
    double w = __builtin_amdgcn_rcp(r2);
    y[i] = w;
    double e = __builtin_fma(-r2, w, 1.0);
    w = __builtin_fma(w, e, w);
    y[n + i] = w;
    e = __builtin_fma(-r2, w, 1.0);
    w = __builtin_fma(w, e, w);
    y[2 * n + i] = w;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s  CUs=%d  clock=%d kHz\n", prop.gcnArchName, ncu, prop.clockRate);
    {
        const int n = 1 << 20;
        std::vector<double> x(n), y(3 * n);
        unsigned long long s = 88172645463325252ull;
        for (int i = 0; i < n; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            x[i] = std::ldexp(1.0 + (double)(s >> 11) * 0x1p-53, (int)(s % 200) - 100);
        }
        double *dx, *dy;
        CK(hipMalloc(&dx, n * sizeof(double))); CK(hipMalloc(&dy, 3 * n * sizeof(double)));
        CK(hipMemcpy(dx, x.data(), n * sizeof(double), hipMemcpyHostToDevice));
        rcp_accuracy<<<n / 256, 256>>>(dx, dy, n);
        CK(hipMemcpy(y.data(), dy, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; ++k) {
            double worst = 0.0;
            for (int i = 0; i < n; ++i) worst = std::max(worst, std::fabs(y[k * n + i] * x[i] - 1.0));
            printf("v_rcp_f64 + %d Newton step(s): max |x*w - 1| = %.3e over 2^20 values in [2^-100, 2^100]\n", k, worst);
        }
        CK(hipFree(dx)); CK(hipFree(dy));
    }
    double* d_out; Stamp* d_st;
    CK(hipMalloc(&d_out, sizeof(double) * ncu * 8 * 256));
    CK(hipMalloc(&d_st, sizeof(Stamp) * ncu * 8));
    for (int i = 0; i < 6; ++i) ub<K_FMA64><<<ncu * 4, 256>>>(d_out, 1000000, d_st);
    CK(hipDeviceSynchronize());
    for (int w : {1, 2, 4}) {
        run<K_FMA64>(w, d_out, d_st, ncu);
        run<K_ADD64>(w, d_out, d_st, ncu);
        run<K_MUL64>(w, d_out, d_st, ncu);
        run<K_RCP64>(w, d_out, d_st, ncu);
        run<K_CMPCND64>(w, d_out, d_st, ncu);
        run<K_CVT64>(w, d_out, d_st, ncu);
        run<K_SQRT32>(w, d_out, d_st, ncu);
        run<K_MIX64>(w, d_out, d_st, ncu);
        printf("\n");
    }
    return 0;
}
