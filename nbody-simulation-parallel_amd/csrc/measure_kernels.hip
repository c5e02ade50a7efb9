// measure_kernels.hip -- MEASUREMENT entry of the library (bench.py, tools/): what this chip, on this box, now, makes of a pure
// packed-fp32 FMA stream -- the ceiling the force kernel's roofline fraction is quoted against -- and the shader clock it holds
// while doing so.  Not on the product path: nothing else in the library calls it.
#include "nbx_ctx.h"

#include <algorithm>
#include <vector>

#include "../../include/nbody_hip.h"

namespace nbx {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kChains = 16;   // independent v_pk_fma_f32 chains per lane: three waves per SIMD never wait on a result

// Every lane runs `iters` x kChains v_pk_fma_f32 (4 flop each); lane 0 of each workgroup stamps the shader clock (s_memtime) and
// the 100 MHz reference clock (s_memrealtime) around the loop.
__global__ __launch_bounds__(256) void pk_fma_stream_kernel(float* out, int iters, unsigned long long* stamps) {
    f2 p[kChains];
    const float b0 = 1.0f + threadIdx.x * 1e-7f, c0 = 0.5f - threadIdx.x * 1e-8f;
    const f2 pb = {b0, b0 * 1.01f}, pc = {c0, c0 * 0.99f};
#pragma unroll
    for (int k = 0; k < kChains; ++k) p[k] = f2{threadIdx.x * 0.001f + k, threadIdx.x * 0.002f + k};
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < kChains; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kChains; ++k) acc += p[k].x + p[k].y;
    out[blockIdx.x * 256u + threadIdx.x] = acc;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

#define M_TRY(expr)                                                         \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) { rc = fail_hip(e_, #expr, __FILE__, __LINE__); goto done; } \
    } while (0)

}  // namespace
}  // namespace nbx

using namespace nbx;

extern "C" int nbx_measure_valu_ceiling(int device, double target_ms, double* tflops, double* shader_mhz) {
    if (!(target_ms >= 1.0 && target_ms <= 2000.0)) return fail(NBX_ERR_INVALID, "target_ms must be in [1, 2000]");
    int rc = NBX_OK;
    float* out = nullptr;
    unsigned long long* stamps = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t stream = nullptr;
    std::vector<unsigned long long> host;
    std::vector<double> mhz;
    hipDeviceProp_t prop;
    int grid = 0, iters = 2500;
    float ms = 0.f;
    M_TRY(hipSetDevice(device));
    M_TRY(hipGetDeviceProperties(&prop, device));
    // 24 workgroups per CU, as many resident as fit (eight: two waves per SIMD each), the rest backfilled as they finish: a grid that
    // fills the chip exactly once leaves the time to the dispatcher's placement (3 workgroups per CU on average, 4 on some: measured
    // 98 TFLOP/s where the same chip makes 123)
    grid = prop.multiProcessorCount * 24;
    M_TRY(hipMalloc((void**)&out, (size_t)grid * 256 * sizeof(float)));
    M_TRY(hipMalloc((void**)&stamps, (size_t)grid * 2 * sizeof(unsigned long long)));
    M_TRY(hipEventCreate(&e0));
    M_TRY(hipEventCreate(&e1));
    M_TRY(take_stream(device, &stream));
    for (int pass = 0; pass < 2; ++pass) {   // a short launch sizes the timed one
        M_TRY(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(pk_fma_stream_kernel, dim3((unsigned)grid), dim3(256), 0, stream, out, iters, stamps);
        M_TRY(hipGetLastError());
        M_TRY(hipEventRecord(e1, stream));
        M_TRY(hipStreamSynchronize(stream));
        M_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 0) iters = (int)std::min(2.0e8, std::max(1000.0, iters * target_ms / std::max((double)ms, 0.01)));
    }
    host.resize((size_t)grid * 2);
    M_TRY(hipMemcpy(host.data(), stamps, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < grid; ++i)
        if (host[2 * i + 1]) mhz.push_back((double)host[2 * i] / (double)host[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    if (tflops) *tflops = (double)grid * 256.0 * (double)iters * kChains * 4.0 / ((double)ms * 1e-3) / 1e12;
    if (shader_mhz) *shader_mhz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
done:
    if (stream) { if (hipStreamSynchronize(stream) == hipSuccess) park_stream(device, stream); else (void)hipStreamDestroy(stream); }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (out) (void)hipFree(out);
    if (stamps) (void)hipFree(stamps);
    return rc;
}
