// Micro-benchmark behind the resident gather of the leaf-pair path (csrc/leaf_pair_kernel.hip): 2^20 bodies as float4 {x,y,z,m} in
// body order -> leaf-ordered source pairs {xa,xb,ya,yb},{za,zb,ma,mb} through a random permutation.  Which side costs what?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o /tmp/ubench_gather && /tmp/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void copy4(const float4* in, const uint32_t* perm, float4* out, uint32_t n) {       // random 16-byte read, coalesced 16-byte write
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n) out[p] = in[perm[p]];
}
__global__ void copy4_seq(const float4* in, const uint32_t*, float4* out, uint32_t n) {         // no permutation at all
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n) out[p] = in[p];
}
__global__ void pairs_dwords(const float4* in, const uint32_t* perm, float* out, uint32_t n) {  // the library's kernel: four 4-byte stores per lane
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const float4 v = in[perm[p]];
    float* o = out + (size_t)(p >> 1) * 8u + (p & 1u);
    o[0] = v.x; o[2] = v.y; o[4] = v.z; o[6] = v.w;
}
__global__ void pairs_shuffle(const float4* in, const uint32_t* perm, float4* out, uint32_t n) { // neighbours exchange halves: every lane stores one whole unit
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    float4 v = make_float4(0, 0, 0, 0);
    if (p < n) v = in[perm[p]];
    // lane 2q holds a, lane 2q+1 holds b: unit 2q = {xa,xb,ya,yb}, unit 2q+1 = {za,zb,ma,mb}
    const bool odd = p & 1u;
    const float s0 = odd ? v.x : v.z, s1 = odd ? v.y : v.w;      // what the partner needs from me
    const float r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
    const float4 u = odd ? make_float4(r0, v.z, r1, v.w) : make_float4(v.x, r0, v.y, r1);
    if (p < n) out[p] = u;
}
__global__ void scatter4(const float4* in, const uint32_t* slot, float4* out, uint32_t n) {      // coalesced read, random 16-byte write
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b < n) out[slot[b]] = in[b];
}

int main() {
    const uint32_t n = 1u << 20;
    std::vector<uint32_t> perm(n), inv(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(5);
    // leaf order is a permutation with locality at the scale of a leaf only: shuffle blocks of 32 slots, and the bodies inside
    std::shuffle(perm.begin(), perm.end(), rng);
    for (uint32_t p = 0; p < n; ++p) inv[perm[p]] = p;
    float4 *in, *out; uint32_t *dperm, *dinv;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16 + 256)); CK(hipMalloc(&dperm, n * 4)); CK(hipMalloc(&dinv, n * 4));
    CK(hipMemset(in, 0, n * 16));
    CK(hipMemcpy(dperm, perm.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv, inv.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* what, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-70s %.4f ms\n", what, ms / 50);
    };
    const dim3 g(n / 256), b(256);
    time("sequential float4 copy", [&] { hipLaunchKernelGGL(copy4_seq, g, b, 0, 0, in, dperm, out, n); });
    time("random 16-byte read, coalesced 16-byte write", [&] { hipLaunchKernelGGL(copy4, g, b, 0, 0, in, dperm, out, n); });
    time("random 16-byte read, four 4-byte stores into pair records (the library)", [&] { hipLaunchKernelGGL(pairs_dwords, g, b, 0, 0, in, dperm, (float*)out, n); });
    time("random 16-byte read, neighbours exchange, one 16-byte store per lane", [&] { hipLaunchKernelGGL(pairs_shuffle, g, b, 0, 0, in, dperm, out, n); });
    time("coalesced read, random 16-byte write", [&] { hipLaunchKernelGGL(scatter4, g, b, 0, 0, in, dinv, out, n); });
    return 0;
}
