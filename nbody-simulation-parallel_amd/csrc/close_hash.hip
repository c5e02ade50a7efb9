// close_hash.hip -- O(n log n) form of the close-set refinement (force_kernel.hip / nbx_internal.h) for systems in
// which MOST bodies are candidates: small-coordinate boxes such as a unit-scale Plummer sphere at the origin, where every
// coordinate is below kCloseCoord and the all-pairs check of refine_close_kernel would be the O(N^2) it exists to avoid.
// Round 1 handed such systems to the guarded kernel (46 % of peak instead of 62 %); with this they keep the fast kernel.
//
// The pass's candidate sources are binned into cells of edge h = 2^-9 (> sqrt(kBadR2) = 1e-3, so a partner below kBadR2
// sits in the same or an adjacent cell), keyed by a 32-bit hash of the integer cell coordinates and sorted by key with
// the slot carried along.  The sort is the library's own (device_sort.h): a stable LSD radix sort, four passes of 8 bits.  The number of candidates lives on the device and is never read
// back: every kernel is launched for the whole capacity and looks at counters[2]; tiles beyond it do nothing.
// Every candidate target then binary-searches its 3^D neighbour cells and compares distances with the same fp32 r^2 as
// the kernels; hash collisions only cost extra comparisons.
#include "nbx_internal.h"
#include "device_sort.h"

namespace nbx {
namespace {

constexpr float kInvCell = 512.0f;   // 1/h, h = 2^-9: scaling by a power of two is exact, so floorf gives the true cell

__device__ __forceinline__ unsigned cell_key(long long cx, long long cy, long long cz) {
    unsigned long long k = (unsigned long long)cx * 0x9E3779B97F4A7C15ull + (unsigned long long)cy * 0xC2B2AE3D27D4EB4Full +
                           (unsigned long long)cz * 0x165667B19E3779F9ull;
    k ^= k >> 31; k *= 0xD6E8FEB86659FD93ull; k ^= k >> 29;
    return (unsigned)(k >> 32);
}

template <int D>
__global__ __launch_bounds__(256) void hash_keys_kernel(KArgs a, HashWork h) {
    const unsigned s = blockIdx.x * 256u + threadIdx.x;
    if (s >= a.counters[2] || s >= h.capacity) return;
    const long long cx = (long long)floorf(a.src_cand_pos[s] * kInvCell);
    const long long cy = (long long)floorf(a.src_cand_pos[(size_t)a.src_stride + s] * kInvCell);
    const long long cz = (D == 3) ? (long long)floorf(a.src_cand_pos[2 * (size_t)a.src_stride + s] * kInvCell) : 0;
    h.keys[s] = cell_key(cx, cy, cz);
    h.vals[s] = s;
}

template <int D>
__global__ __launch_bounds__(256) void hash_refine_kernel(KArgs a, HashWork h) {
    const unsigned n = a.counters[0];
    const unsigned ns = a.counters[2] < h.capacity ? a.counters[2] : h.capacity;   // sorted candidate sources
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
        const float x = a.cand_pos[t], y = a.cand_pos[(size_t)a.pad + t], z = (D == 3) ? a.cand_pos[2 * (size_t)a.pad + t] : 0.0f;
        const long long cx = (long long)floorf(x * kInvCell), cy = (long long)floorf(y * kInvCell);
        const long long cz = (D == 3) ? (long long)floorf(z * kInvCell) : 0;
        bool bad = false;
        for (int oz = (D == 3 ? -1 : 0); oz <= (D == 3 ? 1 : 0) && !bad; ++oz)
            for (int oy = -1; oy <= 1 && !bad; ++oy)
                for (int ox = -1; ox <= 1 && !bad; ++ox) {
                    const unsigned key = cell_key(cx + ox, cy + oy, cz + oz);
                    unsigned lo = 0, hi = ns;          // lower_bound over the sorted keys
                    while (lo < hi) {
                        const unsigned mid = lo + (hi - lo) / 2;
                        if (h.keys[mid] < key) lo = mid + 1; else hi = mid;
                    }
                    for (unsigned i = lo; i < ns && h.keys[i] == key; ++i) {
                        const unsigned j = h.vals[i];
                        const float dx = a.src_cand_pos[j] - x, dy = a.src_cand_pos[(size_t)a.src_stride + j] - y;
                        float r2 = __builtin_fmaf(dy, dy, dx * dx);
                        if (D == 3) { const float dz = a.src_cand_pos[2 * (size_t)a.src_stride + j] - z; r2 = __builtin_fmaf(dz, dz, r2); }
                        if (r2 > 0.0f && r2 < kBadR2) { bad = true; break; }
                    }
                }
        if (bad) {
            const unsigned i = a.cand_list[t];
            if (atomicExch(&a.bad_flag[i], 1u) == 0u) a.bad_list[atomicAdd(&a.counters[1], 1u)] = i;
        }
    }
}

}  // namespace

size_t hash_temp_bytes(unsigned capacity) { return nbx_sort::radix_temp_bytes(capacity); }

hipError_t hash_refine(int dim, const KArgs& a, const HashWork& h, hipStream_t stream) {
    if (!h.keys || !h.keys_alt || !h.vals || !h.vals_alt || !h.temp || h.capacity == 0 || h.capacity > 0x7fffffffu ||
        h.temp_bytes < hash_temp_bytes(h.capacity))
        return hipErrorInvalidValue;
    const dim3 block(256, 1, 1), grid((h.capacity + 255u) / 256u, 1, 1);
    if (dim == 3) hipLaunchKernelGGL(hash_keys_kernel<3>, grid, block, 0, stream, a, h);
    else hipLaunchKernelGGL(hash_keys_kernel<2>, grid, block, 0, stream, a, h);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    unsigned* hist = static_cast<unsigned*>(h.temp);
    const unsigned* count = a.counters + 2;
    unsigned *ka = h.keys, *kb = h.keys_alt, *va = h.vals, *vb = h.vals_alt;
    for (int pass = 0; pass < 4; ++pass) {   // an even number of passes: the sorted arrays end up in keys / vals again
        if ((e = nbx_sort::radix_pass(ka, va, kb, vb, count, h.capacity, 8 * pass, hist, stream)) != hipSuccess) return e;
        unsigned* t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    if (dim == 3) hipLaunchKernelGGL(hash_refine_kernel<3>, dim3(2048, 1, 1), block, 0, stream, a, h);
    else hipLaunchKernelGGL(hash_refine_kernel<2>, dim3(2048, 1, 1), block, 0, stream, a, h);
    return hipGetLastError();
}

}  // namespace nbx
