// device_sort.h -- the library's own device-side exclusive scan and stable LSD radix sort (32-bit keys, a 32-bit value carried
// along), shared by close_hash.hip (sorted cells of the close-set refinement) and leaf_plan_device.h (the leaf plan laid out on
// the device).  No library underneath: plain HIP kernels, every one launched for an upper bound of elements and reading the
// actual count from device memory, so that a pipeline of them needs no host read-back in between.  Included into the
// translation units that use it (the kernels have internal linkage).
//
//   radix pass (8 bits) = radix_hist_kernel   per-tile digit histograms           hist[digit][tile]
//                       + radix_scan_kernel   exclusive scan of the whole table   (digit-major = output order), one workgroup
//                       + radix_scatter_kernel  stable scatter: a tile's elements are ranked in index order -- round by round
//                                               (256 consecutive elements), wave by wave, lane by lane (ballots over the
//                                               digit's bits + per-wave counts in LDS)
//   scan = scan_tiles_kernel (each workgroup scans its tile, leaves the tile's sum) + scan_sums_kernel (one workgroup scans the
//          tile sums) + scan_add_kernel (adds them back; writes the grand total behind the last element)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbx_sort {
namespace {

constexpr unsigned kSortTile = 4096;     // elements per workgroup and radix pass: 16 rounds of 256
constexpr unsigned kScanThreads = 1024;
constexpr unsigned kScanTile = 2048;     // elements per workgroup of scan_tiles_kernel: 8 per lane

__host__ __device__ inline unsigned sort_tiles(unsigned capacity) { return (capacity + kSortTile - 1u) / kSortTile; }
inline size_t radix_temp_bytes(unsigned capacity) { return (size_t)256 * sort_tiles(capacity) * sizeof(unsigned); }
__host__ __device__ inline unsigned scan_tiles(unsigned capacity) { return (capacity + kScanTile - 1u) / kScanTile; }

// hist[digit][tile] = number of this tile's keys with that digit (zero for tiles beyond the count)
__global__ __launch_bounds__(256) void radix_hist_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ count, unsigned capacity,
                                                         int shift, unsigned* __restrict__ hist, unsigned tiles) {
    __shared__ unsigned bins[256];
    const unsigned n = *count < capacity ? *count : capacity;
    const unsigned base = blockIdx.x * kSortTile;
    bins[threadIdx.x] = 0;
    __syncthreads();
    for (unsigned r = 0; r < kSortTile / 256u; ++r) {
        const unsigned i = base + r * 256u + threadIdx.x;
        if (i < n) atomicAdd(&bins[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * tiles + blockIdx.x] = bins[threadIdx.x];
}

// exclusive scan of `entries` words in place, one workgroup
__global__ __launch_bounds__(kScanThreads) void radix_scan_kernel(unsigned* __restrict__ hist, unsigned entries) {
    __shared__ unsigned part[kScanThreads];
    const unsigned per = (entries + kScanThreads - 1u) / kScanThreads;
    const unsigned lo = threadIdx.x * per < entries ? threadIdx.x * per : entries, hi = lo + per < entries ? lo + per : entries;
    unsigned sum = 0;
    for (unsigned i = lo; i < hi; ++i) sum += hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned d = 1; d < kScanThreads; d <<= 1) {   // Hillis-Steele over the 1,024 partial sums
        const unsigned v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
    for (unsigned i = lo; i < hi; ++i) { const unsigned v = hist[i]; hist[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void radix_scatter_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
                                                            unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out,
                                                            const unsigned* __restrict__ count, unsigned capacity, int shift,
                                                            const unsigned* __restrict__ hist, unsigned tiles) {
    __shared__ unsigned next[256];        // where the tile's next element of each digit goes
    __shared__ unsigned wave_cnt[4][256]; // this round's count per wave and digit
    const unsigned n = *count < capacity ? *count : capacity;
    const unsigned base = blockIdx.x * kSortTile;
    if (base >= n) return;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    next[threadIdx.x] = hist[(size_t)threadIdx.x * tiles + blockIdx.x];
    for (unsigned r = 0; r < kSortTile / 256u && base + r * 256u < n; ++r) {
        for (unsigned w = 0; w < 4; ++w) wave_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const unsigned i = base + r * 256u + threadIdx.x;
        const bool live = i < n;
        const unsigned key = live ? keys[i] : 0u, val = live ? vals[i] : 0u;
        const unsigned digit = (key >> shift) & 255u;
        unsigned long long same = __ballot(live);   // lanes of this wave with my digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        const unsigned rank_in_wave = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        if (live && rank_in_wave == 0) wave_cnt[wave][digit] = (unsigned)__popcll(same);
        __syncthreads();
        if (live) {
            unsigned before = next[digit];
            for (unsigned w = 0; w < wave; ++w) before += wave_cnt[w][digit];
            keys_out[before + rank_in_wave] = key;
            vals_out[before + rank_in_wave] = val;
        }
        __syncthreads();
        next[threadIdx.x] += wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x];
        __syncthreads();
    }
}

// One 8-bit pass over keys / vals -> keys_out / vals_out.  hist: radix_temp_bytes(capacity); afterwards hist[digit * tiles] is the
// first output index of `digit` (what a stable partition by a small key wants to know).
inline hipError_t radix_pass(const unsigned* keys, const unsigned* vals, unsigned* keys_out, unsigned* vals_out, const unsigned* count,
                             unsigned capacity, int shift, unsigned* hist, hipStream_t stream) {
    const unsigned tiles = sort_tiles(capacity);
    hipLaunchKernelGGL(radix_hist_kernel, dim3(tiles), dim3(256), 0, stream, keys, count, capacity, shift, hist, tiles);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, hist, 256u * tiles);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(tiles), dim3(256), 0, stream, keys, vals, keys_out, vals_out, count, capacity, shift, hist, tiles);
    return hipGetLastError();
}

// ---- exclusive scan of in[0 .. n) -> out[0 .. n), out[n] = total; n = min(*count, capacity); in == out allowed ----
__global__ __launch_bounds__(256) void scan_tiles_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, const unsigned* __restrict__ count,
                                                         unsigned capacity, unsigned* __restrict__ tile_sums) {
    __shared__ unsigned part[256];
    const unsigned n = *count < capacity ? *count : capacity;
    const unsigned base = blockIdx.x * kScanTile + threadIdx.x * 8u;
    unsigned v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 256u; d <<= 1) {
        const unsigned t = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255u) tile_sums[blockIdx.x] = part[255];
}

__global__ __launch_bounds__(256) void scan_add_kernel(unsigned* __restrict__ out, const unsigned* __restrict__ count, unsigned capacity,
                                                       const unsigned* __restrict__ tile_sums /* scanned */, unsigned tiles) {
    const unsigned n = *count < capacity ? *count : capacity;
    const unsigned add = tile_sums[blockIdx.x];
    const unsigned base = blockIdx.x * kScanTile + threadIdx.x * 8u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (base + k < n) out[base + k] += add;
    // the grand total behind the last element: the tile that holds element n - 1 (or tile 0 when n == 0) knows it
    if (threadIdx.x == 0u && blockIdx.x == (n ? (n - 1u) / kScanTile : 0u)) out[n] = tile_sums[tiles];
}

// tile_sums: scan_tiles(capacity) + 1 words (the scanned tile sums and, behind them, the grand total)
inline hipError_t exclusive_scan(const unsigned* in, unsigned* out, const unsigned* count, unsigned capacity, unsigned* tile_sums, hipStream_t stream) {
    const unsigned tiles = scan_tiles(capacity ? capacity : 1u);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(256), 0, stream, in, out, count, capacity, tile_sums);
    // the one-workgroup scan leaves the total of its `entries` words nowhere: scan tiles + 1 words with a zero behind the sums
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, tile_sums, tiles + 1u);
    hipLaunchKernelGGL(scan_add_kernel, dim3(tiles), dim3(256), 0, stream, out, count, capacity, tile_sums, tiles);
    return hipGetLastError();
}

}  // namespace
}  // namespace nbx_sort
