// Test driver for csrc/device_sort.h (built with hipcc and run on the GPU box by tests/test_gpu_device_sort.py): the library's own
// stable LSD radix sort and exclusive scan against std::stable_sort / a serial scan, at sizes around the tile boundaries, with the
// element count living on the device (smaller than the launch's capacity) as the library uses it.  Prints "ok ..." and returns 0.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "../nbody-simulation-parallel_amd/csrc/device_sort.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)

int main() {
    using namespace nbx_sort;
    std::mt19937 rng(12345);
    const unsigned sizes[] = {0u, 1u, 2u, 255u, 256u, 257u, 2047u, 2048u, 2049u, 4095u, 4096u, 4097u, 70001u, (1u << 20) + 3u};
    unsigned checked = 0;
    for (unsigned n : sizes) {
        for (int key_bits : {3, 14, 32}) {
            const unsigned capacity = n + (n % 3u) * 1000u + 5u;     // the launch's upper bound; the count lives on the device
            std::vector<unsigned> keys(capacity), vals(capacity);
            for (unsigned i = 0; i < capacity; ++i) { keys[i] = key_bits == 32 ? (unsigned)rng() : (unsigned)rng() & ((1u << key_bits) - 1u); vals[i] = i; }
            unsigned *dk, *dv, *dk2, *dv2, *hist, *dcount, *tiles;
            CK(hipMalloc((void**)&dk, capacity * 4)); CK(hipMalloc((void**)&dv, capacity * 4));
            CK(hipMalloc((void**)&dk2, capacity * 4)); CK(hipMalloc((void**)&dv2, capacity * 4));
            CK(hipMalloc((void**)&hist, radix_temp_bytes(capacity))); CK(hipMalloc((void**)&dcount, 4));
            CK(hipMalloc((void**)&tiles, (scan_tiles(capacity + 1) + 2) * 4));
            CK(hipMemcpy(dk, keys.data(), capacity * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, vals.data(), capacity * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dcount, &n, 4, hipMemcpyHostToDevice));
            CK(hipMemset(dk2, 0xee, capacity * 4)); CK(hipMemset(dv2, 0xee, capacity * 4));
            const int passes = (key_bits + 7) / 8;
            unsigned *a = dk, *av = dv, *b = dk2, *bv = dv2;
            for (int p = 0; p < passes; ++p) {
                CK(radix_pass(a, av, b, bv, dcount, capacity, 8 * p, hist, 0));
                std::swap(a, b); std::swap(av, bv);
            }
            CK(hipDeviceSynchronize());
            std::vector<unsigned> gk(capacity), gv(capacity);
            CK(hipMemcpy(gk.data(), a, capacity * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gv.data(), av, capacity * 4, hipMemcpyDeviceToHost));
            std::vector<unsigned> order(n);
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return keys[x] < keys[y]; });
            for (unsigned i = 0; i < n; ++i)
                if (gk[i] != keys[order[i]] || gv[i] != order[i]) { printf("sort differs: n %u key_bits %d at %u\n", n, key_bits, i); return 1; }
            // the first output index of each digit of the LAST pass, as the stable partition reads it: hist[digit * tiles]
            if (passes == 1) {
                std::vector<unsigned> h(256u * sort_tiles(capacity));
                CK(hipMemcpy(h.data(), hist, h.size() * 4, hipMemcpyDeviceToHost));
                unsigned at = 0;
                for (unsigned d = 0; d < 8u; ++d) {
                    if (h[(size_t)d * sort_tiles(capacity)] != at) { printf("class base differs: n %u digit %u\n", n, d); return 1; }
                    for (unsigned i = 0; i < n; ++i) at += keys[i] == d;
                }
            }
            // exclusive scan of the (unsorted) keys' low bits, in place, with the total behind the last element
            std::vector<unsigned> in(capacity + 1, 7u);
            for (unsigned i = 0; i < capacity; ++i) in[i] = keys[i] & 1023u;
            unsigned* ds;
            CK(hipMalloc((void**)&ds, (capacity + 1) * 4));
            CK(hipMemcpy(ds, in.data(), (capacity + 1) * 4, hipMemcpyHostToDevice));
            CK(exclusive_scan(ds, ds, dcount, capacity, tiles, 0));
            CK(hipDeviceSynchronize());
            std::vector<unsigned> out(capacity + 1);
            CK(hipMemcpy(out.data(), ds, (capacity + 1) * 4, hipMemcpyDeviceToHost));
            unsigned run = 0;
            for (unsigned i = 0; i < n; ++i) { if (out[i] != run) { printf("scan differs: n %u at %u\n", n, i); return 1; } run += in[i]; }
            if (out[n] != run) { printf("scan total differs: n %u (%u vs %u)\n", n, out[n], run); return 1; }
            for (unsigned i = n + 1; i <= capacity; ++i) if (out[i] != in[i]) { printf("scan wrote past the count: n %u at %u\n", n, i); return 1; }
            CK(hipFree(dk)); CK(hipFree(dv)); CK(hipFree(dk2)); CK(hipFree(dv2)); CK(hipFree(hist)); CK(hipFree(dcount)); CK(hipFree(tiles)); CK(hipFree(ds));
            ++checked;
        }
    }
    printf("ok: %u (size, key width) cases: stable radix sort, class bases, exclusive scan\n", checked);
    return 0;
}
