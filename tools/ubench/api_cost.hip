// api_cost.hip -- what the HIP runtime calls around a one-shot force evaluation cost on this box (host wall time).
// hipcc --offload-arch=gfx950 -O2 tools/ubench/api_cost.hip -o gpurun_out/api_cost && gpurun_out/api_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(label, reps, ...) do { double best = 1e30; for (int r_ = 0; r_ < (reps); ++r_) { double t0 = now(); __VA_ARGS__; double t = now() - t0; if (t < best) best = t; } std::printf("%-44s %9.3f ms\n", label, best); } while (0)
__global__ void nop() {}
int main() {
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    nop<<<1, 64>>>(); (void)hipDeviceSynchronize();
    hipDeviceProp_t prop; int v = 0;
    T("hipGetDeviceProperties", 3, (void)hipGetDeviceProperties(&prop, 0));
    T("hipDeviceGetAttribute(multiProcessorCount)", 3, (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0));
    hipStream_t s;
    T("hipStreamCreateWithFlags + Destroy", 5, { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); (void)hipStreamDestroy(s); });
    T("hipStreamCreateWithFlags", 1, (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    T("first launch + sync on new stream", 1, { nop<<<1, 64, 0, s>>>(); (void)hipStreamSynchronize(s); });
    T("launch + sync on that stream", 5, { nop<<<1, 64, 0, s>>>(); (void)hipStreamSynchronize(s); });
    hipEvent_t e;
    T("hipEventCreate + Destroy", 5, { (void)hipEventCreate(&e); (void)hipEventDestroy(e); });
    for (size_t mb : {1, 16, 64, 600}) {
        void* p = nullptr; char lab[64];
        std::snprintf(lab, sizeof lab, "hipMalloc %zu MiB", mb);
        T(lab, 1, (void)hipMalloc(&p, mb << 20));
        std::snprintf(lab, sizeof lab, "hipFree   %zu MiB", mb);
        T(lab, 1, (void)hipFree(p));
        std::snprintf(lab, sizeof lab, "hipMalloc+hipFree %zu MiB (again)", mb);
        T(lab, 3, { (void)hipMalloc(&p, mb << 20); (void)hipFree(p); });
    }
    std::vector<char> h(64 << 20);
    void* d = nullptr; (void)hipMalloc(&d, 64 << 20);
    for (size_t kb : {1, 64, 1024, 4096, 57344}) {
        char lab[64];
        std::snprintf(lab, sizeof lab, "H2D pageable %zu KiB (first)", kb);
        T(lab, 1, { (void)hipMemcpyAsync(d, h.data(), kb << 10, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); });
        std::snprintf(lab, sizeof lab, "H2D pageable %zu KiB (again)", kb);
        T(lab, 3, { (void)hipMemcpyAsync(d, h.data(), kb << 10, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); });
        std::snprintf(lab, sizeof lab, "D2H pageable %zu KiB (first)", kb);
        T(lab, 1, { (void)hipMemcpyAsync(h.data(), d, kb << 10, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
        std::snprintf(lab, sizeof lab, "D2H pageable %zu KiB (again)", kb);
        T(lab, 3, { (void)hipMemcpyAsync(h.data(), d, kb << 10, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
    }
    T("hipStreamDestroy", 1, (void)hipStreamDestroy(s));
    return 0;
}
