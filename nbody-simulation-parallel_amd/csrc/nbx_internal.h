// nbx_internal.h -- shared between the HIP translation units of libnbody_hip.so (not installed).
#ifndef NBX_INTERNAL_H
#define NBX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbx {

// Source tile = bodies staged per LDS fill (BASELINE config 2: "LDS tile=256") = workgroup size.
constexpr int kTile = 256;
// Shard arrays are padded to a multiple of this many bodies so that every force-kernel variant
// (up to 4 targets per lane x 256 lanes) sees whole target blocks and whole source tiles.
constexpr int kPadQuantum = 1024;

// Smallest fp32 that is >= the reference's fp64 skip threshold 1e-10 (methods.cpp:24): for any fp32
// r2, (r2 < kR2SkipF) == ((double)r2 < 1e-10).   0x2edbe6ff = 1.00000001335e-10f.
constexpr float kR2SkipF = 1.0e-10f;
static_assert((double)kR2SkipF >= 1e-10, "fp32 threshold must not round below the fp64 one");

// How one force launch walks the exchange buffer  pos_all[n_shards][dim][pad] / mass_all[n_shards][pad].
struct AccelLaunch {
    const float* pos_all;
    const float* mass_all;
    float* acc;          // [splits][dim][pad] partial accelerations of the target shard
    unsigned pad;        // bodies per chunk (multiple of kPadQuantum)
    int tgt_chunk;       // chunk whose bodies are the targets
    int chunk_first;     // first real chunk of the virtual source list
    int vchunks;         // number of chunks in the virtual source list
    int chunk_skip;      // real chunk left out of the list (INT_MAX: none)
    int splits;          // gridDim.y: slices of the virtual tile list
    int accumulate;      // 0: acc = result, 1: acc += result
    int variant;         // force-kernel variant id (see force_kernel.hip)
};

// Kernel-side view of one force launch (built by launch_accel from an AccelLaunch).
struct KArgs {
    const float* __restrict__ pos_all;
    const float* __restrict__ mass_all;
    float* __restrict__ acc;
    unsigned pad;
    unsigned tiles_per_chunk;
    unsigned total_tiles;      // vchunks * tiles_per_chunk
    unsigned tiles_per_split;
    int tgt_chunk, chunk_first, chunk_skip;
    int accumulate;
};

struct KernelVariant {
    const char* name;
    int tpl;              // targets per lane
    void (*k2)(KArgs);    // D = 2
    void (*k3)(KArgs);    // D = 3
};
// force_kernel.hip, compiled once per code-generation flavour
const KernelVariant* variants_slp(int* count);
const KernelVariant* variants_scalar(int* count);

// force_launch.hip
hipError_t launch_accel(int dim, const AccelLaunch& a, hipStream_t stream);
int num_variants();
const char* variant_name(int variant);
int variant_tpl(int variant);
int default_variant();

// state_kernels.hip
struct PackArgs {
    const double* raw;     // staged Body<D> array, n_total bodies
    size_t stride_d;       // doubles between consecutive bodies
    size_t n_total;
    size_t shard_len;      // bodies per shard (last shard may hold fewer)
    unsigned pad;
    int n_shards, shard, dim;
    float* pos_all;        // [n_shards][dim][pad]
    float* mass_all;       // [n_shards][pad]
    double* x64;           // [dim][pad]   own shard
    double* v64;           // [dim][pad]
    double* m64;           // [pad]
};
hipError_t launch_pack(const PackArgs& p, hipStream_t stream);

struct KickDriftArgs {
    const float* acc;      // [splits][dim][pad]
    int splits, dim;
    unsigned pad;
    size_t count;          // real bodies in this shard
    double G, dt;
    double* x64; double* v64; const double* m64;
    float* pos_chunk;      // this shard's chunk of pos_all: [dim][pad]
};
hipError_t launch_kick_drift(const KickDriftArgs& k, hipStream_t stream);

// forces_out: AoS double[count][dim] on the device
hipError_t launch_export_forces(const float* acc, int splits, int dim, unsigned pad, size_t count,
                                double G, const double* m64, double* forces_out, hipStream_t stream);
// accel_out: SoA float[dim][count] on the device (splits summed in fp64, rounded once)
hipError_t launch_export_accel(const float* acc, int splits, int dim, unsigned pad, size_t count,
                               float* accel_out, hipStream_t stream);
// state_out: AoS double[count][2*dim] = position then velocity
hipError_t launch_export_state(const double* x64, const double* v64, int dim, unsigned pad, size_t count,
                               double* state_out, hipStream_t stream);

}  // namespace nbx
#endif
