// state_kernels.hip -- K3/K4: layout pack/unpack at the boundary and the fused kick+drift step.
//
// All O(N), HBM-streaming kernels; one lane per body, SoA arrays so every access is coalesced.
// The integrator state (position, velocity, mass of the own shard) is kept in fp64 and advanced
// with exactly the reference's arithmetic (nbody-sim-new/methods.cpp:425-450); only the pair sum
// feeding it is fp32.  Each drift also refreshes the shard's fp32 chunk of the exchange buffer.
#include "nbx_internal.h"

namespace nbx {
namespace {

// Sum the per-slice partial accelerations of body l, component k, in slice order (deterministic).
__device__ __forceinline__ double sum_partials(const float* __restrict__ acc, int splits, int dim,
                                               unsigned pad, int k, size_t l) {
    double a = 0.0;
#pragma unroll 8   // eight plane loads in flight (the additions stay in plane order): 64 planes at N = 65,536 are a latency chain otherwise
    for (int s = 0; s < splits; ++s) a += (double)acc[((size_t)s * dim + k) * pad + l];
    return a;
}

// K4: AoS fp64 Body<D> (body.h:8-11) -> SoA fp32 exchange buffers for every shard + fp64 state of
// the own shard.  Pad entries become massless bodies at the origin.  The same pass collects what the host needs to
// know about the bodies before it picks a force kernel (p.facts): a million-body scan on one host thread cost 4 ms.
__global__ __launch_bounds__(256) void pack_kernel(PackArgs p) {
    __shared__ unsigned long long s_mass, s_coord;
    __shared__ unsigned s_close;
    if (threadIdx.x == 0) { s_mass = 0ull; s_coord = 0ull; s_close = 0u; }
    __syncthreads();
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;  // index into [n_shards][pad], or into the own chunk's [pad]
    if (b < (p.only_own ? (size_t)p.pad : (size_t)p.n_shards * p.pad)) {
        const int g = p.only_own ? p.shard : (int)(b / p.pad);
        const size_t l = p.only_own ? b : b - (size_t)g * p.pad;
        const size_t id = p.only_own ? l : (size_t)g * p.shard_len + l;          // index into raw
        const bool real = p.only_own ? (l < p.n_own) : ((l < p.shard_len) && (id < p.n_total));
        const double* __restrict__ src = p.raw + id * p.stride_d;
        double cmin = 0.0, cmax = 0.0;
        for (int k = 0; k < p.dim; ++k) {
            const double x = real ? src[k] : 0.0;
            p.pos_all[((size_t)g * p.dim + k) * p.pad + l] = (float)x;
            if (g == p.shard) {
                p.x64[(size_t)k * p.pad + l] = x;
                p.v64[(size_t)k * p.pad + l] = real ? src[p.dim + k] : 0.0;
            }
            const double ax = fabs(x);
            cmin = (k == 0 || ax < cmin) ? ax : cmin;
            cmax = (cmax != cmax) ? cmax : (!(ax <= cmax) ? ax : cmax);   // sticky NaN: once seen, a later finite coordinate does not replace it
        }
        const double m = real ? src[2 * p.dim] : 0.0;
        p.mass_all[(size_t)g * p.pad + l] = (float)m;
        if (g == p.shard) p.m64[l] = m;
        // non-negative doubles order like their bit patterns, and a NaN's pattern lies above infinity's: maxima over the
        // wave by butterfly shuffles, then one LDS atomic per wave (one per lane made this pass 4x longer)
        unsigned long long mb = real ? (unsigned long long)__double_as_longlong(fabs(m)) : 0ull;
        unsigned long long cb = real ? (unsigned long long)__double_as_longlong(cmax) : 0ull;
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long om = __shfl_xor(mb, off), oc = __shfl_xor(cb, off);
            mb = om > mb ? om : mb;
            cb = oc > cb ? oc : cb;
        }
        const unsigned long long votes = __ballot(real && g == p.shard && cmin < (double)kCloseCoord);
        if ((threadIdx.x & 63u) == 0u) {
            if (mb) atomicMax(&s_mass, mb);
            if (cb) atomicMax(&s_coord, cb);
            if (votes) atomicAdd(&s_close, (unsigned)__popcll(votes));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_mass > *(volatile unsigned long long*)&p.facts[0]) atomicMax(&p.facts[0], s_mass);     // same economy at L2
        if (s_coord > *(volatile unsigned long long*)&p.facts[1]) atomicMax(&p.facts[1], s_coord);
        if (s_close) atomicAdd(&p.facts[2], (unsigned long long)s_close);
    }
}

// K3: fused kick + drift (methods.cpp:436 then :448), fp64:
//   F = -(G m) a ;  v += (F / m) * dt ;  x += v * dt ;  pos32 = (float)x
// One lane per (body, component): grid.y = component.  The components are independent (v_k += (F_k/m) dt; x_k += v_k dt),
// so this is the same arithmetic in the same order as a loop over k in one lane -- with dim times the lanes in flight,
// which is what a latency-bound kernel wants at small N (one lane per body was 25 us at N = 65,536: S = 32 dependent-address
// loads per component and one workgroup per CU).
__global__ __launch_bounds__(256) void kick_drift_kernel(KickDriftArgs a) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= a.count) return;
    const int k = (int)blockIdx.y;
    const double m = a.m64[l];
    const double gm = a.G * m;
    const double acc = sum_partials(a.acc, a.splits, a.dim, a.pad, k, l);
    const double F = -(gm * acc);
    double v = a.v64[(size_t)k * a.pad + l];
    double x = a.x64[(size_t)k * a.pad + l];
    v += (F / m) * a.dt_kick;
    x += v * a.dt_drift;
    a.v64[(size_t)k * a.pad + l] = v;
    a.x64[(size_t)k * a.pad + l] = x;
    a.pos_chunk[(size_t)k * a.pad + l] = (float)x;
}

// The leaf plan's stepping loop: the same two helpers fed the near-field sums of the tree codes (fp64, per padded slot).
__global__ __launch_bounds__(256) void kick_drift_slots_kernel(SlotKickArgs a) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= a.count) return;
    const int k = (int)blockIdx.y;
    const double m = a.m64[l];
    const uint32_t slot = a.body_slot[l];
    const double sum = slot == 0xffffffffu ? 0.0 : a.sums[(size_t)k * a.pslots + slot];
    const double F = (a.signedG * m) * sum;
    double v = a.v64[(size_t)k * a.pad + l];
    double x = a.x64[(size_t)k * a.pad + l];
    v += (F / m) * a.dt;
    x += v * a.dt;
    a.v64[(size_t)k * a.pad + l] = v;
    a.x64[(size_t)k * a.pad + l] = x;
    a.pos_chunk[(size_t)k * a.pad + l] = (float)x;
}

__global__ __launch_bounds__(256) void export_forces_kernel(const float* __restrict__ acc, int splits, int dim,
                                                            unsigned pad, size_t count, double G,
                                                            const double* __restrict__ m64,
                                                            double* __restrict__ out) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= count) return;
    const double gm = G * m64[l];
    for (int k = 0; k < dim; ++k) out[l * dim + k] = -(gm * sum_partials(acc, splits, dim, pad, k, l));
}

__global__ __launch_bounds__(256) void export_accel_kernel(const float* __restrict__ acc, int splits, int dim,
                                                           unsigned pad, size_t count, float* __restrict__ out) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= count) return;
    for (int k = 0; k < dim; ++k) out[(size_t)k * count + l] = (float)sum_partials(acc, splits, dim, pad, k, l);
}

__global__ __launch_bounds__(256) void export_state_kernel(const double* __restrict__ x64,
                                                           const double* __restrict__ v64, int dim,
                                                           unsigned pad, size_t count, double* __restrict__ out) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= count) return;
    for (int k = 0; k < dim; ++k) {
        out[l * 2 * dim + k] = x64[(size_t)k * pad + l];
        out[l * 2 * dim + dim + k] = v64[(size_t)k * pad + l];
    }
}

// Per-body energies for the reference law: kinetic m v^2 / 2 and potential (G m / 4) * phi,
// phi = sum over slices of potential_kernel's output (fp64, slice order).
// Energy reduction: per-body kinetic m v^2 / 2 and potential (G m / 4) * sum_slices phi, summed on the device --
// wave64 butterfly over the lanes (__shfl_down on fp64 = two DPP/permute moves per step, 6 steps), the four wave sums of
// a workgroup through LDS in wave order, one {kinetic, potential} pair per workgroup in energy_out[2][blocks].  The host
// adds the per-workgroup pairs in block order: a fixed summation tree, bit-reproducible, and 16 B per 256 bodies cross
// PCIe instead of 16 B per body.
__global__ __launch_bounds__(256) void energy_partials_kernel(const float* __restrict__ phi, int splits, int dim,
                                                              unsigned pad, size_t count, double G,
                                                              const double* __restrict__ v64,
                                                              const double* __restrict__ m64, double* __restrict__ out) {
    __shared__ double wave_ke[4], wave_pe[4];
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    double ke = 0.0, pe = 0.0;
    if (l < count) {
        double p = 0.0;
        for (int s = 0; s < splits; ++s) p += (double)phi[(size_t)s * pad + l];
        double v2 = 0.0;
        for (int k = 0; k < dim; ++k) { const double v = v64[(size_t)k * pad + l]; v2 += v * v; }
        const double m = m64[l];
        ke = 0.5 * m * v2;
        pe = 0.25 * G * m * p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ke += __shfl_down(ke, off, 64);
        pe += __shfl_down(pe, off, 64);
    }
    if ((threadIdx.x & 63u) == 0) { wave_ke[threadIdx.x >> 6] = ke; wave_pe[threadIdx.x >> 6] = pe; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = ((wave_ke[0] + wave_ke[1]) + wave_ke[2]) + wave_ke[3];
        out[gridDim.x + blockIdx.x] = ((wave_pe[0] + wave_pe[1]) + wave_pe[2]) + wave_pe[3];
    }
}

// The reference's accuracy metric (nbody-sim-new/utils.h:170-219) on the device: a body counts as accurate
// when every force component is within 1 % of the reference's; components with |ref| < 1e-20 are held to
// |f| <= 1e-9 instead.  F = -(G m) a is formed exactly as export_forces_kernel does.
__global__ __launch_bounds__(256) void accuracy_kernel(const float* __restrict__ acc, int splits, int dim, unsigned pad,
                                                       size_t count, double G, const double* __restrict__ m64,
                                                       const double* __restrict__ ref, unsigned* __restrict__ accurate) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool ok = l < count;
    if (ok) {
        const double gm = G * m64[l];
        for (int k = 0; k < dim && ok; ++k) {
            const double f = -(gm * sum_partials(acc, splits, dim, pad, k, l));
            const double r = ref[l * dim + k];
            if (fabs(r) < 1e-20) ok = !(fabs(f) > 1e-9);
            else ok = !(fabs((f - r) / r) > 0.01);
        }
    }
    const unsigned long long votes = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && votes) atomicAdd(accurate, (unsigned)__popcll(votes));
}

__global__ __launch_bounds__(256) void export_aux_kernel(const float* __restrict__ aux, int planes, unsigned pad, size_t count,
                                                         double* __restrict__ out) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= count) return;
    double v = 0.0;
    for (int s = 0; s < planes; ++s) v += (double)aux[(size_t)s * pad + l];
    out[l] = v;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

hipError_t launch_pack(const PackArgs& p, hipStream_t stream) {
    const size_t total = p.only_own ? (size_t)p.pad : (size_t)p.n_shards * p.pad;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_kick_drift(const KickDriftArgs& k, hipStream_t stream) {
    if (k.count == 0) return hipSuccess;
    hipLaunchKernelGGL(kick_drift_kernel, dim3(blocks_for(k.count), (unsigned)k.dim, 1), dim3(256), 0, stream, k);
    return hipGetLastError();
}

hipError_t launch_kick_drift_slots(const SlotKickArgs& k, hipStream_t stream) {
    if (k.count == 0) return hipSuccess;
    hipLaunchKernelGGL(kick_drift_slots_kernel, dim3(blocks_for(k.count), (unsigned)k.dim, 1), dim3(256), 0, stream, k);
    return hipGetLastError();
}

hipError_t launch_export_forces(const float* acc, int splits, int dim, unsigned pad, size_t count, double G,
                                const double* m64, double* forces_out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(export_forces_kernel, dim3(blocks_for(count)), dim3(256), 0, stream, acc, splits, dim, pad,
                       count, G, m64, forces_out);
    return hipGetLastError();
}

hipError_t launch_export_accel(const float* acc, int splits, int dim, unsigned pad, size_t count,
                               float* accel_out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(export_accel_kernel, dim3(blocks_for(count)), dim3(256), 0, stream, acc, splits, dim, pad,
                       count, accel_out);
    return hipGetLastError();
}

hipError_t launch_export_energy(const float* phi, int splits, int dim, unsigned pad, size_t count, double G,
                                const double* v64, const double* m64, double* energy_out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(energy_partials_kernel, dim3(blocks_for(count)), dim3(256), 0, stream, phi, splits, dim, pad, count,
                       G, v64, m64, energy_out);
    return hipGetLastError();
}

hipError_t launch_export_aux(const float* aux, int planes, unsigned pad, size_t count, double* out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(export_aux_kernel, dim3(blocks_for(count)), dim3(256), 0, stream, aux, planes, pad, count, out);
    return hipGetLastError();
}

hipError_t launch_accuracy(const float* acc, int splits, int dim, unsigned pad, size_t count, double G,
                           const double* m64, const double* ref_forces, unsigned* accurate, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(accurate, 0, sizeof(unsigned), stream);
    if (e != hipSuccess || count == 0) return e;
    hipLaunchKernelGGL(accuracy_kernel, dim3(blocks_for(count)), dim3(256), 0, stream, acc, splits, dim, pad, count, G,
                       m64, ref_forces, accurate);
    return hipGetLastError();
}

hipError_t launch_export_state(const double* x64, const double* v64, int dim, unsigned pad, size_t count,
                               double* state_out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(export_state_kernel, dim3(blocks_for(count)), dim3(256), 0, stream, x64, v64, dim, pad,
                       count, state_out);
    return hipGetLastError();
}

}  // namespace nbx
