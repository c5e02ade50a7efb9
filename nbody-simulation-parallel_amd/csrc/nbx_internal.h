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
// (up to 16 targets per lane x 256 lanes) sees whole target blocks and whole source tiles.
constexpr int kPadQuantum = 4096;

// Smallest fp32 that is >= the reference's fp64 skip threshold 1e-10 (methods.cpp:24): for any fp32
// r2, (r2 < kR2SkipF) == ((double)r2 < 1e-10).   0x2edbe6ff = 1.00000001335e-10f.
constexpr float kR2SkipF = 1.0e-10f;
static_assert((double)kR2SkipF >= 1e-10, "fp32 threshold must not round below the fp64 one");
constexpr double kR2Skip64 = 1.0e-10;   // the same threshold for the strict fp64 kernel, compared as the reference compares it

// ---- close-pair bookkeeping of the fast force path (force_kernel.hip) --------------------------------
// The fast kernel carries no per-pair guard: it biases r^2 by kTiny = 2^-47 so that coincident bodies
// (d = 0) contribute exactly 0 and 1/r^2 stays finite.  That reproduces the reference for a target unless
// it has a source with 0 < r^2 < X, where X = max(1e-10 [the reference's skip threshold, methods.cpp:24],
// 2^25 * kTiny = 2.4e-7 [below this the bias moves r^2 by more than half a unit roundoff]) = 2.4e-7.
// Targets that might own such a pair are found exactly, once per position update, in two cheap steps:
//  1. candidates.  Two DISTINCT fp32 coordinates a, b of equal sign differ by at least the fp32 spacing at
//     min(|a|,|b|) (opposite signs: by |a|+|b|).  A pair with 0 < r^2 < 2.4e-7 has some |d_k| in (0, 4.9e-4),
//     hence min(|a_k|,|b_k|) < 2^13 and max < 2^13 + 4.9e-4: BOTH members have a coordinate of magnitude below
//     kCloseCoord = 2^14 (twice what the argument needs).  classify_close_kernel lists the shard's targets with
//     any |coordinate| < kCloseCoord (0.5 % of the reference's uniform bodies).  Conversely a non-candidate
//     differs from every other body, in each coordinate where they differ, by >= 2^-10, so its non-zero r^2
//     are >= 9.5e-7: never skipped by the reference and biased by < 1e-8 relative.
//  2. The same argument holds for the SOURCE member of such a pair, whichever chunk (shard) it lives in, so
//     classify_sources_kernel lists the candidate sources of the pass being launched -- every real body of
//     the pass's chunk list (ALL / LOCAL / REMOTE) with a coordinate below kCloseCoord, read from the exchange
//     buffer as it stands when the pass runs (for REMOTE: after the all-gather) -- and refine_close_kernel
//     checks the shard's candidate targets against that list (all-pairs over two small lists, the same fp32
//     r^2) and keeps the targets with a partner at 0 < r^2 < kBadR2 = 1e-6 (4x the X above) -- almost
//     always none.  Above kRefineLimit candidate targets (or kRefinePairLimit checks) it keeps them all.
//     A pair that straddles a shard boundary is therefore found by the pass that evaluates it.
// Bad targets of a pass are flagged (the fast kernel does not store them) and evaluated against that pass's
// sources with the exact compare-and-select guard by extra workgroups of the same launch (close_set_path),
// then scattered into acc.  The result therefore has the reference's skip semantics for every pair.
constexpr float kTiny = 0x1p-47f;          // 7.1e-15
constexpr float kCloseCoord = 16384.0f;
constexpr float kBadR2 = 1.0e-6f;
constexpr unsigned kRefineLimit = 131072;  // candidate targets beyond this are all treated as bad (O(n^2) check avoided)
constexpr unsigned long long kRefinePairLimit = 1ull << 36;  // same for (candidate targets) x (candidate sources)
// m / (kTiny^2) must stay finite in fp32 for a coincident source: masses above this force the exact path.
constexpr double kFastMaxMass = 1.0e10;
// One-reciprocal fast variant: W = 1/(r2a*r2b) must stay finite.  With every |coordinate| <= X, r^2 <= 12 X^2 and
// the product <= 144 X^4; X = 1e9 would still fit (1.4e38), 1e8 leaves a 10x margin for bodies that drift after
// the upload-time check.  Violations fall back to the two-reciprocal fast kernel (same results to ~2 ulp).
constexpr double kOneRcpMaxCoord = 1.0e8;
constexpr int kCloseBlocksX = 32;          // extra workgroups per source slice that a fast launch adds for bad targets

// Workspace of the sorted-cell form of the close-set refinement (close_hash.hip); all null when not in use.
struct HashWork {
    unsigned* keys = nullptr;          // [capacity] 32-bit cell keys of the pass's candidate sources; sorted in place (four passes, two buffers)
    unsigned* keys_alt = nullptr;      // [capacity] the radix sort's second buffer
    unsigned* vals = nullptr;          // [capacity] slot in src_cand_pos, carried along
    unsigned* vals_alt = nullptr;
    void* temp = nullptr;              // per-tile digit histograms of the radix sort
    size_t temp_bytes = 0;
    unsigned capacity = 0;             // = n_chunks * pad (every body may be a candidate)
};

// How one force evaluation walks the exchange buffer  pos_all[n_shards][dim][pad] / mass_all[n_shards][pad].
struct AccelLaunch {
    const float* pos_all;
    const float* mass_all;
    float* acc;          // [splits][dim][pad] partial accelerations of the target shard
    unsigned pad;        // bodies per chunk (multiple of kPadQuantum)
    unsigned count;      // real targets in the shard (<= pad)
    int tgt_chunk;       // chunk whose bodies are the targets
    int chunk_first;     // first real chunk of the virtual source list
    int vchunks;         // number of chunks in the virtual source list
    int chunk_skip;      // real chunk left out of the list (INT_MAX: none)
    int splits;          // gridDim.y: slices of the virtual tile list
    int accumulate;      // 0: acc = result, 1: acc += result
    int variant;         // force-kernel variant id
    // workspace of the fast path (may be null for exact variants)
    unsigned* cand_list;      // [pad] candidate target indices
    float* cand_pos;          // [dim][pad] their positions, compacted
    unsigned* bad_list;       // [pad] targets that own a pair with 0 < r^2 < kBadR2
    unsigned* bad_flag;  // [pad] 1 for listed targets
    unsigned* counters;       // [0] = candidate targets, [1] = bad targets, [2] = candidate sources
    float* close_acc;         // [splits][dim][pad]
    float* src_cand_pos;      // [dim][n_shards*pad] positions of the pass's candidate sources, compacted
    size_t n_total;           // real bodies over all chunks
    size_t shard_len;         // real bodies per chunk (the last non-empty chunk may hold fewer)
    int n_chunks;             // chunks in the exchange buffer (= n_shards)
    int pass;                 // NBX_SRC_* selector of this launch (cache key of the bad-target list)
    int cacheable;            // the pass's sources change only through this context (single shard, or LOCAL)
    int* tgt_cand_valid;      // host flag owned by the context: cand_list matches the own chunk's positions
    int* bad_list_pass;       // host flag owned by the context: pass whose bad list is current (-1: none)
    float eps2;               // > 0: softened law (fast variants only; no close-set pipeline)
    int law;                  // 0: the reference's r^-4 d law, 1: Newtonian r^-3 d (needs eps2 > 0)
    HashWork hash;            // non-null: refine through sorted cells instead of candidates x candidates
    int lists_only;           // 1: build the close-set lists and return (upload-time probe of the bad-target count)
    // mixed mode (nbx_ctx_set_refine; see RefineLaunch): non-null = the fast kernel also writes, per slice and target, the sum
    // over the slice's tiles of |tile partial sum|^2  -- [grid slices][pad]
    float* qsum = nullptr;
    // optional: recorded on the stream immediately before / after the main force kernel
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // measurement (nbx_ctx_enable_clock_stamps): non-null = every workgroup of the three-level force kernel stamps the shader
    // clock and the 100 MHz reference clock around its work -- [2 x workgroups of the launch] {d s_memtime, d s_memrealtime}
    unsigned long long* clk = nullptr;
    unsigned* clk_slots = nullptr;   // host: receives the number of workgroups of the stamped launch
};

// Kernel-side view of one force launch (built by launch_accel from an AccelLaunch).
struct KArgs {
    const float* __restrict__ pos_all;
    const float* __restrict__ mass_all;
    float* __restrict__ acc;
    unsigned pad;
    unsigned count;
    unsigned tiles_per_chunk;
    unsigned total_tiles;      // vchunks * tiles_per_chunk
    unsigned tiles_per_split;
    int tgt_chunk, chunk_first, chunk_skip;
    int accumulate;
    int splits;                // fp32 planes of acc (= grid slices x planes per slice)
    int grid_slices;           // source slices of the launch = planes of close_acc and of qsum
    unsigned close_blocks;     // fast kernels: workgroups with blockIdx.x < close_blocks run close_set_path
    unsigned* __restrict__ cand_list;
    float* __restrict__ cand_pos;
    unsigned* __restrict__ bad_list;
    unsigned* __restrict__ bad_flag;
    unsigned* __restrict__ counters;
    float* __restrict__ close_acc;
    float* __restrict__ src_cand_pos;
    unsigned src_stride;       // floats between the coordinate planes of src_cand_pos (= n_chunks * pad)
    unsigned n_total, shard_len;
    float eps2;                // softened law: epsilon^2 (0: the reference law)
    // mixed mode / strict kernel with listed targets
    float* __restrict__ qsum;          // [grid slices][pad] or null
    unsigned* __restrict__ strict_list;   // [strict_cap] targets to re-evaluate in fp64; their number is counters[3]
    double* __restrict__ strict_acc;   // [slices][dim][stride] with slices and stride chosen ON THE DEVICE from the list's length (strict_layout)
    unsigned strict_cap;               // = the shard's pad: every target can be listed, nothing overflows
    int strict_slices;                 // most source slices the fp64 pass may use (<= 256, <= total_tiles) = its gridDim.y
    unsigned long long strict_budget;  // doubles in strict_acc: slices x dim x stride never exceeds it
    double refine_c2;                  // a target is listed when |a|^2 < refine_c2 * Q  (Q = sum over slices of qsum)
    unsigned long long* __restrict__ clk;   // null, or [2 x gridDim.x x gridDim.y] clock stamps (AccelLaunch::clk)
};

struct KernelVariant {
    const char* name;
    int tpl;              // targets per lane
    void (*k2)(KArgs);    // D = 2
    void (*k3)(KArgs);    // D = 3
    int fast;             // 1: unguarded fast kernel + close-set pipeline (exact overall)
    int max_tiles_per_slice;  // > 0: the kernel's fp32 second-level sums want at most this many tiles per slice
    int needs_extent;     // 1: only valid while every |coordinate| <= kOneRcpMaxCoord (one-reciprocal kernel)
    void (*soft2)(KArgs); // softened-law build of the same kernel (null: none), D = 2 / 3
    void (*soft3)(KArgs);
    void (*newton2)(KArgs);  // softened Newtonian law (two-reciprocal form of the same kernel)
    void (*newton3)(KArgs);
    void (*qs2)(KArgs);      // the same fast kernel also writing KArgs::qsum (mixed mode; null: none)
    void (*qs3)(KArgs);
    int planes;              // fp32 planes of acc written per source slice: 1, or 2 = {hi, lo} of an fp64 sum (strict kernel)
    int aux;                 // 1: the kernel itself writes KArgs::qsum (the strict kernel's magnitude-sum build)
    int stamps;              // 1: the kernel writes KArgs::clk when it is non-null (measurement)
};
// force_kernel.hip
const KernelVariant* kernel_variants(int* count);
struct CloseKernels { void (*classify[2])(KArgs); void (*classify_src[2])(KArgs); void (*refine[2])(KArgs); void (*scatter[2])(KArgs); void (*potential[2])(KArgs); void (*potential_soft[2])(KArgs); void (*potential_newton[2])(KArgs);
                      void (*refine_select[2])(KArgs); void (*strict_list[2])(KArgs); void (*refine_fold[2])(KArgs); };  // [0]: D=2, [1]: D=3
CloseKernels close_kernels();

// close_hash.hip
size_t hash_temp_bytes(unsigned capacity);
hipError_t hash_refine(int dim, const KArgs& a, const HashWork& h, hipStream_t stream);

// force_launch.hip
hipError_t launch_accel(int dim, const AccelLaunch& a, hipStream_t stream);
// Mixed mode, after the fast kernel has produced acc (all planes) and qsum (all slices) for the FULL source set:
//   select: a target is a suspect when its acceleration is small against the spread of its tile partial sums,
//           |a|^2 < c2 * sum_tiles |a_tile|^2 (or it is a close-set target) -> strict_list, count in counters[3];
//   strict: accel_f64_kernel<LIST> over ALL chunks, the listed targets x strict_slices source slices, fp64 throughout;
//   fold:   the slices added in fp64, the target's planes rewritten (plane 0 = hi, plane 1 = lo, the rest 0).
struct RefineLaunch {
    AccelLaunch base;          // pos/mass/acc/pad/count/tgt_chunk/n_chunks/counters/bad_flag/qsum of the evaluation to refine
    unsigned* strict_list;
    double* strict_acc;
    unsigned strict_cap;
    int strict_slices;
    unsigned long long strict_budget;
    double c2;
    int grid_slices;           // slices of the fast launch (planes of qsum)
};
hipError_t launch_refine(int dim, const RefineLaunch& R, hipStream_t stream);
// phi[splits][pad] = sum_j m_j / r^2 over ALL chunks (L.acc points at the phi buffer; L.splits slices)
hipError_t launch_potential(int dim, const AccelLaunch& a, hipStream_t stream);
int num_variants();
const char* variant_name(int variant);
int variant_tpl(int variant);
int variant_is_fast(int variant);
int variant_max_tiles_per_slice(int variant);
int variant_needs_extent(int variant);
int variant_has_law_builds(int variant);  // softened / Newtonian builds of the kernel exist (the A/B table entries have none)
int variant_planes(int variant);          // fp32 planes of acc per source slice (2: strict fp64 kernel)
int variant_has_qsum(int variant);        // the variant has a build that writes qsum (mixed mode possible)
int variant_writes_aux(int variant);      // the variant's own kernel writes qsum (strict magnitude-sum build)
int variant_has_clock_stamps(int variant);   // the variant's kernel honours AccelLaunch::clk (the three-level kernel)
// demangled symbol of the kernel launch_accel launches for (variant, dim, law, softened, mixed mode) -- the name rocprofv3 prints
int variant_kernel_symbol(int variant, int dim, int law, int soft, int qsum, char* buf, size_t len);
int default_fast_two_rcp_variant();   // fast variant without the extent precondition
int variant_by_name(const char* name);
int default_variant();        // the fast default
int default_exact_variant();  // used when the fast path's preconditions do not hold

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
    int only_own;          // 1: `raw` holds the own shard's bodies only (n_own of them) and only the own chunk is packed
    size_t n_own;
    unsigned long long* facts;  // [3], zeroed by the caller: bit patterns of max |mass| and max |coordinate| over all bodies
                                // (NaN sorts above everything), and the own shard's count of close-set candidates
};
hipError_t launch_pack(const PackArgs& p, hipStream_t stream);

struct KickDriftArgs {
    const float* acc;      // [splits][dim][pad]
    int splits, dim;
    unsigned pad;
    size_t count;          // real bodies in this shard
    double G, dt_kick, dt_drift;   // v += (F/m) * dt_kick;  x += v * dt_drift  (both = dt: the reference's kick then drift)
    double* x64; double* v64; const double* m64;
    float* pos_chunk;      // this shard's chunk of pos_all: [dim][pad]
};
hipError_t launch_kick_drift(const KickDriftArgs& k, hipStream_t stream);
// The same update from fp64 leaf sums indexed by padded slot (leaf plan, leaf_pair_kernel.hip): body l's sum is
// sums[k][body_slot[l]] (0 when body_slot[l] == 0xffffffff: the body belongs to no leaf); F = (signedG m) sum.
struct SlotKickArgs {
    const double* sums;        // [dim][pslots]
    const uint32_t* body_slot; // [count]
    uint32_t pslots;
    int dim;
    unsigned pad;
    size_t count;
    double signedG, dt;
    double* x64; double* v64; const double* m64;
    float* pos_chunk;
};
hipError_t launch_kick_drift_slots(const SlotKickArgs& k, hipStream_t stream);

// forces_out: AoS double[count][dim] on the device
hipError_t launch_export_forces(const float* acc, int splits, int dim, unsigned pad, size_t count,
                                double G, const double* m64, double* forces_out, hipStream_t stream);
// accel_out: SoA float[dim][count] on the device (splits summed in fp64, rounded once)
hipError_t launch_export_accel(const float* acc, int splits, int dim, unsigned pad, size_t count,
                               float* accel_out, hipStream_t stream);
// energy_out: double[2][ceil(count/256)]: per-workgroup sums (wave-level reduction on the device) of the bodies' kinetic
// m v^2 / 2 and potential (G m / 4) * sum_slices phi
hipError_t launch_export_energy(const float* phi, int splits, int dim, unsigned pad, size_t count, double G,
                                const double* v64, const double* m64, double* energy_out, hipStream_t stream);
// *accurate = number of bodies whose force is within the reference's 1 % rule of ref_forces (device AoS double[count][dim])
// out[count] (device, double) = sum over the aux planes of aux[plane][i], in plane order
hipError_t launch_export_aux(const float* aux, int planes, unsigned pad, size_t count, double* out, hipStream_t stream);
hipError_t launch_accuracy(const float* acc, int splits, int dim, unsigned pad, size_t count, double G,
                           const double* m64, const double* ref_forces, unsigned* accurate, hipStream_t stream);
// state_out: AoS double[count][2*dim] = position then velocity
hipError_t launch_export_state(const double* x64, const double* v64, int dim, unsigned pad, size_t count,
                               double* state_out, hipStream_t stream);

}  // namespace nbx
#endif
