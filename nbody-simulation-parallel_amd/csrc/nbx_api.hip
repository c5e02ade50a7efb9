// nbx_api.hip -- the C ABI of include/nbody_hip.h: context management, boundary conversions,
// stream/event plumbing.  No compute happens on the host and there is no CPU fallback: every
// entry point that needs the device fails loudly when HIP does.
#include "../../include/nbody_hip.h"
#include "nbx_internal.h"
#include "nbx_ctx.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

using namespace nbx;

namespace {

thread_local std::string g_detail;
}  // namespace

namespace nbx {
int fail_hip(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%s) at %s:%d", what, hipGetErrorString(e), hipGetErrorName(e), file, line);
    g_detail = buf;
    (void)hipGetLastError();  // reset the runtime's sticky last-error so a later launch check does not see this one
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? NBX_ERR_NO_DEVICE
           : (e == hipErrorOutOfMemory)                           ? NBX_ERR_ALLOC
                                                                  : NBX_ERR_HIP;
}
int fail(int code, const char* msg) {
    g_detail = msg;
    return code;
}

// Streams are the expensive part of a context on this runtime: hipStreamCreateWithFlags 1.4 ms, hipStreamDestroy 1.2-3 ms
// (tools/ubench/api_cost.hip, profiles/r2/api_cost.txt) -- more than a whole force evaluation below N ~ 50,000, and the
// one-shot entry points make and destroy a context per call.  A destroyed context parks its idle stream here; the next
// context on that device takes it.  Parked streams live until the process ends.
namespace {
std::mutex g_stream_pool_mu;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;
constexpr size_t kStreamPoolMax = 16;
}  // namespace

hipError_t take_stream(int device, hipStream_t* out) {
    {
        std::lock_guard<std::mutex> lock(g_stream_pool_mu);
        for (size_t i = 0; i < g_stream_pool.size(); ++i)
            if (g_stream_pool[i].first == device) {
                *out = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void park_stream(int device, hipStream_t s) {   // s is idle: the caller has synchronised it
    {
        std::lock_guard<std::mutex> lock(g_stream_pool_mu);
        if (g_stream_pool.size() < kStreamPoolMax) { g_stream_pool.emplace_back(device, s); return; }
    }
    (void)hipStreamDestroy(s);
}

// The same for a context's device allocation: hipFree of a few MB costs 0.2 ms on this runtime, hipMalloc not much less, and the
// drop-in call makes and destroys a context per call -- at N = 1,000 that was half of the call.  A destroyed context parks its
// arena (up to kArenaParkMaxBytes; two per device); the next context on that device that fits takes it.  Nothing in the library
// relies on fresh memory being zero (hipMalloc does not promise it either).
namespace {
struct ParkedCtxArena { int device; char* p; size_t bytes; };
std::vector<ParkedCtxArena> g_ctx_arenas;   // under g_stream_pool_mu
// 1 GiB: since the mixed mode became the default a context's block carries the fp64 pass's sums (8 x dim x pad doubles: 201 MB at
// N = 2^20, 805 MB at 2^22), and a one-shot call per step at such sizes would otherwise hipMalloc and hipFree hundreds of MB each
// time -- the cost the parking exists to avoid.  288 GB of HBM do not miss two idle blocks; nbx_release_cached() frees them.
constexpr size_t kArenaParkMaxBytes = (size_t)1 << 30;
constexpr size_t kArenasParkedPerDevice = 2;
}  // namespace

hipError_t take_ctx_arena(int device, size_t bytes, char** out, size_t* got) {
    {
        std::lock_guard<std::mutex> lock(g_stream_pool_mu);
        size_t best = g_ctx_arenas.size();
        for (size_t i = 0; i < g_ctx_arenas.size(); ++i)
            if (g_ctx_arenas[i].device == device && g_ctx_arenas[i].bytes >= bytes && g_ctx_arenas[i].bytes <= 4 * bytes + (1u << 20) &&
                (best == g_ctx_arenas.size() || g_ctx_arenas[i].bytes < g_ctx_arenas[best].bytes)) best = i;
        if (best != g_ctx_arenas.size()) {
            *out = g_ctx_arenas[best].p;
            *got = g_ctx_arenas[best].bytes;
            g_ctx_arenas.erase(g_ctx_arenas.begin() + (long)best);
            return hipSuccess;
        }
    }
    *got = bytes;
    void* a = nullptr;
    hipError_t e = hipMalloc(&a, bytes);
    if (e == hipErrorOutOfMemory) {   // parked allocations (here and in the leaf path) may be what is in the way
        (void)hipGetLastError();
        release_parked_ctx_arenas();
        release_parked_leaf_arenas();
        e = hipMalloc(&a, bytes);
    }
    *out = static_cast<char*>(a);
    return e;
}

void park_ctx_arena(int device, char* p, size_t bytes) {   // nothing on the device uses p any more
    char* evicted = nullptr;
    if (bytes <= kArenaParkMaxBytes) {
        std::lock_guard<std::mutex> lock(g_stream_pool_mu);
        g_ctx_arenas.push_back(ParkedCtxArena{device, p, bytes});
        size_t mine = 0, oldest = g_ctx_arenas.size();
        for (size_t i = 0; i < g_ctx_arenas.size(); ++i)
            if (g_ctx_arenas[i].device == device) { if (oldest == g_ctx_arenas.size()) oldest = i; ++mine; }
        if (mine > kArenasParkedPerDevice) { evicted = g_ctx_arenas[oldest].p; g_ctx_arenas.erase(g_ctx_arenas.begin() + (long)oldest); }
    } else {
        evicted = p;
    }
    if (evicted) (void)hipFree(evicted);
}

void release_parked_ctx_arenas() {
    std::vector<ParkedCtxArena> parked;
    {
        std::lock_guard<std::mutex> lock(g_stream_pool_mu);
        parked.swap(g_ctx_arenas);
    }
    int before = 0;
    const bool have = hipGetDevice(&before) == hipSuccess;
    for (auto& e : parked)
        if (hipSetDevice(e.device) == hipSuccess) (void)hipFree(e.p);
    if (have) (void)hipSetDevice(before);
    (void)hipGetLastError();
}

// the ids of the live contexts (a plan's last evaluation names its context by id: the arena of a destroyed context is parked and
// handed to the next one, so an address says nothing)
namespace {
std::mutex g_ctx_mu;
std::vector<unsigned long long> g_live_ctx;
std::atomic<unsigned long long> g_next_ctx_id{1};
}  // namespace
bool ctx_alive(unsigned long long id) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    for (unsigned long long v : g_live_ctx) if (v == id) return true;
    return false;
}

void release_parked_streams() {
    std::vector<std::pair<int, hipStream_t>> parked;
    {
        std::lock_guard<std::mutex> lock(g_stream_pool_mu);
        parked.swap(g_stream_pool);
    }
    int before = 0;
    const bool have = hipGetDevice(&before) == hipSuccess;
    for (auto& e : parked)
        if (hipSetDevice(e.first) == hipSuccess) (void)hipStreamDestroy(e.second);
    if (have) (void)hipSetDevice(before);
    (void)hipGetLastError();
}
}  // namespace nbx

namespace {

#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) return fail_hip(e_, #expr, __FILE__, __LINE__); \
    } while (0)

constexpr int kEventPairs = 512;
constexpr int kMaxSplits = 256;
constexpr int kPhiSlices = 16;
constexpr int kGraphMinSteps = 4;   // nbx_ctx_step replays a captured step from this many steps on
// Mixed mode: selection rule |a|^2 < (sigma u / tol)^2 Q (force_kernel.hip).  The sigma factor is a calibration of the DEFAULT
// kernel's summation (round 4: the three-level kernel, whose rounding errors are ~3x smaller than the two-level kernel's at the same
// spread Q -- the factor a body NEEDS, error / (u sqrt(Q)), peaks at 17-19 over 1-4 million bodies where it peaked at 62), taken
// from all-bodies surveys against the strict kernel on ten inputs (profiles/r4/all_bodies_3l.jsonl): uniform
// 3D N = 2^20 (three seeds) and 2^22, Plummer 2^22, 64 Gaussian clumps, a lattice in index order with random and with equal
// masses, uniform 2D N = 2^20 and 2^22.  With 24 (3D) no body of any 3D input is left above 5.8e-6 (tolerance 1e-5), with 32 (2D)
// none above 4.7e-6; the next smaller factors tried (20 / 24) leave 6.7e-6 / 6.3e-6.  Round 3's factors for the two-level kernel
// were 48 / 64 (7.6e-6 to 9.8e-6 left): the two-level variant, when selected by name, still gets those.
constexpr double kRefineSigmaDefault3D = 24.0;
constexpr double kRefineSigmaDefault2D = 32.0;
constexpr double kRefineSigmaTwoLevel3D = 48.0;
constexpr double kRefineSigmaTwoLevel2D = 64.0;
constexpr double kUnitRoundoffF32 = 0x1p-24;
// Precision default of new contexts (nbx_set_default_refine): the north star's tolerance for every body
std::atomic<double> g_default_refine_tol{1.0e-5};
std::atomic<double> g_default_refine_sigma{0.0};

bool refine_args_ok(double rel_tolerance, double sigma_factor, const char** why) {
    if (!(rel_tolerance == 0.0 || (rel_tolerance >= 1.0e-7 && rel_tolerance <= 1.0e-2))) { *why = "refine tolerance must be 0 (off) or in [1e-7, 1e-2]"; return false; }
    if (!(sigma_factor >= 0.0 && sigma_factor <= 1.0e6)) { *why = "sigma factor must be 0 (default) or in (0, 1e6]"; return false; }
    return true;
}

}  // namespace


namespace {

constexpr size_t kArenaAlign = 256;
size_t arena_round(size_t bytes) { return ((bytes ? bytes : 8) + kArenaAlign - 1) / kArenaAlign * kArenaAlign; }

// Device memory of a context: from the creation-time arena while it lasts, otherwise an allocation of its own.
template <typename T>
int dev_alloc(nbx_ctx* c, T** p, size_t bytes) {
    bytes = arena_round(bytes);
    if (c->arena && c->arena_used + bytes <= c->arena_bytes) {
        *p = reinterpret_cast<T*>(c->arena + c->arena_used);
        c->arena_used += bytes;
        return NBX_OK;
    }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e == hipErrorOutOfMemory) {   // parked allocations (the leaf path's: up to 2 x 2 GiB per device; destroyed contexts') may be what is in the way
        (void)hipGetLastError();
        release_parked_leaf_arenas();
        release_parked_ctx_arenas();
        e = hipMalloc(&q, bytes);
    }
    HIP_TRY(e);
    c->extra.push_back(q);
    *p = static_cast<T*>(q);
    return NBX_OK;
}

// Give a buffer back before the context goes: its own allocation is freed, a piece of the arena stays with the arena
// (only the last piece handed out can be returned to it).
template <typename T>
int dev_release(nbx_ctx* c, T*& p, size_t bytes) {
    if (!p) return NBX_OK;
    char* q = reinterpret_cast<char*>(p);
    p = nullptr;
    if (c->arena && q >= c->arena && q < c->arena + c->arena_bytes) {
        if (q + arena_round(bytes) == c->arena + c->arena_used) c->arena_used -= arena_round(bytes);
        return NBX_OK;
    }
    for (size_t i = 0; i < c->extra.size(); ++i)
        if (c->extra[i] == q) { c->extra.erase(c->extra.begin() + (long)i); break; }
    HIP_TRY(hipFree(q));
    return NBX_OK;
}

int ensure_stage(nbx_ctx* c, size_t bytes) {
    if (bytes <= c->stage_bytes) return NBX_OK;
    int rc = dev_release(c, c->stage, c->stage_bytes);
    c->stage_bytes = 0;
    if (rc) return rc;
    rc = dev_alloc(c, &c->stage, bytes);
    if (rc) return rc;
    c->stage_bytes = bytes;
    return NBX_OK;
}

// The variant actually launched: the caller's choice (or the default), demoted to the exact default when
// the fast path's preconditions do not hold for this context.
int effective_variant(const nbx_ctx* c) {
    int v = c->variant_req >= 0 ? c->variant_req : default_variant();
    if (c->softening > 0.0) {  // softened law: the fast kernels with bias = eps^2; no close set, no guarded twin
        if (!variant_is_fast(v) || !variant_has_law_builds(v))   // A/B table entries and the three-level kernel carry no soft build
            v = variant_has_law_builds(default_variant()) ? default_variant() : default_fast_two_rcp_variant();
        if (variant_needs_extent(v) && !c->extent_ok) v = default_fast_two_rcp_variant();
        return v;
    }
    if (variant_is_fast(v) && c->force_exact) v = default_exact_variant();
    else if (variant_needs_extent(v) && !c->extent_ok) v = default_fast_two_rcp_variant();
    if (c->refine_tol > 0.0 && variant_is_fast(v) && !variant_has_qsum(v)) v = default_fast_two_rcp_variant();
    return v;
}

// Mixed mode runs when it is switched on, the law is the reference's and the kernel is a fast one (the guarded and the
// strict kernels have nothing to refine: the first keeps fp64 second-level sums but fp32 terms -- it is a fallback, not a
// precision mode -- the second is fp64 already).
bool refine_active(const nbx_ctx* c, int variant) {
    return c->refine_tol > 0.0 && !(c->softening > 0.0) && c->law == 0 && variant_is_fast(variant) && variant_has_qsum(variant);
}

// Source slices: enough workgroups that the launch is many "waves" of workgroups deep (a grid that just
// fills the chip once turns every extra workgroup into a full-length tail), and no more tiles per
// slice than the kernel's fp32 second-level sums want.
int auto_splits(const nbx_ctx* c, int variant) {
    const unsigned tgt_blocks = c->pad / (256u * (unsigned)variant_tpl(variant));
    const unsigned want_blocks = (unsigned)c->num_cus * 16u;  // >= 4 rounds of 4 workgroups per CU
    unsigned s = (want_blocks + tgt_blocks - 1) / tgt_blocks;
    const unsigned tiles = (unsigned)c->n_shards * (c->pad / kTile);
    // a slice should keep >= 8 tiles (2048 sources) so its prologue/epilogue stays small; only when that
    // leaves the chip under-filled (< 4 workgroups per CU) are slices cut down to 2 tiles
    unsigned max_s = tiles / 8 ? tiles / 8 : 1;
    const unsigned fill = ((unsigned)c->num_cus * 4u + tgt_blocks - 1) / tgt_blocks;
    if (max_s < fill) max_s = fill < (tiles / 2 ? tiles / 2 : 1) ? fill : (tiles / 2 ? tiles / 2 : 1);
    if (s > max_s) s = max_s;
    const int cap = variant_max_tiles_per_slice(variant);
    if (cap > 0) {
        const unsigned need = (tiles + (unsigned)cap - 1) / (unsigned)cap;
        if (s < need) s = need;
    }
    if (s < 1) s = 1;
    if (s > (unsigned)kMaxSplits) s = (unsigned)kMaxSplits;
    return (int)s;
}

// Source slices of the next launch of `variant`: the caller's count or the automatic one, raised to what the kernel's
// fp32 second-level sums need, and such that slices x planes per slice stays within kMaxSplits planes.
int slices_for(const nbx_ctx* c, int variant) {
    int s = c->splits_user ? c->user_slices : auto_splits(c, variant);
    if (const int cap = variant_max_tiles_per_slice(variant)) {  // a caller's slice count is a lower bound
        const unsigned tiles = (unsigned)c->n_shards * (c->pad / kTile);
        const int need = (int)((tiles + (unsigned)cap - 1) / (unsigned)cap);
        if (s < need) s = need;
    }
    const int planes = variant_planes(variant);
    if (s * planes > kMaxSplits) s = kMaxSplits / planes;
    return s < 1 ? 1 : s;
}

// Mixed mode's fp64 pass: a list with room for EVERY target of the shard, and sums for up to 256 source slices of a list of
// typical length (1/64 of the shard: the rule lists well under a percent of uniform or Plummer 3D bodies, a few percent in
// 2D) -- at least the whole shard in eight slices (201 MB at N = 2^20).  The device picks slices x stride within this budget from the list's actual
// length (force_kernel.hip strict_layout): a long list costs time, never accuracy.
void strict_sizes(nbx_ctx* c, size_t* list_bytes, size_t* acc_bytes) {
    c->strict_cap = c->pad;
    const unsigned tiles = (unsigned)c->n_shards * (c->pad / kTile);
    c->strict_slices = tiles < 256u ? (int)tiles : 256;
    size_t typical = ((size_t)c->pad / 64 + 255) / 256 * 256;
    if (typical < 256) typical = 256;
    size_t per_comp = typical * (size_t)c->strict_slices;
    const size_t whole = (size_t)c->pad * (c->strict_slices < 8 ? (size_t)c->strict_slices : 8);   // every target listed: still 8 slices x 32 rows of workgroups
    if (per_comp < whole) per_comp = whole;
    c->strict_budget = (unsigned long long)c->dim * per_comp;
    *list_bytes = (size_t)c->strict_cap * sizeof(unsigned);
    *acc_bytes = (size_t)c->strict_budget * sizeof(double);
}

int ensure_acc(nbx_ctx* c) {
    c->variant = effective_variant(c);
    const int slices = slices_for(c, c->variant);
    c->splits = slices * variant_planes(c->variant);
    // Partial sums cost 12 B x slices x pad, and the slice count grows with the source count (<= 256 tiles per slice): a
    // single shard of more than ~16 M bodies would ask for > 50 GB here.  Say so instead of failing in hipMalloc.
    if ((size_t)c->splits * c->dim * c->pad * sizeof(float) > ((size_t)48 << 30))
        return fail(NBX_ERR_ALLOC, "shard too large for the per-slice partial sums (12 B x slices x bodies > 48 GiB): split the bodies over more shards");
    int rc = NBX_OK;
    if (!(c->acc && c->acc_splits_alloc >= c->splits)) {
        if ((rc = dev_release(c, c->acc, (size_t)c->acc_splits_alloc * c->dim * c->pad * sizeof(float)))) return rc;
        if ((rc = dev_alloc(c, &c->acc, (size_t)c->splits * c->dim * c->pad * sizeof(float)))) return rc;
        c->acc_splits_alloc = c->splits;
    }
    if (variant_is_fast(c->variant)) {
        if (!c->cand_list) {
            if ((rc = dev_alloc(c, &c->cand_list, (size_t)c->pad * sizeof(unsigned)))) return rc;
            if ((rc = dev_alloc(c, &c->cand_pos, (size_t)c->dim * c->pad * sizeof(float)))) return rc;
            if ((rc = dev_alloc(c, &c->bad_list, (size_t)c->pad * sizeof(unsigned)))) return rc;
            if ((rc = dev_alloc(c, &c->bad_flag, (size_t)c->pad * sizeof(unsigned)))) return rc;
            if ((rc = dev_alloc(c, &c->counters, 4 * sizeof(unsigned)))) return rc;
            if ((rc = dev_alloc(c, &c->src_cand_pos, (size_t)c->dim * c->n_shards * c->pad * sizeof(float)))) return rc;
            c->tgt_cand_valid = 0; c->bad_list_pass = -1;
        }
        if (c->hash_refine && !c->hash.keys) {
            const unsigned cap = (unsigned)c->n_shards * c->pad;
            c->hash.capacity = cap;
            c->hash.temp_bytes = hash_temp_bytes(cap);
            if ((rc = dev_alloc(c, &c->hash.keys, (size_t)cap * sizeof(unsigned)))) return rc;
            if ((rc = dev_alloc(c, &c->hash.keys_alt, (size_t)cap * sizeof(unsigned)))) return rc;
            if ((rc = dev_alloc(c, &c->hash.vals, (size_t)cap * sizeof(unsigned)))) return rc;
            if ((rc = dev_alloc(c, &c->hash.vals_alt, (size_t)cap * sizeof(unsigned)))) return rc;
            char* temp = nullptr;
            if ((rc = dev_alloc(c, &temp, c->hash.temp_bytes))) return rc;
            c->hash.temp = temp;
        }
        if (!(c->close_acc && c->close_splits_alloc >= c->splits)) {
            if ((rc = dev_release(c, c->close_acc, (size_t)c->close_splits_alloc * c->dim * c->pad * sizeof(float)))) return rc;
            if ((rc = dev_alloc(c, &c->close_acc, (size_t)c->splits * c->dim * c->pad * sizeof(float)))) return rc;
            c->close_splits_alloc = c->splits;
        }
    }
    if (refine_active(c, c->variant) || variant_writes_aux(c->variant)) {   // aux planes: one per source slice
        if (!(c->qsum && c->qsum_slices_alloc >= c->splits)) {
            if ((rc = dev_release(c, c->qsum, (size_t)c->qsum_slices_alloc * c->pad * sizeof(float)))) return rc;
            if ((rc = dev_alloc(c, &c->qsum, (size_t)c->splits * c->pad * sizeof(float)))) return rc;
            c->qsum_slices_alloc = c->splits;
        }
        if (refine_active(c, c->variant) && !c->strict_list) {
            size_t list_bytes = 0, acc_bytes = 0;
            strict_sizes(c, &list_bytes, &acc_bytes);
            if ((rc = dev_alloc(c, &c->strict_list, list_bytes))) return rc;
            if ((rc = dev_alloc(c, &c->strict_acc, acc_bytes))) return rc;
        }
    }
    return NBX_OK;
}

// The refinement of the evaluation that just finished (mixed mode): see RefineLaunch.
RefineLaunch refine_launch(const nbx_ctx* c) {
    RefineLaunch R = {};
    R.base.pos_all = c->pos_all; R.base.mass_all = c->mass_all; R.base.acc = c->acc; R.base.pad = c->pad;
    R.base.count = (unsigned)c->count; R.base.tgt_chunk = c->shard; R.base.splits = c->splits; R.base.n_chunks = c->n_shards;
    R.base.counters = c->counters; R.base.bad_flag = c->bad_flag; R.base.qsum = c->qsum;
    R.strict_list = c->strict_list; R.strict_acc = c->strict_acc; R.strict_cap = c->strict_cap; R.strict_slices = c->strict_slices;
    R.strict_budget = c->strict_budget;
    const bool three_level = variant_planes(c->variant) == 2;   // of the fast kernels only the three-level one writes {hi, lo} planes
    const double sigma = c->refine_sigma > 0.0 ? c->refine_sigma
                         : three_level ? (c->dim == 2 ? kRefineSigmaDefault2D : kRefineSigmaDefault3D)
                                       : (c->dim == 2 ? kRefineSigmaTwoLevel2D : kRefineSigmaTwoLevel3D);
    const double r = sigma * kUnitRoundoffF32 / c->refine_tol;
    R.c2 = r * r;
    R.grid_slices = c->splits / variant_planes(c->variant);
    return R;
}

constexpr int kPollEverySteps = 16;

// Long runs: the refinement mode and the 1/8 bad-target limit are decided at upload, but bodies move.  Every
// kPollEverySteps steps the three close-set counters are copied to pinned host memory asynchronously; whenever a copy has
// landed (event query, never a wait) the host re-evaluates: many candidates -> sorted cells, few again -> candidates x
// candidates, too many targets that really own a close pair -> the guarded kernel.
int poll_close_counters(nbx_ctx* c, int steps) {
    if (c->capturing || c->softening > 0.0 || c->force_exact || !c->counters || !variant_is_fast(c->variant)) return NBX_OK;
    if (c->counters_pending && hipEventQuery(c->counters_ev) == hipSuccess) {
        c->counters_pending = false;
        c->last_cand = c->counters_host[0];
        c->last_bad = c->counters_host[1];
        if ((size_t)c->last_bad * 8 > c->count) { c->force_exact = true; c->hash_refine = false; }
        else if (!c->hash_refine && (size_t)c->last_cand * 8 > c->count) c->hash_refine = true;
        else if (c->hash_refine && (size_t)c->last_cand * 16 < c->count) c->hash_refine = false;   // hysteresis
    }
    (void)hipGetLastError();   // hipEventQuery leaves hipErrorNotReady behind
    c->steps_since_poll += steps;
    if (!c->counters_pending && c->steps_since_poll >= kPollEverySteps) {
        if (!c->counters_host) {
            HIP_TRY(hipHostMalloc((void**)&c->counters_host, 4 * sizeof(unsigned), hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&c->counters_ev, hipEventDisableTiming));
        }
        HIP_TRY(hipMemcpyAsync(c->counters_host, c->counters, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->counters_ev, c->stream));
        c->counters_pending = true;
        c->steps_since_poll = 0;
    }
    return NBX_OK;
}

int set_device(const nbx_ctx* c) {
    (void)hipGetLastError();  // launches are checked with hipGetLastError(): start from a clean slate
    HIP_TRY(hipSetDevice(c->device));
    return NBX_OK;
}

}  // namespace

namespace nbx {

// Upload, first half: the caller's Body<D> array (or, only_own, just this shard's slice of it) to the staging buffer, the
// pack pass, and what that pass learnt about the bodies (facts: PackArgs::facts).  With only_own the other chunks of the
// exchange buffers are NOT filled: the node layer brings them in from the ranks that own them (nbx_node.hip), so that R
// ranks move the array over the host links once instead of R times.
int upload_stage(nbx_ctx* c, const void* bodies, size_t stride_bytes, bool only_own, unsigned long long facts[3], bool bodies_is_own_slice) {
    if (!c || (!bodies && c->n_total)) return fail(NBX_ERR_INVALID, "null argument");
    const size_t min_stride = (size_t)(2 * c->dim + 1) * sizeof(double);
    if (stride_bytes < min_stride || stride_bytes % sizeof(double) != 0)
        return fail(NBX_ERR_INVALID, "body stride must be a multiple of 8 and >= sizeof(Body<dim>)");
    int rc = set_device(c);
    if (rc) return rc;
    if (!c->pos_all) {
        if ((rc = dev_alloc(c, &c->pos_all, (size_t)c->n_shards * c->dim * c->pad * sizeof(float)))) return rc;
        if ((rc = dev_alloc(c, &c->mass_all, (size_t)c->n_shards * c->pad * sizeof(float)))) return rc;
        c->own_gather = true;
    }
    const size_t first = (only_own && !bodies_is_own_slice) ? (size_t)c->shard * c->shard_len : 0;   // offset into the caller's array
    const size_t nb = only_own ? c->count : c->n_total;
    const size_t bytes = nb * stride_bytes;
    rc = ensure_stage(c, bytes ? bytes : 8);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(c->stage, static_cast<const char*>(bodies) + first * stride_bytes, bytes, hipMemcpyHostToDevice, c->stream));
    PackArgs p;
    p.raw = c->stage; p.stride_d = stride_bytes / sizeof(double); p.n_total = c->n_total;
    p.shard_len = c->shard_len; p.pad = c->pad; p.n_shards = c->n_shards; p.shard = c->shard; p.dim = c->dim;
    p.pos_all = c->pos_all; p.mass_all = c->mass_all; p.x64 = c->x64; p.v64 = c->v64; p.m64 = c->m64;
    p.only_own = only_own ? 1 : 0; p.n_own = c->count;
    p.facts = c->facts;
    HIP_TRY(hipMemsetAsync(c->facts, 0, 3 * sizeof(unsigned long long), c->stream));
    HIP_TRY(launch_pack(p, c->stream));
    HIP_TRY(hipMemcpyAsync(facts, c->facts, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the caller's host array is borrowed only for this call
    return NBX_OK;
}

// Upload, second half: the preconditions of the fast (unguarded) force path from the facts -- facts[0], facts[1] over ALL
// bodies (the node layer combines its ranks' values), facts[2] this shard's candidate count -- and, for a small-coordinate
// system, the probe.  Every chunk of the exchange buffers must be in place by now.
int upload_finish(nbx_ctx* c, const unsigned long long facts[3]) {
    int rc = set_device(c);
    if (rc) return rc;
    {
        double mmax, cmax;
        std::memcpy(&mmax, &facts[0], sizeof mmax);
        std::memcpy(&cmax, &facts[1], sizeof cmax);
        const size_t close = (size_t)facts[2];
        c->mass_max = mmax;
        c->force_exact = !(mmax <= kFastMaxMass);   // every mass small enough for the kTiny bias (a NaN fails the comparison)
        // More than 1/8 of the shard in the candidate set (a small-coordinate system): the candidates x candidates check
        // would be O(N^2); the fast path then refines through sorted cells, provided the probe below finds few enough
        // targets that really own a close pair.
        c->hash_refine = !c->force_exact && close * 8 > c->count;
        c->extent_ok = cmax <= kOneRcpMaxCoord;   // the one-reciprocal kernel's product r2a*r2b stays finite
    }
    c->uploaded = true;
    c->have_accel = false;
    c->tgt_cand_valid = 0; c->bad_list_pass = -1;
    c->probe_bad = 0;
    c->counters_pending = false; c->steps_since_poll = 0; c->last_cand = c->last_bad = 0;
    if (c->hash_refine) {
        // Probe: build the close-set lists once against ALL sources and read the number of bad targets back (this call
        // synchronises anyway).  Too many (> 1/8 of the shard: the guarded side path would dominate) -> guarded kernel.
        rc = ensure_acc(c);
        if (rc) return rc;
        if (variant_is_fast(c->variant)) {
            AccelLaunch L = {};
            L.pos_all = c->pos_all; L.mass_all = c->mass_all; L.acc = c->acc; L.pad = c->pad; L.count = (unsigned)c->count;
            L.tgt_chunk = c->shard; L.splits = c->splits; L.variant = c->variant;
            L.cand_list = c->cand_list; L.cand_pos = c->cand_pos; L.bad_list = c->bad_list; L.bad_flag = c->bad_flag;
            L.counters = c->counters; L.close_acc = c->close_acc; L.src_cand_pos = c->src_cand_pos;
            L.n_total = c->n_total; L.shard_len = c->shard_len; L.n_chunks = c->n_shards;
            L.pass = NBX_SRC_ALL; L.cacheable = 0; L.tgt_cand_valid = &c->tgt_cand_valid; L.bad_list_pass = &c->bad_list_pass;
            L.chunk_skip = INT_MAX; L.chunk_first = 0; L.vchunks = c->n_shards; L.hash = c->hash; L.lists_only = 1;
            HIP_TRY(launch_accel(c->dim, L, c->stream));
            unsigned counts[3] = {0, 0, 0};
            HIP_TRY(hipMemcpyAsync(counts, c->counters, sizeof counts, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->probe_bad = counts[1];
            c->tgt_cand_valid = 0; c->bad_list_pass = -1;
            if ((size_t)counts[1] * 8 > c->count) { c->force_exact = true; c->hash_refine = false; }
        }
    }
    return NBX_OK;
}

}  // namespace nbx

extern "C" {

int nbx_abi_version(void) { return NBX_ABI_VERSION; }

const char* nbx_strerror(int status) {
    switch (status) {
        case NBX_OK: return "ok";
        case NBX_ERR_INVALID: return "invalid argument";
        case NBX_ERR_NO_DEVICE: return "no usable HIP device (libnbody_hip has no CPU fallback)";
        case NBX_ERR_HIP: return "HIP runtime error";
        case NBX_ERR_ALLOC: return "allocation failed";
        case NBX_ERR_STATE: return "call made in the wrong state";
        default: return "unknown nbx status";
    }
}

const char* nbx_last_error_detail(void) { return g_detail.c_str(); }

int nbx_num_variants(void) { return num_variants(); }
const char* nbx_variant_name(int variant) { return variant_name(variant); }
int nbx_default_variant(void) { return default_variant(); }

int nbx_device_count(int* count) {
    if (!count) return fail(NBX_ERR_INVALID, "count is null");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail_hip(e, "hipGetDeviceCount", __FILE__, __LINE__);
    *count = n;
    return n > 0 ? NBX_OK : fail(NBX_ERR_NO_DEVICE, "hipGetDeviceCount returned 0 devices");
}

int nbx_set_default_refine(double rel_tolerance, double sigma_factor) {
    const char* why = "";
    if (!refine_args_ok(rel_tolerance, sigma_factor, &why)) return fail(NBX_ERR_INVALID, why);
    g_default_refine_tol.store(rel_tolerance);
    g_default_refine_sigma.store(sigma_factor);
    return NBX_OK;
}

int nbx_get_default_refine(double* rel_tolerance, double* sigma_factor) {
    if (rel_tolerance) *rel_tolerance = g_default_refine_tol.load();
    if (sigma_factor) *sigma_factor = g_default_refine_sigma.load();
    return NBX_OK;
}

double nbx_refine_sigma_default(int dim) { return dim == 2 ? kRefineSigmaDefault2D : kRefineSigmaDefault3D; }

int nbx_release_cached(void) {
    release_parked_communicators();
    release_parked_leaf_arenas();
    release_parked_ctx_arenas();
    release_parked_streams();
    return NBX_OK;
}

int nbx_warmup(int device) {
    int ndev = 0;
    int rc = nbx_device_count(&ndev);
    if (rc != NBX_OK) return rc;
    if (device < 0 || device >= ndev) return fail(NBX_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));  // forces runtime + context creation
    // A small evaluation loads the code object and touches every kernel of the default path -- and is big enough
    // (8,192 bodies: 448 KiB in, 192 KiB out) that the runtime sets up its host<->device copy machinery now: the first
    // copy of more than a few KiB costs ~15 ms on this runtime (profiles/r2/api_cost.txt), which a 1-body call never meets.
    constexpr size_t n = 8192;
    std::vector<double> b(n * 7), f(n * 3);
    for (size_t i = 0; i < n; ++i) {
        double* r = &b[i * 7];
        r[0] = 1.0e5 + 37.0 * (double)(i % 97); r[1] = 2.0e5 + 11.0 * (double)(i / 97); r[2] = 3.0e5 + (double)i;
        r[3] = r[4] = r[5] = 0.0; r[6] = 1.0;
    }
    return nbx_brute_force_forces(b.data(), n, 3, 7 * sizeof(double), NBX_REFERENCE_G, device, f.data(), nullptr);
}

int nbx_ctx_create(nbx_ctx** out, int device, int dim, size_t n_total, int n_shards, int shard) {
    if (!out) return fail(NBX_ERR_INVALID, "out is null");
    *out = nullptr;
    if (dim != 2 && dim != 3) return fail(NBX_ERR_INVALID, "dim must be 2 or 3");
    if (n_shards < 1 || shard < 0 || shard >= n_shards) return fail(NBX_ERR_INVALID, "bad shard / n_shards");
    if (n_total > (size_t)1 << 31) return fail(NBX_ERR_INVALID, "n_total too large");
    int ndev = 0;
    int rc = nbx_device_count(&ndev);
    if (rc != NBX_OK) return rc;
    if (device < 0 || device >= ndev) return fail(NBX_ERR_NO_DEVICE, "device ordinal out of range");
    nbx_ctx* c = new (std::nothrow) nbx_ctx();
    if (c) {
        c->id = g_next_ctx_id.fetch_add(1);
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        try { g_live_ctx.push_back(c->id); } catch (...) { delete c; c = nullptr; }
    }
    if (!c) return fail(NBX_ERR_ALLOC, "host allocation failed");
    c->device = device; c->dim = dim; c->n_total = n_total; c->n_shards = n_shards; c->shard = shard;
    c->shard_len = (n_total + (size_t)n_shards - 1) / (size_t)n_shards;
    const size_t lo = (size_t)shard * c->shard_len;
    c->count = lo >= n_total ? 0 : ((n_total - lo < c->shard_len) ? n_total - lo : c->shard_len);
    size_t pad = (c->shard_len + kPadQuantum - 1) / kPadQuantum * kPadQuantum;
    if (pad == 0) pad = kPadQuantum;
    c->pad = (unsigned)pad;
    c->variant = default_variant();
    c->variant_req = -1;
    c->refine_tol = g_default_refine_tol.load();       // mixed mode unless the process default says plain fp32
    c->refine_sigma = g_default_refine_sigma.load();
    { const char* e = std::getenv("NBODY_HIP_NO_GRAPHS"); c->no_graphs = e && *e && *e != '0'; }
#define CTX_TRY(expr)                                                            \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) { int r_ = fail_hip(e_, #expr, __FILE__, __LINE__); nbx_ctx_destroy(c); return r_; } \
    } while (0)
    CTX_TRY(hipSetDevice(device));
    int cus = 0;
    CTX_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    c->num_cus = cus > 0 ? cus : 256;
    CTX_TRY(take_stream(device, &c->own_stream));
    c->stream = c->own_stream;
    {
        // One allocation for what a default run of this shape will ask for (dev_alloc): the fp64 state, the boundary staging
        // buffer at Body<dim>'s own stride, the partial sums of the default variant at its automatic slice count (twice:
        // the close-set side path has its own), the close-set lists, and -- for an unsharded context, whose exchange
        // buffers are always its own -- the fp32 source arrays.  Anything else falls through to its own allocation.
        const size_t body = (size_t)(2 * dim + 1) * sizeof(double);
        const int splits = auto_splits(c, c->variant);
        size_t want = 2 * arena_round((size_t)dim * pad * sizeof(double)) + arena_round(pad * sizeof(double));
        want += arena_round(n_total * body + 16) + arena_round(3 * sizeof(unsigned long long));
        want += 2 * arena_round((size_t)splits * variant_planes(c->variant) * dim * pad * sizeof(float));   // acc and the close-set side path's
        want += 3 * arena_round(pad * sizeof(unsigned)) + arena_round((size_t)dim * pad * sizeof(float)) + arena_round(16)
                + arena_round((size_t)dim * n_shards * pad * sizeof(float));
        if (n_shards == 1) want += arena_round((size_t)dim * pad * sizeof(float)) + arena_round(pad * sizeof(float));
        if (c->refine_tol > 0.0) {   // mixed mode (the default): spread sums per slice, the list, the fp64 pass's sums
            size_t list_bytes = 0, acc_bytes = 0;
            strict_sizes(c, &list_bytes, &acc_bytes);
            want += arena_round((size_t)splits * variant_planes(c->variant) * pad * sizeof(float)) + arena_round(list_bytes) + arena_round(acc_bytes);
        }
        if (want <= ((size_t)64 << 30)) {   // beyond that the pieces are allocated one by one, and fail one by one
            CTX_TRY(take_ctx_arena(device, want, &c->arena, &c->arena_bytes));   // a parked one of a destroyed context, or a fresh one
        }
    }
#define CTX_ALLOC(ptr, bytes) do { int r_ = dev_alloc(c, &(ptr), (bytes)); if (r_) { nbx_ctx_destroy(c); return r_; } } while (0)
    CTX_ALLOC(c->x64, (size_t)dim * pad * sizeof(double));
    CTX_ALLOC(c->v64, (size_t)dim * pad * sizeof(double));
    CTX_ALLOC(c->m64, pad * sizeof(double));
    CTX_ALLOC(c->facts, 3 * sizeof(unsigned long long));
#undef CTX_ALLOC
    c->ev0.assign(kEventPairs, nullptr);  // event pairs are created on first use
    c->ev1.assign(kEventPairs, nullptr);
    c->ev2.assign(kEventPairs, nullptr);  // third mark of a mixed-mode evaluation: after its select / fp64 / fold kernels
    c->ev2_set.assign(kEventPairs, 0);
#undef CTX_TRY
    *out = c;
    return NBX_OK;
}

int nbx_ctx_destroy(nbx_ctx* c) {
    if (!c) return NBX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : c->extra) (void)hipFree(p);   // every device buffer is a piece of the arena or one of these
    if (c->arena) {   // the stream(s) were synchronised above: nothing on the device uses it any more
        const bool idle = (!c->stream || hipStreamSynchronize(c->stream) == hipSuccess) && (!c->own_stream || hipStreamSynchronize(c->own_stream) == hipSuccess);
        if (idle) park_ctx_arena(c->device, c->arena, c->arena_bytes);
        else (void)hipFree(c->arena);
    }
    if (c->counters_host) (void)hipHostFree(c->counters_host);
    if (c->counters_ev) (void)hipEventDestroy(c->counters_ev);
    if (c->step_exec) (void)hipGraphExecDestroy(c->step_exec);
    if (c->bulk0) (void)hipEventDestroy(c->bulk0);
    if (c->bulk1) (void)hipEventDestroy(c->bulk1);
    for (auto e : c->ev0) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev1) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev2) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) {
        if (hipStreamSynchronize(c->own_stream) == hipSuccess) park_stream(c->device, c->own_stream);
        else (void)hipStreamDestroy(c->own_stream);
    }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        for (size_t i = 0; i < g_live_ctx.size(); ++i)
            if (g_live_ctx[i] == c->id) { g_live_ctx[i] = g_live_ctx.back(); g_live_ctx.pop_back(); break; }
    }
    delete c;
    return NBX_OK;
}

int nbx_ctx_set_stream(nbx_ctx* c, void* hip_stream) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return NBX_OK;
}

int nbx_ctx_set_gather_buffers(nbx_ctx* c, void* pos_all, void* mass_all) {
    if (!c || !pos_all || !mass_all) return fail(NBX_ERR_INVALID, "null argument");
    if (c->uploaded) return fail(NBX_ERR_STATE, "gather buffers must be set before upload");
    if (c->own_gather) {
        (void)dev_release(c, c->mass_all, (size_t)c->n_shards * c->pad * sizeof(float));
        (void)dev_release(c, c->pos_all, (size_t)c->n_shards * c->dim * c->pad * sizeof(float));
    }
    c->pos_all = (float*)pos_all;
    c->mass_all = (float*)mass_all;
    c->own_gather = false;
    return NBX_OK;
}

int nbx_ctx_gather_layout(const nbx_ctx* c, size_t* shard_len, size_t* shard_pad, void** pos_all, void** mass_all) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (shard_len) *shard_len = c->shard_len;
    if (shard_pad) *shard_pad = c->pad;
    if (pos_all) *pos_all = c->pos_all;
    if (mass_all) *mass_all = c->mass_all;
    return NBX_OK;
}

int nbx_ctx_upload_bodies(nbx_ctx* c, const void* bodies, size_t stride_bytes) {
    unsigned long long facts[3] = {0, 0, 0};
    int rc = upload_stage(c, bodies, stride_bytes, /*only_own=*/false, facts);
    if (rc) return rc;
    return upload_finish(c, facts);
}

int nbx_ctx_upload_shard(nbx_ctx* c, const void* shard_bodies, size_t stride_bytes, double* max_abs_mass, double* max_abs_coord) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (!shard_bodies && c->count) return fail(NBX_ERR_INVALID, "null argument");
    unsigned long long facts[3] = {0, 0, 0};
    c->uploaded = false;
    // a rank whose shard is empty still packs its (all-pad) chunk: massless bodies at the origin
    static const double dummy[8] = {0};
    int rc = upload_stage(c, c->count ? shard_bodies : dummy, stride_bytes, /*only_own=*/true, facts, /*bodies_is_own_slice=*/true);
    if (rc) return rc;
    std::memcpy(c->pending_facts, facts, sizeof facts);
    c->shard_staged = true;
    double m = 0.0, x = 0.0;
    std::memcpy(&m, &facts[0], sizeof m);
    std::memcpy(&x, &facts[1], sizeof x);
    if (max_abs_mass) *max_abs_mass = m;
    if (max_abs_coord) *max_abs_coord = x;
    return NBX_OK;
}

int nbx_ctx_upload_finish(nbx_ctx* c, double max_abs_mass_all, double max_abs_coord_all) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (!c->shard_staged) return fail(NBX_ERR_STATE, "nbx_ctx_upload_shard first");
    unsigned long long facts[3];
    std::memcpy(facts, c->pending_facts, sizeof facts);
    // the fast path's preconditions are properties of ALL bodies: the caller hands in the maxima over every shard (a NaN stays a
    // NaN through a max that is written as !(a <= b) ? ... -- whatever arrives here fails the comparisons it has to fail)
    double m = 0.0, x = 0.0;
    std::memcpy(&m, &facts[0], sizeof m);
    std::memcpy(&x, &facts[1], sizeof x);
    if (!(max_abs_mass_all <= m)) m = max_abs_mass_all;
    if (!(max_abs_coord_all <= x)) x = max_abs_coord_all;
    std::memcpy(&facts[0], &m, sizeof m);
    std::memcpy(&facts[1], &x, sizeof x);
    c->shard_staged = false;
    return upload_finish(c, facts);
}

int nbx_ctx_set_tuning(nbx_ctx* c, int source_splits, int variant) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (source_splits < 0 || source_splits > kMaxSplits) return fail(NBX_ERR_INVALID, "source_splits must be in [0,256]");
    if (variant < -1 || variant >= num_variants()) return fail(NBX_ERR_INVALID, "unknown kernel variant");
    c->splits_user = source_splits > 0;
    c->user_slices = source_splits;
    c->variant_req = variant;
    c->have_accel = false;
    return NBX_OK;
}

int nbx_ctx_set_softening(nbx_ctx* c, double epsilon) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    // eps^2 takes the place of the kTiny bias of r^2 in the fast kernels; below ~1e-6 it would be lost next to fp32
    // pair terms of the reference's coordinate range, and 1/eps^4 must stay far from the fp32 range limits
    if (!(epsilon == 0.0 || (epsilon >= 1.0e-6 && epsilon <= 1.0e15))) return fail(NBX_ERR_INVALID, "softening must be 0 or in [1e-6, 1e15]");
    c->softening = epsilon;
    c->have_accel = false;
    c->tgt_cand_valid = 0; c->bad_list_pass = -1;
    return NBX_OK;
}

int nbx_ctx_close_set_mode(nbx_ctx* c, int* mode, unsigned* candidates_seen, unsigned* bad_seen) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    const int v = effective_variant(c);
    if (mode) *mode = (c->softening > 0.0) ? NBX_CLOSE_NONE : !variant_is_fast(v) ? NBX_CLOSE_GUARDED_KERNEL
                      : c->hash_refine ? NBX_CLOSE_SORTED_CELLS : NBX_CLOSE_CANDIDATE_PAIRS;
    if (candidates_seen) *candidates_seen = c->last_cand;
    if (bad_seen) *bad_seen = c->last_bad ? c->last_bad : c->probe_bad;
    return NBX_OK;
}

int nbx_ctx_set_law(nbx_ctx* c, int law) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (law != NBX_FORCE_LAW_REFERENCE && law != NBX_FORCE_LAW_NEWTON) return fail(NBX_ERR_INVALID, "unknown force law");
    c->law = law;
    c->have_accel = false;
    c->tgt_cand_valid = 0; c->bad_list_pass = -1;
    return NBX_OK;
}

int nbx_ctx_set_refine(nbx_ctx* c, double rel_tolerance, double sigma_factor) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    const char* why = "";
    if (!refine_args_ok(rel_tolerance, sigma_factor, &why)) return fail(NBX_ERR_INVALID, why);
    c->refine_tol = rel_tolerance;
    c->refine_sigma = sigma_factor;
    c->have_accel = false;
    return NBX_OK;
}

int nbx_ctx_refine_stats(nbx_ctx* c, unsigned* selected, unsigned* refined) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (!c->have_accel || !refine_active(c, c->variant) || !c->qsum) return fail(NBX_ERR_STATE, "no mixed-mode evaluation on the device (nbx_ctx_set_refine, then compute)");
    if (!c->refined) return fail(NBX_ERR_STATE, "the accelerations on the device are a LOCAL pass only");
    int rc = set_device(c);
    if (rc) return rc;
    unsigned n = 0;
    HIP_TRY(hipMemcpyAsync(&n, c->counters + 3, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (selected) *selected = n;
    if (refined) *refined = n;   // the list holds every target of the shard: whatever was selected was re-evaluated
    return NBX_OK;
}

int nbx_ctx_get_aux(nbx_ctx* c, double* out) {
    if (!c || (!out && c->count)) return fail(NBX_ERR_INVALID, "null argument");
    if (!c->have_accel || !c->qsum || !(refine_active(c, c->variant) || variant_writes_aux(c->variant)))
        return fail(NBX_ERR_STATE, "the last force evaluation wrote no per-target statistic (mixed mode or the strict_f64_t4_mag variant)");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t bytes = c->count * sizeof(double);
    rc = ensure_stage(c, bytes ? bytes : 8);
    if (rc) return rc;
    HIP_TRY(launch_export_aux(c->qsum, c->splits / variant_planes(c->variant), c->pad, c->count, c->stage, c->stream));
    if (bytes) HIP_TRY(hipMemcpyAsync(out, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBX_OK;
}

int nbx_ctx_effective_tuning(nbx_ctx* c, int* variant, int* source_splits) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    const int v = effective_variant(c);
    if (variant) *variant = v;
    if (source_splits) *source_splits = slices_for(c, v);
    return NBX_OK;
}

int nbx_variant_kernel_symbol(int variant, int dim, int mixed_mode, char* buf, size_t len) {
    if (!buf || len < 16 || (dim != 2 && dim != 3) || variant < 0 || variant >= num_variants()) return fail(NBX_ERR_INVALID, "bad argument");
    if (mixed_mode && !(variant_is_fast(variant) && variant_has_qsum(variant))) return fail(NBX_ERR_INVALID, "the variant has no mixed-mode build");
    if (variant_kernel_symbol(variant, dim, 0, 0, mixed_mode || variant_writes_aux(variant), buf, len))
        return fail(NBX_ERR_STATE, "the kernel's symbol is not known to the runtime");
    return NBX_OK;
}

int nbx_ctx_enable_clock_stamps(nbx_ctx* c, int on) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "upload bodies first");
    int rc = set_device(c);
    if (rc) return rc;
    if (on && !c->clk) {
        const int v = effective_variant(c);
        if (!variant_has_clock_stamps(v)) return fail(NBX_ERR_STATE, "the force kernel that would run carries no clock stamps (three-level variants only)");
        // room for the largest grid this variant launches on this shard: target blocks + close-set blocks, times the slices
        const size_t cap = ((size_t)c->pad / (256u * (unsigned)variant_tpl(v)) + (size_t)kCloseBlocksX) * (size_t)(kMaxSplits / variant_planes(v));
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, cap * 2 * sizeof(unsigned long long)));
        c->extra.push_back(p);
        c->clk = static_cast<unsigned long long*>(p);
        c->clk_cap = cap;
    }
    c->clk_on = on != 0;
    c->clk_slots = 0;
    return NBX_OK;
}

int nbx_ctx_shader_clock(nbx_ctx* c, double* median_mhz, double* min_mhz, double* max_mhz, int* workgroups) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (!c->clk || !c->clk_slots) return fail(NBX_ERR_STATE, "no stamped force launch yet (nbx_ctx_enable_clock_stamps, then an evaluation)");
    int rc = set_device(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> h;
    try { h.resize((size_t)c->clk_slots * 2); } catch (...) { return fail(NBX_ERR_ALLOC, "host staging allocation failed"); }
    HIP_TRY(hipMemcpy(h.data(), c->clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (unsigned i = 0; i < c->clk_slots; ++i)
        if (h[2 * i + 1] >= 1000) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);   // >= 10 us of work: close-set and empty workgroups drop out
    if (mhz.empty()) return fail(NBX_ERR_STATE, "the stamped launch had no workgroup long enough to read a clock from");
    std::sort(mhz.begin(), mhz.end());
    if (median_mhz) *median_mhz = mhz[mhz.size() / 2];
    if (min_mhz) *min_mhz = mhz.front();
    if (max_mhz) *max_mhz = mhz.back();
    if (workgroups) *workgroups = (int)mhz.size();
    return NBX_OK;
}

int nbx_ctx_compute_accel(nbx_ctx* c, int which) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "upload bodies before computing accelerations");
    if (which < NBX_SRC_ALL || which > NBX_SRC_REMOTE) return fail(NBX_ERR_INVALID, "bad source selector");
    if (which == NBX_SRC_REMOTE && !c->have_accel) return fail(NBX_ERR_STATE, "REMOTE pass needs a preceding LOCAL pass");
    int rc = set_device(c);
    if (rc) return rc;
    if (which != NBX_SRC_REMOTE) { rc = ensure_acc(c); if (rc) return rc; }
    AccelLaunch L;
    L.pos_all = c->pos_all; L.mass_all = c->mass_all; L.acc = c->acc; L.pad = c->pad; L.count = (unsigned)c->count;
    L.tgt_chunk = c->shard; L.splits = c->splits; L.variant = c->variant;
    L.cand_list = c->cand_list; L.cand_pos = c->cand_pos; L.bad_list = c->bad_list; L.bad_flag = c->bad_flag;
    L.counters = c->counters; L.close_acc = c->close_acc; L.src_cand_pos = c->src_cand_pos;
    L.n_total = c->n_total; L.shard_len = c->shard_len; L.n_chunks = c->n_shards;
    L.pass = which; L.cacheable = (c->n_shards == 1 || which == NBX_SRC_LOCAL) ? 1 : 0;
    L.tgt_cand_valid = &c->tgt_cand_valid; L.bad_list_pass = &c->bad_list_pass;
    L.eps2 = (float)(c->softening * c->softening);
    L.law = c->law;
    L.lists_only = 0;
    const bool refine = refine_active(c, c->variant);
    if (c->clk_on && c->clk && variant_has_clock_stamps(c->variant) &&
        ((size_t)c->pad / (256u * (unsigned)variant_tpl(c->variant)) + (size_t)kCloseBlocksX) * (size_t)(c->splits / variant_planes(c->variant)) <= c->clk_cap) {
        L.clk = c->clk; L.clk_slots = &c->clk_slots;
    }
    L.qsum = (refine || variant_writes_aux(c->variant)) ? c->qsum : nullptr;
    if (c->hash_refine && variant_is_fast(c->variant) && !(c->softening > 0.0)) L.hash = c->hash;
    if (c->law != 0 && !(c->softening > 0.0)) return fail(NBX_ERR_STATE, "the Newtonian law needs a softening length (nbx_ctx_set_softening)");
    if (c->softening > 0.0 && !(c->mass_max / ((double)L.eps2 * (double)L.eps2) < 1.0e38))
        return fail(NBX_ERR_INVALID, "softening too small for these masses: m / eps^4 must stay finite in fp32");
    // ... and too LARGE a softening length makes every pair weight underflow to zero in fp32 (the result would be an
    // all-zero force field with status OK): the heaviest body's weight at zero distance must be a normal fp32 number
    if (c->softening > 0.0 && c->mass_max > 0.0 &&
        !(c->mass_max / (c->law ? (double)L.eps2 * c->softening : (double)L.eps2 * (double)L.eps2) > 1.0e-30))
        return fail(NBX_ERR_INVALID, "softening too large for these masses: m / eps^4 (Newtonian law: m / eps^3) underflows in fp32");
    L.chunk_skip = INT_MAX; L.accumulate = 0;
    if (which == NBX_SRC_ALL) { L.chunk_first = 0; L.vchunks = c->n_shards; }
    else if (which == NBX_SRC_LOCAL) { L.chunk_first = c->shard; L.vchunks = 1; }
    else { L.chunk_first = 0; L.vchunks = c->n_shards - 1; L.chunk_skip = c->shard; L.accumulate = 1; }
    if (L.vchunks == 0) return NBX_OK;  // REMOTE with a single shard: nothing to add
    if (!c->capturing && c->ev_used == kEventPairs) {
        // event log full (512 launches since the last nbx_ctx_kernel_time): fold it into the running totals instead of
        // silently dropping later launches.  One stream synchronisation per 512 launches.
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int i = 0; i < c->ev_used; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
            c->bulk_ms_done += ms;
            if (c->ev2_set[i]) { HIP_TRY(hipEventElapsedTime(&ms, c->ev1[i], c->ev2[i])); c->refine_ms_done += ms; }
        }
        c->bulk_steps_done += c->ev_used;
        c->ev_used = 0;
    }
    const bool timed = !c->capturing;
    if (timed && !c->ev0[c->ev_used]) {
        HIP_TRY(hipEventCreate(&c->ev0[c->ev_used]));
        HIP_TRY(hipEventCreate(&c->ev1[c->ev_used]));
    }
    // mixed mode: once acc holds the sum over ALL sources (an ALL pass, or the REMOTE pass on top of LOCAL), the suspects
    // are re-evaluated in fp64 against all chunks; a third event marks the end of that (nbx_ctx_refine_time)
    const bool refine_now = refine && (which != NBX_SRC_LOCAL || c->n_shards == 1);
    if (timed) { L.ev_start = c->ev0[c->ev_used]; L.ev_stop = c->ev1[c->ev_used]; c->ev2_set[c->ev_used] = 0; }
    HIP_TRY(launch_accel(c->dim, L, c->stream));
    ++c->launches_since_query;
    c->have_accel = true;
    c->refined = false;
    if (refine_now) {
        HIP_TRY(launch_refine(c->dim, refine_launch(c), c->stream));
        if (timed) {
            if (!c->ev2[c->ev_used]) HIP_TRY(hipEventCreate(&c->ev2[c->ev_used]));
            HIP_TRY(hipEventRecord(c->ev2[c->ev_used], c->stream));
            c->ev2_set[c->ev_used] = 1;
        }
        c->refined = true;
    }
    if (timed) ++c->ev_used;
    return NBX_OK;
}

int nbx_ctx_kick_drift2(nbx_ctx* c, double G, double dt_kick, double dt_drift) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (!c->have_accel) return fail(NBX_ERR_STATE, "compute accelerations before kick_drift");
    int rc = set_device(c);
    if (rc) return rc;
    KickDriftArgs k;
    // attractive (Newtonian) law: F = +(G m) a.  The kernels compute F = -(G m) a, so G goes in negated (exact in fp64).
    k.acc = c->acc; k.splits = c->splits; k.dim = c->dim; k.pad = c->pad; k.count = c->count; k.G = c->law ? -G : G;
    k.dt_kick = dt_kick; k.dt_drift = dt_drift;
    k.x64 = c->x64; k.v64 = c->v64; k.m64 = c->m64;
    k.pos_chunk = c->pos_all + (size_t)c->shard * c->dim * c->pad;
    if (dt_drift != 0.0) {
        rc = poll_close_counters(c, 1);   // before the lists are invalidated on the host side: the counters belong to this step
        if (rc) return rc;
    }
    HIP_TRY(launch_kick_drift(k, c->stream));
    if (dt_drift != 0.0) {               // a pure kick leaves the positions -- and with them accelerations and lists -- valid
        c->have_accel = false;
        c->tgt_cand_valid = 0; c->bad_list_pass = -1;  // positions moved
    }
    return NBX_OK;
}

int nbx_ctx_kick_drift(nbx_ctx* c, double G, double dt) { return nbx_ctx_kick_drift2(c, G, dt, dt); }

// Capture one step on the context's stream into an executable graph (launch-bound regime: seven
// launches per step cost ~7 % at N = 65,536).  Returns false -- leaving no capture open -- if anything
// about capture is unavailable; the caller then steps eagerly.
static bool capture_step(nbx_ctx* c, double G, double dt) {
    if (c->step_exec && c->graph_G == G && c->graph_dt == dt && c->graph_variant == c->variant &&
        c->graph_splits == c->splits && c->graph_stream == c->stream && c->graph_eps == c->softening && c->graph_law == c->law && c->graph_hash == c->hash_refine &&
        c->graph_refine_tol == c->refine_tol && c->graph_refine_sigma == c->refine_sigma && c->graph_clk == (c->clk_on ? c->clk : nullptr))
        return true;
    if (c->step_exec) { (void)hipGraphExecDestroy(c->step_exec); c->step_exec = nullptr; }
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return false; }
    c->capturing = true;
    c->tgt_cand_valid = 0; c->bad_list_pass = -1;
    int rc = nbx_ctx_compute_accel(c, NBX_SRC_ALL);
    if (!rc) rc = nbx_ctx_kick_drift(c, G, dt);
    c->capturing = false;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); return false; }
    const hipError_t ei = hipGraphInstantiate(&c->step_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { c->step_exec = nullptr; (void)hipGetLastError(); return false; }
    c->graph_eps = c->softening; c->graph_law = c->law; c->graph_hash = c->hash_refine;
    c->graph_refine_tol = c->refine_tol; c->graph_refine_sigma = c->refine_sigma;
    c->graph_clk = c->clk_on ? c->clk : nullptr;
    c->graph_G = G; c->graph_dt = dt; c->graph_variant = c->variant; c->graph_splits = c->splits; c->graph_stream = c->stream;
    return true;
}

int nbx_ctx_step(nbx_ctx* c, double G, double dt, int nsteps) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (c->n_shards != 1) return fail(NBX_ERR_STATE, "nbx_ctx_step drives single-shard contexts only");
    if (nsteps < 0) return fail(NBX_ERR_INVALID, "nsteps < 0");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "upload bodies before stepping");
    int s = 0;
    if (nsteps >= kGraphMinSteps && !c->no_graphs) {
        int rc = set_device(c);
        if (rc) return rc;
        rc = ensure_acc(c);  // every allocation happens before capture
        if (rc) return rc;
        if (capture_step(c, G, dt)) {
            if (!c->bulk0) { HIP_TRY(hipEventCreate(&c->bulk0)); HIP_TRY(hipEventCreate(&c->bulk1)); }
            if (c->bulk_steps) {  // resolve the previous replay sequence before reusing the event pair
                float ms = 0.f;
                HIP_TRY(hipEventSynchronize(c->bulk1));
                HIP_TRY(hipEventElapsedTime(&ms, c->bulk0, c->bulk1));
                c->bulk_ms_done += ms; c->bulk_steps_done += c->bulk_steps; c->bulk_steps = 0;
            }
            HIP_TRY(hipEventRecord(c->bulk0, c->stream));
            // Replays go out in blocks of kPollEverySteps with a look at the close-set counters in between, so that one long
            // call follows the bodies like a sequence of short ones does: when the host's view of the counters (an event
            // query, never a wait) flips the refinement mode or demotes the kernel, the step is captured again.
            bool replay = true;
            while (s < nsteps && replay) {
                const int block = nsteps - s < kPollEverySteps ? nsteps - s : kPollEverySteps;
                for (int k = 0; k < block; ++k) HIP_TRY(hipGraphLaunch(c->step_exec, c->stream));
                s += block;
                c->have_accel = false;
                c->tgt_cand_valid = 0; c->bad_list_pass = -1;
                rc = poll_close_counters(c, block);
                if (rc) return rc;
                if (s < nsteps && (c->graph_hash != c->hash_refine || c->graph_variant != effective_variant(c))) {
                    // the replays queued so far still use the executable graph and, possibly, buffers that ensure_acc is about to
                    // give back: drain the stream first (a rare branch -- the close-set regime of the system changed)
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    rc = ensure_acc(c);
                    if (rc) return rc;
                    replay = capture_step(c, G, dt);   // false: the remaining steps run eagerly below
                }
            }
            HIP_TRY(hipEventRecord(c->bulk1, c->stream));
            c->bulk_steps = s;
        }
    }
    for (; s < nsteps; ++s) {
        int rc = nbx_ctx_compute_accel(c, NBX_SRC_ALL);
        if (rc) return rc;
        rc = nbx_ctx_kick_drift(c, G, dt);
        if (rc) return rc;
    }
    return NBX_OK;
}

int nbx_ctx_step_kdk(nbx_ctx* c, double G, double dt, int nsteps) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    if (c->n_shards != 1) return fail(NBX_ERR_STATE, "nbx_ctx_step_kdk drives single-shard contexts only");
    if (nsteps < 0) return fail(NBX_ERR_INVALID, "nsteps < 0");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "upload bodies before stepping");
    if (nsteps == 0) return NBX_OK;
    // K(dt/2) D(dt) [F K(dt) D(dt)]^(n-1) F K(dt/2): adjacent half-kicks merged, n + 1 force evaluations for n steps
    int rc = c->have_accel ? NBX_OK : nbx_ctx_compute_accel(c, NBX_SRC_ALL);
    for (int s = 0; s < nsteps && !rc; ++s) {
        rc = nbx_ctx_kick_drift2(c, G, s == 0 ? 0.5 * dt : dt, dt);
        if (!rc) rc = nbx_ctx_compute_accel(c, NBX_SRC_ALL);
    }
    if (!rc) rc = nbx_ctx_kick_drift2(c, G, 0.5 * dt, 0.0);
    return rc;
}

int nbx_ctx_get_forces(nbx_ctx* c, double G, double* out) {
    if (!c || (!out && c->count)) return fail(NBX_ERR_INVALID, "null argument");
    if (!c->have_accel) return fail(NBX_ERR_STATE, "no accelerations computed");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t bytes = c->count * c->dim * sizeof(double);
    rc = ensure_stage(c, bytes ? bytes : 8);
    if (rc) return rc;
    HIP_TRY(launch_export_forces(c->acc, c->splits, c->dim, c->pad, c->count, c->law ? -G : G, c->m64, c->stage, c->stream));
    if (bytes) HIP_TRY(hipMemcpyAsync(out, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBX_OK;
}

int nbx_ctx_accuracy(nbx_ctx* c, double G, const double* reference_forces, double* percent) {
    if (!c || !percent || (!reference_forces && c->count)) return fail(NBX_ERR_INVALID, "null argument");
    if (!c->have_accel) return fail(NBX_ERR_STATE, "no accelerations computed");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t bytes = c->count * c->dim * sizeof(double);
    rc = ensure_stage(c, bytes + 16);
    if (rc) return rc;
    unsigned* counter = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(c->stage) + bytes);  // 8-byte aligned tail
    if (bytes) HIP_TRY(hipMemcpyAsync(c->stage, reference_forces, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_accuracy(c->acc, c->splits, c->dim, c->pad, c->count, c->law ? -G : G, c->m64, c->stage, counter, c->stream));
    unsigned good = 0;
    HIP_TRY(hipMemcpyAsync(&good, counter, sizeof good, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *percent = c->count ? 100.0 * (double)(int)good / (double)c->count : 0.0;
    return NBX_OK;
}

int nbx_ctx_get_accel(nbx_ctx* c, float* out) {
    if (!c || (!out && c->count)) return fail(NBX_ERR_INVALID, "null argument");
    if (!c->have_accel) return fail(NBX_ERR_STATE, "no accelerations computed");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t bytes = c->count * c->dim * sizeof(float);
    rc = ensure_stage(c, bytes ? bytes : 8);
    if (rc) return rc;
    HIP_TRY(launch_export_accel(c->acc, c->splits, c->dim, c->pad, c->count, (float*)c->stage, c->stream));
    if (bytes) HIP_TRY(hipMemcpyAsync(out, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBX_OK;
}

int nbx_ctx_download_bodies(nbx_ctx* c, void* bodies, size_t stride_bytes) {
    if (!c || (!bodies && c->count)) return fail(NBX_ERR_INVALID, "null argument");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "nothing uploaded");
    const size_t min_stride = (size_t)(2 * c->dim + 1) * sizeof(double);
    if (stride_bytes < min_stride || stride_bytes % sizeof(double) != 0)
        return fail(NBX_ERR_INVALID, "body stride must be a multiple of 8 and >= sizeof(Body<dim>)");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t w = 2 * (size_t)c->dim;
    const size_t bytes = c->count * w * sizeof(double);
    rc = ensure_stage(c, bytes ? bytes : 8);
    if (rc) return rc;
    HIP_TRY(launch_export_state(c->x64, c->v64, c->dim, c->pad, c->count, c->stage, c->stream));
    std::vector<double> host;
    try { host.resize(c->count * w); } catch (...) { return fail(NBX_ERR_ALLOC, "host staging allocation failed"); }
    if (bytes) HIP_TRY(hipMemcpyAsync(host.data(), c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    char* base = (char*)bodies + (size_t)c->shard * c->shard_len * stride_bytes;
    for (size_t l = 0; l < c->count; ++l) std::memcpy(base + l * stride_bytes, &host[l * w], w * sizeof(double));
    return NBX_OK;
}

int nbx_ctx_energy(nbx_ctx* c, double G, double* kinetic, double* potential) {
    if (!c || !kinetic || !potential) return fail(NBX_ERR_INVALID, "null argument");
    if (!c->uploaded) return fail(NBX_ERR_STATE, "nothing uploaded");
    int rc = set_device(c);
    if (rc) return rc;
    if (!c->phi && (rc = dev_alloc(c, &c->phi, (size_t)kPhiSlices * c->pad * sizeof(float)))) return rc;
    AccelLaunch L = {};
    L.pos_all = c->pos_all; L.mass_all = c->mass_all; L.acc = c->phi; L.pad = c->pad; L.count = (unsigned)c->count;
    L.tgt_chunk = c->shard; L.chunk_first = 0; L.vchunks = c->n_shards; L.chunk_skip = INT_MAX; L.splits = kPhiSlices;
    L.eps2 = (float)(c->softening * c->softening);
    L.law = c->law;
    if (c->law != 0 && !(c->softening > 0.0)) return fail(NBX_ERR_STATE, "the Newtonian law needs a softening length (nbx_ctx_set_softening)");
    HIP_TRY(launch_potential(c->dim, L, c->stream));
    const size_t nblk = (c->count + 255) / 256;
    const size_t bytes = 2 * nblk * sizeof(double);
    rc = ensure_stage(c, bytes ? bytes : 8);
    if (rc) return rc;
    // per-body potential = (G m / 4) phi for the reference law (U = sum_{i<j} G m m / (2 r^2)); Newtonian:
    // U = -sum_{i<j} G m m / sqrt(r^2+eps^2), i.e. -(G m / 2) phi = (G' m / 4) phi with G' = -2 G
    HIP_TRY(launch_export_energy(c->phi, kPhiSlices, c->dim, c->pad, c->count, c->law ? -2.0 * G : G, c->v64, c->m64, c->stage, c->stream));
    std::vector<double> host;
    try { host.resize(2 * nblk); } catch (...) { return fail(NBX_ERR_ALLOC, "host staging allocation failed"); }
    if (bytes) HIP_TRY(hipMemcpyAsync(host.data(), c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    long double ke = 0.0L, pe = 0.0L;  // workgroup order over the device's per-workgroup sums: a fixed tree, deterministic
    for (size_t l = 0; l < nblk; ++l) { ke += host[l]; pe += host[nblk + l]; }
    *kinetic = (double)ke;
    *potential = (double)pe;
    return NBX_OK;
}

int nbx_ctx_synchronize(nbx_ctx* c) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    int rc = set_device(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBX_OK;
}

int nbx_ctx_kernel_time(nbx_ctx* c, float* mean_ms, int* launches) {
    if (!c) return fail(NBX_ERR_INVALID, "ctx is null");
    int rc = set_device(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double sum = 0.0;
    double refine_sum = c->refine_ms_done;
    for (int i = 0; i < c->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
        sum += ms;
        if (c->ev2_set[i]) { HIP_TRY(hipEventElapsedTime(&ms, c->ev1[i], c->ev2[i])); refine_sum += ms; }
    }
    c->refine_ms_done = 0.0;
    int count = c->ev_used;
    if (c->bulk_steps) {  // graph-replayed steps: whole-step time (force kernel + the ~0.3 % of small kernels around it)
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->bulk0, c->bulk1));
        c->bulk_ms_done += ms; c->bulk_steps_done += c->bulk_steps; c->bulk_steps = 0;
    }
    sum += c->bulk_ms_done; count += c->bulk_steps_done;
    c->bulk_ms_done = 0.0; c->bulk_steps_done = 0;
    if (mean_ms) *mean_ms = count ? (float)(sum / count) : 0.f;
    if (launches) *launches = count;
    c->last_refine_ms_total = refine_sum;
    c->ev_used = 0;
    c->launches_since_query = 0;
    return NBX_OK;
}

int nbx_ctx_refine_time(nbx_ctx* c, float* total_ms) {
    if (!c || !total_ms) return fail(NBX_ERR_INVALID, "null argument");
    *total_ms = (float)c->last_refine_ms_total;
    return NBX_OK;
}

// ---- one-shot entry points -------------------------------------------------------------------------

int nbx_brute_force_forces_ex(const void* bodies, size_t n, int dim, size_t stride_bytes, double G, int device,
                              double rel_tolerance, double* forces_out, nbx_eval_info* info) {
    if ((!bodies || !forces_out) && n) return fail(NBX_ERR_INVALID, "null argument");
    if (info) *info = nbx_eval_info{};
    nbx_ctx* c = nullptr;
    int rc = nbx_ctx_create(&c, device, dim, n, 1, 0);
    if (rc) return rc;
    if (rel_tolerance >= 0.0) rc = nbx_ctx_set_refine(c, rel_tolerance, 0.0);   // < 0: the process default the context was made with
    if (!rc) rc = nbx_ctx_upload_bodies(c, bodies, stride_bytes);
    if (!rc) rc = nbx_ctx_compute_accel(c, NBX_SRC_ALL);
    if (!rc) rc = nbx_ctx_get_forces(c, G, forces_out);
    if (!rc && info) {
        rc = nbx_ctx_kernel_time(c, &info->kernel_ms, nullptr);
        info->refine_ms = (float)c->last_refine_ms_total;
        info->variant = c->variant;
        if (!rc) rc = nbx_ctx_close_set_mode(c, &info->close_set_mode, nullptr, nullptr);
        if (!rc && c->refined) {
            info->refine_tolerance = c->refine_tol;
            rc = nbx_ctx_refine_stats(c, &info->refine_selected, &info->refine_refined);
        }
    }
    nbx_ctx_destroy(c);
    return rc;
}

int nbx_brute_force_forces(const void* bodies, size_t n, int dim, size_t stride_bytes, double G, int device,
                           double* forces_out, float* kernel_ms) {
    nbx_eval_info info;
    const int rc = nbx_brute_force_forces_ex(bodies, n, dim, stride_bytes, G, device, -1.0, forces_out, kernel_ms ? &info : nullptr);
    if (!rc && kernel_ms) *kernel_ms = info.kernel_ms;
    return rc;
}

int nbx_leapfrog(void* bodies, size_t n, int dim, size_t stride_bytes, double G, double dt, int nsteps, int device,
                 float* kernel_ms_total) {
    if (!bodies && n) return fail(NBX_ERR_INVALID, "null argument");
    nbx_ctx* c = nullptr;
    int rc = nbx_ctx_create(&c, device, dim, n, 1, 0);
    if (rc) return rc;
    rc = nbx_ctx_upload_bodies(c, bodies, stride_bytes);
    if (!rc) rc = nbx_ctx_step(c, G, dt, nsteps);
    if (!rc) rc = nbx_ctx_download_bodies(c, bodies, stride_bytes);
    if (!rc && kernel_ms_total) {
        float mean = 0.f; int cnt = 0;
        rc = nbx_ctx_kernel_time(c, &mean, &cnt);
        *kernel_ms_total = mean * (float)cnt;
    }
    nbx_ctx_destroy(c);
    return rc;
}

}  // extern "C"
