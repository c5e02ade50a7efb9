// nbx_ctx.h -- private definition of the opaque context of include/nbody_hip.h, shared by nbx_api.hip and
// nbx_node.hip (not installed).
#ifndef NBX_CTX_H
#define NBX_CTX_H

#include <vector>

#include "nbx_internal.h"

struct nbx_ctx {
    unsigned long long id = 0;   // unique per context ever created in this process (never reused: a leaf plan remembers WHICH context its last evaluation read, not where it lived)
    int device = 0, dim = 3, n_shards = 1, shard = 0;
    size_t n_total = 0, shard_len = 0, count = 0;  // count = real bodies in this shard
    unsigned pad = 0;
    int splits = 1, variant = 0;   // splits = fp32 planes of acc = source slices x planes per slice (2 for the strict kernel)
    bool splits_user = false;
    int user_slices = 0;           // the caller's source_splits (nbx_ctx_set_tuning), a lower bound on the slice count
    bool uploaded = false, have_accel = false;
    bool shard_staged = false;                 // nbx_ctx_upload_shard done, nbx_ctx_upload_finish pending
    unsigned long long pending_facts[3] = {0, 0, 0};
    hipStream_t own_stream = nullptr, stream = nullptr;
    // exchange buffers (own or caller's)
    float* pos_all = nullptr;
    float* mass_all = nullptr;
    bool own_gather = true;
    // own shard
    double *x64 = nullptr, *v64 = nullptr, *m64 = nullptr;
    unsigned long long* facts = nullptr;   // [3] what pack_kernel learnt about the uploaded bodies (PackArgs::facts)
    float* acc = nullptr;
    int acc_splits_alloc = 0;
    // fast-path workspace (close-set pipeline) and its preconditions
    unsigned* cand_list = nullptr;
    float* cand_pos = nullptr;
    unsigned* bad_list = nullptr;
    unsigned* bad_flag = nullptr;
    unsigned* counters = nullptr;
    float* close_acc = nullptr;
    int close_splits_alloc = 0;
    float* src_cand_pos = nullptr;   // [dim][n_shards*pad] candidate sources of the pass being launched
    nbx::HashWork hash;              // sorted-cell refinement workspace (allocated only in hash mode)
    bool hash_refine = false;        // most of the shard is in the candidate set: refine through sorted cells
    unsigned probe_bad = 0;          // bad targets found by the upload-time probe (hash mode)
    // periodic, asynchronous look at the close-set counters during long runs (never blocks: event query)
    unsigned* counters_host = nullptr;   // pinned [4]
    hipEvent_t counters_ev = nullptr;
    bool counters_pending = false;
    int steps_since_poll = 0;
    unsigned last_cand = 0, last_bad = 0;   // most recent counts seen by the host
    // mixed mode (nbx_ctx_set_refine): fp32 for all targets, the strict fp64 kernel for the suspects
    double refine_tol = 0.0;         // requested relative tolerance (0: mixed mode off)
    double refine_sigma = 0.0;       // sigma factor of the selection rule (force_kernel.hip refine_select_kernel)
    float* qsum = nullptr;           // [slices][pad] spread sums written by the fast kernel's QS build
    int qsum_slices_alloc = 0;
    unsigned* strict_list = nullptr; // [strict_cap]
    double* strict_acc = nullptr;    // [strict_slices][dim][strict_cap]
    unsigned strict_cap = 0;         // = pad: room for every target of the shard
    int strict_slices = 0;           // most source slices of the fp64 pass (the device may use fewer for a long list)
    unsigned long long strict_budget = 0;   // doubles in strict_acc
    bool refined = false;            // the accelerations on the device went through the refinement
    float* phi = nullptr;        // [kPhiSlices][pad] potential partials (energy diagnostic)
    // one captured step {rebuild lists, force, scatter, kick+drift} replayed by nbx_ctx_step
    hipGraphExec_t step_exec = nullptr;
    double graph_G = 0.0, graph_dt = 0.0, graph_eps = 0.0, graph_refine_tol = 0.0, graph_refine_sigma = 0.0;
    int graph_law = 0;
    bool graph_hash = false;
    int graph_variant = -1, graph_splits = 0;
    hipStream_t graph_stream = nullptr;
    bool capturing = false;
    // timing of graph-replayed steps: one event pair around a whole replay sequence
    hipEvent_t bulk0 = nullptr, bulk1 = nullptr;
    int bulk_steps = 0;          // steps covered by the pending (bulk0, bulk1) pair
    double bulk_ms_done = 0.0;   // already-resolved replay time not yet reported
    int bulk_steps_done = 0;
    bool no_graphs = false;      // NBODY_HIP_NO_GRAPHS=1: always step eagerly
    int tgt_cand_valid = 0;     // the device's candidate-target list matches the own chunk's positions
    int bad_list_pass = -1;     // NBX_SRC_* pass whose bad-target list is on the device (-1: none / stale)
    int law = 0;                // 0: the reference's law, 1: softened Newtonian (extension; needs softening > 0)
    double softening = 0.0;     // epsilon of the softened law (0: the reference law, no softening)
    double mass_max = 0.0;      // largest |mass| seen at upload (the softened law needs m / eps^4 finite in fp32)
    bool extent_ok = false;     // every |coordinate| <= kOneRcpMaxCoord at upload (one-reciprocal kernel allowed)
    bool force_exact = false;   // masses too large for the kTiny bias, or most of the shard in the close set
    int variant_req = -1;       // what the caller asked for (-1: library default)
    // device memory: ONE allocation made at creation holds everything a default run needs (a one-shot call otherwise pays
    // ~17 hipMalloc and as many synchronising hipFree, several milliseconds, for work that takes microseconds at the
    // reference sweep's small sizes); what does not fit -- hash-mode workspace, a later change of tuning -- is allocated
    // on its own and remembered in `extra`
    char* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    std::vector<void*> extra;
    // boundary staging
    double* stage = nullptr;
    size_t stage_bytes = 0;
    // kernel timing
    std::vector<hipEvent_t> ev0, ev1, ev2;   // around the force kernel; ev2: after the mixed mode's kernels of the same evaluation
    std::vector<char> ev2_set;
    double refine_ms_done = 0.0;        // mixed-mode time of evaluations already folded out of the event log
    double last_refine_ms_total = 0.0;  // mixed-mode time of the evaluations the last nbx_ctx_kernel_time call covered
    int ev_used = 0;
    int launches_since_query = 0;
    // measurement: in-kernel clock stamps of the force kernel (nbx_ctx_enable_clock_stamps / nbx_ctx_shader_clock)
    bool clk_on = false;
    unsigned long long* clk = nullptr;   // [2 x clk_cap], its own allocation
    size_t clk_cap = 0;                  // workgroups it has room for
    unsigned clk_slots = 0;              // workgroups of the last stamped launch
    unsigned long long* graph_clk = nullptr;   // the stamp buffer baked into the captured step (null: none)
    int num_cus = 256;
};

namespace nbx {
// error plumbing of the C ABI (nbx_api.hip): record the detail text, return the status code
int fail_hip(hipError_t e, const char* what, const char* file, int line);
int fail(int code, const char* msg);
// non-blocking streams from a per-device pool of parked ones (nbx_api.hip: creating and destroying a stream costs
// milliseconds on this runtime); park_stream takes an IDLE stream
hipError_t take_stream(int device, hipStream_t* out);
void park_stream(int device, hipStream_t s);
void release_parked_streams();
// a destroyed context's device allocation is kept for the next context on that device (nbx_api.hip)
hipError_t take_ctx_arena(int device, size_t bytes, char** out, size_t* got);
void park_ctx_arena(int device, char* p, size_t bytes);
void release_parked_ctx_arenas();
// the two halves of nbx_ctx_upload_bodies (nbx_api.hip), apart so that the node layer can upload each rank's own shard
// only and fill the other chunks device to device in between
int upload_stage(nbx_ctx* c, const void* bodies, size_t stride_bytes, bool only_own, unsigned long long facts[3],
                 bool bodies_is_own_slice = false);   // true: `bodies` points at this shard's first body, not at body 0
int upload_finish(nbx_ctx* c, const unsigned long long facts[3]);
bool ctx_alive(unsigned long long id);  // nbx_api.hip: a context with this id exists (created, not yet destroyed)
void release_parked_communicators();   // nbx_node.hip
void release_parked_leaf_arenas();     // leaf_pair_kernel.hip
}  // namespace nbx

#define NBX_HIP_TRY(expr)                                                           \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) return nbx::fail_hip(e_, #expr, __FILE__, __LINE__);  \
    } while (0)

#endif
