// Micro-benchmark 2: does VGPR bank placement / operand reuse change the issue rate of the packed-fp32
// instructions of the fast force kernel on gfx950?  Explicit registers; 64-bit operands are even-aligned,
// so a pair sits on banks {0,1} (index % 4 == 0, "A") or {2,3} (index % 4 == 2, "B").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_banks.hip -o tools/ubench_banks
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long cyc, real; };

// 16 instructions per body; destinations rotate over 8 accumulators to stay independent
#define REP16(X) X X X X X X X X X X X X X X X X
#define BODY_BEGIN asm volatile(
#define CLOB : : : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87")

enum { T_FMA_AAB = 0, T_FMA_AAA, T_FMA_ABA, T_FMA_SQ_AB, T_FMA_SQ_AA, T_FMA_BCAST, T_ADD_AB, T_ADD_AA, T_MUL_AB, T_MUL_AA, T_FMA_DSTSRC_DIFF, T_S_FMA_012, T_S_FMA_000, T_N };
static const char* tname[] = {
    "pk_fma d(A)+= a(A)*b(B)", "pk_fma d(A)+= a(A)*b(A)", "pk_fma d(B)+= a(A)*b(A)... d=A a=B b=A", "pk_fma d(B)+= a(A)*a(A)  (square)",
    "pk_fma d(A)+= a(A)*a(A)  (square)", "pk_fma d(A)+= a(A)*b(B) op_sel bcast lo", "pk_add d(A)= a(A)-b(B)", "pk_add d(A)= a(A)-b(A)",
    "pk_mul d(A)= a(A)*b(B)", "pk_mul d(A)= a(A)*b(A)", "pk_fma d2(A)= a(A)*b(B)+c(B) (dst!=src2)", "v_fma d= a(b0)*b(b1)+d(b2)", "v_fma d= a(b0)*b(b0)+d(b0)"};

template <int T>
__global__ __launch_bounds__(256) void ub(float* out, int iters, Stamp* st) {
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // destinations: A-pairs v[48:49],v[52:53],v[56:57],v[60:61]; B-pairs v[50:51],v[54:55],v[58:59],v[62:63]
        // sources: A: v[40:41], v[44:45], v[64:65];  B: v[42:43], v[46:47], v[66:67]
        if constexpr (T == T_FMA_AAB) { BODY_BEGIN REP16("v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[48:49]\n\tv_pk_fma_f32 v[52:53], v[44:45], v[46:47], v[52:53]\n\tv_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[56:57]\n\tv_pk_fma_f32 v[60:61], v[44:45], v[42:43], v[60:61]\n\t") CLOB; }
        else if constexpr (T == T_FMA_AAA) { BODY_BEGIN REP16("v_pk_fma_f32 v[48:49], v[40:41], v[44:45], v[48:49]\n\tv_pk_fma_f32 v[52:53], v[44:45], v[64:65], v[52:53]\n\tv_pk_fma_f32 v[56:57], v[40:41], v[64:65], v[56:57]\n\tv_pk_fma_f32 v[60:61], v[44:45], v[40:41], v[60:61]\n\t") CLOB; }
        else if constexpr (T == T_FMA_ABA) { BODY_BEGIN REP16("v_pk_fma_f32 v[48:49], v[42:43], v[40:41], v[48:49]\n\tv_pk_fma_f32 v[52:53], v[46:47], v[44:45], v[52:53]\n\tv_pk_fma_f32 v[56:57], v[66:67], v[40:41], v[56:57]\n\tv_pk_fma_f32 v[60:61], v[42:43], v[64:65], v[60:61]\n\t") CLOB; }
        else if constexpr (T == T_FMA_SQ_AB) { BODY_BEGIN REP16("v_pk_fma_f32 v[50:51], v[40:41], v[40:41], v[50:51]\n\tv_pk_fma_f32 v[54:55], v[44:45], v[44:45], v[54:55]\n\tv_pk_fma_f32 v[58:59], v[64:65], v[64:65], v[58:59]\n\tv_pk_fma_f32 v[62:63], v[40:41], v[40:41], v[62:63]\n\t") CLOB; }
        else if constexpr (T == T_FMA_SQ_AA) { BODY_BEGIN REP16("v_pk_fma_f32 v[48:49], v[40:41], v[40:41], v[48:49]\n\tv_pk_fma_f32 v[52:53], v[44:45], v[44:45], v[52:53]\n\tv_pk_fma_f32 v[56:57], v[64:65], v[64:65], v[56:57]\n\tv_pk_fma_f32 v[60:61], v[40:41], v[40:41], v[60:61]\n\t") CLOB; }
        else if constexpr (T == T_FMA_BCAST) { BODY_BEGIN REP16("v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[48:49] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[52:53], v[44:45], v[46:47], v[52:53] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[56:57] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[60:61], v[44:45], v[42:43], v[60:61] op_sel_hi:[0,1,1]\n\t") CLOB; }
        else if constexpr (T == T_ADD_AB) { BODY_BEGIN REP16("v_pk_add_f32 v[48:49], v[40:41], v[42:43] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 v[52:53], v[44:45], v[46:47] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 v[56:57], v[40:41], v[46:47] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 v[60:61], v[44:45], v[42:43] neg_lo:[0,1] neg_hi:[0,1]\n\t") CLOB; }
        else if constexpr (T == T_ADD_AA) { BODY_BEGIN REP16("v_pk_add_f32 v[48:49], v[40:41], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 v[52:53], v[44:45], v[64:65] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 v[56:57], v[40:41], v[64:65] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 v[60:61], v[64:65], v[40:41] neg_lo:[0,1] neg_hi:[0,1]\n\t") CLOB; }
        else if constexpr (T == T_MUL_AB) { BODY_BEGIN REP16("v_pk_mul_f32 v[48:49], v[40:41], v[42:43]\n\tv_pk_mul_f32 v[52:53], v[44:45], v[46:47]\n\tv_pk_mul_f32 v[56:57], v[40:41], v[46:47]\n\tv_pk_mul_f32 v[60:61], v[44:45], v[42:43]\n\t") CLOB; }
        else if constexpr (T == T_MUL_AA) { BODY_BEGIN REP16("v_pk_mul_f32 v[48:49], v[40:41], v[44:45]\n\tv_pk_mul_f32 v[52:53], v[44:45], v[64:65]\n\tv_pk_mul_f32 v[56:57], v[40:41], v[64:65]\n\tv_pk_mul_f32 v[60:61], v[64:65], v[40:41]\n\t") CLOB; }
        else if constexpr (T == T_FMA_DSTSRC_DIFF) { BODY_BEGIN REP16("v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[46:47]\n\tv_pk_fma_f32 v[52:53], v[44:45], v[46:47], v[42:43]\n\tv_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[66:67]\n\tv_pk_fma_f32 v[60:61], v[44:45], v[42:43], v[66:67]\n\t") CLOB; }
        else if constexpr (T == T_S_FMA_012) { BODY_BEGIN REP16("v_fma_f32 v50, v40, v41, v50\n\tv_fma_f32 v51, v44, v45, v51\n\tv_fma_f32 v54, v40, v45, v54\n\tv_fma_f32 v55, v44, v41, v55\n\t") CLOB; }
        else if constexpr (T == T_S_FMA_000) { BODY_BEGIN REP16("v_fma_f32 v48, v40, v44, v48\n\tv_fma_f32 v52, v44, v64, v52\n\tv_fma_f32 v56, v40, v64, v56\n\tv_fma_f32 v60, v64, v40, v60\n\t") CLOB; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(t1 - t0);
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = t1 - t0; st[blockIdx.x].real = r1 - r0; }
}

template <int T>
static void run(int bpc, float* d_out, Stamp* d_st, int ncu) {
    int grid = ncu * bpc, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int pass = 0; pass < 3; ++pass) {
        CK(hipEventRecord(e0)); ub<T><<<grid, 256>>>(d_out, iters, d_st); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass < 2) iters = (int)std::min(4.0e7, std::max(500.0, iters * 120.0 / std::max(ms, 0.01f)));
    }
    std::vector<Stamp> st(grid); CK(hipMemcpy(st.data(), d_st, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk(grid); for (int i = 0; i < grid; ++i) clk[i] = (double)st[i].cyc / (double)st[i].real * 100.0;
    std::sort(clk.begin(), clk.end());
    double instr = (double)iters * 64.0;  // 16 x 4 per body
    double ns = (ms * 1e6) / (instr * bpc);
    printf("%-46s w/SIMD=%d  clk=%5.0f MHz  wall ns/instr/SIMD=%6.3f  = %5.2f cycles\n", tname[T], bpc, clk[grid / 2], ns, ns * clk[grid / 2] * 1e-3);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    float* d_out; Stamp* d_st;
    CK(hipMalloc(&d_out, sizeof(float) * ncu * 8 * 256)); CK(hipMalloc(&d_st, sizeof(Stamp) * ncu * 8));
    for (int i = 0; i < 10; ++i) ub<T_FMA_AAB><<<ncu * 8, 256>>>(d_out, 200000, d_st);
    CK(hipDeviceSynchronize());
    for (int w : {3, 4, 8}) {
        run<T_FMA_AAB>(w, d_out, d_st, ncu); run<T_FMA_AAA>(w, d_out, d_st, ncu); run<T_FMA_ABA>(w, d_out, d_st, ncu);
        run<T_FMA_SQ_AB>(w, d_out, d_st, ncu); run<T_FMA_SQ_AA>(w, d_out, d_st, ncu); run<T_FMA_BCAST>(w, d_out, d_st, ncu);
        run<T_FMA_DSTSRC_DIFF>(w, d_out, d_st, ncu);
        run<T_ADD_AB>(w, d_out, d_st, ncu); run<T_ADD_AA>(w, d_out, d_st, ncu); run<T_MUL_AB>(w, d_out, d_st, ncu); run<T_MUL_AA>(w, d_out, d_st, ncu);
        run<T_S_FMA_012>(w, d_out, d_st, ncu); run<T_S_FMA_000>(w, d_out, d_st, ncu);
        printf("\n");
    }
    return 0;
}
