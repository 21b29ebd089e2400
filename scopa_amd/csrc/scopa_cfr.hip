// scopa_cfr.hip -- vanilla CFR with the reference's exact sequential semantics.
//
// Reference behaviour: CFRTrainer._cfr_recursive / .train (src/algorithms/vanilla_cfr.py:56-110).  The reference
// refreshes node.local_strategy at the end of EVERY node visit (:97) and 302 of the 738 infosets are visited more
// than once per traversal, so its tables are a function of DFS visit order: the exact path is inherently one
// sequential walk per solve ("replicas only" -- independent solves go to independent workgroups / GPUs).
// The walk runs on one lane as a compile-time recursion over the 8 plies (frames in registers) with the three tables,
// the node->infoset map, the payoffs and the first-visit flags in LDS (~80 KB for 738 infosets), so a visit costs a
// few LDS latencies rather than HBM round trips; the other lanes of the workgroup stage tables in and out.  float64 arithmetic follows numpy operation by operation (no FMA
// contraction: built with -ffp-contract=off), which is what makes the result bit-identical to the reference.
#include "scopa_ctx.h"

using namespace scopa;

namespace {

// InfoNode.get_strategy, vanilla_cfr.py:23-30
template <int N>
__device__ __forceinline__ void regret_match(const double *R, double *out) {
    double pos[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < N; i++) pos[i] = R[i] > 0.0 ? R[i] : 0.0;
    double s = pos[0];
    for (int i = 1; i < N; i++) s += pos[i];  // np.sum, n < 8: left-to-right
    for (int i = 0; i < N; i++) out[i] = s > 0.0 ? pos[i] / s : 1.0 / (double)N;
}

struct ExactWalk {
    double *R, *S, *L;          // tables (LDS when they fit, else HBM)
    const uint16_t *inf;        // node -> infoset (LDS)
    const int8_t *pay;          // leaf payoffs x2 (LDS)
    uint32_t *visit;            // first-visit sequence numbers (LDS)
    uint32_t seq;
    unsigned long long dvis, tvis;
};

// CFRTrainer._cfr_recursive (vanilla_cfr.py:56-99) as a compile-time recursion over the 8 plies: the frames (action
// utilities, reaches, the node's strategy row) live in registers, only table rows and the two maps are memory operations.
// The strategy row is read once on entry: local_strategy of an infoset cannot change before this visit ends, because all
// of its other nodes are on the same ply.
template <int D, int TRAV>
__device__ __forceinline__ double exact_rec(ExactWalk &w, int idx, double r0, double r1) {
    if constexpr (D == kPlies) {  // terminal (:58-59)
        w.tvis++;
        const int p0 = w.pay[idx];
        return 0.5 * (double)(TRAV == 0 ? p0 : -p0);
    } else {
        constexpr int n = 4 - (D >> 1), p = D & 1;
        w.dvis++;
        const int I = w.inf[level_offset(D) + idx];
        if (w.visit[I] == 0u) w.visit[I] = ++w.seq;  // dict insertion on first visit (:51-54)
        double *Lr = w.L + I * 4;
        double ls[4], au[4];
        for (int i = 0; i < n; i++) ls[i] = Lr[i];
#pragma unroll 1
        for (int i = 0; i < n; i++)  // recurse into legal action i (:79-85)
            au[i] = exact_rec<D + 1, TRAV>(w, idx * n + i, p == 0 ? r0 * ls[i] : r0, p == 1 ? r1 * ls[i] : r1);
        double v = ls[0] * au[0];  // np.sum(local_strategy * action_utils) (:87)
        for (int i = 1; i < n; i++) v += ls[i] * au[i];
        double *Rr = w.R + I * 4;
        if constexpr (p == TRAV) {  // (:89-95)
            const double reach = TRAV == 0 ? r0 : r1, opp = TRAV == 0 ? r1 : r0;
            double *Sr = w.S + I * 4;
            for (int i = 0; i < n; i++) {
                Rr[i] += opp * (au[i] - v);
                Sr[i] += reach * ls[i];
            }
        }
        regret_match<n>(Rr, Lr);  // local_strategy refresh on EVERY visit (:97)
        return v;
    }
}

template <int TRAV>
__device__ double exact_from(ExactWalk &w, int depth, int idx, double r0, double r1) {
    switch (depth) {
        case 0: return exact_rec<0, TRAV>(w, idx, r0, r1);
        case 1: return exact_rec<1, TRAV>(w, idx, r0, r1);
        case 2: return exact_rec<2, TRAV>(w, idx, r0, r1);
        case 3: return exact_rec<3, TRAV>(w, idx, r0, r1);
        case 4: return exact_rec<4, TRAV>(w, idx, r0, r1);
        case 5: return exact_rec<5, TRAV>(w, idx, r0, r1);
        case 6: return exact_rec<6, TRAV>(w, idx, r0, r1);
        case 7: return exact_rec<7, TRAV>(w, idx, r0, r1);
        default: return exact_rec<8, TRAV>(w, idx, r0, r1);  // a terminal state was passed in
    }
}

}  // namespace

__global__ void __launch_bounds__(256)
k_cfr_exact(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, double *__restrict__ g_regret,
            double *__restrict__ g_strat, double *__restrict__ g_local, int n_infosets /* <= 0: multi-deal */, int n_traversals, int first_traverser,
            double *__restrict__ root_values, unsigned long long *__restrict__ g_counters, int use_lds,
            uint32_t *__restrict__ g_visit, int32_t *__restrict__ g_meta, int start_depth, int start_idx, double start_r0,
            double start_r1) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint16_t s_inf[1656];
    __shared__ int8_t s_pay[kTerminal];
    __shared__ uint32_t s_visit[kDecision];
    if (n_infosets <= 0) {  // multi-deal mode: one workgroup per deal, shapes from the deal's meta block
        const size_t deal = blockIdx.x;
        g_infoset += deal * kDecision; g_payoff += deal * kTerminal;
        g_regret += deal * kDecision * 4; g_strat += deal * kDecision * 4; g_local += deal * kDecision * 4;
        g_visit += deal * kDecision; g_meta += deal * 8; g_counters += deal * 8;
        if (root_values) root_values += deal * (size_t)n_traversals;
        n_infosets = g_meta[0];
    }
    const int tid = threadIdx.x, cells = n_infosets * 4;
    double *R = g_regret, *S = g_strat, *L = g_local;
    if (use_lds) {
        R = reinterpret_cast<double *>(smem);
        S = R + cells;
        L = S + cells;
        for (int i = tid; i < cells; i += blockDim.x) { R[i] = g_regret[i]; S[i] = g_strat[i]; L[i] = g_local[i]; }
    }
    for (int i = tid; i < kDecision; i += blockDim.x) s_inf[i] = g_infoset[i];
    for (int i = tid; i < kTerminal; i += blockDim.x) s_pay[i] = g_payoff[i];
    for (int i = tid; i < n_infosets; i += blockDim.x) s_visit[i] = g_visit[i];
    __syncthreads();

    if (tid == 0) {
        ExactWalk w;
        w.R = R; w.S = S; w.L = L; w.inf = s_inf; w.pay = s_pay; w.visit = s_visit;
        w.seq = (uint32_t)g_meta[1]; w.dvis = 0; w.tvis = 0;
#pragma unroll 1
        for (int t = 0; t < n_traversals; t++) {
            const int trav = (first_traverser + t) & 1;  // train(): for i in range(num_players) (:108-110)
            const double ret = trav == 0 ? exact_from<0>(w, start_depth, start_idx, start_r0, start_r1)
                                         : exact_from<1>(w, start_depth, start_idx, start_r0, start_r1);
            if (root_values) root_values[t] = ret;
        }
        g_counters[0] += w.dvis;
        g_counters[1] += w.tvis;
        g_meta[1] = (int32_t)w.seq;
    }
    __syncthreads();
    for (int i = tid; i < n_infosets; i += blockDim.x) g_visit[i] = s_visit[i];
    if (use_lds)
        for (int i = tid; i < cells; i += blockDim.x) { g_regret[i] = R[i]; g_strat[i] = S[i]; g_local[i] = L[i]; }
}

static int32_t run_exact(scopa_ctx *ctx, int n_traversals, int first_traverser, double *h_values, int start_depth = 0,
                         int start_idx = 0, double r0 = 1.0, double r1 = 1.0) {
    SC_HIP(ctx, hipSetDevice(ctx->device));
    { const int32_t rc = ensure_scratch(ctx, (size_t)(n_traversals > 0 ? n_traversals : 1) * sizeof(double)); if (rc != SCOPA_OK) return rc; }
    const size_t lds = (size_t)ctx->n_infosets * 4 * 8 * 3;
    const size_t static_lds = 1656 * 2 + kTerminal + sizeof(uint32_t) * kDecision + 256;
    const int use_lds = lds + static_lds <= (size_t)ctx->lds_limit ? 1 : 0;
    SC_LDS_ATTR(ctx, scopa::kLdsCfrExact, k_cfr_exact, ctx->lds_limit - (int)static_lds);
    hipLaunchKernelGGL(k_cfr_exact, dim3(1), dim3(256), use_lds ? lds : 0, ctx->stream, ctx->d_infoset, ctx->d_payoff,
                       ctx->d_regret, ctx->d_strat, ctx->d_local, ctx->n_infosets, n_traversals, first_traverser,
                       ctx->d_scratch, ctx->d_counters, use_lds, ctx->d_visit, ctx->d_meta, start_depth, start_idx, r0, r1);
    SC_HIP(ctx, hipGetLastError());
    if (h_values && n_traversals > 0)
        SC_HIP(ctx, hipMemcpyAsync(h_values, ctx->d_scratch, (size_t)n_traversals * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sigcdf_valid = false;
    return SCOPA_OK;
}

extern "C" {

int32_t scopa_cfr_exact_iterate(scopa_ctx *ctx, int32_t n_iters, double *h_root_values) {
    if (!ctx || n_iters < 0 || n_iters > (1 << 24)) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_cfr_exact_iterate: no deal set");
    if (n_iters == 0) return SCOPA_OK;
    return run_exact(ctx, n_iters * 2, 0, h_root_values);
}

int32_t scopa_cfr_exact_traverse(scopa_ctx *ctx, int32_t traverser, double *h_value) {
    if (!ctx || traverser < 0 || traverser > 1) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_cfr_exact_traverse: no deal set");
    return run_exact(ctx, 1, traverser, h_value);
}

int32_t scopa_cfr_exact_traverse_from(scopa_ctx *ctx, int32_t traverser, int32_t depth, const int32_t *path, double reach_p0,
                                      double reach_p1, double *h_value) {
    if (!ctx || traverser < 0 || traverser > 1 || depth < 0 || depth > kPlies || (depth > 0 && !path)) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_cfr_exact_traverse_from: no deal set");
    int idx = 0;
    for (int d = 0; d < depth; d++) {
        SC_REQUIRE(ctx, path[d] >= 0 && path[d] < nlegal_at(d), SCOPA_EINVAL, "scopa_cfr_exact_traverse_from: path index out of range");
        idx = idx * nlegal_at(d) + path[d];
    }
    return run_exact(ctx, 1, traverser, h_value, depth, idx, reach_p0, reach_p1);
}

}  // extern "C"
