// scopa_cfr.hip -- vanilla CFR with the reference's exact sequential semantics.
//
// Reference behaviour: CFRTrainer._cfr_recursive / .train (src/algorithms/vanilla_cfr.py:56-110).  The reference
// refreshes node.local_strategy at the end of EVERY node visit (:97) and 302 of the 738 infosets are visited more
// than once per traversal, so its tables are a function of DFS visit order: the exact path is inherently one
// sequential walk per solve ("replicas only" -- independent solves go to independent workgroups / GPUs).
// The walk runs on one lane with the three tables, the node->infoset map, the payoffs and the DFS frames all in
// LDS (~75 KB for 738 infosets), so a visit costs LDS latencies rather than HBM round trips; the other lanes of
// the workgroup stage tables in and out.  float64 arithmetic follows numpy operation by operation (no FMA
// contraction: built with -ffp-contract=off), which is what makes the result bit-identical to the reference.
#include "scopa_ctx.h"

using namespace scopa;

namespace {

struct CfrFrame {
    int idx, I, i;
    double au[4], r0, r1;
};

// InfoNode.get_strategy, vanilla_cfr.py:23-30
__device__ __forceinline__ void regret_match(const double *R, int n, double *out) {
    double pos[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < n; i++) pos[i] = R[i] > 0.0 ? R[i] : 0.0;
    double s = pos[0];
    for (int i = 1; i < n; i++) s += pos[i];  // np.sum, n < 8: left-to-right
    for (int i = 0; i < n; i++) out[i] = s > 0.0 ? pos[i] / s : 1.0 / (double)n;
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
    __shared__ CfrFrame fr[kPlies + 1];
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
    __syncthreads();

    if (tid == 0) {
        unsigned long long dvis = 0, tvis = 0;
        uint32_t seq = (uint32_t)g_meta[1];
        for (int t = 0; t < n_traversals; t++) {
            const int trav = (first_traverser + t) & 1;  // train(): for i in range(num_players) (:108-110)
            int d = start_depth;
            fr[d].idx = start_idx; fr[d].i = -1; fr[d].r0 = start_r0; fr[d].r1 = start_r1;
            double ret = 0.0;
            if (d == kPlies) {  // a terminal state was passed in (:58-59)
                const int p0 = s_pay[start_idx];
                ret = 0.5 * (double)(trav == 0 ? p0 : -p0);
                tvis++;
                d = start_depth - 1;
            }
            while (d >= start_depth) {
                if (d == kPlies) {  // terminal (:58-59)
                    const int p0 = s_pay[fr[d].idx];
                    ret = 0.5 * (double)(trav == 0 ? p0 : -p0);
                    tvis++;
                    d--;
                    fr[d].au[fr[d].i] = ret;  // d >= start_depth here: a terminal start never enters the loop
                    continue;
                }
                CfrFrame &f = fr[d];
                const int n = 4 - (d >> 1), p = d & 1;
                if (f.i < 0) {  // node entry (:73-77)
                    dvis++;
                    f.I = s_inf[level_offset(d) + f.idx];
                    f.i = 0;
                    if (g_visit[f.I] == 0u) g_visit[f.I] = ++seq;  // dict insertion on first visit (:51-54)
                } else {
                    f.i++;  // child f.i returned into au[f.i]
                }
                double *ls = L + f.I * 4;
                if (f.i < n) {  // recurse into legal action i with the CURRENT local_strategy[i] (:79-85)
                    CfrFrame &c = fr[d + 1];
                    c.idx = f.idx * n + f.i;
                    c.i = -1;
                    c.r0 = p == 0 ? f.r0 * ls[f.i] : f.r0;
                    c.r1 = p == 1 ? f.r1 * ls[f.i] : f.r1;
                    d++;
                    continue;
                }
                double v = ls[0] * f.au[0];  // np.sum(local_strategy * action_utils) (:87)
                for (int i = 1; i < n; i++) v += ls[i] * f.au[i];
                if (p == trav) {  // (:89-95)
                    const double reach = trav == 0 ? f.r0 : f.r1, opp = trav == 0 ? f.r1 : f.r0;
                    for (int i = 0; i < n; i++) {
                        const double regret = f.au[i] - v;
                        R[f.I * 4 + i] += opp * regret;
                        S[f.I * 4 + i] += reach * ls[i];
                    }
                }
                regret_match(R + f.I * 4, n, ls);  // local_strategy refresh on EVERY visit (:97)
                ret = v;
                d--;
                if (d >= start_depth) fr[d].au[fr[d].i] = ret;
            }
            if (root_values) root_values[t] = ret;
        }
        g_counters[0] += dvis;
        g_counters[1] += tvis;
        g_meta[1] = (int32_t)seq;
    }
    __syncthreads();
    if (use_lds)
        for (int i = tid; i < cells; i += blockDim.x) { g_regret[i] = R[i]; g_strat[i] = S[i]; g_local[i] = L[i]; }
}

static int32_t run_exact(scopa_ctx *ctx, int n_traversals, int first_traverser, double *h_values, int start_depth = 0,
                         int start_idx = 0, double r0 = 1.0, double r1 = 1.0) {
    SC_HIP(ctx, hipSetDevice(ctx->device));
    { const int32_t rc = ensure_scratch(ctx, (size_t)(n_traversals > 0 ? n_traversals : 1) * sizeof(double)); if (rc != SCOPA_OK) return rc; }
    const size_t lds = (size_t)ctx->n_infosets * 4 * 8 * 3;
    const size_t static_lds = 1656 * 2 + kTerminal + sizeof(CfrFrame) * (kPlies + 1) + 256;
    const int use_lds = lds + static_lds <= (size_t)ctx->lds_limit ? 1 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_cfr_exact), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        ctx->lds_limit - (int)static_lds));
        attr_set = true;
    }
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
