// scopa_eval.hip -- policy value and exploitability on the flat tree.
//
// The reference's only exploitability code is a call into OpenSpiel (src/algorithms/vanilla_cfr.py:112-118),
// off by default and absent from this image: there is no reference value to match ("parity unpinned").
// This is the build's own implementation of the procedural definition OpenSpiel uses: the best responder picks,
// per infoset STRING, argmax_a sum_{h in I} opp_reach(h) * value(h.a), deepest infosets first;
// exploitability = NashConv / 2 = (BR_0 + BR_1) / 2 for this zero-sum game.  The arithmetic and summation orders
// are the oracle's (oracle/scopa_oracle.c og_exploitability), so the two agree bit-for-bit.
//
// One workgroup, level-synchronous over the 9 plies; reach, value, policy and q tables live in LDS (~85 KB).
#include "scopa_ctx.h"

using namespace scopa;

__global__ void __launch_bounds__(1024)
k_exploitability(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, const uint64_t *__restrict__ g_key,
                 const double *__restrict__ g_strat, const double *__restrict__ g_policy_in, int n_infosets,
                 double *__restrict__ out4, double *__restrict__ g_policy_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int I = n_infosets, tid = threadIdx.x, nt = blockDim.x;
    double *s_pol = reinterpret_cast<double *>(smem);   // [I][4]
    double *s_q = s_pol + (size_t)I * 4;                // [I][4]
    double *s_reach = s_q + (size_t)I * 4;              // [kNodes] BFS order
    double *s_val = s_reach + kNodes;                   // [kNodes]
    int *s_choice = reinterpret_cast<int *>(s_val + kNodes);  // [I]
    uint16_t *s_inf = reinterpret_cast<uint16_t *>(s_choice + I);  // [1653]

    // policy: given, or the average policy S / sum(S), uniform where nothing was accumulated (vanilla_cfr.py:32-39)
    for (int r = tid; r < I; r += nt) {
        const int n = (int)((g_key[r] >> 1) & 7);
        double p[4] = {0.0, 0.0, 0.0, 0.0};
        if (g_policy_in) {
            for (int c = 0; c < 4; c++) p[c] = g_policy_in[r * 4 + c];
        } else {
            double s = g_strat[r * 4];
            for (int c = 1; c < n; c++) s += g_strat[r * 4 + c];
            for (int c = 0; c < n; c++) p[c] = s > 0.0 ? g_strat[r * 4 + c] / s : 1.0 / (double)n;
        }
        for (int c = 0; c < 4; c++) { s_pol[r * 4 + c] = p[c]; if (g_policy_out) g_policy_out[r * 4 + c] = p[c]; }
    }
    for (int i = tid; i < kDecision; i += nt) s_inf[i] = g_infoset[i];
    __syncthreads();

    for (int pass = 0; pass < 3; pass++) {  // 0: BR of player 0, 1: BR of player 1, 2: plain value of the policy for P0
        const int br = pass;                // pass 2: nobody best-responds
        // top-down: reach of everyone but the best responder
        if (tid == 0) s_reach[0] = 1.0;
        __syncthreads();
        for (int d = 0; d < kPlies; d++) {
            const int n = nlegal_at(d), w1 = level_width(d + 1), p = d & 1;
            for (int j = tid; j < w1; j += nt) {
                const int par = j / n, a = j - par * n;
                const double r = s_reach[level_offset(d) + par];
                s_reach[level_offset(d + 1) + j] = p == br ? r : r * s_pol[s_inf[level_offset(d) + par] * 4 + a];
            }
            __syncthreads();
        }
        // terminals
        for (int j = tid; j < kTerminal; j += nt) {
            const int p0 = g_payoff[j];
            s_val[level_offset(8) + j] = 0.5 * (double)(br == 1 ? -p0 : p0);
        }
        __syncthreads();
        // bottom-up
        for (int d = kPlies - 1; d >= 0; d--) {
            const int n = nlegal_at(d), w = level_width(d), off = level_offset(d), p = d & 1;
            if (p == br) {
                // q[I][a] = sum over the infoset's nodes, in node order, of reach * value(child a): one lane per
                // (infoset, action) scans the ply (<= 576 nodes) so the order is fixed
                for (int cell = tid; cell < I * 4; cell += nt) {
                    const int r = cell >> 2, a = cell & 3;
                    if ((int)(g_key[r] & 1) != p || (int)((g_key[r] >> 1) & 7) != n || a >= n) continue;
                    double q = 0.0;
                    for (int j = 0; j < w; j++)
                        if (s_inf[off + j] == r) q += s_reach[off + j] * s_val[level_offset(d + 1) + j * n + a];
                    s_q[cell] = q;
                }
                __syncthreads();
                for (int r = tid; r < I; r += nt) {
                    if ((int)(g_key[r] & 1) != p || (int)((g_key[r] >> 1) & 7) != n) continue;
                    int best = 0;
                    for (int a = 1; a < n; a++) if (s_q[r * 4 + a] > s_q[r * 4 + best]) best = a;
                    s_choice[r] = best;
                }
                __syncthreads();
                for (int j = tid; j < w; j += nt) s_val[off + j] = s_val[level_offset(d + 1) + j * n + s_choice[s_inf[off + j]]];
            } else {
                for (int j = tid; j < w; j += nt) {
                    const int r = s_inf[off + j];
                    double v = 0.0;
                    for (int a = 0; a < n; a++) v += s_pol[r * 4 + a] * s_val[level_offset(d + 1) + j * n + a];
                    s_val[off + j] = v;
                }
            }
            __syncthreads();
        }
        if (tid == 0) out4[1 + pass] = s_val[0];
        __syncthreads();
    }
    if (tid == 0) out4[0] = 0.5 * (out4[1] + out4[2]);
}

extern "C" int32_t scopa_exploitability(scopa_ctx *ctx, const double *h_policy, double *h_out4, double *h_policy_out) {
    if (!ctx || !h_out4) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_exploitability: no deal set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const int I = ctx->n_infosets;
    const size_t pol_bytes = (size_t)I * 4 * sizeof(double);
    { const int32_t rc = ensure_scratch(ctx, 64 + 2 * pol_bytes); if (rc != SCOPA_OK) return rc; }
    double *d_out = ctx->d_scratch, *d_pin = ctx->d_scratch + 8, *d_pout = d_pin + (size_t)I * 4;
    if (h_policy) SC_HIP(ctx, hipMemcpyAsync(d_pin, h_policy, pol_bytes, hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = pol_bytes * 2 + sizeof(double) * kNodes * 2 + sizeof(int) * (size_t)I + 1656 * 2;
    SC_REQUIRE(ctx, lds <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_exploitability: tables do not fit in LDS");
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_exploitability), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        ctx->lds_limit));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_exploitability, dim3(1), dim3(1024), lds, ctx->stream, ctx->d_infoset, ctx->d_payoff, ctx->d_key,
                       ctx->d_strat, h_policy ? d_pin : nullptr, I, d_out, h_policy_out ? d_pout : nullptr);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipMemcpyAsync(h_out4, d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (h_policy_out) SC_HIP(ctx, hipMemcpyAsync(h_policy_out, d_pout, pol_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}
