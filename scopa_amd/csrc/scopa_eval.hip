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
#include "scopa_philox.h"

using namespace scopa;

__global__ void __launch_bounds__(1024)
k_exploitability(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, const uint64_t *__restrict__ g_key,
                 const double *__restrict__ g_strat, const double *__restrict__ g_policy_in, int n_infosets,
                 double *__restrict__ out4, double *__restrict__ g_policy_out, const int32_t *__restrict__ multi_meta) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (n_infosets <= 0) {  // multi-deal mode: one workgroup per deal; meta[0] of the deal = its infoset count
        const size_t deal = blockIdx.x;
        n_infosets = -n_infosets;  // the caller passes -max(n_infosets) so that LDS carving below is the same for all deals
        g_infoset += deal * kDecision; g_payoff += deal * kTerminal; g_key += deal * kDecision;
        g_strat += deal * kDecision * 4; out4 += deal * 4;
        if (g_policy_in) g_policy_in += deal * kDecision * 4;
        if (g_policy_out) g_policy_out += deal * kDecision * 4;
        // rows beyond this deal's own count are never referenced: every loop below is bounded by the deal's keys
        n_infosets = multi_meta ? multi_meta[deal * 8] : n_infosets;
    }
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
    SC_LDS_ATTR(ctx, scopa::kLdsExploit, k_exploitability, ctx->lds_limit);
    hipLaunchKernelGGL(k_exploitability, dim3(1), dim3(1024), lds, ctx->stream, ctx->d_infoset, ctx->d_payoff, ctx->d_key,
                       ctx->d_strat, h_policy ? d_pin : nullptr, I, d_out, h_policy_out ? d_pout : nullptr, (const int32_t *)nullptr);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipMemcpyAsync(h_out4, d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (h_policy_out) SC_HIP(ctx, hipMemcpyAsync(h_policy_out, d_pout, pol_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

// =====================================================================================================================
// Synchronous ("frozen strategy") CFR -- the parallel variant SURVEY §8b lists next to the exact one (H1): sigma =
// regret-match(regret) is frozen for the whole iteration and BOTH players' regrets are updated from one sweep.  It is
// NOT the reference's algorithm (whose tables depend on DFS visit order, vanilla_cfr.py:97); it is the textbook
// simultaneous-update CFR, defined by oracle/scopa_oracle.c og_cfr_sync and matched bit-for-bit: every float64 sum
// runs in the oracle's order (children left to right; an infoset's nodes in ply order).
// One workgroup per deal; regret and sigma (2 x 23.6 KB at 738 infosets), reach x2 + value (53 KB) in LDS -- fits every
// deal up to the 1653-infoset maximum -- n_iters per launch.
__global__ void __launch_bounds__(1024)
k_cfr_sync(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, const uint64_t *__restrict__ g_key,
           double *__restrict__ g_regret, double *__restrict__ g_strat, int n_infosets, int n_iters,
           unsigned long long *__restrict__ g_counters, uint32_t *__restrict__ g_visit, int32_t *__restrict__ g_meta) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (n_infosets <= 0) {  // multi-deal mode: one workgroup per deal
        const size_t deal = blockIdx.x;
        g_infoset += deal * kDecision; g_payoff += deal * kTerminal; g_key += deal * kDecision;
        g_regret += deal * kDecision * 4; g_strat += deal * kDecision * 4;
        g_visit += deal * kDecision; g_meta += deal * 8; g_counters += deal * 8;
        n_infosets = g_meta[0];
    }
    const int I = n_infosets, tid = threadIdx.x, nt = blockDim.x;
    double *s_R = reinterpret_cast<double *>(smem);   // [I][4]
    double *s_sig = s_R + (size_t)I * 4;              // [I][4]   (strategy sums stay in HBM: one RMW per cell per iteration)
    double *s_r0 = s_sig + (size_t)I * 4;             // [kNodes] reach of player 0 (BFS order)
    double *s_r1 = s_r0 + kNodes;                     // [kNodes]
    double *s_val = s_r1 + kNodes;                    // [kNodes] value for player 0
    uint16_t *s_inf = reinterpret_cast<uint16_t *>(s_val + kNodes);  // [1653]
    for (int i = tid; i < I * 4; i += nt) s_R[i] = g_regret[i];
    for (int i = tid; i < kDecision; i += nt) s_inf[i] = g_infoset[i];
    __syncthreads();
    for (int it = 0; it < n_iters; it++) {
        for (int r = tid; r < I; r += nt) {  // InfoNode.get_strategy (vanilla_cfr.py:23-30)
            const int n = (int)((g_key[r] >> 1) & 7);
            double pos[4] = {0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < n; c++) pos[c] = s_R[r * 4 + c] > 0.0 ? s_R[r * 4 + c] : 0.0;
            double s = pos[0];
            for (int c = 1; c < n; c++) s += pos[c];
            for (int c = 0; c < 4; c++) s_sig[r * 4 + c] = c < n ? (s > 0.0 ? pos[c] / s : 1.0 / (double)n) : 0.0;
        }
        if (tid == 0) { s_r0[0] = 1.0; s_r1[0] = 1.0; }
        __syncthreads();
        for (int d = 0; d < kPlies; d++) {  // reach probabilities, top down
            const int n = nlegal_at(d), w1 = level_width(d + 1), p = d & 1;
            for (int j = tid; j < w1; j += nt) {
                const int par = j / n, a = j - par * n;
                const double sg = s_sig[s_inf[level_offset(d) + par] * 4 + a];
                const double a0 = s_r0[level_offset(d) + par], a1 = s_r1[level_offset(d) + par];
                s_r0[level_offset(d + 1) + j] = p == 0 ? a0 * sg : a0;
                s_r1[level_offset(d + 1) + j] = p == 1 ? a1 * sg : a1;
            }
            __syncthreads();
        }
        for (int j = tid; j < kTerminal; j += nt) s_val[level_offset(8) + j] = 0.5 * (double)g_payoff[j];
        __syncthreads();
        for (int d = kPlies - 1; d >= 0; d--) {  // values bottom up, then this ply's regret / strategy increments
            const int n = nlegal_at(d), w = level_width(d), off = level_offset(d), p = d & 1;
            for (int j = tid; j < w; j += nt) {
                const int r = s_inf[off + j];
                double v = 0.0;
                for (int a = 0; a < n; a++) v += s_sig[r * 4 + a] * s_val[level_offset(d + 1) + j * n + a];
                s_val[off + j] = v;
            }
            __syncthreads();
            const double sgn = p == 0 ? 1.0 : -1.0;
            for (int cell = tid; cell < I * 4; cell += nt) {
                const int r = cell >> 2, a = cell & 3;
                if ((int)(g_key[r] & 1) != p || (int)((g_key[r] >> 1) & 7) != n || a >= n) continue;
                double dR = 0.0, dS = 0.0;
                const double sg = s_sig[cell];
                for (int j = 0; j < w; j++) {
                    if (s_inf[off + j] != r) continue;
                    const double reach = p == 0 ? s_r0[off + j] : s_r1[off + j], opp = p == 0 ? s_r1[off + j] : s_r0[off + j];
                    dR += opp * (sgn * (s_val[level_offset(d + 1) + j * n + a] - s_val[off + j]));
                    dS += reach * sg;
                }
                s_R[cell] += dR;   // sigma is already frozen in s_sig, so the tables can be updated in place
                g_strat[cell] += dS;
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < I * 4; i += nt) g_regret[i] = s_R[i];
    for (int r = tid; r < I; r += nt) if (n_iters > 0 && g_visit[r] == 0u) g_visit[r] = 0x40000000u + (uint32_t)r;
    if (tid == 0) { g_counters[0] += (unsigned long long)kDecision * n_iters; g_counters[1] += (unsigned long long)kTerminal * n_iters; }
    (void)g_meta;
}

// =====================================================================================================================
// Batched evaluation of a TABULAR policy against uniform random (evaluate_agent, vanilla_cfr.py:157-216 /
// mc_cfr.py:146-206): n episodes of the context's deal in lockstep.  Each lane keeps the packed state (advanced
// with the same device step as everything else) and its tree index, so the trained seat's policy row is one table lookup.
// Sampling thresholds of a tabular policy, once per evaluation: np.random.choice(actions, p = probs) is index = #{q : cdf_q / cdf_last <= u}, and every u an
// episode draws is N * 2^-53 with an integer N (u53 of two Philox words); x * 2^53 is exact in float64, so x <= u  <=>  ceil(x * 2^53) <= N -- the three
// float64 divisions of a visit become three integer compares against rows computed here, the same answer bit for bit (the SDCFR policy table's idea).
// thr[r][k] = 2^53 (never counted) for k >= n - 1 and for rows whose probabilities sum to 0 or NaN (the division gives NaN there and no compare holds).
__global__ void __launch_bounds__(256)
k_eval_thresholds(const uint64_t *__restrict__ g_key, const double *__restrict__ policy /*[I][4]*/, int n_infosets, unsigned long long *__restrict__ thr /*[I][3]*/) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_infosets) return;
    const int n = (int)((g_key[r] >> 1) & 7);
    const double *row = policy + (size_t)r * 4;
    double c = 0.0, cdf[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < n; q++) { c = q ? c + row[q] : row[0]; cdf[q] = c; }
    const double last = n > 0 ? cdf[n - 1] : 0.0;
    for (int k = 0; k < 3; k++) {
        unsigned long long t = 1ull << 53;
        if (k < n - 1) {
            const double x = cdf[k] / last;
            if (x <= 0.0) t = 0ull;                                  // x <= u for every u >= 0
            else if (x < 1.0) t = (unsigned long long)ceil(x * 9007199254740992.0);
            // x >= 1 or NaN: never <= u (u < 1)
        }
        thr[(size_t)r * 3 + k] = t;
    }
}

// THR: the trained seat samples by the prepared integer thresholds (scopa_eval_tabular_prepare) instead of by float64 divisions of the policy row
template <bool THR>
__global__ void __launch_bounds__(256)
k_eval_tabular_step(scopa_state *__restrict__ states, int32_t *__restrict__ node_idx, long long n, int ply,
                    const uint16_t *__restrict__ g_infoset, const double *__restrict__ policy /*[I][4]*/, const unsigned long long *__restrict__ thr /*[I][3]*/,
                    const int32_t *__restrict__ trained_seat, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 sw = reinterpret_cast<const uint4 *>(states)[i];     // the state as its four words (scopa_rules.h step_words): no struct member indexed by the mover
    if ((sw.z & 0xFFFFu) == 0u || ((sw.z >> 24) & SCOPA_STEP_COUNT_MASK) >= (((sw.z >> 24) & SCOPA_STEP_CLONED) ? 16u : 8u)) return;   // terminal
    const int p = (int)((sw.z >> 24) & 1u), nl = (int)((sw.z >> (8 * p)) & 255u);
    const int idx = node_idx[i];
    const philox_out x = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)ply, stream, seed_lo, seed_hi);
    int k = nl - 1;
    if (p == trained_seat[i]) {  // np.random.choice(actions, p=probs): cumsum, normalise, searchsorted right
        int a = 0;
        if (THR) {
            const unsigned long long *t = thr + (size_t)g_infoset[level_offset(ply) + idx] * 3;
            const unsigned long long N = ((unsigned long long)(x.x0 >> 5) << 26) | (unsigned long long)(x.x1 >> 6);   // u = N * 2^-53
            a = (int)(t[0] <= N) + (int)(t[1] <= N) + (int)(t[2] <= N);
        } else {
            const double u = u53(x.x0, x.x1);
            const double *row = policy + (size_t)g_infoset[level_offset(ply) + idx] * 4;
            double c = 0.0, cdf[4];
            for (int q = 0; q < nl; q++) { c = q ? c + row[q] : row[0]; cdf[q] = c; }
            const double last = cdf[nl - 1];
            for (int q = 0; q < nl; q++) if (cdf[q] / last <= u) a = q + 1;
        }
        k = a < nl - 1 ? a : nl - 1;
    } else {  // uniform opponent
        const double u = u53(x.x0, x.x1);
        const int a = (int)(u * (double)nl);
        k = a < nl - 1 ? a : nl - 1;
    }
    step_words(sw.x, sw.y, sw.z, sw.w, nib((sw.x >> (16 * p)) & 0xFFFFu, k));
    reinterpret_cast<uint4 *>(states)[i] = sw;
    node_idx[i] = idx * nl + k;
}

// The whole seat-swapped match in ONE launch.  Every episode plays the context's deal, so its state after any prefix of moves is a node of the deal's tree: an
// episode is a walk over node indices -- per ply one Philox draw (the per-ply kernels' stream: episode, ply, stream_id) and, for the trained seat, three integer
// compares against its infoset's thresholds, both tables in LDS; plies 6 and 7 have one legal card and draw nothing.  No state is read or written per ply (the
// per-ply form moves 44 B per episode-ply and is bound by the vector instructions of the move itself); the episode's final state is the tree's terminal node,
// and what evaluate_agent reports is summed here as integers (rewards x2: exact), per seat half: stats[seat][0..4] = episodes, sum r2, sum r2^2, sum of the
// trained side's scopas, sum of the opponent's.  Episode i < n_seat0 has the trained policy in seat 0.
__global__ void __launch_bounds__(256)
k_eval_tabular_match(long long n, long long n_seat0, const uint16_t *__restrict__ g_infoset, const unsigned long long *__restrict__ g_thr /*[I][3]*/, int n_infosets,
                     const scopa_state *__restrict__ tree_states, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream,
                     scopa_state *__restrict__ out_states, int32_t *__restrict__ out_idx, unsigned long long *__restrict__ stats /*[2][5]*/) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *s_thr = reinterpret_cast<unsigned long long *>(smem);                 // [n_infosets][3]
    uint16_t *s_inf = reinterpret_cast<uint16_t *>(smem + (size_t)n_infosets * 24);           // [kDecision]
    __shared__ unsigned long long s_stats[10];
    for (int t = threadIdx.x; t < n_infosets * 3; t += blockDim.x) s_thr[t] = g_thr[t];
    for (int t = threadIdx.x; t < kDecision; t += blockDim.x) s_inf[t] = g_infoset[t];
    if (threadIdx.x < 10) s_stats[threadIdx.x] = 0ull;
    __syncthreads();
    long long acc[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int seat = i >= n_seat0;
        int idx = 0;
#pragma unroll
        for (int ply = 0; ply < 6; ply++) {
            const int nl = nlegal_at(ply);
            const philox_out x = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)ply, stream, seed_lo, seed_hi);
            int a;
            if ((ply & 1) == seat) {  // np.random.choice(actions, p=probs) through the prepared thresholds (k_eval_tabular_step<true>)
                const unsigned long long *t = s_thr + (size_t)s_inf[level_offset(ply) + idx] * 3;
                const unsigned long long N = ((unsigned long long)(x.x0 >> 5) << 26) | (unsigned long long)(x.x1 >> 6);
                a = (int)(t[0] <= N) + (int)(t[1] <= N) + (int)(t[2] <= N);
            } else {                  // uniform opponent
                a = (int)(u53(x.x0, x.x1) * (double)nl);
            }
            idx = idx * nl + (a < nl - 1 ? a : nl - 1);
        }
        const uint4 tw = reinterpret_cast<const uint4 *>(tree_states)[kDecision + idx];      // terminal node idx of ply 8 (plies 6, 7: one child each)
        const int r0 = (int)(tw.w & 255u) + 2 * (int)((tw.w >> 16) & 255u), r1 = (int)((tw.w >> 8) & 255u) + 2 * (int)(tw.w >> 24);
        const int mine = seat ? r1 - r0 : r0 - r1;                                           // evaluate_game x 2: 2 r_i - (r_0 + r_1)
        const int sc_t = (int)((tw.w >> (16 + 8 * seat)) & 255u), sc_o = (int)((tw.w >> (24 - 8 * seat)) & 255u);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const bool on = seat == h;
            acc[h][0] += on ? 1 : 0; acc[h][1] += on ? mine : 0; acc[h][2] += on ? mine * mine : 0; acc[h][3] += on ? sc_t : 0; acc[h][4] += on ? sc_o : 0;
        }
        if (out_states) reinterpret_cast<uint4 *>(out_states)[i] = tw;
        if (out_idx) out_idx[i] = idx;
    }
#pragma unroll
    for (int j = 0; j < 10; j++) {
        long long v = acc[j / 5][j % 5];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v != 0) atomicAdd(&s_stats[j], (unsigned long long)v);
    }
    __syncthreads();
    if (threadIdx.x < 10 && s_stats[threadIdx.x] != 0ull) atomicAdd(&stats[threadIdx.x], s_stats[threadIdx.x]);
}

extern "C" {

int32_t scopa_cfr_sync_iterate(scopa_ctx *ctx, int32_t n_iters) {
    if (!ctx || n_iters < 0 || n_iters > (1 << 24)) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_cfr_sync_iterate: no deal set");
    if (n_iters == 0) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds = (size_t)ctx->n_infosets * 4 * 8 * 2 + sizeof(double) * kNodes * 3 + 1656 * 2;
    SC_REQUIRE(ctx, lds <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_cfr_sync_iterate: tables do not fit in LDS");
    SC_LDS_ATTR(ctx, scopa::kLdsCfrSync, k_cfr_sync, ctx->lds_limit);
    hipLaunchKernelGGL(k_cfr_sync, dim3(1), dim3(1024), lds, ctx->stream, ctx->d_infoset, ctx->d_payoff, ctx->d_key, ctx->d_regret,
                       ctx->d_strat, ctx->n_infosets, (int)n_iters, ctx->d_counters, ctx->d_visit, ctx->d_meta);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sigcdf_valid = false;
    return SCOPA_OK;
}

int32_t scopa_eval_tabular_prepare(scopa_ctx *ctx, const double *d_policy) {
    if (!ctx || !d_policy) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_eval_tabular_prepare: no deal set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_eval_thr) SC_HIP(ctx, hipMalloc(&ctx->d_eval_thr, sizeof(unsigned long long) * 3 * (size_t)kDecision));
    hipLaunchKernelGGL(k_eval_thresholds, dim3((ctx->n_infosets + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_key, d_policy, ctx->n_infosets,
                       (unsigned long long *)ctx->d_eval_thr);
    SC_HIP(ctx, hipGetLastError());
    ctx->eval_thr_valid = true;
    return SCOPA_OK;
}

int32_t scopa_eval_tabular_step(scopa_ctx *ctx, scopa_state *d_states, int32_t *d_node_idx, int64_t n, int32_t ply,
                                const double *d_policy, const int32_t *d_trained_seat, uint32_t stream_id) {
    if (!ctx || n < 0 || ply < 0 || ply >= kPlies || (n && (!d_states || !d_node_idx || !d_trained_seat))) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_eval_tabular_step: no deal set");
    SC_REQUIRE(ctx, d_policy || ctx->eval_thr_valid, SCOPA_ESTATE, "scopa_eval_tabular_step: no policy given and none prepared (scopa_eval_tabular_prepare)");
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (d_policy)
        hipLaunchKernelGGL(k_eval_tabular_step<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_states, d_node_idx, (long long)n, (int)ply,
                           ctx->d_infoset, d_policy, (const unsigned long long *)nullptr, d_trained_seat, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), stream_id);
    else
        hipLaunchKernelGGL(k_eval_tabular_step<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_states, d_node_idx, (long long)n, (int)ply,
                           ctx->d_infoset, (const double *)nullptr, (const unsigned long long *)ctx->d_eval_thr, d_trained_seat, (uint32_t)ctx->seed,
                           (uint32_t)(ctx->seed >> 32), stream_id);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_eval_tabular_match(scopa_ctx *ctx, int64_t n, int64_t n_seat0, uint32_t stream_id, scopa_state *d_states_out, int32_t *d_node_idx_out,
                                 int64_t h_stats[10]) {
    if (!ctx || !h_stats || n < 0 || n_seat0 < 0 || n_seat0 > n) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_eval_tabular_match: no deal set");
    SC_REQUIRE(ctx, ctx->eval_thr_valid, SCOPA_ESTATE, "scopa_eval_tabular_match: no policy prepared (scopa_eval_tabular_prepare)");
    SC_REQUIRE(ctx, !d_states_out || ((uintptr_t)d_states_out & 15) == 0, SCOPA_EINVAL, "scopa_eval_tabular_match: states must be 16-byte aligned");
    for (int j = 0; j < 10; j++) h_stats[j] = 0;
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    { const int32_t rc = ensure_scratch(ctx, 128); if (rc != SCOPA_OK) return rc; }
    unsigned long long *d_stats = reinterpret_cast<unsigned long long *>(ctx->d_scratch);   // 80 bytes of the context's scratch
    SC_HIP(ctx, hipMemsetAsync(d_stats, 0, sizeof(unsigned long long) * 10, ctx->stream));
    const size_t lds = (size_t)ctx->n_infosets * 24 + sizeof(uint16_t) * kDecision;
    const long long blocks = std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_eval_tabular_match, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, (long long)n, (long long)n_seat0, ctx->d_infoset,
                       (const unsigned long long *)ctx->d_eval_thr, ctx->n_infosets, ctx->d_states, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), stream_id,
                       d_states_out, d_node_idx_out, d_stats);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipMemcpyAsync(h_stats, d_stats, sizeof(int64_t) * 10, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

}  // extern "C"
