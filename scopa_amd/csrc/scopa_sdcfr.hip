// scopa_sdcfr.hip -- external-sampling traversal for Single Deep CFR, level-synchronous over the 8 plies.  Three forms, same rows:
//   (1) ply by ply around a PyTorch forward (k_sdcfr_features / _expand / _terminal / _backward, first part of this file);
//   (2) one launch with a forward pass per visit on the matrix cores (k_sdcfr_traverse);
//   (3) the default: every decision node of the deal evaluated once per launch (k_sdcfr_policy), the traversals as walks over that
//       policy table (k_sdcfr_walk) -- the nets are frozen during a launch and a node's features depend on the tree node alone.
//
// Reference behaviour: DeepCFR._external_sampling_cfr / ._state_to_features / ._get_legal_actions_mask
// (src/algorithms/deep_cfr/deep_cfr.py:213-365), AdvantageNetwork.add_experience (:70-75),
// positive_regret_policy (src/algorithms/deep_cfr/nets.py:93-101).  The reference walks ONE traversal depth-first
// and runs a batch-1 MLP forward per node (its whole traversal time).  Here B traversals advance together: per ply
//   k_sdcfr_features : frontier nodes -> feats[N][34], mask[N][16]            (then ONE batched MLP forward in PyTorch)
//   k_sdcfr_expand   : advantages -> regret-matching policy; traverser ply: all legal children,
//                      opponent ply: one sampled child
// and after the last ply the values flow back up with k_sdcfr_backward, which also emits the advantage-memory rows
// (features, max-abs-normalised regrets incl. the reference's "-value at illegal slots" quirk, mask) straight into
// the caller's ring buffers in the reference's append order (DFS post-order within a traversal).
// A traversal's frontier is regular: widths 1,4,4,12,12,24,24,24,24 (traverser 0) / 1,1,4,4,12,12,24,24,24
// (traverser 1); slot s of ply d has children s*n+k (traverser ply) or s (opponent ply).  All arithmetic is float32
// in the reference's operation order (NEP-50 scalar rules make `value` float32 there).
#include "scopa_ctx.h"
#include "scopa_philox.h"

using namespace scopa;

namespace {

__host__ __device__ __forceinline__ int frontier_width(int traverser, int ply) {
    int w = 1;
    for (int d = 0; d < ply; d++)
        if ((d & 1) == traverser) w *= nlegal_at(d);
    return w;
}

}  // namespace

// ---- features + mask (deep_cfr.py:213-282) ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sdcfr_features(const scopa_state *__restrict__ g_states, int ply, long long n, const int32_t *__restrict__ idx,
                 float *__restrict__ feats, float *__restrict__ mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const scopa_state s = idx ? g_states[level_offset(ply) + idx[i]] : g_states[i];  // tree node, or a free-standing state
    const int p = s.step & 1;                                                           // == ply & 1 on the tree
    uint32_t hand_bits = 0, table_bits = 0;
    for (int k = 0; k < s.nh[p]; k++) hand_bits |= 1u << nib(s.hand[p], k);
    for (int k = 0; k < s.nt; k++) table_bits |= 1u << nib(s.table, k);
    float *f = feats + i * 34;
    float *m = mask + i * 16;
    for (int c = 0; c < 16; c++) {
        const float h = (float)((hand_bits >> c) & 1u);
        f[c] = h;                                   // hand one-hot by card id (order-free)
        f[16 + c] = (float)((table_bits >> c) & 1u);  // table multi-hot
        m[c] = h;                                   // legal actions = cards in hand (openspiel_mini_scopa.py:22-47)
    }
    f[32] = 1.0f;  // float(player == state.current_player()): features are always taken for the player to move
    f[33] = 0.0f;
}

// ---- policy + expansion (deep_cfr.py:315-365) --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sdcfr_expand(const scopa_state *__restrict__ g_states, int ply, int traverser, long long n, const int32_t *__restrict__ idx,
               const float *__restrict__ adv, int32_t *__restrict__ child_idx, float *__restrict__ pol,
               const double *__restrict__ uniforms, uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t b0, int width) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const scopa_state s = g_states[level_offset(ply) + idx[i]];
    const int p = ply & 1, nl = s.nh[p];  // == nlegal_at(ply) on the tree
    // positive_regret_policy (nets.py:93-101) on advantages*mask - 1e6*(1-mask): relu kills the illegal slots
    float pos[4], z = 0.0f;
    {
        uint32_t hand_bits = 0;
        for (int k = 0; k < nl; k++) hand_bits |= 1u << nib(s.hand[p], k);
        for (int c = 0; c < 16; c++) {  // sum over all 16 outputs in index order
            const float a = adv[i * 16 + c];
            z += ((hand_bits >> c) & 1u) ? (a > 0.0f ? a : 0.0f) : 0.0f;
        }
    }
    const float zc = z > 1e-8f ? z : 1e-8f;  // clamp_min(eps)
    for (int k = 0; k < 4; k++) {
        float v = 0.0f;
        if (k < nl) { const float a = adv[i * 16 + nib(s.hand[p], k)]; v = (a > 0.0f ? a : 0.0f) / zc; }
        pos[k] = v;
        pol[i * 4 + k] = v;
    }
    if (p == traverser) {  // recurse on ALL legal actions, hand order (:326-336)
        for (int k = 0; k < nl; k++) child_idx[i * nl + k] = idx[i] * nl + k;
        return;
    }
    // opponent: sample ONE action (:347-365)
    float sum = pos[0];
    for (int k = 1; k < nl; k++) sum += pos[k];  // action_probs.sum(), float32, left to right
    double u;
    if (uniforms) u = uniforms[i];
    else {
        const uint32_t b = b0 + (uint32_t)(i / width), slot = (uint32_t)(i % width);
        const philox_out x = philox4x32_10(slot + 1024u * (uint32_t)ply, b, iteration, 4u + (uint32_t)traverser, seed_lo, seed_hi);
        u = u53(x.x0, x.x1);
    }
    int a;
    if (sum == 0.0f) {  // np.random.choice(legal_actions): uniform
        a = (int)(u * (double)nl);
        a = a < nl - 1 ? a : nl - 1;
    } else {  // np.random.choice(legal, p=action_probs / sum): float32 p, float64 cdf
        double c = 0.0, cdf[4];
        for (int k = 0; k < nl; k++) { const double pk = (double)(pos[k] / sum); c = k ? c + pk : pk; cdf[k] = c; }
        const double last = cdf[nl - 1];
        a = 0;
        for (int k = 0; k < nl; k++) if (cdf[k] / last <= u) a = k + 1;
        a = a < nl - 1 ? a : nl - 1;
    }
    child_idx[i] = idx[i] * nl + a;
}

// ---- terminal values: float(rewards[player]) (:286-293) -----------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sdcfr_terminal(const int8_t *__restrict__ g_payoff, int traverser, long long n, const int32_t *__restrict__ idx, float *__restrict__ val) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p0 = g_payoff[idx[i]];
    val[i] = 0.5f * (float)(traverser == 0 ? p0 : -p0);
}

// ---- backward: node values and advantage-memory rows (:321-346, :70-75) ----------------------------------------------------
__global__ void __launch_bounds__(256)
k_sdcfr_backward(const scopa_state *__restrict__ g_states, int ply, int traverser, long long n, const int32_t *__restrict__ idx,
                 const float *__restrict__ pol, const float *__restrict__ child_val, float *__restrict__ val,
                 const float *__restrict__ feats, const float *__restrict__ mask, float *__restrict__ mem_feat,
                 float *__restrict__ mem_regret, float *__restrict__ mem_mask, long long capacity, long long write_base, int width) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = ply & 1;
    if (p != traverser) { val[i] = child_val[i]; return; }  // the sampled child's value is returned unchanged (:363-365)
    const scopa_state s = g_states[level_offset(ply) + idx[i]];
    const int nl = s.nh[p];
    float value = 0.0f, cfv[16];
    for (int c = 0; c < 16; c++) cfv[c] = 0.0f;  // counterfactual_values = zeros(16) (:324)
    for (int k = 0; k < nl; k++) {
        const float av = child_val[i * nl + k];
        value += pol[i * 4 + k] * av;            // value += policy[action] * action_value, float32 (:335)
        cfv[nib(s.hand[p], k)] = av;
    }
    val[i] = value;
    float mx = 0.0f, reg[16];
    for (int c = 0; c < 16; c++) { reg[c] = cfv[c] - value; const float a = fabsf(reg[c]); mx = a > mx ? a : mx; }  // illegal slots = -value
    if (mx > 0.0f) { const float den = mx + 1e-8f; for (int c = 0; c < 16; c++) reg[c] = reg[c] / den; }        // add_experience (:73-74)
    // ring position: the reference appends in DFS post-order; rank of this traverser node within its traversal
    const int m = (ply - traverser) >> 1;      // 0..3: which traverser ply
    const long long b = i / width;
    int j = (int)(i % width);
    const int T[4] = {41, 10, 3, 1};           // traverser nodes in the subtree of a traverser node of ply index m
    int rank = T[m] - 1;
    for (int q = m - 1; q >= 0; q--) {         // digits of j, radices 4,3,2 for q = 0,1,2
        const int radix = 4 - q;
        rank += (j % radix) * T[q + 1];
        j /= radix;
    }
    long long row = (write_base + b * 41 + rank) % capacity;
    for (int c = 0; c < 34; c++) mem_feat[row * 34 + c] = feats[i * 34 + c];
    for (int c = 0; c < 16; c++) mem_regret[row * 16 + c] = reg[c];
    if (mem_mask) for (int c = 0; c < 16; c++) mem_mask[row * 16 + c] = mask[i * 16 + c];   // no mask array: the row's mask is features[0..16) (sd_mask_note)
}

// ---- batched play vs a uniform-random opponent: states advance with the same device step as everything else -------------
// (evaluate_vs_random, deep_cfr.py:367-429).  probs[N][16]: the trained seat's policy at the lanes where it
// moves (ignored elsewhere); one lane per episode.
__global__ void __launch_bounds__(256)
k_eval_step(scopa_state *__restrict__ states, long long n, const float *__restrict__ probs, const int32_t *__restrict__ trained_seat,
            uint32_t seed_lo, uint32_t seed_hi, uint32_t stream, uint32_t ply_tag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    scopa_state s = states[i];
    if (is_terminal(s)) return;
    const int p = s.step & 1, nl = s.nh[p];
    const philox_out x = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), ply_tag, stream, seed_lo, seed_hi);
    const double u = u53(x.x0, x.x1);
    int k;
    double w[4], tot = 0.0;
    for (int q = 0; q < nl; q++) { w[q] = (p == trained_seat[i] && probs) ? (double)probs[i * 16 + nib(s.hand[p], q)] : 1.0; if (!(w[q] > 0.0)) w[q] = 0.0; tot += w[q]; }
    if (!(tot > 0.0)) { for (int q = 0; q < nl; q++) w[q] = 1.0; tot = (double)nl; }  // nan / non-positive -> uniform (:394-395)
    double c = 0.0;
    k = nl - 1;
    for (int q = 0; q < nl; q++) { c += w[q] / tot; if (u < c) { k = q; break; } }
    step(s, nib(s.hand[p], k));
    states[i] = s;
}

extern "C" {

int32_t scopa_sdcfr_frontier_width(int32_t traverser, int32_t ply) {
    if (traverser < 0 || traverser > 1 || ply < 0 || ply > kPlies) return SCOPA_EINVAL;
    return frontier_width(traverser, ply);
}

#define SC_GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, ctx->stream

int32_t scopa_sdcfr_features(scopa_ctx *ctx, int32_t ply, int64_t n, const int32_t *d_idx, float *d_feats, float *d_mask) {
    if (!ctx || ply < 0 || ply >= kPlies || n < 0 || (n && (!d_idx || !d_feats || !d_mask))) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_sdcfr_features: no deal set");
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sdcfr_features, SC_GRID(n), ctx->d_states, (int)ply, (long long)n, d_idx, d_feats, d_mask);
    SC_HIP(ctx, hipGetLastError());
    ctx->sdcfr_visits += (uint64_t)n;
    return SCOPA_OK;
}

int32_t scopa_features_from_states(scopa_ctx *ctx, const scopa_state *d_states, int64_t n, float *d_feats, float *d_mask) {
    /* DeepCFR.get_policy's encoder (deep_cfr.py:497-504) on arbitrary (non-terminal) states, e.g. evaluation episodes */
    if (!ctx || n < 0 || (n && (!d_states || !d_feats || !d_mask))) return SCOPA_EINVAL;
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sdcfr_features, SC_GRID(n), d_states, 0, (long long)n, (const int32_t *)nullptr, d_feats, d_mask);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_sdcfr_visits(scopa_ctx *ctx, uint64_t *decision_visits) {
    if (!ctx || !decision_visits) return SCOPA_EINVAL;
    *decision_visits = ctx->sdcfr_visits;
    // the fused kernel's team barriers give up after about a second and say so here (d_counters[5]): waits for the stream
    unsigned long long err = 0;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    SC_HIP(ctx, hipMemcpyAsync(&err, ctx->d_counters + 5, sizeof err, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_REQUIRE(ctx, err == 0, SCOPA_ETIMEOUT, "scopa_sdcfr_visits: a team barrier of k_sdcfr_traverse timed out -- the traversal results since the last check are invalid");
    return SCOPA_OK;
}

int32_t scopa_sdcfr_expand(scopa_ctx *ctx, int32_t ply, int32_t traverser, int64_t n, const int32_t *d_idx, const float *d_adv,
                           int32_t *d_child_idx, float *d_pol, const double *d_uniforms, uint32_t iteration, uint32_t b0) {
    if (!ctx || ply < 0 || ply >= kPlies || traverser < 0 || traverser > 1 || n < 0 || (n && (!d_idx || !d_adv || !d_child_idx || !d_pol)))
        return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_sdcfr_expand: no deal set");
    const int width = frontier_width(traverser, ply);
    SC_REQUIRE(ctx, n % width == 0, SCOPA_EINVAL, "scopa_sdcfr_expand: n is not a multiple of the ply's frontier width");
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sdcfr_expand, SC_GRID(n), ctx->d_states, (int)ply, (int)traverser, (long long)n, d_idx, d_adv, d_child_idx,
                       d_pol, d_uniforms, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), iteration, b0, width);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_sdcfr_terminal_values(scopa_ctx *ctx, int32_t traverser, int64_t n, const int32_t *d_idx, float *d_val) {
    if (!ctx || traverser < 0 || traverser > 1 || n < 0 || (n && (!d_idx || !d_val))) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_sdcfr_terminal_values: no deal set");
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sdcfr_terminal, SC_GRID(n), ctx->d_payoff, (int)traverser, (long long)n, d_idx, d_val);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_sdcfr_backward(scopa_ctx *ctx, int32_t ply, int32_t traverser, int64_t n, const int32_t *d_idx, const float *d_pol,
                             const float *d_child_val, float *d_val, const float *d_feats, const float *d_mask, float *d_mem_feat,
                             float *d_mem_regret, float *d_mem_mask, int64_t capacity, int64_t write_base) {
    if (!ctx || ply < 0 || ply >= kPlies || traverser < 0 || traverser > 1 || n < 0 || (n && (!d_idx || !d_pol || !d_child_val || !d_val)))
        return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_sdcfr_backward: no deal set");
    const bool trav_ply = (ply & 1) == traverser;
    if (trav_ply && n)
        SC_REQUIRE(ctx, d_feats && d_mask && d_mem_feat && d_mem_regret && capacity >= 41 && write_base >= 0, SCOPA_EINVAL,
                   "scopa_sdcfr_backward: memory buffers required at a traverser ply");
    const int width = frontier_width(traverser, ply);
    SC_REQUIRE(ctx, n % width == 0, SCOPA_EINVAL, "scopa_sdcfr_backward: n is not a multiple of the ply's frontier width");
    if (trav_ply) SC_REQUIRE(ctx, (n / width) * 41 <= capacity, SCOPA_EINVAL, "scopa_sdcfr_backward: batch larger than the memory ring");
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sdcfr_backward, SC_GRID(n), ctx->d_states, (int)ply, (int)traverser, (long long)n, d_idx, d_pol, d_child_val, d_val,
                       d_feats, d_mask, d_mem_feat, d_mem_regret, d_mem_mask, (long long)capacity, (long long)write_base, width);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_eval_init_states(scopa_ctx *ctx, scopa_state *d_states, int64_t n) {
    if (!ctx || n < 0 || (n && !d_states)) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_eval_init_states: no deal set");
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    // every episode starts from the context's deal: broadcast the root state (d_states[0] of the tree)
    scopa_state root;
    state_init(root, ctx->perm);
    std::vector<scopa_state> h((size_t)(n < 65536 ? n : 65536), root);
    for (int64_t off = 0; off < n; off += (int64_t)h.size()) {
        const size_t cnt = (size_t)((n - off) < (int64_t)h.size() ? (n - off) : (int64_t)h.size());
        SC_HIP(ctx, hipMemcpyAsync(d_states + off, h.data(), cnt * sizeof(scopa_state), hipMemcpyHostToDevice, ctx->stream));
    }
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_eval_step(scopa_ctx *ctx, scopa_state *d_states, int64_t n, const float *d_probs, const int32_t *d_trained_seat,
                        uint32_t stream_id, uint32_t ply_tag) {
    if (!ctx || n < 0 || (n && (!d_states || !d_trained_seat))) return SCOPA_EINVAL;
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_eval_step, SC_GRID(n), d_states, (long long)n, d_probs, d_trained_seat, (uint32_t)ctx->seed,
                       (uint32_t)(ctx->seed >> 32), stream_id, ply_tag);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

}  // extern "C"

// =====================================================================================================================
// Fused traversal: ONE launch per traversal batch instead of ~65 (8 plies x {features, 3 GEMMs + activations, expand} +
// 8 backward steps).  Both players' advantage MLPs (34-128-64-16, float32) sit in LDS for the whole launch and the forward
// pass runs on the matrix cores with v_mfma_f32_16x16x4_f32 on tiles of SIXTEEN frontier nodes:
//
//   * a WAVEFRONT walks T traversals together (T = 4: frontier widths 4,16,16,48,48,96,96 -> 21 tiles at 96 % fill; one
//     traversal alone is 1,4,4,12,12,24,24 wide and would leave most of a 16-node tile empty), level-synchronously;
//   * MFMA roles: rows i = units (weights = A operand), columns j = nodes (activations = B operand).  The result layout of
//     this instruction -- lane l, register r holds row 4*(l/16)+r of column l%16 -- is exactly its B-operand layout for a
//     K-step that covers the four units {4q + r : q = 0..3}: so the 32 accumulator registers of layer 1 ARE the B operands of
//     layer 2's 32 K-steps and the 16 of layer 2 are layer 3's.  Activations never leave the registers; the only LDS traffic
//     of the forward pass is the weights, as one conflict-free 16-byte read per lane per four MFMAs (the weight image is laid
//     out for that by k_sdcfr_pack: scopa_sdcfr_pack_weights);
//   * per 16-node tile: 64 + 128 + 16 MFMAs (8, 4 and 2 independent accumulator chains), 16 + 32 + 4 weight reads, 12 bias
//     reads.  The round-2 kernel (v_mfma_f32_4x4x1, four nodes per group, activations through LDS) needed one 16-byte LDS
//     read per MFMA and its matrix-core time equalled its LDS-array time; this form reads 6.4 x less per FLOP.
//   layer 1  K = 32 dense 0/1 inputs from the node's hand/table bit masks (a zero input adds an exact zero); the bias and the
//            constant-1 feature's column are the initial accumulator (feature 33 is always 0)
//   layer 2  K = 128, 64 units; layer 3  K = 64, 16 outputs; regret matching (sum of relu(adv)*mask over a node's 16 outputs)
//            in the accumulator layout + two cross-lane adds
// Opponent nodes with a single legal action skip the forward pass: their child is forced and nothing else of them is used.
// Expansion / sampling is exactly k_sdcfr_expand's (same Philox keying, so the two paths sample identical actions); after
// ply 7 the values flow back up inside the wave and the memory rows go straight to the caller's ring (8- / 16-byte stores).
namespace {
// the net image in LDS, per player (float32), written by k_sdcfr_pack from the torch tensors W[out][in]:
constexpr int kImgW1 = 0;                 // [8 mt][2 g][64 lanes][4 c] : W1[16 mt + lane%16][4 (4g + c) + lane/16]       4096
constexpr int kImgC1 = kImgW1 + 4096;     // [128]                      : b1[u] + W1[u][32]  (feature 32 is the constant 1)  128
constexpr int kImgW2 = kImgC1 + 128;      // [4 nt][8 mt][64 lanes][4 r]: W2[16 nt + lane%16][16 mt + 4 (lane/16) + r]     8192
constexpr int kImgB2 = kImgW2 + 8192;     // [64]                                                                            64
constexpr int kImgW3 = kImgB2 + 64;       // [4 nt][64 lanes][4 r]      : W3[lane%16][16 nt + 4 (lane/16) + r]             1024
constexpr int kImgB3 = kImgW3 + 1024;     // [16]                                                                            16
constexpr int kImgFloats = kImgB3 + 16;   // 13 520 (the 13 776 parameters less W1's columns 32, 33)
static_assert(kImgFloats == SCOPA_SDCFR_IMAGE_FLOATS, "include/scopa.h states the image size");

template <int T, int W>
struct alignas(16) SdTeam {     // scratch of one TEAM (W wavefronts walking one task = T traversals); a frontier node is addressed by its POSITION f = t * width + j
    float pol_trav[T][64];      // policy (legal actions, hand order) of every traverser node, packed: a node of traverser ply m = 0..3 has 4 - m legal
                                // actions, so the 1, 4, 12, 24 nodes of a traversal take 4 + 12 + 24 + 24 = 64 floats, ply m from offset 0, 4, 16, 40
    uint16_t hand_trav[T][41];  // the same nodes' hand nibbles (all the backward pass needs of a state), plies m = 0..3 at offsets 0, 1, 5, 17
    uint16_t pad0[T == 4 ? 4 : 6];
    float val[W > 1 ? 2 : 1][W > 1 ? T * 24 : 4];   // values of the frontier flowing back up, by position.  A team's wavefronts work on a ply's positions
                                // side by side: two buffers here.  A SOLO wavefront keeps them in its `pos` area (dead once the forward pass is over) and
                                // replaces them IN PLACE: node f reads its children f nl + k >= f, the 64 lanes read before any of them writes (LDS executes
                                // a wavefront's operations in order), and a second round (positions >= 64) reads positions >= 64 nl, which the first did not write
    uint16_t idx[2][T * 24];    // tree index of the frontier nodes of the current / the next ply
    uint32_t bar;               // arrivals at the team's barriers so far (monotonic)
    int32_t task;               // the task the team walks next
    uint32_t pad[2];
};
struct alignas(16) SdPos { float pos[16][16]; };   // per wavefront: relu(adv) * mask of the tile in flight, [node][output]
static_assert(sizeof(SdTeam<4, 1>) % 16 == 0 && sizeof(SdTeam<2, 1>) % 16 == 0 && sizeof(SdTeam<4, 2>) % 16 == 0 && sizeof(SdTeam<2, 2>) % 16 == 0, "SdTeam alignment");
constexpr int kSdNodeSlots = (kDecision + 1) & ~1;   // the node table in LDS, padded to 16 bytes
static_assert(kTerminal % 16 == 0, "the payoff table in LDS keeps the team scratch behind it 16-byte aligned and is copied four bytes at a time");
// wavefronts per workgroup: twelve (three per SIMD, 168 registers each) -- as solo wavefronts (W = 1; 2.8 KB of scratch each beside the two
// nets: policies packed, values in the dead `pos` area; with eight wavefronts the matrix pipes idled a quarter of the time), as six teams of
// two or four teams of three
__host__ __device__ constexpr int sd_waves(int W) { return 12; }
// DFS post-order rank of traverser node j of traverser ply m within its traversal = its memory row (the reference appends in that order):
// (41, 10, 3, 1)[m] - 1 + the digits of j (radices 4, 3, 2 from the top) times (10, 3, 1), written out per m with constant divisors -- as a
// loop over a table it cost a scalar memory load per digit (each draining the LDS queue with it: one lgkm counter) and a division by a
// run-time radix
__device__ __forceinline__ int sd_rank(int m, int j) {
    const int jh = m == 3 ? j >> 1 : j, j1 = m >= 2 ? jh / 3 : jh, j2 = jh - 3 * j1;
    return m == 0 ? 40 : m == 1 ? 9 + 10 * j : m == 2 ? 2 + 3 * j2 + 10 * j1 : (j & 1) + 3 * j2 + 10 * j1;
}

// A wavefront's LDS operations execute in order, so lane A's store is seen by lane B's later load without any wait; this only
// keeps the compiler from moving LDS accesses across the point (the wait for a load's data is the compiler's own s_waitcnt).
__device__ __forceinline__ void sd_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Barrier of the W wavefronts of a team (W = 1: only the compiler-level ordering).  Arrival = one LDS atomic; LDS executes a
// wavefront's operations in order, so everything a wavefront wrote to LDS before arriving is in place when its arrival is seen.
// All W wavefronts of a team belong to one workgroup (resident together) and execute the same sequence of barriers.
// The wait is bounded (about a second): a wavefront that gives up reports it in *g_err and goes on, so that a defect can end in a
// wrong result that the host refuses (scopa_sdcfr_visits / scopa_ctx_synchronize report SCOPA_ETIMEOUT) but never in a hung GPU.
template <int W>
__device__ __forceinline__ void team_barrier(uint32_t *bar, uint32_t &phase, int lane, uint32_t *g_err) {
    sd_order();
    if (W > 1) {
        phase += W;
        if (lane == 0) atomicAdd(bar, 1u);
        int spins = 0;
        while ((int32_t)(*reinterpret_cast<volatile uint32_t *>(bar) - phase) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 23)) { if (lane == 0) atomicOr(g_err, 1u); break; }
        }
        sd_order();
    }
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f to_v4f(float4 x) { v4f r = {x.x, x.y, x.z, x.w}; return r; }
__device__ __forceinline__ float relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_huge_valf()); }   // one v_max (fmaxf costs two: it quiets NaNs first)
// x + (the same register of lane ^ 16), then + lane ^ 32: gfx950's row swaps instead of two trips through the LDS crossbar
__device__ __forceinline__ float sum_row_groups(float x) {
    const unsigned xi = __float_as_uint(x);
    const v2u a = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);     // rows (16 lanes) 1 <-> 0 and 3 <-> 2 of the two copies
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned si = __float_as_uint(s);
    const v2u b = __builtin_amdgcn_permlane32_swap(si, si, false, false);     // upper half <-> lower half
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
}  // namespace

// feature bits (hand one-hot | table multi-hot << 16) and the mover's hand nibbles of every decision node, BFS order: what a
// traversal needs of a node's 16-byte state, ready-made (k_sdcfr_features derives them per visit)
__global__ void __launch_bounds__(256)
k_sdcfr_nodeinfo(const scopa_state *__restrict__ g_states, uint2 *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kDecision) return;
    const scopa_state s = g_states[i];
    const int p = s.step & 1;
    uint32_t hand_bits = 0, table_bits = 0;
    for (int k = 0; k < s.nh[p]; k++) hand_bits |= 1u << nib(s.hand[p], k);
    for (int k = 0; k < s.nt; k++) table_bits |= 1u << nib(s.table, k);
    out[i] = make_uint2(hand_bits | (table_bits << 16), (uint32_t)s.hand[p]);
}

// torch tensors (W[out][in] row-major, float32) of one advantage net -> its LDS image (layout above)
__global__ void __launch_bounds__(256)
k_sdcfr_pack(const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
             const float *__restrict__ w3, const float *__restrict__ b3, float *__restrict__ img) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kImgFloats) return;
    float v;
    if (e < kImgC1) {
        const int mt = e >> 9, g = (e >> 8) & 1, lane = (e >> 2) & 63, c = e & 3;
        v = w1[(16 * mt + (lane & 15)) * 34 + 4 * (4 * g + c) + (lane >> 4)];
    } else if (e < kImgW2) {
        const int u = e - kImgC1;
        v = b1[u] + w1[u * 34 + 32];
    } else if (e < kImgB2) {
        const int i = e - kImgW2, nt = i >> 11, mt = (i >> 8) & 7, lane = (i >> 2) & 63, r = i & 3;
        v = w2[(16 * nt + (lane & 15)) * 128 + 16 * mt + 4 * (lane >> 4) + r];
    } else if (e < kImgW3) {
        v = b2[e - kImgB2];
    } else if (e < kImgB3) {
        const int i = e - kImgW3, nt = i >> 8, lane = (i >> 2) & 63, r = i & 3;
        v = w3[(lane & 15) * 64 + 16 * nt + 4 * (lane >> 4) + r];
    } else {
        v = b3[e - kImgB3];
    }
    img[e] = v;
}

#ifdef SCOPA_WALK_STAMPS   // development build only: shader-clock stamps of wavefront 0 of workgroup 0 (tests/tools/sdcfr_stamps.py)
__device__ unsigned long long g_sd_stamps[16];   // 1 layer 1 | 2 layer 2 | 3 layer 3 + policy | 4 expand / sample | 6 leaves + backward | 7 take | 13 shader clocks, 14 100 MHz ticks of whole tasks | 15 tasks
// stamps accumulate in registers and reach memory once per task (a global read-modify-write per stamp cost more than the stages it
// timed); sched_barrier pins the clock read where it is written (the scheduler otherwise moves MFMAs across it)
#define SD_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = clock64(); sd_acc_[i] += now_ - sd_prev_; sd_prev_ = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SD_STAMP(i) do { } while (0)
#endif

// REPLAY: the opponent draws come from the caller's uniforms (tests) instead of Philox.  A template parameter and not a run-time
// branch because the compiler cannot tell a global LOAD pending in one arm from none: with the branch in the loop it waited for
// vmcnt(0) at the end of every tile and before every draw -- i.e. for the memory-row STORES of the tiles before (one counter).
template <int T, int W, bool REPLAY>
__global__ void __launch_bounds__(sd_waves(W) * 64)
k_sdcfr_traverse(const uint2 *__restrict__ g_ninfo, const int8_t *__restrict__ g_payoff, const float *__restrict__ g_image,
                 int traverser, int batch, float *__restrict__ mem_feat, float *__restrict__ mem_regret, float *__restrict__ mem_mask,
                 uint32_t capacity, uint32_t write_base, float *__restrict__ root_values, const double *__restrict__ uniforms,
                 uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t b0, uint32_t *__restrict__ g_err) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_next[1];                                               // next task of this workgroup not taken yet
    float *s_w = reinterpret_cast<float *>(smem);                           // [2][kImgFloats]
    uint2 *s_node = reinterpret_cast<uint2 *>(s_w + 2 * kImgFloats);        // [kDecision (+1 pad)]: feature bits | hand nibbles of every decision node
    constexpr int kTeams = sd_waves(W) / W;
    int8_t *s_payoff = reinterpret_cast<int8_t *>(s_node + kSdNodeSlots);   // [kTerminal]: rewards x 2 of player 0 at the leaves
    SdTeam<T, W> *s_team = reinterpret_cast<SdTeam<T, W> *>(s_payoff + kTerminal);   // [kTeams]
    SdPos *s_pos = reinterpret_cast<SdPos *>(s_team + kTeams);               // [wavefronts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int team = wave / W, wr = wave % W;                               // consecutive wavefronts form a team: they sit on different SIMDs
    if (tid == 0) s_next[0] = kTeams;
    if (tid < kTeams) s_team[tid].bar = 0;
    for (int i = tid; i < 2 * kImgFloats / 4; i += blockDim.x)
        reinterpret_cast<float4 *>(s_w)[i] = reinterpret_cast<const float4 *>(g_image)[i];
    for (int i = tid; i < kDecision; i += blockDim.x) s_node[i] = g_ninfo[i];   // 13 KB + 576 B: with them in LDS a task makes no global load at all, so
    for (int i = tid; i < kTerminal / 4; i += blockDim.x)                       // nothing ever waits behind the memory-row stores (one vmcnt queue)
        reinterpret_cast<uint32_t *>(s_payoff)[i] = reinterpret_cast<const uint32_t *>(g_payoff)[i];
    __syncthreads();
    SdTeam<T, W> &ws = s_team[team];
    float (*wpos)[16] = s_pos[wave].pos;
    float *vals = W > 1 ? &ws.val[0][0] : &wpos[0][0];   // the frontier's values on the way back up (a solo wavefront: in its pos area, free by then)
    uint32_t phase = 0;                        // arrivals the team's barrier counter shows once everybody has reached this wavefront's latest barrier
    const int nj = lane & 15, q = lane >> 4;   // this lane's column (node of the tile) and K / row group

    // A TASK is T consecutive traversals (the last one of a batch may be short), walked by a TEAM of W wavefronts that share the
    // tiles of every ply (tile i goes to wavefront i mod W) and meet at a team barrier between plies: a ply's tiles are independent
    // of each other, and a single wavefront's task is a chain of 21 tiles of which a small batch (one task per SIMD at 4096
    // traversals) hides nothing.  The workgroup owns tasks [first, first + count) and its teams TAKE them from a counter in LDS
    // (the first one is static): the SIMD's arbiter favours its oldest wavefront, so equal shares would leave the workgroup
    // waiting for its slowest (scopa_mccfr.hip, main loop, has the measurement).  Who walks a traversal does not matter: draws
    // and memory-row positions are keyed by its id.
    const int n_tasks = (batch + T - 1) / T;
    const int per_wg = (n_tasks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int first = (int)blockIdx.x * per_wg, count = first < n_tasks ? (n_tasks - first < per_wg ? n_tasks - first : per_wg) : 0;
    for (int c = team; c < count;) {
        const int tb0 = (first + c) * T;                                    // the task's first traversal (local id within the batch)
        const int n_live = batch - tb0 < T ? batch - tb0 : T;               // traversals t >= n_live are walked like the others but write nothing
        // ring row of the task's first memory row: write_base < capacity and 41 * batch <= capacity (checked on the host), so one
        // conditional subtraction each replaces the 64-bit modulo per row
        uint32_t row0 = write_base + 41u * (uint32_t)tb0;
        row0 = row0 >= capacity ? row0 - capacity : row0;
#ifdef SCOPA_WALK_STAMPS
        unsigned long long sd_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long sd_t0_ = clock64(), sd_r0_ = wall_clock64();
        unsigned long long sd_prev_ = sd_t0_;
#endif
        if (wr == 0 && lane < T) ws.idx[0][lane] = 0;
        team_barrier<W>(&ws.bar, phase, lane, g_err);
        int width = 1, cb = 0;                                              // a traversal's frontier width at the current ply; which half of ws.idx holds it
        // ---- forward: plies 0..7 ----------------------------------------------------------------------------------------
#pragma unroll 1
        for (int d = 0; d < kPlies; d++) {
            const int p = d & 1, nl = 4 - (d >> 1);
            const bool trav_ply = p == traverser;
            // opponent node with ONE legal action (plies 6/7): whatever the advantages are, regret matching either puts all mass on it
            // or falls back to uniform over it (deep_cfr.py:353-359) -- the child is forced (same position, same index within the
            // next ply) and nothing else of this node is used (no memory row, no value weight): its forward pass is skipped, 24
            // of the 105 / 82 node evaluations of a traversal
            if (!trav_ply && nl == 1) continue;
            const int n_nodes = T * width;
            const int m = (d - traverser) >> 1;                             // traverser-ply index when trav_ply
            const int moff = m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 5 : 17, poff = m == 0 ? 0 : m == 1 ? 4 : m == 2 ? 16 : 40;
            const uint2 *nodes_d = s_node + (d == 0 ? 0 : d == 1 ? 1 : d == 2 ? 5 : d == 3 ? 21 : d == 4 ? 69 : d == 5 ? 213 : d == 6 ? 501 : 1077);   // level_offset(d) as
                                                                                  // selects: the table lookup is a scalar memory load per ply
            const float *Wn = s_w + p * kImgFloats;
            const float4 *w1 = reinterpret_cast<const float4 *>(Wn + kImgW1) + lane;
            const float4 *c1 = reinterpret_cast<const float4 *>(Wn + kImgC1) + q;
            const float4 *w2 = reinterpret_cast<const float4 *>(Wn + kImgW2) + lane;
            const float4 *b2 = reinterpret_cast<const float4 *>(Wn + kImgB2) + q;
            const float4 *w3 = reinterpret_cast<const float4 *>(Wn + kImgW3) + lane;
            const float4 *b3 = reinterpret_cast<const float4 *>(Wn + kImgB3) + q;
            // layer 1's first step and bias do not depend on the tile: they are fetched for the NEXT tile under the current tile's
            // last MFMAs, so that a tile's first MFMA does not wait for them behind the previous tile's expansion
            float4 wa[8], c1v[8];
#pragma unroll
            for (int mt = 0; mt < 8; mt++) wa[mt] = w1[(mt * 2 + 0) * 64];
#pragma unroll
            for (int mt = 0; mt < 8; mt++) c1v[mt] = c1[mt * 4];
            // ... and so is the next tile's node (tree index, then its feature bits / hand nibbles: two dependent LDS reads)
            uint32_t node_nx = ws.idx[cb][16 * wr + nj < n_nodes ? 16 * wr + nj : n_nodes - 1];
            uint2 inf_nx = nodes_d[node_nx];
#pragma unroll 1
            for (int f0 = 16 * wr; f0 < n_nodes; f0 += 16 * W) {
                const int f = f0 + nj;
                const bool live = f < n_nodes;
                const uint32_t xbits = inf_nx.x, hand = inf_nx.y, node = node_nx;   // lanes beyond the frontier compute a copy of its last node and store nothing
                // Weights are fetched one step AHEAD of the MFMAs that use them (a step = 32 / 16 MFMAs = 1024 / 512 matrix-pipe
                // cycles, an LDS read returns in ~130): written in that order and pinned with sched_barrier, because left alone the
                // scheduler issues a step's reads only two or three MFMAs before their first use and every step stalls on them.
                // ---- layer 1: K-step s = 4 g + c covers features 4 s .. 4 s + 3 (k index = lane / 16) --------------------------
                v4f h1[8];
                float4 wb[8];
#pragma unroll
                for (int mt = 0; mt < 8; mt++) h1[mt] = to_v4f(c1v[mt]);
#pragma unroll
                for (int mt = 0; mt < 8; mt++) wb[mt] = w1[(mt * 2 + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t xs = xbits >> q;
                {
                    const float x0 = (float)(xs & 1u), x1 = (float)((xs >> 4) & 1u), x2 = (float)((xs >> 8) & 1u), x3 = (float)((xs >> 12) & 1u);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wa[mt].x, x0, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wa[mt].y, x1, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wa[mt].z, x2, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wa[mt].w, x3, h1[mt]);
                }
                // layer 2's first step and its bias
                v4f h2[4];
                float4 u0[4], u1[4];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) u0[nt] = w2[(nt * 8 + 0) * 64];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) h2[nt] = to_v4f(b2[nt * 4]);
                __builtin_amdgcn_sched_barrier(0);
                {
                    const float x0 = (float)((xs >> 16) & 1u), x1 = (float)((xs >> 20) & 1u), x2 = (float)((xs >> 24) & 1u), x3 = (float)((xs >> 28) & 1u);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wb[mt].x, x0, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wb[mt].y, x1, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wb[mt].z, x2, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(wb[mt].w, x3, h1[mt]);
                }
                SD_STAMP(1);
                // ---- layer 2: K-step (mt, r) covers the units 16 mt + 4 k + r, k = lane / 16: B operand = relu(h1[mt][r]) as it stands ----
                // (register r of tile mt = unit 16 mt + 4 q + r of node nj)
                float4 w3r[4];
                v4f o0, o1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int mt = 0; mt < 8; mt += 2) {
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) u1[nt] = w2[(nt * 8 + mt + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const float a0 = relu(h1[mt][0]), a1 = relu(h1[mt][1]), a2 = relu(h1[mt][2]), a3 = relu(h1[mt][3]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u0[nt].x, a0, h2[nt]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u0[nt].y, a1, h2[nt]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u0[nt].z, a2, h2[nt]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u0[nt].w, a3, h2[nt]);
                    }
                    if (mt + 2 < 8) {
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) u0[nt] = w2[(nt * 8 + mt + 2) * 64];
                    } else {
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) w3r[nt] = w3[nt * 64];
                        o0 = to_v4f(b3[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const float a0 = relu(h1[mt + 1][0]), a1 = relu(h1[mt + 1][1]), a2 = relu(h1[mt + 1][2]), a3 = relu(h1[mt + 1][3]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u1[nt].x, a0, h2[nt]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u1[nt].y, a1, h2[nt]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u1[nt].z, a2, h2[nt]);
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(u1[nt].w, a3, h2[nt]);
                    }
                }
                SD_STAMP(2);
                // ---- layer 3: 16 outputs, K-step (nt, r); two accumulator chains ------------------------------------------------------
                float adv[4];
                {
                    float g[4][4];
#pragma unroll
                    for (int nt = 0; nt < 4; nt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) g[nt][r] = relu(h2[nt][r]);
                    o0 = mfma16(w3r[0].x, g[0][0], o0); o1 = mfma16(w3r[1].x, g[1][0], o1);
                    o0 = mfma16(w3r[0].y, g[0][1], o0); o1 = mfma16(w3r[1].y, g[1][1], o1);
                    o0 = mfma16(w3r[0].z, g[0][2], o0); o1 = mfma16(w3r[1].z, g[1][2], o1);
                    o0 = mfma16(w3r[0].w, g[0][3], o0); o1 = mfma16(w3r[1].w, g[1][3], o1);
                    o0 = mfma16(w3r[2].x, g[2][0], o0); o1 = mfma16(w3r[3].x, g[3][0], o1);
                    o0 = mfma16(w3r[2].y, g[2][1], o0); o1 = mfma16(w3r[3].y, g[3][1], o1);
                    o0 = mfma16(w3r[2].z, g[2][2], o0); o1 = mfma16(w3r[3].z, g[3][2], o1);
                    o0 = mfma16(w3r[2].w, g[2][3], o0); o1 = mfma16(w3r[3].w, g[3][3], o1);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) wa[mt] = w1[(mt * 2 + 0) * 64];     // the next tile's first step (above)
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) c1v[mt] = c1[mt * 4];
                    { const int fn = f + 16 * W; node_nx = ws.idx[cb][fn < n_nodes ? fn : n_nodes - 1]; inf_nx = nodes_d[node_nx]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; r++) adv[r] = o0[r] + o1[r];     // output 4 q + r of node nj
                }
                // positive_regret_policy over the node's 16 outputs (nets.py:93-101): relu kills the illegal slots' -1e6
                float z = 0.0f;
                {
                    float4 pv;
                    float *pp = &pv.x;
#pragma unroll
                    for (int r = 0; r < 4; r++) { pp[r] = (((xbits >> (4 * q + r)) & 1u) && adv[r] > 0.0f) ? adv[r] : 0.0f; z += pp[r]; }
                    *reinterpret_cast<float4 *>(&wpos[nj][4 * q]) = pv;
                    z = sum_row_groups(z);                                 // the four row groups of the node's column
                }
                const float den = z > 1e-8f ? z : 1e-8f;                   // clamp_min(eps)
                sd_order();
                SD_STAMP(3);
                int t = 0;
#pragma unroll
                for (int k = 1; k < T; k++) t += f >= k * width;
                const int j = f - t * width;
                if (trav_ply) {
                    // recurse on ALL legal actions, hand order (:326-336): lane (q, node) takes action q
                    const float pk = q < nl ? wpos[nj][(hand >> (4 * q)) & 15u] / den : 0.0f;
                    if (live) {
                        if (q < nl) ws.pol_trav[t][poff + j * nl + q] = pk;
                        if (q < nl) ws.idx[cb ^ 1][f * nl + q] = (uint16_t)(node * nl + q);
                        // the node's memory row (:339-346): where, and everything of it that does not wait for the values -- features
                        // and mask -- now, four lanes to a row; the regrets follow in the backward pass.
                        const int rank = sd_rank(m, j);
                        if (q == 0) ws.hand_trav[t][moff + j] = (uint16_t)hand;
                        if (t < n_live) {
                            uint32_t row = row0 + 41u * (uint32_t)t + (uint32_t)rank;
                            row = row >= capacity ? row - capacity : row;
                            float2 *mf = reinterpret_cast<float2 *>(mem_feat + (size_t)row * 34);      // 136-byte rows: 8-byte aligned
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const int ch = q + 4 * i;                  // 8-byte piece ch = features 2 ch, 2 ch + 1
                                mf[ch] = make_float2((float)((xbits >> (2 * ch)) & 1u), (float)((xbits >> (2 * ch + 1)) & 1u));
                            }
                            if (q == 0) mf[16] = make_float2(1.0f, 0.0f);  // float(player == current_player), unused feature
                            if (mem_mask) reinterpret_cast<float4 *>(mem_mask + (size_t)row * 16)[q] =
                                make_float4((float)((xbits >> (4 * q)) & 1u), (float)((xbits >> (4 * q + 1)) & 1u), (float)((xbits >> (4 * q + 2)) & 1u), (float)((xbits >> (4 * q + 3)) & 1u));
                        }
                    }
                } else {
                    // opponent: sample ONE action (:347-365); the four lanes of a node draw the same number
                    float pk[4];
                    {
                        float pr[4];                                        // all four reads before any use: one LDS round trip (unused hand nibbles are 0: a valid slot)
#pragma unroll
                        for (int k = 0; k < 4; k++) pr[k] = wpos[nj][(hand >> (4 * k)) & 15u];
#pragma unroll
                        for (int k = 0; k < 4; k++) pk[k] = k < nl ? pr[k] / den : 0.0f;
                    }
                    float sum = pk[0];
#pragma unroll
                    for (int k = 1; k < 4; k++) if (k < nl) sum += pk[k];  // action_probs.sum(), float32, left to right
                    const int tb = tb0 + t;
                    double u;
                    if (REPLAY) u = (live && t < n_live) ? uniforms[((size_t)tb * kPlies + d) * 24 + j] : 0.0;
                    else {
                        const philox_out x = philox4x32_10((uint32_t)j + 1024u * (uint32_t)d, b0 + (uint32_t)tb, iteration, 4u + (uint32_t)traverser, seed_lo, seed_hi);
                        u = u53(x.x0, x.x1);
                    }
                    int a;
                    if (sum == 0.0f) { a = (int)(u * (double)nl); a = a < nl - 1 ? a : nl - 1; }   // np.random.choice(legal_actions): uniform
                    else {                                                                          // p = action_probs / sum: float32 p, float64 cdf
                        double cs = 0.0, cdf[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) { const double pq = (double)(pk[k] / sum); cs = k ? cs + pq : pq; cdf[k] = cs; }
                        double last = cdf[0];
#pragma unroll
                        for (int k = 1; k < 4; k++) last = k < nl ? cdf[k] : last;
                        // a = #{k < nl - 1 : cdf[k] / last <= u}  (cdf[nl - 1] / last = 1 > u: never); nl is wave-uniform, so the float64
                        // divisions a ply does not need are branched over, not computed and masked (nl = 2 at the widest plies: one)
                        a = cdf[0] / last <= u ? 1 : 0;
                        if (nl > 2) a = cdf[1] / last <= u ? 2 : a;
                        if (nl > 3) a = cdf[2] / last <= u ? 3 : a;
                    }
                    if (live && q == 0) ws.idx[cb ^ 1][f] = (uint16_t)(node * nl + a);
                }
                sd_order();
                SD_STAMP(4);
            }
            cb ^= 1;
            if (trav_ply) width *= nl;
            team_barrier<W>(&ws.bar, phase, lane, g_err);                          // the next ply reads what every wavefront of the team expanded
        }
        // ---- leaves, then backward ---------------------------------------------------------------------------------------
        for (int f = lane + 64 * wr; f < T * width; f += 64 * W) {
            const int p0 = s_payoff[ws.idx[cb][f]];
            vals[f] = 0.5f * (float)(traverser == 0 ? p0 : -p0);
        }
        team_barrier<W>(&ws.bar, phase, lane, g_err);
        int cur = 0;                                                        // (teams) which half of the value buffer holds the children's values
#pragma unroll 1
        for (int d = kPlies - 1; d >= 0; d--) {
            const int p = d & 1, nl = 4 - (d >> 1);
            if (p != traverser) continue;                                  // opponent ply: the sampled child's value is returned unchanged (:363-365), same position
            width /= nl;
            const int m = (d - traverser) >> 1;
            const int moff = m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 5 : 17, poff = m == 0 ? 0 : m == 1 ? 4 : m == 2 ? 16 : 40;
            const float *vin = vals + (W > 1 ? cur * T * 24 : 0);
            float *vout = vals + (W > 1 ? (cur ^ 1) * T * 24 : 0);
            for (int f = lane + 64 * wr; f < T * width; f += 64 * W) {
                int t = 0;
#pragma unroll
                for (int k = 1; k < T; k++) t += f >= k * width;
                const int j = f - t * width;
                const uint32_t hand = ws.hand_trav[t][moff + j], rank = (uint32_t)sd_rank(m, j);
                float pl[4];
#pragma unroll
                for (int k = 0; k < 4; k++) pl[k] = ws.pol_trav[t][poff + j * nl + (k < nl ? k : 0)];
                float value = 0.0f, cfv[16];
#pragma unroll
                for (int cc = 0; cc < 16; cc++) cfv[cc] = 0.0f;           // counterfactual_values = zeros(16) (:324)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < nl) {
                        const float av = vin[f * nl + k];
                        value += pl[k] * av;                                 // value += policy[action] * action_value, float32 (:335)
                        const int c = (int)((hand >> (4 * k)) & 15u);
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) if (cc == c) cfv[cc] = av;
                    }
                }
                vout[f] = value;
                if (t < n_live) {
                    float mx = 0.0f, reg[16];
#pragma unroll
                    for (int cc = 0; cc < 16; cc++) { reg[cc] = cfv[cc] - value; const float a = fabsf(reg[cc]); mx = a > mx ? a : mx; }   // illegal slots = -value
                    const float den = mx + 1e-8f;
                    if (mx > 0.0f) {
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) reg[cc] = reg[cc] / den;    // add_experience (:73-74)
                    }
                    uint32_t row = row0 + 41u * (uint32_t)t + rank;
                    row = row >= capacity ? row - capacity : row;
                    float4 *mr = reinterpret_cast<float4 *>(mem_regret + (size_t)row * 16);
#pragma unroll
                    for (int i = 0; i < 4; i++) mr[i] = make_float4(reg[4 * i], reg[4 * i + 1], reg[4 * i + 2], reg[4 * i + 3]);
                }
            }
            cur ^= (W > 1);
            team_barrier<W>(&ws.bar, phase, lane, g_err);
        }
        if (wr == 0 && lane < n_live) root_values[tb0 + lane] = vals[(W > 1 ? cur * T * 24 : 0) + lane];
        SD_STAMP(6);
        // the team's next task: its first wavefront takes it (everybody has passed the barrier above, so the scratch is free)
        if (W == 1) {
            int got = 0;
            if (lane == 0) got = atomicAdd(s_next, 1);
            c = __builtin_amdgcn_readfirstlane(got);
        } else {
            if (wr == 0 && lane == 0) ws.task = atomicAdd(s_next, 1);
            team_barrier<W>(&ws.bar, phase, lane, g_err);
            c = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int32_t *>(&ws.task));
            team_barrier<W>(&ws.bar, phase, lane, g_err);                          // nobody overwrites ws.task before all have read it
        }
        SD_STAMP(7);
#ifdef SCOPA_WALK_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            for (int i = 0; i < 8; i++) g_sd_stamps[i] += sd_acc_[i];
            g_sd_stamps[13] += clock64() - sd_t0_; g_sd_stamps[14] += wall_clock64() - sd_r0_; g_sd_stamps[15] += 1;
        }
#endif
    }
}

// =====================================================================================================================
// The same traversal for the reference's own workload -- ONE deal per solver -- without a forward pass per VISIT.
// The advantage nets are frozen while a launch runs and a node's features are a function of the tree node alone, so the policy at a
// decision node is the same for every traversal that reaches it.  The deal's 1 653 decision nodes are therefore evaluated ONCE per
// launch (k_sdcfr_policy: 105 tiles of 16 nodes on the matrix cores, the tile arithmetic of k_sdcfr_traverse instruction for
// instruction, so the same bits) and the traversals become walks over a 26 KB policy table held in LDS (k_sdcfr_walk): sample, expand,
// write memory rows, carry values back up.  At 4 096 traversals a launch's 331 776 forward passes (81 per traversal) become 1 653; what
// is left is bound by the memory rows it must write (41 x 264 B per traversal) and by the sampling arithmetic.  This is the frozen-table
// idea of the batched MCCFR path (sigma | threshold rows per iteration) applied to Deep CFR.  k_sdcfr_traverse stays: it is the form
// for batches whose traversals do NOT share a deal (nothing to share then), it pins the tile arithmetic to the reference fixture through
// the replayed-draws tests, and the two paths are tested against each other row for row.
namespace {
__host__ __device__ constexpr int sd_level_off(int d) { return d == 0 ? 0 : d == 1 ? 1 : d == 2 ? 5 : d == 3 ? 21 : d == 4 ? 69 : d == 5 ? 213 : d == 6 ? 501 : 1077; }   // level_offset as selects (no table load)
constexpr int kPolicyTiles = 1 + 1 + 1 + 3 + 9 + 18 + 36 + 36;   // 16-node tiles per ply: widths 1, 4, 16, 48, 144, 288, 576, 576
constexpr int kPolicyWaves = 4;                                  // one tile per workgroup, its layers split over four wavefronts
__host__ __device__ constexpr int sd_tile_off(int d) { return d == 0 ? 0 : d == 1 ? 1 : d == 2 ? 2 : d == 3 ? 3 : d == 4 ? 6 : d == 5 ? 15 : d == 6 ? 33 : 69; }
static_assert(sd_tile_off(7) + 36 == kPolicyTiles, "tiles by ply");
}  // namespace

// regret-matching policy (legal actions in hand order, zeros beyond) of EVERY decision node of the deal under the current nets, and what
// sampling from it needs.  np.random.choice(legal, p = probs / probs.sum()) (deep_cfr.py:347-365) is index = #{k : cdf_k / cdf_last <= u}
// with float32 p and a float64 cdf; every u the traversals draw is N * 2^-53 with an integer N < 2^53 (u53 of two Philox words), and
// x * 2^53 is exact in float64, so  x <= u  <=>  ceil(x * 2^53) <= N: the node's three thresholds are stored as those integers and a
// visit compares 64-bit integers -- the same answer as the float64 compare, bit for bit, with the float32 / float64 divisions done once per
// node instead of once per visit.  thr[0] = ~0 marks probs.sum() == 0 (uniform choice: the visit keeps numpy's arithmetic for that case).
__global__ void __launch_bounds__(kPolicyWaves * 64)
k_sdcfr_policy(const uint2 *__restrict__ g_ninfo, const float *__restrict__ g_image, float4 *__restrict__ g_pol, unsigned long long *__restrict__ g_thr) {
    // One 16-node tile per workgroup of four wavefronts.  The launch is a latency chain (105 tiles on 256 compute units), so the tile's MLP is cut
    // ACROSS wavefronts -- layer 1: two of the eight 16-unit blocks each; layer 2: one of the four blocks each; layer 3: its two accumulator chains on
    // wavefronts 0 and 1 -- with the activations handed over through LDS (an MFMA result IS the next layer's B operand, lane for lane: a float4 per lane
    // and block), and every wavefront takes its weights straight from the image in global memory (16 float4 per lane, all in flight at once; no staging
    // of the 54 KB net, no barrier in front of the first MFMA).  Every accumulator sees the K order of k_sdcfr_traverse: results are bit-identical.
    __shared__ float4 s_h1[8][64], s_h2[4][64], s_o[2][64];
    __shared__ SdPos s_pos1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile = (int)blockIdx.x, d = 0;
#pragma unroll
    for (int k = 1; k < kPlies; k++) d += tile >= sd_tile_off(k);
    tile -= sd_tile_off(d);
    const int p = d & 1;
    const float *Wn = g_image + p * kImgFloats;
    const int wd = d == 0 ? 1 : d == 1 ? 4 : d == 2 ? 16 : d == 3 ? 48 : d == 4 ? 144 : d == 5 ? 288 : 576, nl = 4 - (d >> 1);
    const int nj = lane & 15, q = lane >> 4, j = tile * 16 + nj;
    const bool live = j < wd;
    const uint2 inf = g_ninfo[sd_level_off(d) + (live ? j : wd - 1)];
    const uint32_t xbits = inf.x, hand = inf.y;
    float4 w1r[2][2], c1r[2], w2r[8], w3r[2];
    {
        const float4 *w1 = reinterpret_cast<const float4 *>(Wn + kImgW1) + lane, *c1 = reinterpret_cast<const float4 *>(Wn + kImgC1) + q;
        const float4 *w2 = reinterpret_cast<const float4 *>(Wn + kImgW2) + lane, *w3 = reinterpret_cast<const float4 *>(Wn + kImgW3) + lane;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int mt = 2 * wave + h;
            c1r[h] = c1[mt * 4];
#pragma unroll
            for (int g = 0; g < 2; g++) w1r[h][g] = w1[(mt * 2 + g) * 64];
        }
#pragma unroll
        for (int mt = 0; mt < 8; mt++) w2r[mt] = w2[(wave * 8 + mt) * 64];
#pragma unroll
        for (int h = 0; h < 2; h++) w3r[h] = w3[((wave & 1) + 2 * h) * 64];
    }
    const float4 b2r = (reinterpret_cast<const float4 *>(Wn + kImgB2) + q)[wave * 4], b3r = (reinterpret_cast<const float4 *>(Wn + kImgB3) + q)[0];
    {   // layer 1: unit blocks 2 wave, 2 wave + 1
        v4f ha = to_v4f(c1r[0]), hb = to_v4f(c1r[1]);
        const uint32_t xs = xbits >> q;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const float x0 = (float)((xs >> (16 * g)) & 1u), x1 = (float)((xs >> (16 * g + 4)) & 1u);
            const float x2 = (float)((xs >> (16 * g + 8)) & 1u), x3 = (float)((xs >> (16 * g + 12)) & 1u);
            ha = mfma16(w1r[0][g].x, x0, ha); hb = mfma16(w1r[1][g].x, x0, hb);
            ha = mfma16(w1r[0][g].y, x1, ha); hb = mfma16(w1r[1][g].y, x1, hb);
            ha = mfma16(w1r[0][g].z, x2, ha); hb = mfma16(w1r[1][g].z, x2, hb);
            ha = mfma16(w1r[0][g].w, x3, ha); hb = mfma16(w1r[1][g].w, x3, hb);
        }
        s_h1[2 * wave][lane] = make_float4(relu(ha[0]), relu(ha[1]), relu(ha[2]), relu(ha[3]));
        s_h1[2 * wave + 1][lane] = make_float4(relu(hb[0]), relu(hb[1]), relu(hb[2]), relu(hb[3]));
    }
    __syncthreads();
    {   // layer 2: unit block `wave`
        v4f h2 = to_v4f(b2r);
#pragma unroll
        for (int mt = 0; mt < 8; mt++) {
            const float4 a = s_h1[mt][lane];
            h2 = mfma16(w2r[mt].x, a.x, h2);
            h2 = mfma16(w2r[mt].y, a.y, h2);
            h2 = mfma16(w2r[mt].z, a.z, h2);
            h2 = mfma16(w2r[mt].w, a.w, h2);
        }
        s_h2[wave][lane] = make_float4(relu(h2[0]), relu(h2[1]), relu(h2[2]), relu(h2[3]));
    }
    __syncthreads();
    if (wave < 2) {   // layer 3: chain 0 = bias + blocks 0, 2 (wavefront 0), chain 1 = blocks 1, 3 (wavefront 1)
        v4f o = {0.0f, 0.0f, 0.0f, 0.0f};
        if (wave == 0) o = to_v4f(b3r);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float4 a = s_h2[wave + 2 * h][lane];
            o = mfma16(w3r[h].x, a.x, o);
            o = mfma16(w3r[h].y, a.y, o);
            o = mfma16(w3r[h].z, a.z, o);
            o = mfma16(w3r[h].w, a.w, o);
        }
        s_o[wave][lane] = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    if (wave != 0) return;
    float adv[4];
    {
        const float4 o0 = s_o[0][lane], o1 = s_o[1][lane];
        adv[0] = o0.x + o1.x; adv[1] = o0.y + o1.y; adv[2] = o0.z + o1.z; adv[3] = o0.w + o1.w;
    }
    float (*wpos)[16] = s_pos1.pos;
    float z = 0.0f;
    {
        float4 pv;
        float *pp = &pv.x;
#pragma unroll
        for (int r = 0; r < 4; r++) { pp[r] = (((xbits >> (4 * q + r)) & 1u) && adv[r] > 0.0f) ? adv[r] : 0.0f; z += pp[r]; }
        *reinterpret_cast<float4 *>(&wpos[nj][4 * q]) = pv;
        z = sum_row_groups(z);
    }
    const float den = z > 1e-8f ? z : 1e-8f;
    sd_order();
    const float pk = q < nl ? wpos[nj][(hand >> (4 * q)) & 15u] / den : 0.0f;   // lane (q, node): action q of the node, hand order
    if (live) reinterpret_cast<float *>(g_pol + sd_level_off(d) + j)[q] = pk;
    // the node's sampling thresholds: its four probabilities meet in lane (0, node) through the tile's pos area (free by now)
    sd_order();
    wpos[nj][q] = pk;
    sd_order();
    if (q == 0 && live) {
        const float4 p4 = *reinterpret_cast<const float4 *>(&wpos[nj][0]);
        const float pv[4] = {p4.x, p4.y, p4.z, p4.w};
        float sum = pv[0];
#pragma unroll
        for (int k = 1; k < 4; k++) if (k < nl) sum += pv[k];              // action_probs.sum(), float32, left to right
        unsigned long long thr[3] = {1ull << 53, 1ull << 53, 1ull << 53};   // 2^53 > every N: never counted (k >= nl - 1: cdf_k / cdf_last = 1 > u)
        if (sum == 0.0f) thr[0] = ~0ull;
        else {
            double cs = 0.0, cdf[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const double pq = (double)(pv[k] / sum); cs = k ? cs + pq : pq; cdf[k] = cs; }   // float32 p, float64 cdf
            double last = cdf[0];
#pragma unroll
            for (int k = 1; k < 4; k++) last = k < nl ? cdf[k] : last;
#pragma unroll
            for (int k = 0; k < 3; k++) if (k < nl - 1) thr[k] = (unsigned long long)ceil((cdf[k] / last) * 9007199254740992.0);
        }
        unsigned long long *out = g_thr + (size_t)(sd_level_off(d) + j) * 3;
        out[0] = thr[0]; out[1] = thr[1]; out[2] = thr[2];
    }
}

namespace {
// What a walk keeps in LDS is per TRAVERSER: policies and node words of the traverser's plies only (the walk recurses there and needs the policy on the way
// back), sampling thresholds of the opponent's three sampled plies only, nl - 1 of them per node -- 22-24 KB instead of the 80 KB of every node's
// everything, so that TWO workgroups of twelve wavefronts fit a compute unit beside their per-wavefront scratch (measured: the staging per workgroup shrinks
// with the tables; the doubled occupancy by itself changed nothing -- at 32768 traversals the launch is bound by its row stores)
constexpr int kWalkTravNodes = 916;        // traverser 1's plies 1, 3, 5, 7: 4 + 48 + 288 + 576  (traverser 0's plies 0, 2, 4, 6: 1 + 16 + 144 + 576 = 737)
constexpr int kWalkThr = 396;              // traverser 0's opponent plies 1, 3, 5: 4 x 3 + 48 x 2 + 288 x 1  (traverser 1's plies 0, 2, 4: 1 x 3 + 16 x 2 + 144 x 1 = 179)
__host__ __device__ constexpr int sd_trav_off(int traverser, int m) {   // the traverser's ply m = 0..3 within its compact tables
    return traverser == 0 ? (m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 17 : 161) : (m == 0 ? 0 : m == 1 ? 4 : m == 2 ? 52 : 340);
}
__host__ __device__ constexpr int sd_thr_off(int traverser, int k) {    // the opponent's sampled ply k = 0..2 (nl = 4, 3, 2: three, two, one thresholds per node)
    return traverser == 0 ? (k == 0 ? 0 : k == 1 ? 12 : 108) : (k == 0 ? 0 : k == 1 ? 3 : 35);
}
static_assert(sd_trav_off(0, 3) + 576 == 737 && sd_trav_off(1, 3) + 576 == kWalkTravNodes && sd_thr_off(0, 2) + 288 == kWalkThr && sd_thr_off(1, 2) + 144 == 179, "compact tables");
template <int T>
struct alignas(16) SdWalk {       // per wavefront: T traversals in flight, a frontier node addressed by its position f = t * width + j
    float val[T * 24];            // values of the frontier on the way back up, replaced in place (see SdTeam::val)
    uint16_t idx[2][T * 24];      // tree index of the frontier nodes of the current / the next ply
    uint32_t xb[T][41];           // feature bits of every traverser node BY MEMORY-ROW RANK: features and masks of the task's rows are written from
                                  // here in one sweep of consecutive addresses (a lane per row piece, rows in memory order) instead of a lane per row
    union alignas(16) {
        float regs[T * 41][16];            // way back: the task's regret rows by memory-row rank, assembled a lane per row, stored in one sweep of consecutive addresses
        unsigned long long draw[T * 40];   // forward pass only: the task's opponent draws N (u = N * 2^-53), all of them taken before the walk in full 64-lane rounds
    };
};
struct __attribute__((aligned(8))) SdF4A8 { float x, y, z, w; };   // sixteen bytes of a 136-byte feature row: rows alternate between 16- and 8-byte alignment
__host__ __device__ constexpr int sd_walk_waves(int T) { return T == 1 ? 12 : T == 2 ? 16 : T == 4 ? 8 : 4; }   // wavefronts per workgroup by LDS (a task's regret rows: 2.6 KB per traversal); T = 1: two workgroups per compute unit
}  // namespace

#ifdef SCOPA_WALK_STAMPS   // development build only: the same stamps for wavefront 0 of workgroup 0 of k_sdcfr_walk
__device__ unsigned long long g_wk_stamps[16];   // 0 staging the tables | 1 draws | 2 forward | 3 feature / mask sweep | 4 leaves + backward | 5 take | 13 shader clocks, 14 100 MHz ticks of whole tasks | 15 tasks
#endif
#ifdef SCOPA_WALK_ROWMASK   // development builds only: every memory row lands in a ring of SCOPA_WALK_ROWMASK + 1 rows (L2-resident) -- the same instruction stream without its HBM traffic
#define SD_WALK_ROWMASK(row) (row) &= (uint32_t)(SCOPA_WALK_ROWMASK)
#else
#define SD_WALK_ROWMASK(row) (void)0
#endif
// The traverser is a template parameter and the ply loops are unrolled: every per-ply quantity (legal actions, frontier width, table offsets, which
// plies sample) is then a constant, the frontier loops have known trip counts, and the scalar selects and branches that decided them per ply at run time
// -- about 600 scalar and 250 vector instructions per traversal of a chain that a lone wavefront executes at one instruction per 8 clocks -- are gone.
template <int T, int TR>
__global__ void __launch_bounds__(sd_walk_waves(T) * 64, T == 1 ? 6 : 4)   // (second figure, HIP: wavefronts per SIMD to stay eligible for: two workgroups of twelve -- at most 80 registers)
k_sdcfr_walk(const uint2 *__restrict__ g_ninfo, const int8_t *__restrict__ g_payoff, const float4 *__restrict__ g_pol, const unsigned long long *__restrict__ g_thr,
             int batch,
             float *__restrict__ mem_feat, float *__restrict__ mem_regret, float *__restrict__ mem_mask, uint32_t capacity, uint32_t write_base,
             float *__restrict__ root_values, uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t b0) {
    constexpr int traverser = TR;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_next[1];
    float4 *s_pol = reinterpret_cast<float4 *>(smem);                        // [kWalkTravNodes] policy of the traverser's nodes
    uint2 *s_node = reinterpret_cast<uint2 *>(s_pol + kWalkTravNodes);       // [kWalkTravNodes] feature bits | hand nibbles of the traverser's nodes
    unsigned long long *s_thr = reinterpret_cast<unsigned long long *>(s_node + kWalkTravNodes);   // [kWalkThr] sampling thresholds of the opponent's sampled plies (k_sdcfr_policy)
    int8_t *s_payoff = reinterpret_cast<int8_t *>(s_thr + kWalkThr);         // [kTerminal]
    SdWalk<T> *s_wave = reinterpret_cast<SdWalk<T> *>(s_payoff + kTerminal); // [wavefronts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SCOPA_WALK_STAMPS
    const unsigned long long wk_k0_ = clock64();
#endif
    if (tid == 0) s_next[0] = sd_walk_waves(T);
    for (int i = tid; i < sd_trav_off(traverser, 3) + 576; i += blockDim.x) {
        const int m = (int)(i >= sd_trav_off(traverser, 1)) + (int)(i >= sd_trav_off(traverser, 2)) + (int)(i >= sd_trav_off(traverser, 3));
        const int g = sd_level_off(2 * m + traverser) + i - sd_trav_off(traverser, m);
        s_pol[i] = g_pol[g];
        s_node[i] = g_ninfo[g];
    }
    for (int i = tid; i < sd_thr_off(traverser, 2) + (traverser == 0 ? 288 : 144); i += blockDim.x) {
        const int k = (int)(i >= sd_thr_off(traverser, 1)) + (int)(i >= sd_thr_off(traverser, 2)), per = 3 - k, e = i - sd_thr_off(traverser, k);
        const int node = per == 3 ? e / 3 : per == 2 ? e >> 1 : e, which = e - node * per;
        s_thr[i] = g_thr[(size_t)(sd_level_off(2 * k + 1 - traverser) + node) * 3 + which];
    }
    for (int i = tid; i < kTerminal / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(s_payoff)[i] = reinterpret_cast<const uint32_t *>(g_payoff)[i];
    __syncthreads();
#ifdef SCOPA_WALK_STAMPS
    if (blockIdx.x == 0 && tid == 0) g_wk_stamps[0] += clock64() - wk_k0_;
#endif
    SdWalk<T> &ws = s_wave[wave];
    const int n_tasks = (batch + T - 1) / T;
    const int per_wg = (n_tasks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int first = (int)blockIdx.x * per_wg, count = first < n_tasks ? (n_tasks - first < per_wg ? n_tasks - first : per_wg) : 0;
    for (int c = wave; c < count;) {
        const int tb0 = (first + c) * T;
        const int n_live = batch - tb0 < T ? batch - tb0 : T;               // traversals t >= n_live are walked like the others but write nothing
        uint32_t row0 = write_base + 41u * (uint32_t)tb0;                   // ring row of the task's first memory row (see k_sdcfr_traverse)
        row0 = row0 >= capacity ? row0 - capacity : row0;
#ifdef SCOPA_WALK_STAMPS
        unsigned long long sd_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long sd_t0_ = clock64(), sd_r0_ = wall_clock64();
        unsigned long long sd_prev_ = sd_t0_;
#endif
        if (lane < T) ws.idx[0][lane] = 0;
        // ---- the task's draws: a draw is keyed by (position within the traversal's frontier + 1024 ply, traversal id, iteration, stream) and by nothing the walk
        // decides, so all of them -- the three sampled opponent plies, 4 + 12 + 24 positions per traversal for traverser 0, 1 + 4 + 12 for traverser 1 -- are
        // taken here with every lane busy instead of in three rounds of 2..48 lanes; slot = T * (positions of the earlier sampled plies) + f
        {
            const int w0 = traverser == 0 ? 4 : 1, w1 = traverser == 0 ? 12 : 4, w2 = traverser == 0 ? 24 : 12, d0 = 1 - traverser;
#pragma unroll 1
            for (int sl = lane; sl < T * (w0 + w1 + w2); sl += 64) {
                const int k = (int)(sl >= T * w0) + (int)(sl >= T * (w0 + w1));
                const int wd = k == 0 ? w0 : k == 1 ? w1 : w2, f = sl - (k == 0 ? 0 : k == 1 ? T * w0 : T * (w0 + w1)), d = d0 + 2 * k;
                int t = 0;
#pragma unroll
                for (int k2 = 1; k2 < T; k2++) t += f >= k2 * wd;
                const int j = f - t * wd;
                const philox_out x = philox4x32_10((uint32_t)j + 1024u * (uint32_t)d, b0 + (uint32_t)(tb0 + t), iteration, 4u + (uint32_t)traverser, seed_lo, seed_hi);
                ws.draw[sl] = ((unsigned long long)(x.x0 >> 5) << 26) | (unsigned long long)(x.x1 >> 6);   // u = N * 2^-53 (u53)
            }
        }
        sd_order();
        SD_STAMP(1);
        int width = 1, cb = 0, dbase = 0;
        uint32_t kept[4][(T * 24 + 63) / 64];
        // ---- forward --------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int d = 0; d < kPlies; d++) {
            const int p = d & 1, nl = 4 - (d >> 1);
            const bool trav_ply = p == traverser;
            if (!trav_ply && nl == 1) continue;                             // forced child, same position, same index within the next ply
            const int n_nodes = T * width, m = (d - traverser) >> 1;
            const int toff = sd_trav_off(traverser, m & 3), per = nl - 1, thr_off = sd_thr_off(traverser, (d >> 1) < 2 ? (d >> 1) : 2);
#pragma unroll
            for (int r = 0; r * 64 < n_nodes; r++) {
                const int f = r * 64 + lane;
                if (f >= n_nodes) continue;
                const uint32_t node = ws.idx[cb][f];
                if (trav_ply) {
                    const uint2 inf = s_node[toff + (int)node];
                    const uint32_t xbits = inf.x;
                    kept[m & 3][r] = node | (inf.y << 16);                 // the way back visits position f of this ply from the same lane: tree index and hand nibbles stay in a register
                    int t = 0;
#pragma unroll
                    for (int k = 1; k < T; k++) t += f >= k * width;
                    const int j = f - t * width;
                    // recurse on ALL legal actions, hand order (:326-336)
#pragma unroll
                    for (int k = 0; k < 4; k++) if (k < nl) ws.idx[cb ^ 1][f * nl + k] = (uint16_t)(node * nl + k);
                    ws.xb[t][sd_rank(m, j)] = xbits;                       // its memory row's features and mask follow in the sweep below
                } else {
                    // opponent: sample ONE action (:347-365): k_sdcfr_expand's Philox keying, its float comparisons as integer ones (k_sdcfr_policy)
                    const unsigned long long *th = s_thr + thr_off + (int)node * per;
                    const unsigned long long t0 = th[0], t1 = per > 1 ? th[1] : 1ull << 53, t2 = per > 2 ? th[2] : 1ull << 53;   // 2^53 > every N: never counted
                    const unsigned long long N = ws.draw[dbase + f];
                    int a = (int)(t0 <= N) + (int)(t1 <= N) + (int)(t2 <= N);
                    if (t0 == ~0ull) {                                     // probs.sum() == 0: np.random.choice(legal_actions), uniform, numpy's float64 arithmetic
                        const double u = (double)N * 0x1p-53;              // (exact: u53 of the two Philox words)
                        a = (int)(u * (double)nl);
                        a = a < nl - 1 ? a : nl - 1;
                    }
                    ws.idx[cb ^ 1][f] = (uint16_t)(node * nl + a);
                }
            }
            cb ^= 1;
            if (trav_ply) width *= nl; else dbase += n_nodes;
            sd_order();
        }
        SD_STAMP(2);
        // ---- the task's memory rows (:339-346), features and masks: 41 n_live rows, consecutive in the ring (up to its wrap) ------------------
        // A lane per piece, pieces in memory order, a row's nine pieces in nine consecutive lanes of ONE store instruction: the L2 sees each 136-byte row
        // whole.  (Measured and not kept: the [32], [33] pair of every row in a loop of its own -- 8-byte stores 136 bytes apart -- and the loops unrolled
        // four times or fully: a lone wavefront's sweep got 35 % shorter, the launch at 32768 traversals 25 % LONGER -- the stores arrive in a worse order;
        // the 8-byte-aligned 16-byte pieces as ONE store each (inline asm; the compiler emits two 8-byte stores): no difference.)
        {
            const uint32_t *xbv = &ws.xb[0][0];
            for (int e = lane; e < n_live * 41 * 9; e += 64) {              // features: eight 16-byte pieces and one of 8 bytes per 136-byte row
                const int rr = e / 9, i = e - rr * 9;
                const uint32_t hb = xbv[rr] >> (4 * i);
                uint32_t row = row0 + (uint32_t)rr;
                row = row >= capacity ? row - capacity : row;
                SD_WALK_ROWMASK(row);
                float *dst = mem_feat + (size_t)row * 34 + 4 * i;
                if (i < 8) *reinterpret_cast<SdF4A8 *>(dst) = SdF4A8{(float)(hb & 1u), (float)((hb >> 1) & 1u), (float)((hb >> 2) & 1u), (float)((hb >> 3) & 1u)};
                else *reinterpret_cast<float2 *>(dst) = make_float2(1.0f, 0.0f);   // [32] = float(player == current_player), [33] unused
            }
            if (mem_mask != nullptr)                                        // no mask array (the default ring): a row's mask IS features[0..16) -- sd_mask_note
            for (int e = lane; e < n_live * 41 * 4; e += 64) {              // masks: 4 sixteen-byte pieces per 64-byte row
                const int rr = e >> 2, i = e & 3;
                const uint32_t hb = xbv[rr] >> (4 * i);
                uint32_t row = row0 + (uint32_t)rr;
                row = row >= capacity ? row - capacity : row;
                SD_WALK_ROWMASK(row);
                reinterpret_cast<float4 *>(mem_mask + (size_t)row * 16)[i] = make_float4((float)(hb & 1u), (float)((hb >> 1) & 1u), (float)((hb >> 2) & 1u), (float)((hb >> 3) & 1u));
            }
        }
        SD_STAMP(3);
        // ---- leaves, then backward -------------------------------------------------------------------------------------------------
        for (int f = lane; f < T * width; f += 64) {
            const int p0 = s_payoff[ws.idx[cb][f]];
            ws.val[f] = 0.5f * (float)(traverser == 0 ? p0 : -p0);
        }
        sd_order();
#pragma unroll
        for (int d = kPlies - 1; d >= 0; d--) {
            const int p = d & 1, nl = 4 - (d >> 1);
            if (p != traverser) continue;                                  // opponent ply: the sampled child's value is returned unchanged (:363-365), same position
            width /= nl;
            const int m = (d - traverser) >> 1, off_d = sd_trav_off(traverser, m);
#pragma unroll
            for (int r = 0; r * 64 < T * width; r++) {
            const int f0 = r * 64, f = f0 + lane;
            if (f < T * width) {
                int t = 0;
#pragma unroll
                for (int k = 1; k < T; k++) t += f >= k * width;
                const int j = f - t * width;
                const int node = (int)(kept[m][r] & 0xFFFFu);
                const uint32_t hand = kept[m][r] >> 16;
                const float4 pol = s_pol[off_d + node];
                const float pl[4] = {pol.x, pol.y, pol.z, pol.w};
                // value = sum policy * action value (float32, hand order, :335); regrets = counterfactual_values - value over all 16 slots, where
                // counterfactual_values is zero at the illegal slots (:324, :339): a row holds at most nl + 1 DISTINCT numbers -- av_k - value at the
                // legal cards, 0 - value everywhere else -- so the max-abs normalisation (add_experience :73-74) divides those and not 16 slots
                float value = 0.0f, av[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    av[k] = k < nl ? ws.val[f * nl + k] : 0.0f;
                    if (k < nl) value += pl[k] * av[k];
                }
                __builtin_amdgcn_wave_barrier();                            // (every lane of the round has read its children before any writes)
                ws.val[f] = value;
                float rv[4], ri = 0.0f - value, mx = fabsf(ri);
#pragma unroll
                for (int k = 0; k < 4; k++) { rv[k] = av[k] - value; if (k < nl) { const float a = fabsf(rv[k]); mx = a > mx ? a : mx; } }
                if (mx > 0.0f) {
                    const float den = mx + 1e-8f;
                    ri = ri / den;
#pragma unroll
                    for (int k = 0; k < 4; k++) if (k < nl) rv[k] = rv[k] / den;
                }
                // the row is assembled in LDS at its rank within the task -- sixteen times the illegal slots' value, then the legal cards' values over it (a
                // wavefront's LDS writes land in program order) -- and leaves for memory with the task's other regret rows in ONE sweep after the pass
                float *srow = &ws.regs[41 * t + sd_rank(m, j)][0];
#pragma unroll
                for (int i = 0; i < 4; i++) reinterpret_cast<float4 *>(srow)[i] = make_float4(ri, ri, ri, ri);
#pragma unroll
                for (int k = 0; k < 4; k++) if (k < nl) srow[(hand >> (4 * k)) & 15u] = rv[k];
            }
            }
            sd_order();
        }
        // the task's regret rows: 41 n_live rows of 64 bytes, consecutive in the ring like its feature and mask rows, four lanes to a row
        for (int e = lane; e < n_live * 41 * 4; e += 64) {
            const int rr = e >> 2, i = e & 3;
            uint32_t row = row0 + (uint32_t)rr;
            row = row >= capacity ? row - capacity : row;
            SD_WALK_ROWMASK(row);
            reinterpret_cast<float4 *>(mem_regret + (size_t)row * 16)[i] = reinterpret_cast<const float4 *>(&ws.regs[rr][0])[i];
        }
        if (lane < n_live) root_values[tb0 + lane] = ws.val[lane];
        sd_order();
        SD_STAMP(4);
        int got = 0;
        if (lane == 0) got = atomicAdd(s_next, 1);
        c = __builtin_amdgcn_readfirstlane(got);
        SD_STAMP(5);
#ifdef SCOPA_WALK_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            for (int i = 1; i < 8; i++) g_wk_stamps[i] += sd_acc_[i];
            g_wk_stamps[13] += clock64() - sd_t0_; g_wk_stamps[14] += wall_clock64() - sd_r0_; g_wk_stamps[15] += 1;
        }
#endif
    }
}

#ifdef SCOPA_WALK_STAMPS
extern "C" int scopa_debug_sdwalk_stamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wk_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wk_stamps), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
extern "C" int scopa_debug_sdcfr_stamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sd_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sd_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

extern "C" {

int32_t scopa_sdcfr_image_floats(void) { return kImgFloats; }

int32_t scopa_sdcfr_pack_weights(scopa_ctx *ctx, int32_t player, const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2,
                                 const float *d_w3, const float *d_b3, float *d_image) {
    if (!ctx || player < 0 || player > 1 || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_w3 || !d_b3 || !d_image) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ((uintptr_t)d_image & 15) == 0, SCOPA_EINVAL, "scopa_sdcfr_pack_weights: the image must be 16-byte aligned");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sdcfr_pack, dim3((kImgFloats + 255) / 256), dim3(256), 0, ctx->stream, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3,
                       d_image + (size_t)player * kImgFloats);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_sdcfr_mode(scopa_ctx *ctx, int32_t forward_per_visit) {
    if (!ctx || forward_per_visit < 0 || forward_per_visit > 1) return SCOPA_EINVAL;
    ctx->sdcfr_mode = forward_per_visit;
    return SCOPA_OK;
}

int32_t scopa_sdcfr_tuning(scopa_ctx *ctx, int32_t traversals_per_task, int32_t wavefronts_per_task) {
    if (!ctx || (traversals_per_task != 0 && traversals_per_task != 1 && traversals_per_task != 2 && traversals_per_task != 4 && traversals_per_task != 8) || wavefronts_per_task < 0 || wavefronts_per_task > 3)
        return SCOPA_EINVAL;
    ctx->sdcfr_tile_t = traversals_per_task;
    ctx->sdcfr_team_w = wavefronts_per_task;
    return SCOPA_OK;
}

int32_t scopa_sdcfr_traverse_fused(scopa_ctx *ctx, int32_t traverser, int32_t batch, const float *d_image, float *d_mem_feat,
                                   float *d_mem_regret, float *d_mem_mask, int64_t capacity, int64_t write_base,
                                   float *d_root_values, const double *d_uniforms, uint32_t iteration, uint32_t b0) {
    // sd_mask_note -- d_mem_mask may be NULL: at a traverser node the legal actions are the cards of the mover's hand (openspiel_mini_scopa.py:36-45), and
    // the first sixteen features are that hand's one-hot (deep_cfr.py:213-275), so a row this call writes has mask == features[0..16) and the 64-byte mask
    // stream is redundant: the caller keeps masks as a strided view of the feature array (DeviceMemory.mask) and the row shrinks from 264 to 200 bytes.
    if (!ctx || traverser < 0 || traverser > 1 || batch < 0 || (batch && (!d_image || !d_mem_feat || !d_mem_regret || !d_root_values)))
        return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_sdcfr_traverse_fused: no deal set");
    SC_REQUIRE(ctx, capacity >= 41 && (int64_t)batch * 41 <= capacity, SCOPA_EINVAL, "scopa_sdcfr_traverse_fused: memory ring too small for the batch");
    SC_REQUIRE(ctx, write_base >= 0 && write_base < capacity && capacity < ((int64_t)1 << 30), SCOPA_EINVAL,
               "scopa_sdcfr_traverse_fused: write_base must lie in [0, capacity) and capacity below 2^30 rows");
    SC_REQUIRE(ctx, ((uintptr_t)d_image & 15) == 0, SCOPA_EINVAL, "scopa_sdcfr_traverse_fused: the weight image must be 16-byte aligned");
    SC_REQUIRE(ctx, ((uintptr_t)d_mem_feat & 7) == 0 && ((uintptr_t)d_mem_regret & 15) == 0 && ((uintptr_t)d_mem_mask & 15) == 0, SCOPA_EINVAL,
               "scopa_sdcfr_traverse_fused: memory rows are stored 8 / 16 bytes at a time: d_mem_feat must be 8-byte, d_mem_regret / d_mem_mask 16-byte aligned");
    if (!batch) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    scopa::Range range_("scopa sdcfr traverse");
    if (!ctx->d_sdnode) SC_HIP(ctx, hipMalloc(&ctx->d_sdnode, sizeof(uint2) * kDecision));
    if (!ctx->sdnode_valid) {
        hipLaunchKernelGGL(k_sdcfr_nodeinfo, dim3((kDecision + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_states, (uint2 *)ctx->d_sdnode);
        SC_HIP(ctx, hipGetLastError());
        ctx->sdnode_valid = true;
    }
    if (!d_uniforms && ctx->sdcfr_mode == 0) {
        // the default: every decision node of the deal evaluated once (k_sdcfr_policy), then the traversals as walks over that table
        if (!ctx->d_sdpol) SC_HIP(ctx, hipMalloc(&ctx->d_sdpol, (sizeof(float4) + 3 * sizeof(unsigned long long)) * kDecision));   // policies, then thresholds
        unsigned long long *d_thr = reinterpret_cast<unsigned long long *>(reinterpret_cast<float4 *>(ctx->d_sdpol) + kDecision);
        hipLaunchKernelGGL(k_sdcfr_policy, dim3(kPolicyTiles), dim3(kPolicyWaves * 64), 0, ctx->stream,
                           (const uint2 *)ctx->d_sdnode, d_image, (float4 *)ctx->d_sdpol, d_thr);
        SC_HIP(ctx, hipGetLastError());
        // traversals per wavefront: more would use the 64 lanes better (frontiers T .. 24 T wide), but a task keeps its regret rows in LDS until its last
        // ply (2.6 KB per traversal), and ONE traversal per wavefront is what lets two workgroups of twelve wavefronts share a compute unit: measured
        // (policy + walk, one call) 24.2 / 86.2 us at 4096 / 32768 traversals with 1, 28.0 / 85.1 with 2 (one workgroup of sixteen per compute unit)
        int Tw = ctx->sdcfr_tile_t;
        if (Tw != 1 && Tw != 2 && Tw != 4 && Tw != 8) Tw = 1;
        const int wgs_per_cu = Tw == 1 ? 2 : 1;
        const int tasks_w = (batch + Tw - 1) / Tw, grid_w = tasks_w < wgs_per_cu * ctx->n_cus ? tasks_w : wgs_per_cu * ctx->n_cus;
        const size_t wave_w = Tw == 8 ? sizeof(SdWalk<8>) : Tw == 4 ? sizeof(SdWalk<4>) : Tw == 2 ? sizeof(SdWalk<2>) : sizeof(SdWalk<1>);
        const size_t lds_w = (size_t)kWalkTravNodes * (sizeof(float4) + sizeof(uint2)) + (size_t)kWalkThr * sizeof(unsigned long long) + (size_t)kTerminal + (size_t)sd_walk_waves(Tw) * wave_w;
        SC_REQUIRE(ctx, lds_w + 64 <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_sdcfr_traverse_fused: LDS (walk kernel)");
#define SD_WALK(TT, TR, BIT)                                                                                                                      \
    do {                                                                                                                                          \
        SC_LDS_ATTR(ctx, BIT, (k_sdcfr_walk<TT, TR>), ctx->lds_limit - 64);                                                                       \
        hipLaunchKernelGGL((k_sdcfr_walk<TT, TR>), dim3(grid_w), dim3(sd_walk_waves(TT) * 64), lds_w, ctx->stream, (const uint2 *)ctx->d_sdnode, ctx->d_payoff, \
                           (const float4 *)ctx->d_sdpol, (const unsigned long long *)d_thr, (int)batch, d_mem_feat, d_mem_regret, d_mem_mask, (uint32_t)capacity, \
                           (uint32_t)write_base, d_root_values, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), iteration, b0);                  \
    } while (0)
        if (traverser == 0) {
            if (Tw == 8) SD_WALK(8, 0, scopa::kLdsSdWalk8);
            else if (Tw == 4) SD_WALK(4, 0, scopa::kLdsSdWalk4);
            else if (Tw == 2) SD_WALK(2, 0, scopa::kLdsSdWalk2);
            else SD_WALK(1, 0, scopa::kLdsSdWalk1);
        } else {
            if (Tw == 8) SD_WALK(8, 1, scopa::kLdsSdWalk8b);
            else if (Tw == 4) SD_WALK(4, 1, scopa::kLdsSdWalk4b);
            else if (Tw == 2) SD_WALK(2, 1, scopa::kLdsSdWalk2b);
            else SD_WALK(1, 1, scopa::kLdsSdWalk1b);
        }
#undef SD_WALK
        SC_HIP(ctx, hipGetLastError());
        ctx->sdcfr_visits += (uint64_t)batch * (traverser == 0 ? 105 : 82);
        return SCOPA_OK;
    }
    // a forward pass per visit (scopa_sdcfr_mode 1, and the replayed draws of the tests).
    // T traversals per task: 4 fills the 16-node tiles best (21 tiles for traverser 0's 324 evaluated nodes; 2: 26 tiles for twice
    // the tasks).  W wavefronts per task: 1 = solo wavefronts, eight per compute unit; 2 / 3 = teams, twelve wavefronts per compute unit.
    const int T = (ctx->sdcfr_tile_t == 2 && !d_uniforms) ? 2 : 4;                // (replayed draws: the default shape only)
    const int W = (ctx->sdcfr_team_w && !d_uniforms) ? ctx->sdcfr_team_w : 1;   // measured: solo wavefronts 110 / 620 us at 4096 / 32768 traversals, teams of two 121 / 788, of three 117 / 753
    const int n_waves = sd_waves(W), n_teams = n_waves / W;
    const size_t team_bytes = T == 4 ? (W == 1 ? sizeof(SdTeam<4, 1>) : sizeof(SdTeam<4, 2>)) : (W == 1 ? sizeof(SdTeam<2, 1>) : sizeof(SdTeam<2, 2>));
    const size_t lds = (size_t)2 * kImgFloats * sizeof(float) + (size_t)kSdNodeSlots * sizeof(uint2) + (size_t)kTerminal + (size_t)n_teams * team_bytes + (size_t)n_waves * sizeof(SdPos);
    SC_REQUIRE(ctx, lds + 64 <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_sdcfr_traverse_fused: LDS");
    const int n_tasks = (batch + T - 1) / T;
    const int grid = n_tasks < ctx->n_cus ? n_tasks : ctx->n_cus;
#define SD_LAUNCH(TT, WW, RR, BIT)                                                                                                                      \
    do {                                                                                                                                              \
        SC_LDS_ATTR(ctx, BIT, (k_sdcfr_traverse<TT, WW, RR>), ctx->lds_limit - 64);   /* 64: the kernel's static LDS, beside the dynamic part */      \
        hipLaunchKernelGGL((k_sdcfr_traverse<TT, WW, RR>), dim3(grid), dim3(n_waves * 64), lds, ctx->stream, (const uint2 *)ctx->d_sdnode, ctx->d_payoff, \
                           d_image, (int)traverser, (int)batch, d_mem_feat, d_mem_regret, d_mem_mask, (uint32_t)capacity, (uint32_t)write_base,       \
                           d_root_values, d_uniforms, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), iteration, b0,                                \
                           reinterpret_cast<uint32_t *>(ctx->d_counters + 5));                                                                        \
    } while (0)
    if (d_uniforms) SD_LAUNCH(4, 1, true, scopa::kLdsSdcfr7);
    else if (T == 4 && W == 1) SD_LAUNCH(4, 1, false, scopa::kLdsSdcfr);
    else if (T == 4 && W == 2) SD_LAUNCH(4, 2, false, scopa::kLdsSdcfr2);
    else if (T == 4 && W == 3) SD_LAUNCH(4, 3, false, scopa::kLdsSdcfr3);
    else if (T == 2 && W == 1) SD_LAUNCH(2, 1, false, scopa::kLdsSdcfr4);
    else if (T == 2 && W == 2) SD_LAUNCH(2, 2, false, scopa::kLdsSdcfr5);
    else SD_LAUNCH(2, 3, false, scopa::kLdsSdcfr6);
#undef SD_LAUNCH
    SC_HIP(ctx, hipGetLastError());
    ctx->sdcfr_visits += (uint64_t)batch * (traverser == 0 ? 105 : 82);
    return SCOPA_OK;
}

}  // extern "C"
