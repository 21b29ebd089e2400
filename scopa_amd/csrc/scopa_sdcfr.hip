// scopa_sdcfr.hip -- external-sampling traversal for Single Deep CFR, level-synchronous over the 8 plies.
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
    for (int c = 0; c < 16; c++) { mem_regret[row * 16 + c] = reg[c]; mem_mask[row * 16 + c] = mask[i * 16 + c]; }
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
        SC_REQUIRE(ctx, d_feats && d_mask && d_mem_feat && d_mem_regret && d_mem_mask && capacity >= 41 && write_base >= 0, SCOPA_EINVAL,
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

template <int T>
struct SdWave {                 // per-wavefront scratch: T traversals in flight
    float pol_trav[T][41][4];   // policy (legal actions, hand order) of every traverser node: plies m = 0..3 at offsets 0,1,5,17
    float val[2][T * 24];       // values of the frontier flowing back up (position t * width + j)
    float pos[16][16];          // relu(adv) * mask of the tile in flight, [node][output]
    uint32_t ninfo[T * 24][2];  // the current ply's frontier: feature bits | hand nibbles + tree index << 16
    uint16_t idx[T][136];       // tree index of every frontier node, per ply (ply 8 = leaves), at idx_at(ply): a traversal's frontier is 1, <= 4, 4,
                                // <= 12, 12, <= 24, 24, 24, 24 wide (either traverser)
};
__host__ __device__ constexpr int idx_at(int d) { return d == 0 ? 0 : d == 1 ? 1 : d == 2 ? 5 : d == 3 ? 9 : d == 4 ? 21 : d == 5 ? 33 : d == 6 ? 57 : d == 7 ? 81 : 105; }
static_assert(idx_at(1) - idx_at(0) >= 1 && idx_at(2) - idx_at(1) >= 4 && idx_at(3) - idx_at(2) >= 4 && idx_at(4) - idx_at(3) >= 12 && idx_at(5) - idx_at(4) >= 12 &&
              idx_at(6) - idx_at(5) >= 24 && idx_at(7) - idx_at(6) >= 24 && idx_at(8) - idx_at(7) >= 24 && idx_at(8) + 24 <= 136,
              "every ply's frontier (widest over the two traversers) fits its slice of SdWave::idx");
static_assert(sizeof(SdWave<4>) % 16 == 0 && sizeof(SdWave<2>) % 16 == 0, "SdWave alignment");
constexpr int kSdWaves = 8;     // wavefronts per workgroup: two per SIMD (one walks its tiles' matrix phases while the other samples / expands)

__device__ __forceinline__ void sd_sync() {   // a wavefront's LDS operations execute in order: this only stops the compiler (and drains the queue)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f to_v4f(float4 x) { v4f r = {x.x, x.y, x.z, x.w}; return r; }
__device__ __forceinline__ v4f relu4(v4f x) { v4f r = {fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f), fmaxf(x[2], 0.0f), fmaxf(x[3], 0.0f)}; return r; }
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
__device__ unsigned long long g_sd_stamps[16];   // 0 frontier info | 1 layer 1 | 2 layer 2 | 3 layer 3 + policy | 4 expand / sample | 5 skipped plies | 6 leaves + backward | 7 take | 15 tasks
#define SD_STAMP(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_sd_stamps[i] += now_ - sd_prev_; sd_prev_ = now_; } while (0)
#else
#define SD_STAMP(i) do { } while (0)
#endif

template <int T>
__global__ void __launch_bounds__(kSdWaves * 64)
k_sdcfr_traverse(const uint2 *__restrict__ g_ninfo, const int8_t *__restrict__ g_payoff, const float *__restrict__ g_image,
                 int traverser, int batch, float *__restrict__ mem_feat, float *__restrict__ mem_regret, float *__restrict__ mem_mask,
                 long long capacity, long long write_base, float *__restrict__ root_values, const double *__restrict__ uniforms,
                 uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t b0) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_next[1];                                               // next task of this workgroup not taken yet
    float *s_w = reinterpret_cast<float *>(smem);                           // [2][kImgFloats]
    SdWave<T> *s_wave = reinterpret_cast<SdWave<T> *>(s_w + 2 * kImgFloats);   // [wavefronts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    if (tid == 0) s_next[0] = n_waves;
    for (int i = tid; i < 2 * kImgFloats / 4; i += blockDim.x)
        reinterpret_cast<float4 *>(s_w)[i] = reinterpret_cast<const float4 *>(g_image)[i];
    __syncthreads();
    SdWave<T> &ws = s_wave[wave];
    const int nj = lane & 15, q = lane >> 4;   // this lane's column (node of the tile) and K / row group

    // A TASK is T consecutive traversals (the last one of a batch may be short).  The workgroup owns tasks [first, first + count)
    // and its wavefronts TAKE them from a counter in LDS (the first one is static): the SIMD's arbiter favours its oldest
    // wavefront, so equal shares would leave the workgroup waiting for its slowest (scopa_mccfr.hip, main loop, has the
    // measurement).  Which wavefront walks a traversal does not matter: draws and memory-row positions are keyed by its id.
    const int n_tasks = (batch + T - 1) / T;
    const int per_wg = (n_tasks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int first = (int)blockIdx.x * per_wg, count = first < n_tasks ? (n_tasks - first < per_wg ? n_tasks - first : per_wg) : 0;
    for (int c = wave; c < count;) {
        const int tb0 = (first + c) * T;                                    // the task's first traversal (local id within the batch)
        const int n_live = batch - tb0 < T ? batch - tb0 : T;               // traversals t >= n_live are walked like the others but write nothing
#ifdef SCOPA_WALK_STAMPS
        unsigned long long sd_prev_ = clock64();
#endif
        if (lane < T) ws.idx[lane][idx_at(0)] = 0;
        sd_sync();
        int width = 1;                                                      // a traversal's frontier width at the current ply
        // ---- forward: plies 0..7 ----------------------------------------------------------------------------------------
#pragma unroll 1
        for (int d = 0; d < kPlies; d++) {
            const int p = d & 1, nl = 4 - (d >> 1);
            const bool trav_ply = p == traverser;
            const int n_nodes = T * width;                                  // position f = t * width + j
            const int m = (d - traverser) >> 1;                             // traverser-ply index when trav_ply
            const int moff = m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 5 : 17;
            if (!trav_ply && nl == 1) {
                // opponent node with ONE legal action (plies 6/7): whatever the advantages are, regret matching either puts all mass
                // on it or falls back to uniform over it (deep_cfr.py:353-359) -- the child is forced and nothing else of this
                // node is used (no memory row, no value weight), so its forward pass is skipped: 24 of the 105 / 82 node
                // evaluations of a traversal
                for (int f = lane; f < n_nodes; f += 64) {
                    int t = 0;
#pragma unroll
                    for (int k = 1; k < T; k++) t += f >= k * width;
                    const int j = f - t * width;
                    ws.idx[t][idx_at(d + 1) + j] = ws.idx[t][idx_at(d) + j];
                }
                sd_sync();
                SD_STAMP(5);
                continue;
            }
            // the ply's frontier: feature bits, hand nibbles and tree index of every node, one global load each (the tiles below read LDS)
            for (int f = lane; f < n_nodes; f += 64) {
                int t = 0;
#pragma unroll
                for (int k = 1; k < T; k++) t += f >= k * width;
                const int j = f - t * width;
                const uint32_t node = ws.idx[t][idx_at(d) + j];
                const uint2 inf = g_ninfo[level_offset(d) + (int)node];
                ws.ninfo[f][0] = inf.x;
                ws.ninfo[f][1] = inf.y | (node << 16);
            }
            sd_sync();
            SD_STAMP(0);
            const float *W = s_w + p * kImgFloats;
            const float4 *w1 = reinterpret_cast<const float4 *>(W + kImgW1) + lane;
            const float4 *c1 = reinterpret_cast<const float4 *>(W + kImgC1) + q;
            const float4 *w2 = reinterpret_cast<const float4 *>(W + kImgW2) + lane;
            const float4 *b2 = reinterpret_cast<const float4 *>(W + kImgB2) + q;
            const float4 *w3 = reinterpret_cast<const float4 *>(W + kImgW3) + lane;
            const float4 *b3 = reinterpret_cast<const float4 *>(W + kImgB3) + q;
#pragma unroll 1
            for (int f0 = 0; f0 < n_nodes; f0 += 16) {
                const int f = f0 + nj;
                const bool live = f < n_nodes;
                const int fc = live ? f : n_nodes - 1;                      // lanes beyond the frontier compute a copy of its last node and store nothing
                const uint32_t xbits = ws.ninfo[fc][0], hn = ws.ninfo[fc][1];
                const uint32_t hand = hn & 0xFFFFu, node = hn >> 16;
                // ---- layer 1: K-step s = 4 g + c covers features 4 s .. 4 s + 3 (k index = lane / 16) --------------------------
                v4f h1[8];
#pragma unroll
                for (int mt = 0; mt < 8; mt++) h1[mt] = to_v4f(c1[mt * 4]);
                const uint32_t xs = xbits >> q;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    float4 w[8];
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) w[mt] = w1[(mt * 2 + g) * 64];
                    const float x0 = (float)((xs >> (16 * g)) & 1u), x1 = (float)((xs >> (16 * g + 4)) & 1u);
                    const float x2 = (float)((xs >> (16 * g + 8)) & 1u), x3 = (float)((xs >> (16 * g + 12)) & 1u);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(w[mt].x, x0, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(w[mt].y, x1, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(w[mt].z, x2, h1[mt]);
#pragma unroll
                    for (int mt = 0; mt < 8; mt++) h1[mt] = mfma16(w[mt].w, x3, h1[mt]);
                }
#pragma unroll
                for (int mt = 0; mt < 8; mt++) h1[mt] = relu4(h1[mt]);      // register r of tile mt = unit 16 mt + 4 q + r of node nj
                SD_STAMP(1);
                // ---- layer 2: K-step (mt, r) covers the units 16 mt + 4 k + r, k = lane / 16: B operand = h1[mt][r] as it stands -----
                v4f h2[4];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) h2[nt] = to_v4f(b2[nt * 4]);
#pragma unroll
                for (int mt = 0; mt < 8; mt++) {
                    float4 w[4];
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) w[nt] = w2[(nt * 8 + mt) * 64];
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(w[nt].x, h1[mt][0], h2[nt]);
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(w[nt].y, h1[mt][1], h2[nt]);
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(w[nt].z, h1[mt][2], h2[nt]);
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) h2[nt] = mfma16(w[nt].w, h1[mt][3], h2[nt]);
                }
#pragma unroll
                for (int nt = 0; nt < 4; nt++) h2[nt] = relu4(h2[nt]);
                SD_STAMP(2);
                // ---- layer 3: 16 outputs, K-step (nt, r); two accumulator chains ------------------------------------------------------
                float adv[4];
                {
                    v4f o0 = to_v4f(b3[0]), o1 = {0.0f, 0.0f, 0.0f, 0.0f};
                    float4 w[4];
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) w[nt] = w3[nt * 64];
                    o0 = mfma16(w[0].x, h2[0][0], o0); o1 = mfma16(w[1].x, h2[1][0], o1);
                    o0 = mfma16(w[0].y, h2[0][1], o0); o1 = mfma16(w[1].y, h2[1][1], o1);
                    o0 = mfma16(w[0].z, h2[0][2], o0); o1 = mfma16(w[1].z, h2[1][2], o1);
                    o0 = mfma16(w[0].w, h2[0][3], o0); o1 = mfma16(w[1].w, h2[1][3], o1);
                    o0 = mfma16(w[2].x, h2[2][0], o0); o1 = mfma16(w[3].x, h2[3][0], o1);
                    o0 = mfma16(w[2].y, h2[2][1], o0); o1 = mfma16(w[3].y, h2[3][1], o1);
                    o0 = mfma16(w[2].z, h2[2][2], o0); o1 = mfma16(w[3].z, h2[3][2], o1);
                    o0 = mfma16(w[2].w, h2[2][3], o0); o1 = mfma16(w[3].w, h2[3][3], o1);
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
                    z += __shfl_xor(z, 16); z += __shfl_xor(z, 32);        // the four row groups
                    *reinterpret_cast<float4 *>(&ws.pos[nj][4 * q]) = pv;
                }
                const float den = z > 1e-8f ? z : 1e-8f;                   // clamp_min(eps)
                sd_sync();
                SD_STAMP(3);
                int t = 0;
#pragma unroll
                for (int k = 1; k < T; k++) t += f >= k * width;
                const int j = f - t * width;
                if (trav_ply) {
                    // recurse on ALL legal actions, hand order (:326-336): lane (q, node) takes action q
                    const float pk = q < nl ? ws.pos[nj][(hand >> (4 * q)) & 15u] / den : 0.0f;
                    if (live) {
                        ws.pol_trav[t][moff + j][q] = pk;
                        if (q < nl) ws.idx[t][idx_at(d + 1) + j * nl + q] = (uint16_t)(node * nl + q);
                    }
                } else {
                    // opponent: sample ONE action (:347-365); the four lanes of a node draw the same number
                    float pk[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) pk[k] = k < nl ? ws.pos[nj][(hand >> (4 * k)) & 15u] / den : 0.0f;
                    float sum = pk[0];
#pragma unroll
                    for (int k = 1; k < 4; k++) if (k < nl) sum += pk[k];  // action_probs.sum(), float32, left to right
                    const int tb = tb0 + t;
                    double u;
                    if (uniforms) u = (live && t < n_live) ? uniforms[((size_t)tb * kPlies + d) * 24 + j] : 0.0;
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
                        a = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) if (k < nl && cdf[k] / last <= u) a = k + 1;
                        a = a < nl - 1 ? a : nl - 1;
                    }
                    if (live && q == 0) ws.idx[t][idx_at(d + 1) + j] = (uint16_t)(node * nl + a);
                }
                sd_sync();
                SD_STAMP(4);
            }
            if (trav_ply) width *= nl;
        }
        // ---- leaves, then backward ---------------------------------------------------------------------------------------
        for (int f = lane; f < T * width; f += 64) {
            int t = 0;
#pragma unroll
            for (int k = 1; k < T; k++) t += f >= k * width;
            const int p0 = g_payoff[ws.idx[t][idx_at(8) + f - t * width]];
            ws.val[0][f] = 0.5f * (float)(traverser == 0 ? p0 : -p0);
        }
        sd_sync();
        int cur = 0;
#pragma unroll 1
        for (int d = kPlies - 1; d >= 0; d--) {
            const int p = d & 1, nl = 4 - (d >> 1);
            const bool trav_ply = p == traverser;
            if (trav_ply) width /= nl;
            const int m = (d - traverser) >> 1;
            const int moff = m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 5 : 17;
            for (int f = lane; f < T * width; f += 64) {
                if (!trav_ply) { ws.val[cur ^ 1][f] = ws.val[cur][f]; continue; }   // the sampled child's value is returned unchanged (:363-365)
                int t = 0;
#pragma unroll
                for (int k = 1; k < T; k++) t += f >= k * width;
                const int j = f - t * width;
                const uint2 inf = g_ninfo[level_offset(d) + ws.idx[t][idx_at(d) + j]];
                const uint32_t xbits = inf.x, hand = inf.y;
                float value = 0.0f, cfv[16];
#pragma unroll
                for (int cc = 0; cc < 16; cc++) cfv[cc] = 0.0f;           // counterfactual_values = zeros(16) (:324)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < nl) {
                        const float av = ws.val[cur][(t * width + j) * nl + k];
                        value += ws.pol_trav[t][moff + j][k] * av;           // value += policy[action] * action_value, float32 (:335)
                        const int c = (int)((hand >> (4 * k)) & 15u);
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) if (cc == c) cfv[cc] = av;
                    }
                }
                ws.val[cur ^ 1][f] = value;
                if (t < n_live) {
                    float mx = 0.0f, reg[16];
#pragma unroll
                    for (int cc = 0; cc < 16; cc++) { reg[cc] = cfv[cc] - value; const float a = fabsf(reg[cc]); mx = a > mx ? a : mx; }   // illegal slots = -value
                    const float den = mx + 1e-8f;
                    if (mx > 0.0f) {
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) reg[cc] = reg[cc] / den;    // add_experience (:73-74)
                    }
                    // ring position: the reference appends in DFS post-order; rank of this traverser node within its traversal
                    const int Tn[4] = {41, 10, 3, 1};          // traverser nodes in the subtree of a traverser node of ply index m
                    int jj = j, rank = Tn[m] - 1;
                    for (int qd = m - 1; qd >= 0; qd--) { const int radix = 4 - qd; rank += (jj % radix) * Tn[qd + 1]; jj /= radix; }
                    const long long row = (write_base + (long long)(tb0 + t) * 41 + rank) % capacity;
                    float2 *mf = reinterpret_cast<float2 *>(mem_feat + row * 34);          // 136-byte rows: 8-byte aligned
                    float4 *mm = reinterpret_cast<float4 *>(mem_mask + row * 16), *mr = reinterpret_cast<float4 *>(mem_regret + row * 16);
#pragma unroll
                    for (int i = 0; i < 16; i++) mf[i] = make_float2((float)((xbits >> (2 * i)) & 1u), (float)((xbits >> (2 * i + 1)) & 1u));
                    mf[16] = make_float2(1.0f, 0.0f);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        mm[i] = make_float4((float)((xbits >> (4 * i)) & 1u), (float)((xbits >> (4 * i + 1)) & 1u), (float)((xbits >> (4 * i + 2)) & 1u), (float)((xbits >> (4 * i + 3)) & 1u));
                        mr[i] = make_float4(reg[4 * i], reg[4 * i + 1], reg[4 * i + 2], reg[4 * i + 3]);
                    }
                }
            }
            cur ^= 1;
            sd_sync();
        }
        if (lane < n_live) root_values[tb0 + lane] = ws.val[cur][lane];
        sd_sync();
        SD_STAMP(6);
        int got = 0;
        if (lane == 0) got = atomicAdd(s_next, 1);
        c = __builtin_amdgcn_readfirstlane(got);
        SD_STAMP(7);
#ifdef SCOPA_WALK_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_sd_stamps[15] += 1;
#endif
    }
}

#ifdef SCOPA_WALK_STAMPS
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

int32_t scopa_sdcfr_tile_traversals(scopa_ctx *ctx, int32_t traversals_per_wavefront) {
    if (!ctx || (traversals_per_wavefront != 0 && traversals_per_wavefront != 2 && traversals_per_wavefront != 4)) return SCOPA_EINVAL;
    ctx->sdcfr_tile_t = traversals_per_wavefront;
    return SCOPA_OK;
}

int32_t scopa_sdcfr_traverse_fused(scopa_ctx *ctx, int32_t traverser, int32_t batch, const float *d_image, float *d_mem_feat,
                                   float *d_mem_regret, float *d_mem_mask, int64_t capacity, int64_t write_base,
                                   float *d_root_values, const double *d_uniforms, uint32_t iteration, uint32_t b0) {
    if (!ctx || traverser < 0 || traverser > 1 || batch < 0 || (batch && (!d_image || !d_mem_feat || !d_mem_regret || !d_mem_mask || !d_root_values)))
        return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_sdcfr_traverse_fused: no deal set");
    SC_REQUIRE(ctx, capacity >= 41 && (int64_t)batch * 41 <= capacity && write_base >= 0, SCOPA_EINVAL, "scopa_sdcfr_traverse_fused: memory ring too small for the batch");
    SC_REQUIRE(ctx, ((uintptr_t)d_image & 15) == 0, SCOPA_EINVAL, "scopa_sdcfr_traverse_fused: the weight image must be 16-byte aligned");
    SC_REQUIRE(ctx, ((uintptr_t)d_mem_feat & 7) == 0 && ((uintptr_t)d_mem_regret & 15) == 0 && ((uintptr_t)d_mem_mask & 15) == 0, SCOPA_EINVAL,
               "scopa_sdcfr_traverse_fused: memory rows are stored 8 / 16 bytes at a time: d_mem_feat must be 8-byte, d_mem_regret / d_mem_mask 16-byte aligned");
    if (!batch) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_sdnode) SC_HIP(ctx, hipMalloc(&ctx->d_sdnode, sizeof(uint2) * kDecision));
    if (!ctx->sdnode_valid) {
        hipLaunchKernelGGL(k_sdcfr_nodeinfo, dim3((kDecision + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_states, (uint2 *)ctx->d_sdnode);
        SC_HIP(ctx, hipGetLastError());
        ctx->sdnode_valid = true;
    }
    // T traversals per wavefront: 4 fills the 16-node tiles best (21 tiles for traverser 0's 324 evaluated nodes); with fewer than
    // 8 tasks of 4 per compute unit a wavefront per SIMD would walk alone, so small batches take 2 (26 tiles, twice the wavefronts)
    int T = ctx->sdcfr_tile_t;
    if (T == 0) T = 4;
    const size_t wave_bytes = T == 4 ? sizeof(SdWave<4>) : sizeof(SdWave<2>);
    const size_t lds = (size_t)2 * kImgFloats * sizeof(float) + (size_t)kSdWaves * wave_bytes;
    SC_REQUIRE(ctx, lds + 64 <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_sdcfr_traverse_fused: LDS");
    const int n_tasks = (batch + T - 1) / T;
    const int grid = n_tasks < ctx->n_cus ? n_tasks : ctx->n_cus;
    if (T == 4) {
        SC_LDS_ATTR(ctx, scopa::kLdsSdcfr, k_sdcfr_traverse<4>, ctx->lds_limit - 64);   // 64: the kernel's static LDS (s_next), beside the dynamic part
        hipLaunchKernelGGL(k_sdcfr_traverse<4>, dim3(grid), dim3(kSdWaves * 64), lds, ctx->stream, (const uint2 *)ctx->d_sdnode, ctx->d_payoff, d_image,
                           (int)traverser, (int)batch, d_mem_feat, d_mem_regret, d_mem_mask, (long long)capacity, (long long)write_base, d_root_values,
                           d_uniforms, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), iteration, b0);
    } else {
        SC_LDS_ATTR(ctx, scopa::kLdsSdcfr2, k_sdcfr_traverse<2>, ctx->lds_limit - 64);
        hipLaunchKernelGGL(k_sdcfr_traverse<2>, dim3(grid), dim3(kSdWaves * 64), lds, ctx->stream, (const uint2 *)ctx->d_sdnode, ctx->d_payoff, d_image,
                           (int)traverser, (int)batch, d_mem_feat, d_mem_regret, d_mem_mask, (long long)capacity, (long long)write_base, d_root_values,
                           d_uniforms, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), iteration, b0);
    }
    SC_HIP(ctx, hipGetLastError());
    ctx->sdcfr_visits += (uint64_t)batch * (traverser == 0 ? 105 : 82);
    return SCOPA_OK;
}

}  // extern "C"
