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
// 8 backward steps).  One WAVEFRONT walks one traversal level-synchronously; both players' advantage MLPs
// (34-128-64-16, float32, 13 776 parameters = 55 KB each) sit in LDS for the whole launch and the forward pass of the
// <= 24 frontier nodes of a ply runs inside the wave, kG = 4 nodes at a time, on the matrix cores (v_mfma_f32_4x4x1: 16
// independent 4x4 outer products per instruction -- 4 hidden units x 4 nodes per lane quad):
//   layer 1  K = 32 dense 0/1 inputs built from the node's hand/table bit masks (a zero input adds an exact zero), bias and
//            the constant-1 feature's column as the initial accumulator, 128 units = two halves of 16 blocks
//   layer 2  K = 128, 64 units = 16 blocks; per four inputs two 16-byte LDS reads (weights and activations both kept as
//            k-quads) and four MFMAs
//   layer 3  K = 64, 16 outputs = 4 blocks x 4 K-splits, partial sums added with two cross-lane shuffles; regret matching
//            (sum of relu(adv)*mask over a node's 16 outputs) stays in the accumulator layout
// Opponent nodes with a single legal action skip the forward pass: their child is forced and nothing else of them is used.
// then expand / sample exactly as k_sdcfr_expand does (same Philox keying, so the two paths sample identical actions), and
// after ply 7 the values flow back up inside the wave with the memory rows written straight to the caller's ring.
// Weights per net, as one float32 buffer: W1^T [34][128] | b1 [128] | W2^T [128][64] | b2 [64] | W3^T [64][16] | b3 [16].
namespace {
constexpr int kNetFloats = 34 * 128 + 128 + 128 * 64 + 64 + 64 * 16 + 16;  // 13 776
constexpr int kW1 = 0, kB1 = 34 * 128, kW2 = kB1 + 128, kB2 = kW2 + 128 * 64, kW3 = kB2 + 64, kB3 = kW3 + 64 * 16;

constexpr int kG = 4;         // frontier nodes evaluated together (a "group"); measured: 4 nodes x 10 wavefronts 1.18 ms per iteration, 8 nodes x 6 wavefronts (what LDS then allows) 1.76 ms
struct SdWave {               // per-wavefront scratch
    float h1[128][kG];        // hidden layer 1 of the nodes in flight; hidden layer 2 ([64][kG]) takes the first half of the same storage once
                              // layer 2 has read all of layer 1 (a wavefront's LDS operations execute in order) -- 1 KB less per wavefront
    float pol_trav[41][4];    // policy (legal actions, hand order) of every traverser node: plies m = 0..3 at offsets 0,1,5,17
    float val[2][24];
    float polcur[kG][4];
    scopa_state st[kG];       // packed states of the nodes in flight
    uint16_t idx[136];        // tree index of every frontier node, per ply (ply 8 = leaves), at idx_at(ply): the frontier is 1, <= 4, 4, <= 12, 12,
                              // <= 24, 24, 24, 24 wide (either traverser) -- packed, so that SIXTEEN wavefronts fit beside the two nets: 4 per SIMD
};
__host__ __device__ constexpr int idx_at(int d) { return d == 0 ? 0 : d == 1 ? 1 : d == 2 ? 5 : d == 3 ? 9 : d == 4 ? 21 : d == 5 ? 33 : d == 6 ? 57 : d == 7 ? 81 : 105; }
static_assert(idx_at(1) - idx_at(0) >= 1 && idx_at(2) - idx_at(1) >= 4 && idx_at(3) - idx_at(2) >= 4 && idx_at(4) - idx_at(3) >= 12 && idx_at(5) - idx_at(4) >= 12 &&
              idx_at(6) - idx_at(5) >= 24 && idx_at(7) - idx_at(6) >= 24 && idx_at(8) - idx_at(7) >= 24 && idx_at(8) + 24 <= 136,
              "every ply's frontier (widest over the two traversers) fits its slice of SdWave::idx");
static_assert(sizeof(SdWave) % 16 == 0, "SdWave alignment");

__device__ __forceinline__ void sd_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace

#ifdef SCOPA_WALK_STAMPS   // development build only: shader-clock stamps of wavefront 0 of workgroup 0 (tests/tools/sdcfr_stamps.py)
__device__ unsigned long long g_sd_stamps[16];   // 0 state load | 1 layer 1 | 2 layer 2 | 3 layer 3 + policy | 4 expand / sample | 5 skipped plies | 6 leaves + backward | 7 take | 15 traversals
#define SD_STAMP(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_sd_stamps[i] += now_ - sd_prev_; sd_prev_ = now_; } while (0)
#else
#define SD_STAMP(i) do { } while (0)
#endif

__global__ void __launch_bounds__(1024)
k_sdcfr_traverse(const scopa_state *__restrict__ g_states, const int8_t *__restrict__ g_payoff, const float *__restrict__ g_weights,
                 int traverser, int batch, float *__restrict__ mem_feat, float *__restrict__ mem_regret, float *__restrict__ mem_mask,
                 long long capacity, long long write_base, float *__restrict__ root_values, const double *__restrict__ uniforms,
                 uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t b0) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_next[1];                                               // next traversal of this workgroup not taken yet
    float *s_w = reinterpret_cast<float *>(smem);                           // [2][kNetFloats]
    SdWave *s_wave = reinterpret_cast<SdWave *>(s_w + 2 * kNetFloats);      // [wavefronts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    if (tid == 0) s_next[0] = n_waves;
    for (int i = tid; i < 2 * kNetFloats / 4; i += blockDim.x)
        reinterpret_cast<float4 *>(s_w)[i] = reinterpret_cast<const float4 *>(g_weights)[i];
    __syncthreads();
    // The three weight matrices are kept as k-QUADS, Wq[k/4][unit][k%4] (the buffer holds W^T[k][unit]), so that a lane fetches
    // its unit's weights for four inputs with one 16-byte, conflict-free LDS read -- the A operand of four v_mfma_f32_4x4x1.
    // In-place permutation through registers: W1 rows 0..31 (rows 32, 33 stay: constant-1 feature, unused feature), W2, W3.
    {
        constexpr int kQ1 = 8 * 128, kQ2 = 32 * 64, kQ3 = 16 * 16, kQ = kQ1 + kQ2 + kQ3;   // float4 quads per net
        constexpr int kPer = (kQ + 511) / 512;                                            // per thread at >= 512 threads
        float4 t[2][kPer];
        auto src_of = [&](int net, int e) -> const float * {
            const float *base = s_w + net * kNetFloats;
            if (e < kQ1) return base + kW1 + (e >> 7) * 4 * 128 + (e & 127);
            e -= kQ1;
            if (e < kQ2) return base + kW2 + (e >> 6) * 4 * 64 + (e & 63);
            e -= kQ2;
            return base + kW3 + (e >> 4) * 4 * 16 + (e & 15);
        };
        auto stride_of = [&](int e) { return e < kQ1 ? 128 : e < kQ1 + kQ2 ? 64 : 16; };
        auto dst_of = [&](int net, int e) -> float4 * {
            float *base = s_w + net * kNetFloats;
            if (e < kQ1) return reinterpret_cast<float4 *>(base + kW1) + e;
            e -= kQ1;
            if (e < kQ2) return reinterpret_cast<float4 *>(base + kW2) + e;
            return reinterpret_cast<float4 *>(base + kW3) + (e - kQ2);
        };
#pragma unroll
        for (int net = 0; net < 2; net++)
#pragma unroll
            for (int r = 0; r < kPer; r++) {
                const int e = tid + r * (int)blockDim.x;
                if (e < kQ) { const float *src = src_of(net, e); const int st = stride_of(e); t[net][r] = make_float4(src[0], src[st], src[2 * st], src[3 * st]); }
            }
        __syncthreads();
#pragma unroll
        for (int net = 0; net < 2; net++)
#pragma unroll
            for (int r = 0; r < kPer; r++) {
                const int e = tid + r * (int)blockDim.x;
                if (e < kQ) *dst_of(net, e) = t[net][r];
            }
        __syncthreads();
    }
    SdWave &ws = s_wave[wave];

    // The workgroup owns traversals [first, first + count) and its wavefronts TAKE them from a counter in LDS (the first one is
    // static): 10 wavefronts share 4 SIMDs unevenly and the SIMD's arbiter favours its oldest wavefront, so equal shares would leave
    // the workgroup waiting for its slowest wavefront (scopa_mccfr.hip, main loop, has the measurement).  Which wavefront walks a
    // traversal does not matter: draws and memory-row positions are keyed by the traversal id.
    const int per_wg = (batch + (int)gridDim.x - 1) / (int)gridDim.x;
    const int first = (int)blockIdx.x * per_wg, count = first < batch ? (batch - first < per_wg ? batch - first : per_wg) : 0;
    for (int c = wave; c < count;) {
        const int tb = first + c;
#ifdef SCOPA_WALK_STAMPS
        unsigned long long sd_prev_ = clock64();
#endif
        if (lane == 0) ws.idx[idx_at(0)] = 0;
        sd_sync();
        int width = 1;
        // ---- forward: plies 0..7 ----------------------------------------------------------------------------------------
#pragma unroll 1
        for (int d = 0; d < kPlies; d++) {
            const int p = d & 1, nl = 4 - (d >> 1);
            const bool trav_ply = p == traverser;
            const float *W = s_w + p * kNetFloats;
            const int m = (d - traverser) >> 1;                       // traverser-ply index when trav_ply
            const int moff = m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 5 : 17;
            if (!trav_ply && nl == 1) {
                // opponent node with ONE legal action (plies 6/7): whatever the advantages are, regret matching either puts all mass
                // on it or falls back to uniform over it (deep_cfr.py:353-359) -- the child is forced and nothing else of this
                // node is used (no memory row, no value weight), so its forward pass is skipped: 24 of the 105 / 82 node
                // evaluations of a traversal
                if (lane < width) ws.idx[idx_at(d + 1) + lane] = ws.idx[idx_at(d) + lane];
                sd_sync();
                SD_STAMP(5);
                continue;
            }
            typedef float v4f __attribute__((ext_vector_type(4)));
            static_assert(kG == 4, "the 4x4x1 mapping serves four nodes per group");
            // The whole forward pass runs on the matrix cores with v_mfma_f32_4x4x1 (16 independent 4x4 outer products per
            // instruction, K = 1): block b = lane / 4, A = four units' weights for input k (row i = lane % 4), B = input k of the four
            // nodes in flight (node j = lane % 4), D[i] of lane (b, j) = unit 4b+i of node j.
            const int bq = lane >> 2, nj = lane & 3;
            // layer-1 accumulators start from bias + the constant-1 feature's column (feature 33 is 0.0), per 64-unit half
            v4f bias1[2];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int i = 0; i < 4; i++) bias1[h][i] = W[kB1 + 64 * h + 4 * bq + i] + W[kW1 + 32 * 128 + 64 * h + 4 * bq + i];
            const float4 *w1q = reinterpret_cast<const float4 *>(W + kW1), *w2q = reinterpret_cast<const float4 *>(W + kW2) + lane;
            const float4 *w3q = reinterpret_cast<const float4 *>(W + kW3);
            for (int g0 = 0; g0 < width; g0 += kG) {
                // the group's packed states: one global load per node, then every lane reads the state of ITS node (lane % 4)
                if (lane < kG && g0 + lane < width) ws.st[lane] = g_states[level_offset(d) + ws.idx[idx_at(d) + g0 + lane]];
                sd_sync();
                SD_STAMP(0);
                const bool live = g0 + nj < width;
                const scopa_state sj = ws.st[nj];
                uint32_t hand_bits = 0, table_bits = 0;
                if (live) {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (k < sj.nh[p]) hand_bits |= 1u << nib(sj.hand[p], k);
#pragma unroll
                    for (int k = 0; k < 8; k++) if (k < sj.nt) table_bits |= 1u << nib(sj.table, k);
                }
                const uint32_t xbits = hand_bits | (table_bits << 16);       // the 32 one-hot features of this lane's node
                // layer 1: K = 32 dense 0/1 inputs (a zero input adds an exact zero), 128 units = two halves of 16 blocks
                {
                    v4f acc0 = bias1[0], acc1 = bias1[1];
#pragma unroll
                    for (int kq = 0; kq < 8; kq++) {
                        const float4 wa = w1q[kq * 128 + lane], wb = w1q[kq * 128 + 64 + lane];
                        const float x0 = (float)((xbits >> (4 * kq)) & 1u), x1 = (float)((xbits >> (4 * kq + 1)) & 1u);
                        const float x2 = (float)((xbits >> (4 * kq + 2)) & 1u), x3 = (float)((xbits >> (4 * kq + 3)) & 1u);
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.x, x0, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.x, x0, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.y, x1, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.y, x1, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.z, x2, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.z, x2, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.w, x3, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.w, x3, acc1, 0, 0, 0);
                    }
                    // h1 as k-quads for layer 2: h1q[unit / 4][node][unit % 4] -- exactly this lane's accumulator vectors
                    float4 *h1q = reinterpret_cast<float4 *>(&ws.h1[0][0]);
                    h1q[bq * 4 + nj] = make_float4(fmaxf(acc0[0], 0.0f), fmaxf(acc0[1], 0.0f), fmaxf(acc0[2], 0.0f), fmaxf(acc0[3], 0.0f));
                    h1q[(16 + bq) * 4 + nj] = make_float4(fmaxf(acc1[0], 0.0f), fmaxf(acc1[1], 0.0f), fmaxf(acc1[2], 0.0f), fmaxf(acc1[3], 0.0f));
                }
                sd_sync();
                SD_STAMP(1);
                // layer 2: K = 128, 64 units = 16 blocks.  Per four inputs: two 16-byte LDS reads and four MFMAs.
                {
                    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f}, accb = {0.0f, 0.0f, 0.0f, 0.0f};   // two independent chains (even / odd k-quads)
                    const float4 *hq = reinterpret_cast<const float4 *>(&ws.h1[0][0]) + nj;
#pragma unroll 4
                    for (int kq = 0; kq < 32; kq += 2) {
                        const float4 w = w2q[kq * 64], wb = w2q[(kq + 1) * 64];
                        const float4 h = hq[kq * 4], hb = hq[(kq + 1) * 4];
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, h.x, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.x, hb.x, accb, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, h.y, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.y, hb.y, accb, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, h.z, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.z, hb.z, accb, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, h.w, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wb.w, hb.w, accb, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[i] += accb[i];
                    // h2 as k-quads for layer 3: h2q[unit / 4][node][unit % 4]
                    reinterpret_cast<float4 *>(&ws.h1[0][0])[bq * 4 + nj] =
                        make_float4(fmaxf(acc[0] + W[kB2 + 4 * bq], 0.0f), fmaxf(acc[1] + W[kB2 + 4 * bq + 1], 0.0f),
                                    fmaxf(acc[2] + W[kB2 + 4 * bq + 2], 0.0f), fmaxf(acc[3] + W[kB2 + 4 * bq + 3], 0.0f));
                }
                sd_sync();
                SD_STAMP(2);
                // layer 3: K = 64, 16 outputs = 4 blocks; the other factor 4 of the 16 blocks splits K (block = ks * 4 + ob), the four
                // partial sums are added across lanes afterwards
                {
                    const int ks = bq >> 2, ob = bq & 3;
                    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
                    const float4 *hq = reinterpret_cast<const float4 *>(&ws.h1[0][0]);   // = hidden layer 2 now
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int kq = ks * 4 + t;
                        const float4 w = w3q[kq * 16 + 4 * ob + nj];          // A role: output 4*ob + (lane % 4)
                        const float4 h = hq[kq * 4 + nj];                      // B role: node lane % 4
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, h.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, h.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, h.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, h.w, acc, 0, 0, 0);
                    }
                    float adv[4], pos[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float v = acc[i];
                        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);        // the four K-splits
                        adv[i] = v + W[kB3 + 4 * ob + i];
                    }
                    // positive_regret_policy over the node's 16 outputs (nets.py:93-101): this lane holds outputs 4*ob .. 4*ob+3 of node nj
                    float z = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; i++) { pos[i] = (((hand_bits >> (4 * ob + i)) & 1u) && adv[i] > 0.0f) ? adv[i] : 0.0f; z += pos[i]; }
                    z += __shfl_xor(z, 4); z += __shfl_xor(z, 8);              // the four output groups
                    const float den = z > 1e-8f ? z : 1e-8f;
                    if (ks == 0 && live) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int c = nib(sj.hand[p], k);
                            if (k < nl && (c >> 2) == ob) {
                                float pv = pos[0];
#pragma unroll
                                for (int i = 1; i < 4; i++) pv = (c & 3) == i ? pos[i] : pv;
                                ws.polcur[nj][k] = pv / den;
                            }
                        }
                    }
                }
                sd_sync();
                SD_STAMP(3);
                // expand / sample: one lane per node of the group
                if (lane < kG && g0 + lane < width) {
                    const int j = g0 + lane, idx = ws.idx[idx_at(d) + j];
                    float pk[4];
                    for (int k = 0; k < 4; k++) pk[k] = k < nl ? ws.polcur[lane][k] : 0.0f;
                    if (trav_ply) {
                        for (int k = 0; k < nl; k++) ws.idx[idx_at(d + 1) + j * nl + k] = (uint16_t)(idx * nl + k);
                        for (int k = 0; k < 4; k++) ws.pol_trav[moff + j][k] = pk[k];
                    } else {
                        float sum = pk[0];
                        for (int k = 1; k < nl; k++) sum += pk[k];
                        double u;
                        if (uniforms) u = uniforms[((size_t)tb * kPlies + d) * 24 + j];
                        else {
                            const philox_out x = philox4x32_10((uint32_t)j + 1024u * (uint32_t)d, b0 + (uint32_t)tb, iteration, 4u + (uint32_t)traverser, seed_lo, seed_hi);
                            u = u53(x.x0, x.x1);
                        }
                        int a;
                        if (sum == 0.0f) { a = (int)(u * (double)nl); a = a < nl - 1 ? a : nl - 1; }
                        else {
                            double c = 0.0, cdf[4];
                            for (int k = 0; k < nl; k++) { const double pq = (double)(pk[k] / sum); c = k ? c + pq : pq; cdf[k] = c; }
                            const double last = cdf[nl - 1];
                            a = 0;
                            for (int k = 0; k < nl; k++) if (cdf[k] / last <= u) a = k + 1;
                            a = a < nl - 1 ? a : nl - 1;
                        }
                        ws.idx[idx_at(d + 1) + j] = (uint16_t)(idx * nl + a);
                    }
                }
                sd_sync();
                SD_STAMP(4);
            }
            if (trav_ply) width *= nl;
        }
        // ---- leaves, then backward ---------------------------------------------------------------------------------------
        if (lane < width) { const int p0 = g_payoff[ws.idx[idx_at(8) + lane]]; ws.val[0][lane] = 0.5f * (float)(traverser == 0 ? p0 : -p0); }
        sd_sync();
        int cur = 0;
#pragma unroll 1
        for (int d = kPlies - 1; d >= 0; d--) {
            const int p = d & 1, nl = 4 - (d >> 1);
            const bool trav_ply = p == traverser;
            if (trav_ply) width /= nl;
            const int m = (d - traverser) >> 1;
            const int moff = m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 5 : 17;
            if (lane < width) {
                const int j = lane;
                if (!trav_ply) ws.val[cur ^ 1][j] = ws.val[cur][j];
                else {
                    const scopa_state s = g_states[level_offset(d) + ws.idx[idx_at(d) + j]];
                    float value = 0.0f, cfv[16];
                    for (int c = 0; c < 16; c++) cfv[c] = 0.0f;
                    uint32_t hand_bits = 0, table_bits = 0;
                    for (int k = 0; k < nl; k++) {
                        const float av = ws.val[cur][j * nl + k];
                        value += ws.pol_trav[moff + j][k] * av;
                        const int c = nib(s.hand[p], k);
                        hand_bits |= 1u << c;
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) if (cc == c) cfv[cc] = av;
                    }
                    for (int k = 0; k < s.nt; k++) table_bits |= 1u << nib(s.table, k);
                    ws.val[cur ^ 1][j] = value;
                    float mx = 0.0f, reg[16];
#pragma unroll
                    for (int c = 0; c < 16; c++) { reg[c] = cfv[c] - value; const float a = fabsf(reg[c]); mx = a > mx ? a : mx; }
                    const float den = mx + 1e-8f;
                    const int T[4] = {41, 10, 3, 1};
                    int jj = j, rank = T[m] - 1;
                    for (int qd = m - 1; qd >= 0; qd--) { const int radix = 4 - qd; rank += (jj % radix) * T[qd + 1]; jj /= radix; }
                    const long long row = (write_base + (long long)tb * 41 + rank) % capacity;
#pragma unroll
                    for (int c = 0; c < 16; c++) {
                        const float h = (float)((hand_bits >> c) & 1u);
                        mem_feat[row * 34 + c] = h;
                        mem_feat[row * 34 + 16 + c] = (float)((table_bits >> c) & 1u);
                        mem_mask[row * 16 + c] = h;
                        mem_regret[row * 16 + c] = mx > 0.0f ? reg[c] / den : reg[c];
                    }
                    mem_feat[row * 34 + 32] = 1.0f;
                    mem_feat[row * 34 + 33] = 0.0f;
                }
            }
            cur ^= 1;
            sd_sync();
        }
        if (lane == 0) root_values[tb] = ws.val[cur][0];
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

extern "C" int32_t scopa_sdcfr_traverse_fused(scopa_ctx *ctx, int32_t traverser, int32_t batch, const float *d_weights, float *d_mem_feat,
                                              float *d_mem_regret, float *d_mem_mask, int64_t capacity, int64_t write_base,
                                              float *d_root_values, const double *d_uniforms, uint32_t iteration, uint32_t b0) {
    if (!ctx || traverser < 0 || traverser > 1 || batch < 0 || (batch && (!d_weights || !d_mem_feat || !d_mem_regret || !d_mem_mask || !d_root_values)))
        return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_sdcfr_traverse_fused: no deal set");
    SC_REQUIRE(ctx, capacity >= 41 && (int64_t)batch * 41 <= capacity && write_base >= 0, SCOPA_EINVAL, "scopa_sdcfr_traverse_fused: memory ring too small for the batch");
    SC_REQUIRE(ctx, ((uintptr_t)d_weights & 15) == 0, SCOPA_EINVAL, "scopa_sdcfr_traverse_fused: weights must be 16-byte aligned");
    if (!batch) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    // 110 KB of weights + one SdWave per wavefront: as many wavefronts as fit (<= 16, the kernel's launch bound: 4 per SIMD)
    int waves = (int)(((size_t)ctx->lds_limit - 64 - (size_t)2 * kNetFloats * sizeof(float)) / sizeof(SdWave));
    waves = waves > 16 ? 16 : waves;
    SC_REQUIRE(ctx, waves >= 8, SCOPA_ELIMIT, "scopa_sdcfr_traverse_fused: LDS (the weight staging assumes >= 512 threads)");
    const int threads = waves * 64;
    const size_t lds = (size_t)2 * kNetFloats * sizeof(float) + (size_t)waves * sizeof(SdWave);
    SC_REQUIRE(ctx, lds + 64 <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_sdcfr_traverse_fused: LDS");
    SC_LDS_ATTR(ctx, scopa::kLdsSdcfr, k_sdcfr_traverse, ctx->lds_limit - 64);   // 64: the kernel's static LDS (s_next), beside the dynamic part
    const int passes = (batch + waves - 1) / waves;
    const int grid = passes < ctx->n_cus ? passes : ctx->n_cus;
    hipLaunchKernelGGL(k_sdcfr_traverse, dim3(grid), dim3(threads), lds, ctx->stream, ctx->d_states, ctx->d_payoff, d_weights, (int)traverser,
                       (int)batch, d_mem_feat, d_mem_regret, d_mem_mask, (long long)capacity, (long long)write_base, d_root_values, d_uniforms,
                       (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), iteration, b0);
    SC_HIP(ctx, hipGetLastError());
    ctx->sdcfr_visits += (uint64_t)batch * (traverser == 0 ? 105 : 82);
    return SCOPA_OK;
}
