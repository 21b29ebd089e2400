// scopa_tree.hip -- batched game step and on-device construction of the flat game tree.
//
// Reference behaviour replaced: MiniScopaEnv.step (src/envs/mini_scopa_game.py:140-167) applied through
// MiniScopaState.clone()+apply_action() at every edge of every traversal
// (src/algorithms/vanilla_cfr.py:79-85, mc_cfr.py:56-78, deep_cfr/deep_cfr.py:328-363).  The reference
// re-derives the same 2229 states on every traversal; here the tree of a deal is expanded once, level by
// level, by the same device step function that scopa_step_batch exposes, and kept in HBM.
#include <string.h>

#include <vector>

#include "scopa_ctx.h"

using namespace scopa;

// ---- kernel 1: batched step.  One lane per game; a state is one 16-byte (dwordx4) coalesced load/store. --------
#ifndef SCOPA_STEP_UNROLL
#define SCOPA_STEP_UNROLL 2
#endif
__global__ void __launch_bounds__(256) k_step_batch(scopa_state *__restrict__ states,
                                                     const uint8_t *__restrict__ actions, int64_t n) {
    // U games per lane and pass, all loads issued before any state is stepped: U times the bytes in flight per wavefront (the kernel moves 33 bytes
    // per game-step and waits on HBM latency, not on its arithmetic, while the tables are short)
    constexpr int U = SCOPA_STEP_UNROLL;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint4 *st = reinterpret_cast<uint4 *>(states);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += U * stride) {
        uint4 v[U];
        int act[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t j = i + u * stride;
            v[u] = j < n ? st[j] : make_uint4(0u, 0u, 0u, 0u);
            act[u] = j < n ? actions[j] & 15 : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t j = i + u * stride;
            if (j < n) {
                step_words(v[u].x, v[u].y, v[u].z, v[u].w, act[u]);
                st[j] = v[u];
            }
        }
    }
}

// ---- kernel 2: tree build, one workgroup per deal ------------------------------------------------------------------
// DFS index (reference visiting order) of the BFS node (ply d, index j): sum over the path of 1 + a_k * subtree(k+1).
__device__ __forceinline__ int dfs_index(int d, int j) {
    int dfs = 0;
    for (int k = d - 1; k >= 0; k--) {
        const int n = nlegal_at(k);
        const int a = j % n;
        j /= n;
        dfs += 1 + a * subtree_size(k + 1);
    }
    return dfs;
}

__global__ void __launch_bounds__(1024) k_tree_build(const uint8_t *__restrict__ perm16, scopa_state *__restrict__ states,
                                                      uint16_t *__restrict__ infoset_of, int8_t *__restrict__ payoff,
                                                      uint64_t *__restrict__ key_of_infoset, int32_t *__restrict__ meta) {
    __shared__ uint64_t s_key[576];            // keys of the ply being processed
    __shared__ int s_dfs[576];                 // their DFS indices
    __shared__ int s_first[kDecision];         // per decision node (BFS): DFS index of the first node sharing its key
    __shared__ int s_scan[kNodes + 1];         // indexed by DFS index: 1 where an infoset is first visited -> prefix sum
    const int tid = threadIdx.x, nt = blockDim.x;
    {   // one workgroup per deal (multi-deal mode launches a grid of them; a single context uses deal 0)
        const size_t deal = blockIdx.x;
        perm16 += deal * 16; states += deal * kNodes; infoset_of += deal * kDecision; payoff += deal * kTerminal;
        key_of_infoset += deal * kDecision; meta += deal * 8;
    }

    for (int i = tid; i <= kNodes; i += nt) s_scan[i] = 0;
    if (tid == 0) {
        uint8_t p[16];
        for (int i = 0; i < 16; i++) p[i] = perm16[i];
        scopa_state root;
        state_init(root, p);
        states[0] = root;
    }
    __syncthreads();
    // level-synchronous expansion: node j of ply d+1 = step(node j / n of ply d, its (j % n)-th legal action)
    for (int d = 0; d < kPlies; d++) {
        const int n = nlegal_at(d), w1 = level_width(d + 1);
        for (int j = tid; j < w1; j += nt) {
            scopa_state s = states[level_offset(d) + j / n];
            const int action = nib(s.hand[s.step & 1], j % n);  // legal actions are the hand, in hand order
            step(s, action);
            states[level_offset(d + 1) + j] = s;
        }
        __syncthreads();
    }
    // terminal payoffs (player 0's reward x2; zero-sum)
    for (int j = tid; j < kTerminal; j += nt) {
        int r0, r1;
        rewards_x2(states[level_offset(8) + j], r0, r1);
        payoff[j] = (int8_t)r0;
    }
    // infoset identity: nodes of one ply with equal (player, ordered hand, ordered table) keys.  Keys of different
    // plies never collide (the hand size differs), so first-occurrence search runs per ply.
    for (int d = 0; d < kPlies; d++) {
        const int w = level_width(d), off = level_offset(d);
        for (int j = tid; j < w; j += nt) { s_key[j] = infoset_key(states[off + j], d & 1); s_dfs[j] = dfs_index(d, j); }
        __syncthreads();
        for (int j = tid; j < w; j += nt) {
            const uint64_t k = s_key[j];
            int first = s_dfs[j];
            for (int m = 0; m < w; m++) {  // DFS order within a ply is index order: the first match is the minimum
                if (s_key[m] == k) { first = s_dfs[m]; break; }
            }
            s_first[off + j] = first;
            if (first == s_dfs[j]) s_scan[first + 1] = 1;  // this node is where the reference inserts the key
        }
        __syncthreads();
    }
    // dense ids in DFS first-visit order (= dict insertion order, vanilla_cfr.py:51-54): inclusive scan of the flags.
    // 4 elements per lane, wavefront shuffle scan, 16 wavefront totals through LDS.
    {
        __shared__ int s_wsum[16];
        const int lane = tid & 63, wave = tid >> 6, base = tid * 4;
        int v[4], sum = 0;
        for (int q = 0; q < 4; q++) { v[q] = base + q <= kNodes ? s_scan[base + q] : 0; sum += v[q]; }
        int x = sum;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int w = 0; w < 16; w++) { const int t = s_wsum[w]; s_wsum[w] = acc; acc += t; }
            meta[0] = acc;
        }
        __syncthreads();
        int run = x - sum + s_wsum[wave];
        for (int q = 0; q < 4; q++) { run += v[q]; if (base + q <= kNodes) s_scan[base + q] = run; }
    }
    __syncthreads();
    for (int d = 0; d < kPlies; d++) {
        const int w = level_width(d), off = level_offset(d);
        for (int j = tid; j < w; j += nt) {
            const int first = s_first[off + j];
            const int id = s_scan[first];  // flags strictly before `first`
            infoset_of[off + j] = (uint16_t)id;
            if (first == dfs_index(d, j)) key_of_infoset[id] = infoset_key(states[off + j], d & 1);
        }
    }
}

__global__ void k_tables_reset(double *regret, double *strat, double *local, const uint64_t *key_of_infoset,
                               const int32_t *meta) {
    // InfoNode.__post_init__ (vanilla_cfr.py:15-21): zeros, local_strategy = ones(n)/n.  blockIdx.y = deal.
    {
        const size_t deal = blockIdx.y;
        regret += deal * kDecision * 4; strat += deal * kDecision * 4; local += deal * kDecision * 4;
        key_of_infoset += deal * kDecision; meta += deal * 8;
    }
    const int I = meta[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kDecision * 4; i += gridDim.x * blockDim.x) {
        const int row = i >> 2, col = i & 3;
        regret[i] = 0.0;
        strat[i] = 0.0;
        double l = 0.0;
        if (row < I) {
            const int n = (int)((key_of_infoset[row] >> 1) & 7);  // legal count = hand size
            if (col < n) l = 1.0 / (double)n;
        }
        local[i] = l;
    }
}

extern "C" {

int32_t scopa_step_batch(scopa_ctx *ctx, scopa_state *d_states, const uint8_t *d_actions, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!d_states || !d_actions))) return SCOPA_EINVAL;
    if (n == 0) return SCOPA_OK;
    SC_REQUIRE(ctx, ((uintptr_t)d_states & 15) == 0, SCOPA_EINVAL, "scopa_step_batch: states must be 16-byte aligned");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int64_t blocks = (n + 255) / 256;
#ifndef SCOPA_STEP_BLOCKS_PER_CU
#define SCOPA_STEP_BLOCKS_PER_CU 128
#endif
    // grid-stride beyond that many blocks per CU.  (8 -- one resident round of workgroups -- left the launch's tail to its slowest wavefronts and the whole
    // game at 0.64 of the HBM peak; 64-128 give 0.70-0.71, a flat grid 0.69: gpurun_out/step_grid*.log, round 4.)
    const int64_t cap = (int64_t)ctx->n_cus * SCOPA_STEP_BLOCKS_PER_CU;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_step_batch, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_states, d_actions, n);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_step_batch_host(scopa_ctx *ctx, scopa_state *h_states, const uint8_t *h_actions, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!h_states || !h_actions))) return SCOPA_EINVAL;
    if (n == 0) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    scopa_state *d_s = nullptr;
    uint8_t *d_a = nullptr;
    SC_HIP(ctx, hipMalloc(&d_s, (size_t)n * sizeof(scopa_state)));
    hipError_t e = hipMalloc(&d_a, (size_t)n);
    if (e != hipSuccess) { (void)hipFree(d_s); return fail(ctx, SCOPA_EHIP, "hipMalloc(actions)", e); }
    int32_t rc = SCOPA_OK;
    if ((e = hipMemcpyAsync(d_s, h_states, (size_t)n * sizeof(scopa_state), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(d_a, h_actions, (size_t)n, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(H2D)", e);
    if (rc == SCOPA_OK) rc = scopa_step_batch(ctx, d_s, d_a, n);
    if (rc == SCOPA_OK && (e = hipMemcpyAsync(h_states, d_s, (size_t)n * sizeof(scopa_state), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(D2H)", e);
    e = hipStreamSynchronize(ctx->stream);
    if (rc == SCOPA_OK && e != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipStreamSynchronize", e);
    (void)hipFree(d_s);
    (void)hipFree(d_a);
    return rc;
}

int32_t scopa_tables_reset(scopa_ctx *ctx) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_tables_reset: no deal set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_tables_reset, dim3(8), dim3(1024), 0, ctx->stream, ctx->d_regret, ctx->d_strat, ctx->d_local,
                       ctx->d_key, ctx->d_meta);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipMemsetAsync(ctx->d_visit, 0, sizeof(uint32_t) * kDecision, ctx->stream));
    SC_HIP(ctx, hipMemsetAsync(ctx->d_meta + 1, 0, sizeof(int32_t), ctx->stream));
    SC_HIP(ctx, hipMemsetAsync(ctx->d_delta, 0, (size_t)(ctx->d_delta == ctx->d_delta_own ? kDecision : ctx->n_infosets) * 5 * sizeof(double), ctx->stream));
    ctx->iteration = 0;
    ctx->sigcdf_valid = false;
    ctx->mccfr_all_seen = false;
    ctx->mccfr_seen_wait = 0;
    return SCOPA_OK;
}

int32_t scopa_set_deal(scopa_ctx *ctx, const uint8_t perm16[16]) {
    if (!ctx || !perm16) return SCOPA_EINVAL;
    uint32_t seen = 0;
    for (int i = 0; i < 16; i++) { if (perm16[i] > 15) return fail(ctx, SCOPA_EINVAL, "scopa_set_deal: card id > 15"); seen |= 1u << perm16[i]; }
    SC_REQUIRE(ctx, seen == 0xFFFFu, SCOPA_EINVAL, "scopa_set_deal: not a permutation of the 16 cards");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    { const int32_t rc = ensure_scratch(ctx, 4096); if (rc != SCOPA_OK) return rc; }
    memcpy(ctx->perm, perm16, 16);
    SC_HIP(ctx, hipMemcpyAsync(ctx->d_scratch, ctx->perm, 16, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_tree_build, dim3(1), dim3(1024), 0, ctx->stream, (const uint8_t *)ctx->d_scratch, ctx->d_states,
                       ctx->d_infoset, ctx->d_payoff, ctx->d_key, ctx->d_meta);
    SC_HIP(ctx, hipGetLastError());
    int32_t n_inf = 0;
    SC_HIP(ctx, hipMemcpyAsync(&n_inf, ctx->d_meta, sizeof n_inf, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_REQUIRE(ctx, n_inf > 0 && n_inf <= kDecision, SCOPA_EHIP, "scopa_set_deal: tree build produced a bad infoset count");
    ctx->n_infosets = n_inf;
    ctx->has_deal = true;
    ctx->d_delta = ctx->d_delta_own;  // a new deal drops any caller-bound delta buffer
    ctx->sched_valid = false;         // ... and the exact-CFR schedule of the previous deal
    ctx->sdnode_valid = false;        // ... and the SDCFR traversal's per-node feature bits
    ctx->eval_thr_valid = false;      // ... and an evaluation policy's thresholds (indexed by THIS deal's infoset ids)
    scopa::mccfr_graphs_clear(ctx);   // ... and captured iteration graphs (their launches carry the old deal's sizes)
    return scopa_tables_reset(ctx);
}

int32_t scopa_tree_counts(scopa_ctx *ctx, int32_t *n_nodes, int32_t *n_decision, int32_t *n_infosets) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_tree_counts: no deal set");
    if (n_nodes) *n_nodes = kNodes;
    if (n_decision) *n_decision = kDecision;
    if (n_infosets) *n_infosets = ctx->n_infosets;
    return SCOPA_OK;
}

int32_t scopa_tree_export(scopa_ctx *ctx, scopa_state *h_states, int32_t *h_infoset, int8_t *h_r2, uint64_t *h_infoset_key,
                          int8_t *h_infoset_nlegal, int8_t *h_infoset_legal) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_tree_export: no deal set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<scopa_state> st(kNodes);
    std::vector<uint16_t> inf(kDecision);
    std::vector<int8_t> pay(kTerminal);
    std::vector<uint64_t> keys((size_t)ctx->n_infosets);
    SC_HIP(ctx, hipMemcpyAsync(st.data(), ctx->d_states, sizeof(scopa_state) * kNodes, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(inf.data(), ctx->d_infoset, sizeof(uint16_t) * kDecision, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(pay.data(), ctx->d_payoff, kTerminal, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(keys.data(), ctx->d_key, sizeof(uint64_t) * keys.size(), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // BFS (device layout) -> reference DFS order
    for (int d = 0; d <= kPlies; d++) {
        for (int j = 0; j < level_width(d); j++) {
            int dfs = 0, jj = j;
            for (int k = d - 1; k >= 0; k--) { const int n = nlegal_at(k); dfs += 1 + (jj % n) * subtree_size(k + 1); jj /= n; }
            const int bfs = level_offset(d) + j;
            if (h_states) h_states[dfs] = st[(size_t)bfs];
            if (h_infoset) h_infoset[dfs] = d < kPlies ? (int32_t)inf[(size_t)bfs] : -1;
            if (h_r2) {
                const int r0 = d < kPlies ? 0 : pay[(size_t)j];
                h_r2[dfs * 2] = (int8_t)r0;
                h_r2[dfs * 2 + 1] = (int8_t)-r0;
            }
        }
    }
    for (int i = 0; i < ctx->n_infosets; i++) {
        const uint64_t k = keys[(size_t)i];
        if (h_infoset_key) h_infoset_key[i] = k;
        const int n = (int)((k >> 1) & 7);
        if (h_infoset_nlegal) h_infoset_nlegal[i] = (int8_t)n;
        if (h_infoset_legal)
            for (int c = 0; c < 4; c++) h_infoset_legal[i * 4 + c] = (int8_t)(c < n ? (int)((k >> (4 + 4 * c)) & 15) : -1);
    }
    return SCOPA_OK;
}

}  // extern "C"
