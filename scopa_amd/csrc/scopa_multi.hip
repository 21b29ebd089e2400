// scopa_multi.hip -- many independent deals at once: "replicas" parallelism for the exact-semantics solvers.
//
// The reference always solves ONE deal (seed 42, src/envs/mini_scopa_game.py:25-28,131-132) although its API takes a
// seed (MiniScopaEnv(seed=...), :120-132).  Its vanilla CFR is inherently sequential per solve (DESIGN.md §4,
// SURVEY H1), so the way to use a 256-CU GPU for it is one solve per workgroup.  A scopa_multi holds n deals
// resident in HBM -- per deal 35.7 KB of packed states, the node->infoset map, leaf payoffs, infoset keys and three
// [1653][4] float64 tables (~200 KB) -- and runs, one WORKGROUP PER DEAL:
//   k_deal_py_seed   the deal itself: CPython's random.seed(int) + random.shuffle on a per-lane MT19937
//   k_tree_build     level-synchronous expansion with the device step function
//   k_cfr_exact      the reference's sequential vanilla CFR, bit-exact per deal
//   k_cfr_sync       synchronous CFR
//   k_exploitability best responses and policy value
#include <vector>

#include "scopa_kernels.h"

using namespace scopa;

struct scopa_multi {
    scopa_ctx *ctx = nullptr;
    int n = 0;
    bool built = false;
    int max_infosets = 0;
    uint8_t *d_perm = nullptr;       // [n][16]
    scopa_state *d_states = nullptr; // [n][2229]
    uint16_t *d_infoset = nullptr;   // [n][1653]
    int8_t *d_payoff = nullptr;      // [n][576]
    uint64_t *d_key = nullptr;       // [n][1653]
    int32_t *d_meta = nullptr;       // [n][8]
    double *d_regret = nullptr, *d_strat = nullptr, *d_local = nullptr;  // [n][1653][4]
    uint32_t *d_visit = nullptr;     // [n][1653]
    unsigned long long *d_counters = nullptr;  // [n][8]
    double *d_out = nullptr;         // [n][4] exploitability outputs
    int64_t *d_seeds = nullptr;      // [n]
    uint16_t *d_infoset_T = nullptr; // [1653][n]  node-major copies for the lane-per-deal kernel (coalesced across deals)
    int8_t *d_payoff_T = nullptr;    // [576][n]
    double *d_rows = nullptr;        // [n][1653][8]  lane-per-deal kernel's table image: one 64-byte row per infoset = regret[4] strategy[4]
    bool rows_current = false;       // the tables live in d_rows (true) or in d_regret/d_strat/d_local (false)
    uint32_t mccfr_iteration = 0;
};

// ---- CPython random.seed(int) + random.shuffle(16 cards), one lane per deal (MiniDeck.__init__, mini_scopa_game.py:25-28) --------
// MT19937 init_by_array on |seed|'s 32-bit words, then Random._randbelow_with_getrandbits for i = 15..1.  The 624-word state
// lives in per-lane scratch; outputs are produced by twisting word k on demand, in order, which is the reference generator.
__global__ void __launch_bounds__(64) k_deal_py_seed(const int64_t *__restrict__ seeds, uint8_t *__restrict__ perms, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t mt[624];
    const int64_t sd = seeds[i];
    const uint64_t a = sd < 0 ? (uint64_t)0 - (uint64_t)sd : (uint64_t)sd;
    const uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
    const int klen = key[1] ? 2 : 1;
    mt[0] = 19650218u;
    for (int k = 1; k < 624; k++) mt[k] = 1812433253u * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (uint32_t)k;
    {
        int p = 1, j = 0;
        for (int k = 624; k > 0; k--) {
            mt[p] = (mt[p] ^ ((mt[p - 1] ^ (mt[p - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++p >= 624) { mt[0] = mt[623]; p = 1; }
            if (++j >= klen) j = 0;
        }
        for (int k = 623; k > 0; k--) {
            mt[p] = (mt[p] ^ ((mt[p - 1] ^ (mt[p - 1] >> 30)) * 1566083941u)) - (uint32_t)p;
            if (++p >= 624) { mt[0] = mt[623]; p = 1; }
        }
        mt[0] = 0x80000000u;
    }
    int at = 0;  // next word to twist + temper
    uint8_t perm[16];
    for (int c = 0; c < 16; c++) perm[c] = (uint8_t)c;
    for (int hi = 15; hi >= 1; hi--) {
        const uint32_t nn = (uint32_t)hi + 1u;
        const int bits = 32 - __clz(nn);
        uint32_t r;
        do {
            const int k = at % 624;
            const uint32_t y0 = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y0 >> 1) ^ ((y0 & 1u) ? 0x9908b0dfu : 0u);
            uint32_t y = mt[k];
            at++;
            y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
            r = y >> (32 - bits);
        } while (r >= nn);
        const uint8_t t = perm[hi]; perm[hi] = perm[r]; perm[r] = t;
    }
    for (int c = 0; c < 16; c++) perms[(size_t)i * 16 + c] = perm[c];
}

// ---- lane-per-deal exact CFR: the HBM-bound way to run many sequential solves ----------------------------------------
// The reference's vanilla CFR is one sequential DFS per solve (vanilla_cfr.py:56-99), but the tree SHAPE is the same for every
// deal (legal counts 4,4,3,3,2,2,1,1), so 64 deals can share one instruction stream with no divergence: lane = deal, the DFS
// position (ply, node index) is wave-uniform, only the infoset id at that position -- and therefore which table row is
// touched -- differs per lane.  Node -> infoset maps and payoffs are stored node-major ([node][deal]) so those reads are
// coalesced across lanes; the tables stay in HBM and are gathered row by row.
//
// Table image: ONE 64-BYTE ROW PER INFOSET = regret_sum[4] | strategy_sum[4] (one DRAM burst), and no local_strategy at all:
// the reference refreshes local_strategy = get_strategy() at the end of EVERY visit (vanilla_cfr.py:97) and regret_sum only
// changes inside visits, so between visits local_strategy == regret_matching(regret_sum) holds identically (InfoNode starts
// with zeros / uniform, which is the same statement).  The kernel therefore reads the regret row on entry, derives the
// strategy, and only the traverser's nodes write anything (regret | strategy, one full 64-byte store).  Per decision visit
// that is one 64-byte fetch and half a 64-byte store instead of the 3 reads + 1-3 writes of 32 bytes of the literal form.
// k_rows_pack refuses tables for which the invariant does not hold (tables written by another solver).
// Arithmetic and visit order per deal are exactly k_cfr_exact's, hence bit-identical to the reference.
namespace {
struct LaneCtx {
    const uint16_t *inf_T;   // [1653][n]
    const int8_t *pay_T;     // [576][n]
    double *rows;            // this deal's [1653][8] table image
    size_t n;                // deals (stride of the node-major maps)
    size_t deal;
    unsigned long long dvis, tvis;
};

template <int N>
__device__ __forceinline__ void regret_match(const double (&R)[4], double (&out)[4]) {  // InfoNode.get_strategy, vanilla_cfr.py:23-30
    double pos[4];
    for (int i = 0; i < N; i++) pos[i] = R[i] > 0.0 ? R[i] : 0.0;
    double sm = pos[0];
    for (int i = 1; i < N; i++) sm += pos[i];
    for (int i = 0; i < 4; i++) out[i] = i < N ? (sm > 0.0 ? pos[i] / sm : 1.0 / (double)N) : 0.0;
}

template <int D, int TRAV>
__device__ __forceinline__ double lane_rec(LaneCtx &c, int idx, double r0, double r1) {
    if constexpr (D == kPlies) {
        c.tvis++;
        const int p0 = c.pay_T[(size_t)idx * c.n + c.deal];
        return 0.5 * (double)(TRAV == 0 ? p0 : -p0);
    } else {
        constexpr int n = 4 - (D >> 1), p = D & 1;
        c.dvis++;
        const int I = c.inf_T[(size_t)(level_offset(D) + idx) * c.n + c.deal];
        double2 *row = reinterpret_cast<double2 *>(c.rows + (size_t)I * 8);
        double ls[4], au[4];
        {   // this infoset's regret row cannot change before this visit ends: its other nodes are on the same ply
            const double2 a = row[0], b = row[1];
            const double Rr[4] = {a.x, a.y, b.x, b.y};
            regret_match<n>(Rr, ls);
        }
#pragma unroll 1
        for (int i = 0; i < n; i++)
            au[i] = lane_rec<D + 1, TRAV>(c, idx * n + i, p == 0 ? r0 * ls[i] : r0, p == 1 ? r1 * ls[i] : r1);
        double v = ls[0] * au[0];
        for (int i = 1; i < n; i++) v += ls[i] * au[i];
        if constexpr (p == TRAV) {
            const double reach = TRAV == 0 ? r0 : r1, opp = TRAV == 0 ? r1 : r0;
            const double2 a = row[0], b = row[1], e = row[2], f = row[3];   // the row read on entry, usually still in L2
            double Rr[4] = {a.x, a.y, b.x, b.y}, Sr[4] = {e.x, e.y, f.x, f.y};
            for (int i = 0; i < n; i++) { Rr[i] += opp * (au[i] - v); Sr[i] += reach * ls[i]; }
            row[0] = make_double2(Rr[0], Rr[1]); row[1] = make_double2(Rr[2], Rr[3]);
            row[2] = make_double2(Sr[0], Sr[1]); row[3] = make_double2(Sr[2], Sr[3]);
        }
        return v;
    }
}
}  // namespace

__global__ void __launch_bounds__(256) k_transpose_maps(const uint16_t *__restrict__ inf, const int8_t *__restrict__ pay,
                                                        uint16_t *__restrict__ inf_T, int8_t *__restrict__ pay_T, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over n * 1653
    if (i < n * kDecision) { const long long d = i / kDecision, k = i - d * kDecision; inf_T[k * n + d] = inf[i]; }
    if (i < n * kTerminal) { const long long d = i / kTerminal, k = i - d * kTerminal; pay_T[k * n + d] = pay[i]; }
}

// tables <-> row image; one thread per infoset row.  pack: checks local_strategy == regret_matching(regret_sum) bit for bit and
// counts violations in *bad; unpack: writes regret_sum, strategy_sum and the implied local_strategy (zero rows past the deal's
// infoset count, as scopa_tables_reset leaves them).
__global__ void __launch_bounds__(256) k_rows_pack(double *__restrict__ R, double *__restrict__ S, double *__restrict__ L, double *__restrict__ rows,
                                                   const uint64_t *__restrict__ key, const int32_t *__restrict__ meta, long long n_rows, int unpack,
                                                   unsigned int *__restrict__ bad) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const long long deal = r / kDecision;
    const int in_deal = (int)(r - deal * kDecision);
    const bool live = in_deal < meta[deal * 8];
    const int n = live ? (int)((key[r] >> 1) & 7) : 0;  // legal count = hand size
    double2 *line = reinterpret_cast<double2 *>(rows + r * 8);
    double Rr[4], ls[4] = {0.0, 0.0, 0.0, 0.0};
    if (!unpack) {
        const double2 a = reinterpret_cast<const double2 *>(R + r * 4)[0], b = reinterpret_cast<const double2 *>(R + r * 4)[1];
        Rr[0] = a.x; Rr[1] = a.y; Rr[2] = b.x; Rr[3] = b.y;
        line[0] = a; line[1] = b;
        line[2] = reinterpret_cast<const double2 *>(S + r * 4)[0]; line[3] = reinterpret_cast<const double2 *>(S + r * 4)[1];
    } else {
        const double2 a = line[0], b = line[1];
        Rr[0] = a.x; Rr[1] = a.y; Rr[2] = b.x; Rr[3] = b.y;
        reinterpret_cast<double2 *>(R + r * 4)[0] = a; reinterpret_cast<double2 *>(R + r * 4)[1] = b;
        reinterpret_cast<double2 *>(S + r * 4)[0] = line[2]; reinterpret_cast<double2 *>(S + r * 4)[1] = line[3];
    }
    if (n == 4) regret_match<4>(Rr, ls); else if (n == 3) regret_match<3>(Rr, ls); else if (n == 2) regret_match<2>(Rr, ls); else if (n == 1) regret_match<1>(Rr, ls);
    if (!unpack) {
        bool same = true;
        for (int i = 0; i < 4; i++) same = same && __double_as_longlong(L[r * 4 + i]) == __double_as_longlong(ls[i]);
        if (!same) atomicAdd(bad, 1u);
    } else {
        reinterpret_cast<double2 *>(L + r * 4)[0] = make_double2(ls[0], ls[1]); reinterpret_cast<double2 *>(L + r * 4)[1] = make_double2(ls[2], ls[3]);
    }
}

__global__ void __launch_bounds__(64)
k_cfr_exact_lanes(const uint16_t *__restrict__ inf_T, const int8_t *__restrict__ pay_T, double *__restrict__ g_rows, const int32_t *__restrict__ g_meta,
                  uint32_t *__restrict__ g_visit, unsigned long long *__restrict__ g_counters, long long n, int n_iters) {
    const long long deal = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (deal >= n) return;  // the tail wavefront simply has fewer lanes
    LaneCtx c;
    c.inf_T = inf_T; c.pay_T = pay_T; c.n = (size_t)n; c.deal = (size_t)deal;
    c.rows = g_rows + (size_t)deal * kDecision * 8;
    c.dvis = 0; c.tvis = 0;
#pragma unroll 1
    for (int it = 0; it < n_iters; it++) {
        lane_rec<0, 0>(c, 0, 1.0, 1.0);
        lane_rec<0, 1>(c, 0, 1.0, 1.0);
    }
    g_counters[deal * 8] += c.dvis;
    g_counters[deal * 8 + 1] += c.tvis;
    if (n_iters > 0) {  // a full traversal visits every infoset, in id order (ids ARE the DFS first-visit order)
        const int I = g_meta[deal * 8];
        for (int r = 0; r < I; r++) if (g_visit[deal * kDecision + r] == 0u) g_visit[deal * kDecision + r] = (uint32_t)r + 1u;
    }
}

namespace {
int32_t rows_convert(scopa_multi *m, bool to_rows) {
    scopa_ctx *ctx = m->ctx;
    if (m->rows_current == to_rows) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->d_rows) {
        if (hipMalloc(&m->d_rows, (size_t)m->n * kDecision * 64 + 256) != hipSuccess)
            return fail(ctx, SCOPA_ENOMEM, "scopa_multi: no device memory for the row-per-infoset table image");
    }
    unsigned int *d_bad = reinterpret_cast<unsigned int *>(m->d_rows + (size_t)m->n * kDecision * 8);
    const long long n_rows = (long long)m->n * kDecision;
    if (to_rows) SC_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_rows_pack, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, ctx->stream, m->d_regret, m->d_strat, m->d_local, m->d_rows,
                       m->d_key, m->d_meta, n_rows, to_rows ? 0 : 1, d_bad);
    SC_HIP(ctx, hipGetLastError());
    if (to_rows) {
        unsigned int bad = 0;
        SC_HIP(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SC_REQUIRE(ctx, bad == 0, SCOPA_ESTATE,
                   "scopa_multi_cfr_exact_iterate_lanes: local_strategy is not regret-matching of regret_sum (tables written by another solver); use scopa_multi_cfr_exact_iterate");
    }
    m->rows_current = to_rows;
    return SCOPA_OK;
}
}  // namespace

extern "C" {

int32_t scopa_multi_destroy(scopa_multi *m);

int32_t scopa_multi_create(scopa_ctx *ctx, int32_t n_deals, scopa_multi **out) {
    if (!ctx || !out || n_deals <= 0 || n_deals > (1 << 20)) return SCOPA_EINVAL;
    *out = nullptr;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    scopa_multi *m = new (std::nothrow) scopa_multi();
    if (!m) return SCOPA_ENOMEM;
    m->ctx = ctx;
    m->n = n_deals;
    const size_t n = (size_t)n_deals;
    bool ok = hipMalloc(&m->d_perm, n * 16) == hipSuccess && hipMalloc(&m->d_states, n * kNodes * sizeof(scopa_state)) == hipSuccess &&
              hipMalloc(&m->d_infoset, n * kDecision * 2) == hipSuccess && hipMalloc(&m->d_payoff, n * kTerminal) == hipSuccess &&
              hipMalloc(&m->d_key, n * kDecision * 8) == hipSuccess && hipMalloc(&m->d_meta, n * 8 * 4) == hipSuccess &&
              hipMalloc(&m->d_regret, n * kDecision * 32) == hipSuccess && hipMalloc(&m->d_strat, n * kDecision * 32) == hipSuccess &&
              hipMalloc(&m->d_local, n * kDecision * 32) == hipSuccess && hipMalloc(&m->d_visit, n * kDecision * 4) == hipSuccess &&
              hipMalloc(&m->d_counters, n * 8 * 8) == hipSuccess && hipMalloc(&m->d_out, n * 4 * 8) == hipSuccess &&
              hipMalloc(&m->d_seeds, n * 8) == hipSuccess && hipMalloc(&m->d_infoset_T, n * kDecision * 2) == hipSuccess &&
              hipMalloc(&m->d_payoff_T, n * kTerminal) == hipSuccess;
    if (ok) ok = hipMemsetAsync(m->d_counters, 0, n * 64, ctx->stream) == hipSuccess && hipMemsetAsync(m->d_meta, 0, n * 32, ctx->stream) == hipSuccess;
    if (!ok) { scopa_multi_destroy(m); return fail(ctx, SCOPA_ENOMEM, "scopa_multi_create: device allocation failed"); }
    *out = m;
    return SCOPA_OK;
}

int32_t scopa_multi_destroy(scopa_multi *m) {
    if (!m) return SCOPA_EINVAL;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    void *bufs[] = {m->d_perm, m->d_states, m->d_infoset, m->d_payoff, m->d_key, m->d_meta, m->d_regret, m->d_strat, m->d_local,
                    m->d_visit, m->d_counters, m->d_out, m->d_seeds, m->d_infoset_T, m->d_payoff_T, m->d_rows};
    for (void *b : bufs) if (b) (void)hipFree(b);
    delete m;
    return SCOPA_OK;
}

int32_t scopa_multi_deal_py_seeds(scopa_multi *m, const int64_t *h_seeds) {
    if (!m || !h_seeds) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    SC_HIP(ctx, hipMemcpyAsync(m->d_seeds, h_seeds, (size_t)m->n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_deal_py_seed, dim3((m->n + 63) / 64), dim3(64), 0, ctx->stream, m->d_seeds, m->d_perm, m->n);
    SC_HIP(ctx, hipGetLastError());
    m->built = false;
    return SCOPA_OK;
}

int32_t scopa_multi_set_perms(scopa_multi *m, const uint8_t *h_perms) {
    if (!m || !h_perms) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    for (int d = 0; d < m->n; d++) {
        uint32_t seen = 0;
        for (int c = 0; c < 16; c++) { if (h_perms[d * 16 + c] > 15) return fail(ctx, SCOPA_EINVAL, "scopa_multi_set_perms: card id > 15"); seen |= 1u << h_perms[d * 16 + c]; }
        SC_REQUIRE(ctx, seen == 0xFFFFu, SCOPA_EINVAL, "scopa_multi_set_perms: a row is not a permutation of the 16 cards");
    }
    SC_HIP(ctx, hipSetDevice(ctx->device));
    SC_HIP(ctx, hipMemcpyAsync(m->d_perm, h_perms, (size_t)m->n * 16, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->built = false;
    return SCOPA_OK;
}

int32_t scopa_multi_perms_get(scopa_multi *m, uint8_t *h_perms) {
    if (!m || !h_perms) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_HIP(ctx, hipMemcpyAsync(h_perms, m->d_perm, (size_t)m->n * 16, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_multi_build(scopa_multi *m, int32_t *h_n_infosets) {
    if (!m) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_tree_build, dim3(m->n), dim3(1024), 0, ctx->stream, m->d_perm, m->d_states, m->d_infoset, m->d_payoff, m->d_key, m->d_meta);
    SC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_tables_reset, dim3(8, m->n), dim3(1024), 0, ctx->stream, m->d_regret, m->d_strat, m->d_local, m->d_key, m->d_meta);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipMemsetAsync(m->d_visit, 0, (size_t)m->n * kDecision * 4, ctx->stream));
    {
        const long long total = (long long)m->n * kDecision;
        hipLaunchKernelGGL(k_transpose_maps, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, m->d_infoset, m->d_payoff,
                           m->d_infoset_T, m->d_payoff_T, (long long)m->n);
        SC_HIP(ctx, hipGetLastError());
    }
    std::vector<int32_t> meta((size_t)m->n * 8);
    SC_HIP(ctx, hipMemcpyAsync(meta.data(), m->d_meta, meta.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->max_infosets = 0;
    for (int d = 0; d < m->n; d++) {
        const int I = meta[(size_t)d * 8];
        SC_REQUIRE(ctx, I > 0 && I <= kDecision, SCOPA_EHIP, "scopa_multi_build: a tree build produced a bad infoset count");
        if (I > m->max_infosets) m->max_infosets = I;
        if (h_n_infosets) h_n_infosets[d] = I;
    }
    m->built = true;
    m->rows_current = false;
    m->mccfr_iteration = 0;
    return SCOPA_OK;
}

int32_t scopa_multi_cfr_exact_iterate_lanes(scopa_multi *m, int32_t n_iters);

int32_t scopa_multi_cfr_exact_iterate(scopa_multi *m, int32_t n_iters) {
    if (!m || n_iters < 0 || n_iters > (1 << 20)) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_REQUIRE(ctx, m->built, SCOPA_ESTATE, "scopa_multi_cfr_exact_iterate: call scopa_multi_build first");
    if (!n_iters) return SCOPA_OK;
    // Both kernels produce the same bits.  One workgroup per deal costs ~0.86 ms per iteration and round of ~512 resident
    // workgroups; one lane per deal is bound by memory latency below a few thousand deals and by HBM beyond (4e10 visits/s):
    // from 8192 deals on the lane form is taken, unless its precondition or its extra row image is not available.
    if (m->n >= 8192) {
        const int32_t rc = scopa_multi_cfr_exact_iterate_lanes(m, n_iters);
        if (rc == SCOPA_OK) return SCOPA_OK;
        if (rc != SCOPA_ESTATE && rc != SCOPA_ENOMEM) return rc;
    }
    if (int32_t rc = rows_convert(m, false)) return rc;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds = (size_t)m->max_infosets * 4 * 8 * 3;
    const size_t static_lds = 1656 * 2 + kTerminal + sizeof(uint32_t) * kDecision + 256;   // k_cfr_exact: maps + first-visit flags
    const int use_lds = lds + static_lds <= (size_t)ctx->lds_limit ? 1 : 0;
    SC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_cfr_exact), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    ctx->lds_limit - (int)static_lds));
    hipLaunchKernelGGL(k_cfr_exact, dim3(m->n), dim3(256), use_lds ? lds : 0, ctx->stream, m->d_infoset, m->d_payoff, m->d_regret, m->d_strat,
                       m->d_local, 0 /* multi-deal */, n_iters * 2, 0, (double *)nullptr, m->d_counters, use_lds, m->d_visit, m->d_meta, 0, 0, 1.0, 1.0);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_multi_cfr_exact_iterate_lanes(scopa_multi *m, int32_t n_iters) {
    if (!m || n_iters < 0 || n_iters > (1 << 20)) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_REQUIRE(ctx, m->built, SCOPA_ESTATE, "scopa_multi_cfr_exact_iterate_lanes: call scopa_multi_build first");
    if (!n_iters) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (int32_t rc = rows_convert(m, true)) return rc;
    hipLaunchKernelGGL(k_cfr_exact_lanes, dim3((m->n + 63) / 64), dim3(64), 0, ctx->stream, m->d_infoset_T, m->d_payoff_T, m->d_rows, m->d_meta, m->d_visit, m->d_counters, (long long)m->n, (int)n_iters);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_multi_cfr_sync_iterate(scopa_multi *m, int32_t n_iters) {
    if (!m || n_iters < 0 || n_iters > (1 << 20)) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_REQUIRE(ctx, m->built, SCOPA_ESTATE, "scopa_multi_cfr_sync_iterate: call scopa_multi_build first");
    if (int32_t rc = rows_convert(m, false)) return rc;
    if (!n_iters) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds = (size_t)m->max_infosets * 4 * 8 * 2 + sizeof(double) * kNodes * 3 + 1656 * 2;
    SC_REQUIRE(ctx, lds <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_multi_cfr_sync_iterate: a deal's tables do not fit in LDS");
    SC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_cfr_sync), hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit));
    hipLaunchKernelGGL(k_cfr_sync, dim3(m->n), dim3(1024), lds, ctx->stream, m->d_infoset, m->d_payoff, m->d_key, m->d_regret, m->d_strat,
                       0 /* multi-deal */, (int)n_iters, m->d_counters, m->d_visit, m->d_meta);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_multi_mccfr_iterate(scopa_multi *m, uint32_t batch, uint32_t n_iters, uint64_t seed) {
    if (!m || batch == 0 || batch > (1u << 24) || n_iters > (1u << 24)) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_REQUIRE(ctx, m->built, SCOPA_ESTATE, "scopa_multi_mccfr_iterate: call scopa_multi_build first");
    if (int32_t rc = rows_convert(m, false)) return rc;
    if (!n_iters) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const int32_t rc = launch_mccfr_multi(ctx, m->n, m->max_infosets, m->d_infoset, m->d_payoff, m->d_key, m->d_regret, m->d_strat, m->d_meta,
                                          m->d_visit, m->d_counters, seed, m->mccfr_iteration, n_iters, batch);
    if (rc != SCOPA_OK) return rc;
    m->mccfr_iteration += n_iters;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_multi_exploitability(scopa_multi *m, double *h_out4) {
    if (!m || !h_out4) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_REQUIRE(ctx, m->built, SCOPA_ESTATE, "scopa_multi_exploitability: call scopa_multi_build first");
    if (int32_t rc = rows_convert(m, false)) return rc;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds = (size_t)m->max_infosets * 4 * 8 * 2 + sizeof(double) * kNodes * 2 + sizeof(int) * (size_t)m->max_infosets + 1656 * 2;
    SC_REQUIRE(ctx, lds <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "scopa_multi_exploitability: tables do not fit in LDS");
    SC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_exploitability), hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit));
    hipLaunchKernelGGL(k_exploitability, dim3(m->n), dim3(1024), lds, ctx->stream, m->d_infoset, m->d_payoff, m->d_key, m->d_strat,
                       (const double *)nullptr, -m->max_infosets, m->d_out, (double *)nullptr, m->d_meta);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipMemcpyAsync(h_out4, m->d_out, (size_t)m->n * 32, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_multi_tables_get(scopa_multi *m, int32_t deal, double *h_regret, double *h_strategy, double *h_local, uint64_t *h_keys) {
    if (!m || deal < 0 || deal >= m->n) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    SC_REQUIRE(ctx, m->built, SCOPA_ESTATE, "scopa_multi_tables_get: call scopa_multi_build first");
    if (int32_t rc = rows_convert(m, false)) return rc;
    int32_t I = 0;
    SC_HIP(ctx, hipMemcpyAsync(&I, m->d_meta + (size_t)deal * 8, 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t off = (size_t)deal * kDecision * 4, bytes = (size_t)I * 32;
    if (h_regret) SC_HIP(ctx, hipMemcpyAsync(h_regret, m->d_regret + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (h_strategy) SC_HIP(ctx, hipMemcpyAsync(h_strategy, m->d_strat + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (h_local) SC_HIP(ctx, hipMemcpyAsync(h_local, m->d_local + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (h_keys) SC_HIP(ctx, hipMemcpyAsync(h_keys, m->d_key + (size_t)deal * kDecision, (size_t)I * 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_multi_counters(scopa_multi *m, uint64_t *decision_visits, uint64_t *terminal_visits) {
    if (!m) return SCOPA_EINVAL;
    scopa_ctx *ctx = m->ctx;
    std::vector<unsigned long long> h((size_t)m->n * 8);
    SC_HIP(ctx, hipMemcpyAsync(h.data(), m->d_counters, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long a = 0, b = 0;
    for (int d = 0; d < m->n; d++) { a += h[(size_t)d * 8]; b += h[(size_t)d * 8 + 1]; }
    if (decision_visits) *decision_visits = a;
    if (terminal_visits) *terminal_visits = b;
    return SCOPA_OK;
}

}  // extern "C"
