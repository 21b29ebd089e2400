// scopa_mccfr.hip -- external-sampling MCCFR on gfx950.
//
// Reference behaviour: MCCFRTrainer._sample / .iteration (src/algorithms/mc_cfr.py:37-92).  One traversal of the
// reference is a recursion tree: at a traverser node the sampled child is followed (:55-67) and then EVERY legal
// action is re-expanded by a fresh sampled sub-traversal (:69-78); at an opponent node only the sampled child is
// followed.  With the 4,4,3,3,2,2,1,1 legal profile that recursion tree always has 291 (traverser 0) / 172
// (traverser 1) decision visits and 120 leaves: 5*4*3*2 branch combinations at the four traverser nodes of a path.
//
// Kernel design (k_mccfr_traverse): one LANE per LEAF PATH.  A traversal is a "task" of 128 lanes (120 paths + 8
// idle) = 2 wavefronts; lane l decodes its four branch digits (mixed radix 5,4,3,2; digit 0 = follow the sampled
// child, digit i+1 = re-expansion of action i) and walks root -> leaf in 8 dependent steps instead of the
// reference's 231-visit serial DFS.  Lanes that share a prefix recompute the same nodes and agree, because every
// random draw is keyed by the PATH (Philox4x32-10, scopa_philox.h), not by visit order; the draws themselves are
// produced once per task in a dense pass (86 / 26 Philox blocks) and staged in LDS.  Everything the walk touches
// -- sigma and normalised-cdf rows of the frozen regret table, the node -> infoset map, leaf payoffs -- is LDS
// resident (~93 KB for 738 infosets; one persistent 1024-thread workgroup per CU).  Regret deltas are reduced with
// LDS float64 atomics per workgroup; each workgroup then writes its partial table as one coalesced SLAB in HBM
// (plain stores) and k_mccfr_reduce sums the slabs in a fixed order.  (v1 flushed with global float64 atomics:
// 256 workgroups hammering the same 29.5 KB cost ~105 us per launch, 6x the traversal itself -- measured.)
// Strategy sums are accumulated as integer visit counts (sigma is frozen, so strategy_sum += count * sigma).
#include "scopa_ctx.h"
#include "scopa_philox.h"

using namespace scopa;

namespace {

constexpr int kTaskLanes = 128;   // 120 leaf paths, padded to two wavefronts
constexpr int kPaths = 120;
constexpr int kSlots = 86;        // Philox blocks per task (traverser 0); traverser 1 needs the first 26

// InfoNode.current_strategy, mc_cfr.py:20-24  (np.maximum, ndarray.sum left-to-right, elementwise divide)
__device__ __forceinline__ void mc_sigma(const double *R, int n, double *sigma) {
    double pos[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < n; i++) pos[i] = R[i] > 0.0 ? R[i] : 0.0;
    double s = pos[0];
    for (int i = 1; i < n; i++) s += pos[i];
    for (int i = 0; i < 4; i++) sigma[i] = i < n ? (s == 0.0 ? 1.0 / (double)n : pos[i] / s) : 0.0;
}

// np.random.choice(legal, p=sigma): cdf = p.cumsum(); cdf /= cdf[-1]; index = cdf.searchsorted(u, 'right')
__device__ __forceinline__ void choice_cdf(const double *sigma, int n, double *cdf) {
    double c = sigma[0];
    cdf[0] = c;
    for (int i = 1; i < n; i++) { c += sigma[i]; cdf[i] = c; }
    const double last = cdf[n - 1];
    for (int i = 0; i < 4; i++) cdf[i] = i < n ? cdf[i] / last : 2.0;  // 2.0 > any u: padding never counts
}

__device__ __forceinline__ int slot_of(int ntl, uint32_t dig) {
    const int b0 = dig & 7, b1 = (dig >> 3) & 7, b2 = (dig >> 6) & 7;
    return ntl == 0 ? 0 : ntl == 1 ? 1 + b0 : ntl == 2 ? 6 + b0 * 4 + b1 : 26 + (b0 * 4 + b1) * 3 + b2;
}

// One leaf path of traverser TRAV: 8 plies, each { node -> infoset row (sigma[4] | cdf[4], 64 B in LDS); sampled action =
// #{cdf <= u}; at a traverser ply the lane's branch digit may override it }.  Returns the leaf's BFS index; IX / wX
// are the infoset and importance weight (mc_cfr.py:81-82) of the one traverser node this lane updates in phase C.
template <int TRAV>
__device__ __forceinline__ int walk_path(const uint16_t *__restrict__ s_inf, const double *__restrict__ s_sigcdf,
                                         uint8_t *__restrict__ s_seen, const double2 *__restrict__ my_u, uint32_t bpack,
                                         int kstar, int &IX, double &wX) {
    int idx = 0;
    uint32_t dig = 0;
    double reach = 1.0, samp = 1.0, reachX = 0.0, sampX = 0.0;
#pragma unroll
    for (int d = 0; d < kPlies; d++) {
        constexpr int dummy = 0; (void)dummy;
        const int n = 4 - (d >> 1);
        const bool is_trav = (d & 1) == TRAV;
        const int ntl = TRAV == 0 ? (d + 1) >> 1 : d >> 1;  // traverser plies strictly above ply d
        const int In = s_inf[level_offset(d) + idx];
        s_seen[In] = 1;  // benign race: every writer stores 1
        const double *row = s_sigcdf + In * 8;
        int a = 0;
        if (n > 1) {
            const double2 uu = my_u[slot_of(ntl, dig)];
            const double u = is_trav ? uu.y : uu.x;
            a = (row[4] <= u) + (row[5] <= u) + (row[6] <= u) + (row[7] <= u);
            a = a < n - 1 ? a : n - 1;
        }
        if (is_trav) {
            const int dg = (int)((bpack >> (3 * ntl)) & 7u);
            if (dg) a = dg - 1;
            if (ntl == kstar) { IX = In; reachX = reach; sampX = samp; }
            if (n > 1) samp *= row[a];
            dig |= (uint32_t)dg << (3 * ntl);
        } else if (n > 1) {
            reach *= row[a];
        }
        idx = idx * n + a;
    }
    wX = sampX > 0.0 ? reachX / sampX : 0.0;  // weight = opp_reach / sampling_probs[player] if > 0 else 0
    return idx;
}

}  // namespace

// sigma | cdf rows of the frozen regret table, [n_infosets][8] float64, computed once per iteration
__global__ void __launch_bounds__(256)
k_mccfr_prepare(const uint64_t *__restrict__ g_key, const double *__restrict__ g_regret, double *__restrict__ g_sigcdf, int n_infosets) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_infosets) return;
    const int n = (int)((g_key[r] >> 1) & 7);
    double R[4], sg[4], cd[4];
    for (int c = 0; c < 4; c++) R[c] = g_regret[r * 4 + c];
    mc_sigma(R, n, sg);
    choice_cdf(sg, n, cd);
    for (int c = 0; c < 4; c++) { g_sigcdf[r * 8 + c] = sg[c]; g_sigcdf[r * 8 + 4 + c] = cd[c]; }
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_mccfr_traverse(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff,
                 const double *__restrict__ g_sigcdf, double *__restrict__ g_slabs,
                 int n_infosets, uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t b0, uint32_t nb,
                 unsigned long long *__restrict__ g_wg_counts, uint8_t *__restrict__ g_seen_slabs) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned int s_vis[2];
    const int I = n_infosets;
    double *s_sigcdf = reinterpret_cast<double *>(smem);         // [I][8]: sigma[4] | normalised cdf[4]
    double *s_dR = s_sigcdf + (size_t)I * 8;                     // [I][4]
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_dR + (size_t)I * 4);  // [I]
    const int tasks_per_wg = blockDim.x / kTaskLanes;
    double2 *s_u = reinterpret_cast<double2 *>(reinterpret_cast<unsigned char *>(s_cnt) + (((size_t)I * 4 + 15) & ~(size_t)15));
    int *s_px2 = reinterpret_cast<int *>(s_u + (size_t)tasks_per_wg * kSlots);    // [tasks][128]
    uint16_t *s_inf = reinterpret_cast<uint16_t *>(s_px2 + (size_t)tasks_per_wg * kTaskLanes);  // [1653] (+pad)
    int8_t *s_pay = reinterpret_cast<int8_t *>(s_inf + 1656);   // [576]
    uint8_t *s_seen = reinterpret_cast<uint8_t *>(s_pay + kTerminal);  // [I] infoset visited by any lane of this workgroup

    const int tid = threadIdx.x;
    if (tid < 2) s_vis[tid] = 0u;
    // ---- prologue: freeze this iteration's strategy in LDS ----------------------------------------------------------
    for (int i = tid; i < I * 4; i += blockDim.x) {  // 16-byte copies of the prepared rows
        reinterpret_cast<double2 *>(s_sigcdf)[i] = reinterpret_cast<const double2 *>(g_sigcdf)[i];
        s_dR[i] = 0.0;
    }
    for (int r = tid; r < I; r += blockDim.x) { s_cnt[r] = 0u; s_seen[r] = 0; }
    for (int i = tid; i < kDecision; i += blockDim.x) s_inf[i] = g_infoset[i];
    for (int i = tid; i < kTerminal; i += blockDim.x) s_pay[i] = g_payoff[i];
    __syncthreads();

    const int task_in_wg = tid / kTaskLanes, l = tid % kTaskLanes;
    const uint32_t n_tasks = nb * 2u;
    const uint32_t n_groups = (n_tasks + tasks_per_wg - 1) / tasks_per_wg;
    // static per-lane path description
    const int d0 = l / 24, d1 = (l / 6) % 4, d2 = (l >> 1) % 3, d3 = l & 1;
    const uint32_t bpack = (uint32_t)d0 | ((uint32_t)d1 << 3) | ((uint32_t)d2 << 6) | ((uint32_t)d3 << 9);
    const int kstar = l >= kPaths ? -2 : d3 ? 3 : d2 ? 2 : d1 ? 1 : d0 ? 0 : -1;
    unsigned int my_dvis = 0, my_tvis = 0;

    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t task = g * tasks_per_wg + task_in_wg;
        const bool live = task < n_tasks;
        const uint32_t b = b0 + (task >> 1);
        const int trav = (int)(task & 1u);
        double2 *my_u = s_u + (size_t)task_in_wg * kSlots;
        int *my_px2 = s_px2 + (size_t)task_in_wg * kTaskLanes;

        // ---- phase A: this task's random draws, one Philox block per (ntl, branch prefix) ------------------------
        if (live && l < (trav == 0 ? 86 : 26)) {
            int ntl; uint32_t dig;
            if (l == 0) { ntl = 0; dig = 0; }
            else if (l < 6) { ntl = 1; dig = (uint32_t)(l - 1); }
            else if (l < 26) { ntl = 2; const int q = l - 6; dig = (uint32_t)(q / 4) | ((uint32_t)(q % 4) << 3); }
            else { ntl = 3; const int q = l - 26; dig = (uint32_t)(q / 12) | ((uint32_t)((q / 3) % 4) << 3) | ((uint32_t)(q % 3) << 6); }
            const philox_out x = philox4x32_10((uint32_t)ntl + 16u * dig, b, iteration, (uint32_t)trav, seed_lo, seed_hi);
            my_u[l] = make_double2(u53(x.x0, x.x1), u53(x.x2, x.x3));
        }
        __syncthreads();

        // ---- phase B: walk root -> leaf (specialised on the wave-uniform traverser: ply roles are compile-time) ---
        int IX = 0;
        double wX = 0.0;
        if (live && l < kPaths) {
            int leaf;
            if (trav == 0) leaf = walk_path<0>(s_inf, s_sigcdf, s_seen, my_u, bpack, kstar, IX, wX);
            else           leaf = walk_path<1>(s_inf, s_sigcdf, s_seen, my_u, bpack, kstar, IX, wX);
            const int p0 = s_pay[leaf];
            my_px2[l] = trav == 0 ? p0 : -p0;
            my_tvis += 1;
            my_dvis += kstar < 0 ? 8 : 7 - trav - 2 * kstar;
        }
        __syncthreads();

        // ---- phase C: one (traverser node, action) regret update per lane (mc_cfr.py:79-84) ----------------------
        if (live && kstar >= 0) {
            const int nX = 4 - kstar;
            const int stride = kstar == 0 ? 24 : kstar == 1 ? 6 : kstar == 2 ? 2 : 1;
            const int dg = (int)((bpack >> (3 * kstar)) & 7u);
            const int base = l - dg * stride;
            double v = 0.0, mine = 0.0;
            for (int j = 0; j < nX; j++) {
                const double cfv = 0.5 * (double)my_px2[base + (j + 1) * stride];
                v = fma(s_sigcdf[IX * 8 + j], cfv, v);  // np.dot on this numpy build: an fma chain (see oracle)
                if (j == dg - 1) mine = cfv;
            }
            const double delta = wX * (mine - v);
            if (delta != 0.0) atomicAdd(&s_dR[IX * 4 + dg - 1], delta);
            if (dg == 1) atomicAdd(&s_cnt[IX], 1u);
        }
        __syncthreads();
    }

    // ---- epilogue: this workgroup's partial delta table -> its slab, [n_infosets][5] like the delta buffer ----------
    {
        double *slab = g_slabs + (size_t)blockIdx.x * ((size_t)I * 5);
        for (int c = tid; c < I * 5; c += blockDim.x) {
            const int r = c / 5, k = c - r * 5;
            slab[c] = k < 4 ? s_dR[r * 4 + k] : (double)s_cnt[r];
        }
        uint8_t *seen = g_seen_slabs + (size_t)blockIdx.x * kDecision;
        for (int r = tid; r < I; r += blockDim.x) seen[r] = s_seen[r];
    }
    // exact visit counters: wave reduce -> LDS -> ONE plain store per workgroup, summed by k_mccfr_reduce.  (v1 issued
    // one global atomicAdd per wavefront: 8192 same-address atomics at ~12 ns each = the ~100 us fixed cost measured.)
    for (int off = 32; off > 0; off >>= 1) {
        my_dvis += __shfl_down(my_dvis, off);
        my_tvis += __shfl_down(my_tvis, off);
    }
    if ((tid & 63) == 0) { atomicAdd(&s_vis[0], my_dvis); atomicAdd(&s_vis[1], my_tvis); }
    __syncthreads();
    if (tid < 2) g_wg_counts[blockIdx.x * 2 + tid] = s_vis[tid];
}

// delta[c] += sum over slabs, in slab order (deterministic).  A workgroup owns 16 consecutive cells; thread
// (chunk = tid / 16, cell = tid % 16) adds slabs chunk, chunk+16, ... so that 16 lanes read 128 contiguous bytes of
// one slab; the 16 partial sums per cell are then combined in chunk order through LDS.
__global__ void __launch_bounds__(256)
k_mccfr_reduce(const double *__restrict__ g_slabs, int n_slabs, double *__restrict__ g_delta, int n_cells,
               const unsigned long long *__restrict__ g_wg_counts, unsigned long long *__restrict__ g_counters,
               const uint8_t *__restrict__ g_seen_slabs, uint32_t *__restrict__ g_visit) {
    __shared__ double part[16][17];
    __shared__ unsigned long long s_tot[2];
    if (blockIdx.x == gridDim.x - 1) {  // visit counters of this launch: one lane per workgroup record, LDS reduce
        if (threadIdx.x < 2) s_tot[threadIdx.x] = 0ull;
        __syncthreads();
        unsigned long long d = 0ull, t = 0ull;
        for (int w = threadIdx.x; w < n_slabs; w += blockDim.x) { d += g_wg_counts[w * 2]; t += g_wg_counts[w * 2 + 1]; }
        if (d | t) { atomicAdd(&s_tot[0], d); atomicAdd(&s_tot[1], t); }
        __syncthreads();
        if (threadIdx.x < 2) g_counters[threadIdx.x] += s_tot[threadIdx.x];
    }
    {   // infosets first seen by this launch: 16 per workgroup, 16 lanes OR the slabs' flags
        const int r = blockIdx.x * 16 + (threadIdx.x & 15);
        const int n_rows = n_cells / 5;
        unsigned int any = 0u;
        if (r < n_rows)
            for (int w = threadIdx.x >> 4; w < n_slabs; w += 16) any |= g_seen_slabs[(size_t)w * kDecision + r];
        if (any && g_visit[r] == 0u) g_visit[r] = 0x40000000u + (uint32_t)r;  // racing writers store the same value
    }
    const int cell_l = threadIdx.x & 15, chunk = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cell_l;
    double acc = 0.0;
    if (c < n_cells) {
        // all loads of a batch are issued before the first add (16 independent requests in flight per lane; the
        // dependent load->add form of v1 took 35 us for 7.5 MB), then summed in slab order
        for (int base = chunk; base < n_slabs; base += 256) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int sl = base + 16 * j;
                v[j] = sl < n_slabs ? __builtin_nontemporal_load(&g_slabs[(size_t)sl * n_cells + c]) : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 16; j++) acc += v[j];
        }
    }
    part[chunk][cell_l] = acc;
    __syncthreads();
    if (chunk == 0 && c < n_cells) {
        double t = part[0][cell_l];
        for (int k = 1; k < 16; k++) t += part[k][cell_l];
        g_delta[c] += t;
    }
}

// regret += delta; strategy_sum += count * sigma(frozen regret); delta <- 0
__global__ void __launch_bounds__(256)
k_mccfr_apply(const uint64_t *__restrict__ g_key, double *__restrict__ g_regret, double *__restrict__ g_strat,
              double *__restrict__ g_delta, int n_infosets, double *__restrict__ g_sigcdf) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_infosets) return;
    const int n = (int)((g_key[r] >> 1) & 7);
    double R[4], sg[4];
    for (int c = 0; c < 4; c++) R[c] = g_regret[r * 4 + c];
    mc_sigma(R, n, sg);
    const double cnt = g_delta[r * 5 + 4];
    for (int c = 0; c < n; c++) {
        R[c] += g_delta[r * 5 + c];
        g_regret[r * 4 + c] = R[c];
        g_strat[r * 4 + c] += cnt * sg[c];
    }
    for (int c = 0; c < 5; c++) g_delta[r * 5 + c] = 0.0;
    // next iteration's frozen strategy rows (what k_mccfr_prepare would compute)
    double cd[4];
    mc_sigma(R, n, sg);
    choice_cdf(sg, n, cd);
    for (int c = 0; c < 4; c++) { g_sigcdf[r * 8 + c] = sg[c]; g_sigcdf[r * 8 + 4 + c] = cd[c]; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Replay mode: the reference's own sequential semantics (tables are live: every update is seen by the next visit),
// driven by the uniforms np.random.choice would have drawn.  One lane; this is the bit-exactness anchor, not the
// throughput path.
namespace {
struct ReplayFrame {
    int idx, I, phase, a_first;
    double sigma[4], cfv[4], reach_opp, samp_trav, util;
};
}  // namespace

__global__ void __launch_bounds__(64)
k_mccfr_replay(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, const uint64_t *__restrict__ g_key,
               double *__restrict__ g_regret, double *__restrict__ g_strat, const double *__restrict__ uniforms,
               long long n_uniforms, int n_iters, unsigned long long *__restrict__ g_counters, long long *__restrict__ consumed,
               uint32_t *__restrict__ g_visit, int32_t *__restrict__ g_meta) {
    __shared__ ReplayFrame fr[kPlies + 1];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long upos = 0;
    unsigned long long dvis = 0, tvis = 0;
    uint32_t seq = (uint32_t)g_meta[1];
    for (int it = 0; it < n_iters; it++) {
        for (int trav = 0; trav < 2; trav++) {  // iteration(), mc_cfr.py:88-92
            int d = 0;
            fr[0].idx = 0; fr[0].phase = -1; fr[0].reach_opp = 1.0; fr[0].samp_trav = 1.0;
            double ret = 0.0;
            bool returning = false;
            while (d >= 0) {
                if (d == kPlies) {  // terminal: state.rewards()[traversing_player] (:38-39)
                    const int p0 = g_payoff[fr[d].idx];
                    ret = 0.5 * (double)(trav == 0 ? p0 : -p0);
                    tvis++;
                    returning = true;
                    d--;
                    continue;
                }
                ReplayFrame &f = fr[d];
                const int n = 4 - (d >> 1);
                const bool is_trav = (d & 1) == trav;
                if (!returning) {  // node entry (:49-55)
                    dvis++;
                    f.I = g_infoset[level_offset(d) + f.idx];
                    if (g_visit[f.I] == 0u) g_visit[f.I] = ++seq;  // _get_node inserts on first visit (mc_cfr.py:32-35)
                    double R[4];
                    for (int c = 0; c < 4; c++) R[c] = g_regret[f.I * 4 + c];
                    mc_sigma(R, n, f.sigma);
                    double cdf[4];
                    choice_cdf(f.sigma, n, cdf);
                    const double u = upos < n_uniforms ? uniforms[upos] : 0.0;
                    upos++;
                    int a = (cdf[0] <= u) + (cdf[1] <= u) + (cdf[2] <= u) + (cdf[3] <= u);
                    a = a < n - 1 ? a : n - 1;
                    f.a_first = a;
                    f.phase = 0;
                    ReplayFrame &c = fr[d + 1];
                    c.idx = f.idx * n + a;
                    if (is_trav) { c.reach_opp = f.reach_opp; c.samp_trav = f.samp_trav * f.sigma[a]; }
                    else         { c.reach_opp = f.reach_opp * f.sigma[a]; c.samp_trav = f.samp_trav; }
                    d++;
                    continue;
                }
                // a child returned `ret`
                if (!is_trav) { d--; continue; }  // opponent node: pass the sampled child's value up (:86)
                if (f.phase == 0) f.util = ret; else f.cfv[f.phase - 1] = ret;
                if (f.phase < n) {  // re-expand legal action number f.phase (:72-78)
                    const int i = f.phase;
                    f.phase++;
                    ReplayFrame &c = fr[d + 1];
                    c.idx = f.idx * n + i;
                    c.reach_opp = f.reach_opp;
                    c.samp_trav = f.samp_trav * f.sigma[i];
                    returning = false;
                    d++;
                    continue;
                }
                double v = 0.0;  // (:79-84)
                for (int i = 0; i < n; i++) v = fma(f.sigma[i], f.cfv[i], v);
                const double w = f.samp_trav > 0.0 ? f.reach_opp / f.samp_trav : 0.0;
                for (int i = 0; i < n; i++) {
                    g_regret[f.I * 4 + i] += w * (f.cfv[i] - v);
                    g_strat[f.I * 4 + i] += 1.0 * f.sigma[i];  // reach_probs[traverser] stays 1.0 (:61-65)
                }
                ret = f.util;
                d--;
            }
        }
    }
    g_counters[0] += dvis;
    g_counters[1] += tvis;
    g_meta[1] = (int32_t)seq;
    *consumed = upos;
}

// ---------------------------------------------------------------------------------------------------------------------
static size_t traverse_lds_bytes(int n_infosets, int threads) {
    const int tasks = threads / kTaskLanes;
    size_t b = (size_t)n_infosets * 4 * 8 * 3;
    b += ((size_t)n_infosets * 4 + 15) & ~(size_t)15;
    b += (size_t)tasks * kSlots * 16 + (size_t)tasks * kTaskLanes * 4;
    b += 1656 * 2 + 576 + (size_t)n_infosets;
    return (b + 15) & ~(size_t)15;
}

static int32_t launch_traverse(scopa_ctx *ctx, uint32_t iteration, uint32_t b0, uint32_t nb) {
    const int threads = 1024;
    const size_t lds = traverse_lds_bytes(ctx->n_infosets, threads);
    SC_REQUIRE(ctx, lds + 64 <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "mccfr traverse: infoset tables do not fit in LDS");
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mccfr_traverse),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 64));  // 8 B of static LDS (s_vis)
        attr_set = true;
    }
    const uint32_t n_groups = (nb * 2u + (threads / kTaskLanes) - 1) / (threads / kTaskLanes);
    const uint32_t grid = n_groups < (uint32_t)ctx->n_cus ? n_groups : (uint32_t)ctx->n_cus;
    const int n_cells = ctx->n_infosets * 5;
    const size_t slab_bytes = (size_t)grid * n_cells * sizeof(double);
    if (slab_bytes > ctx->slab_bytes) {
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_slabs) SC_HIP(ctx, hipFree(ctx->d_slabs));
        ctx->d_slabs = nullptr; ctx->slab_bytes = 0;
        const size_t want = (size_t)ctx->n_cus * kDecision * 5 * sizeof(double);  // worst case, allocated once (16.9 MB)
        SC_HIP(ctx, hipMalloc(&ctx->d_slabs, want > slab_bytes ? want : slab_bytes));
        ctx->slab_bytes = want > slab_bytes ? want : slab_bytes;
    }
    if (!ctx->sigcdf_valid) {  // tables were changed by another entry point since the last apply
        hipLaunchKernelGGL(k_mccfr_prepare, dim3((ctx->n_infosets + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_key,
                           ctx->d_regret, ctx->d_sigcdf, ctx->n_infosets);
        SC_HIP(ctx, hipGetLastError());
        ctx->sigcdf_valid = true;
    }
    prof_begin(ctx);
    hipLaunchKernelGGL(k_mccfr_traverse, dim3(grid), dim3(threads), lds, ctx->stream, ctx->d_infoset, ctx->d_payoff,
                       ctx->d_sigcdf, ctx->d_slabs, ctx->n_infosets, (uint32_t)ctx->seed,
                       (uint32_t)(ctx->seed >> 32), iteration, b0, nb, ctx->d_counters + 8, ctx->d_seen_slabs);
    prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_mccfr_reduce, dim3((n_cells + 15) / 16), dim3(256), 0, ctx->stream, ctx->d_slabs, (int)grid,
                       ctx->d_delta, n_cells, ctx->d_counters + 8, ctx->d_counters, ctx->d_seen_slabs, ctx->d_visit);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

extern "C" {

int32_t scopa_mccfr_seed(scopa_ctx *ctx, uint64_t seed) {
    if (!ctx) return SCOPA_EINVAL;
    ctx->seed = seed;
    return SCOPA_OK;
}

int32_t scopa_mccfr_traverse(scopa_ctx *ctx, uint32_t iteration, uint32_t b0, uint32_t nb) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_traverse: no deal set");
    SC_REQUIRE(ctx, nb <= (1u << 30), SCOPA_EINVAL, "scopa_mccfr_traverse: batch too large");
    if (nb == 0) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return launch_traverse(ctx, iteration, b0, nb);
}

int32_t scopa_mccfr_delta_buffer(scopa_ctx *ctx, void **d_delta, size_t *bytes) {
    if (!ctx || !d_delta || !bytes) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_delta_buffer: no deal set");
    *d_delta = ctx->d_delta;
    *bytes = (size_t)ctx->n_infosets * 5 * sizeof(double);
    return SCOPA_OK;
}

int32_t scopa_mccfr_bind_delta(scopa_ctx *ctx, void *d_buf, size_t bytes) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_bind_delta: no deal set");
    const size_t need = (size_t)ctx->n_infosets * 5 * sizeof(double);
    if (d_buf) {
        SC_REQUIRE(ctx, bytes >= need, SCOPA_EINVAL, "scopa_mccfr_bind_delta: buffer smaller than n_infosets*5 float64");
        SC_REQUIRE(ctx, ((uintptr_t)d_buf & 15) == 0, SCOPA_EINVAL, "scopa_mccfr_bind_delta: buffer must be 16-byte aligned");
        ctx->d_delta = static_cast<double *>(d_buf);
    } else {
        ctx->d_delta = ctx->d_delta_own;
    }
    SC_HIP(ctx, hipSetDevice(ctx->device));
    SC_HIP(ctx, hipMemsetAsync(ctx->d_delta, 0, need, ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_mccfr_delta_get(scopa_ctx *ctx, double *h_delta) {
    if (!ctx || !h_delta) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_delta_get: no deal set");
    SC_HIP(ctx, hipMemcpyAsync(h_delta, ctx->d_delta, (size_t)ctx->n_infosets * 5 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_mccfr_delta_set(scopa_ctx *ctx, const double *h_delta) {
    if (!ctx || !h_delta) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_delta_set: no deal set");
    SC_HIP(ctx, hipMemcpyAsync(ctx->d_delta, h_delta, (size_t)ctx->n_infosets * 5 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_mccfr_apply(scopa_ctx *ctx) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_apply: no deal set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_mccfr_apply, dim3((ctx->n_infosets + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_key,
                       ctx->d_regret, ctx->d_strat, ctx->d_delta, ctx->n_infosets, ctx->d_sigcdf);
    SC_HIP(ctx, hipGetLastError());
    ctx->sigcdf_valid = true;
    ctx->iteration++;
    return SCOPA_OK;
}

int32_t scopa_mccfr_iteration_counter(scopa_ctx *ctx, uint32_t *iteration) {
    if (!ctx || !iteration) return SCOPA_EINVAL;
    *iteration = ctx->iteration;
    return SCOPA_OK;
}

int32_t scopa_mccfr_iterate(scopa_ctx *ctx, uint32_t batch, uint32_t n_iters) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_iterate: no deal set");
    SC_REQUIRE(ctx, batch > 0 && batch <= (1u << 30), SCOPA_EINVAL, "scopa_mccfr_iterate: bad batch");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t it = 0; it < n_iters; it++) {
        int32_t rc = launch_traverse(ctx, ctx->iteration, 0, batch);
        if (rc != SCOPA_OK) return rc;
        rc = scopa_mccfr_apply(ctx);
        if (rc != SCOPA_OK) return rc;
    }
    return SCOPA_OK;
}

int32_t scopa_mccfr_replay(scopa_ctx *ctx, int32_t n_iters, const double *h_uniforms, int64_t n_uniforms, int64_t *consumed) {
    if (!ctx || n_iters < 0 || n_uniforms < 0 || (n_uniforms > 0 && !h_uniforms)) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_replay: no deal set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ubytes = (size_t)n_uniforms * sizeof(double);
    { const int32_t rc = ensure_scratch(ctx, ubytes + 64); if (rc != SCOPA_OK) return rc; }
    long long *d_consumed = reinterpret_cast<long long *>(ctx->d_scratch);
    double *d_u = ctx->d_scratch + 8;
    if (n_uniforms) SC_HIP(ctx, hipMemcpyAsync(d_u, h_uniforms, ubytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_mccfr_replay, dim3(1), dim3(64), 0, ctx->stream, ctx->d_infoset, ctx->d_payoff, ctx->d_key,
                       ctx->d_regret, ctx->d_strat, d_u, (long long)n_uniforms, (int)n_iters, ctx->d_counters, d_consumed,
                       ctx->d_visit, ctx->d_meta);
    SC_HIP(ctx, hipGetLastError());
    long long used = 0;
    SC_HIP(ctx, hipMemcpyAsync(&used, d_consumed, sizeof used, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sigcdf_valid = false;
    if (consumed) *consumed = used;
    return SCOPA_OK;
}

}  // extern "C"
