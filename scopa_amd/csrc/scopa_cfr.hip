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
//
// k_cfr_exact_sched (round 2) runs the SAME sequential semantics in ~75 parallel steps instead of 1653 visits.  What the reference's
// order actually constrains is the order of the visits of ONE infoset (each visit reads the local_strategy the previous visit of
// that infoset left, :97); visits of different infosets commute.  A visit of node Y is an interval [enter, exit]: children read
// Y's local_strategy row while Y is open, and the row is rewritten at Y's exit -- so Y may open only after the previous node of its
// infoset (in DFS order) has exited, and exits only after its children have.  Nodes of one infoset are never nested (a player's hand
// shrinks along a path), so those are all the constraints.  The schedule is computed once per deal on the host: an as-soon-as-
// possible levelling of the EXIT events under exactly those constraints (enter costs no step: a node's reach probabilities are the
// product, root first, of its ancestors' local_strategy entries, and every ancestor is still open -- its row unchanged -- when the
// node exits).  Per step, one lane per event; a workgroup barrier between steps.  Every infoset sees the same sequence of float64
// operations as in the reference, so the tables stay bit-identical (tests: vanilla_cfr.npz, 5 checkpoints to 200 iterations).
#include <vector>

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

// ---- the scheduled form: whole-tree traversals of one deal -------------------------------------------------------------------
namespace {
constexpr int kSchedThreads = 256;     // 64 events per pass: a quad of lanes per event, one lane per action slot (one wavefront per SIMD: 0.099 s per 1000 iterations of the seed-42 deal; 128 threads 0.108, 512 0.107, 1024 0.162)

// value of lane K (0..3) of the caller's quad (4 consecutive lanes), for every lane of the quad: two DPP moves, no LDS
template <int K>
__device__ __forceinline__ double quad_bcast(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), K * 0x55, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), K * 0x55, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
}  // namespace

// events[e] = node (BFS index, 11 bits) | ply << 11 | (BFS index of its first child, or its first terminal's index at ply 7) << 16, grouped by
// step: events of step s are [steps[s], steps[s + 1]);
// paths[node][k], k < ply: the local_strategy cell (infoset * 4 + action) of the node's ancestor at ply k on the way to the node
__global__ void __launch_bounds__(kSchedThreads)
k_cfr_exact_sched(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, double *__restrict__ g_regret,
                  double *__restrict__ g_strat, double *__restrict__ g_local, int n_infosets, int n_traversals, int first_traverser,
                  double *__restrict__ root_values, unsigned long long *__restrict__ g_counters, uint32_t *__restrict__ g_visit,
                  int32_t *__restrict__ g_meta, const uint32_t *__restrict__ g_events, const uint16_t *__restrict__ g_steps, int n_steps,
                  const uint16_t *__restrict__ g_paths) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, cells = n_infosets * 4;
    double *R = reinterpret_cast<double *>(smem), *S = R + cells, *L = S + cells;
    double *val = L + cells;                                                    // [kDecision] node values of the traversal under way
    uint32_t *s_ev = reinterpret_cast<uint32_t *>(val + kDecision);             // [kDecision + 1]
    uint16_t *s_inf = reinterpret_cast<uint16_t *>(s_ev + kDecision + 1);       // [1656]
    uint16_t *s_st = s_inf + 1656;                                              // [n_steps + 1]
    uint16_t *s_path = s_st + ((n_steps + 2 + 7) & ~7);                         // [kDecision][8]
    int8_t *s_pay = reinterpret_cast<int8_t *>(s_path + kDecision * 8);         // [kTerminal]
    for (int i = tid; i < cells; i += kSchedThreads) { R[i] = g_regret[i]; S[i] = g_strat[i]; L[i] = g_local[i]; }
    for (int i = tid; i < kDecision; i += kSchedThreads) s_ev[i] = g_events[i];   // (the infoset ids ride in the path records' eighth slot)
    if (tid == 0) *reinterpret_cast<double *>(s_inf) = 1.0;                         // the cell path entries beyond a node's ply point at (sched_one_cell)
    for (int i = tid; i <= n_steps; i += kSchedThreads) s_st[i] = g_steps[i];
    for (int i = tid; i < kDecision * 4; i += kSchedThreads) reinterpret_cast<uint32_t *>(s_path)[i] = reinterpret_cast<const uint32_t *>(g_paths)[i];
    for (int i = tid; i < kTerminal; i += kSchedThreads) s_pay[i] = g_payoff[i];
    // dict insertion on first visit (:51-54): a whole-tree traversal meets the infosets in id order (ids ARE the DFS first-visit order)
    if (tid == 0 && n_traversals > 0) {
        uint32_t seq = (uint32_t)g_meta[1];
        for (int I = 0; I < n_infosets; I++)
            if (g_visit[I] == 0u) g_visit[I] = ++seq;
        g_meta[1] = (int32_t)seq;
        g_counters[0] += (unsigned long long)n_traversals * kDecision;
        g_counters[1] += (unsigned long long)n_traversals * kTerminal;
    }
    __syncthreads();
    // The records of a step's first 64 events are read one step AHEAD (they are static, and stored per event): the step's own chain then starts at the
    // table cells it reads, not three dependent LDS reads earlier (step offsets -> event -> path cells / infoset): 0.099 -> 0.086 s per 1000 iterations (a step
    // stays a chain of ~150 dependent-issue instructions on wavefronts alone on their SIMD: 0.6 us).
    const int quad = tid >> 2, slot = tid & 3;
    int e0 = s_st[0], e1 = s_st[1];
    int n0 = s_st[1 < n_steps ? 1 : 0], n1 = s_st[(1 < n_steps ? 1 : 0) + 1];       // the offsets themselves two steps ahead: no address of a step waits on a read
    uint32_t cur_ev = s_ev[e0 + quad < e1 ? e0 + quad : e0];
    uint4 cur_pw = *reinterpret_cast<const uint4 *>(s_path + (e0 + quad < e1 ? e0 + quad : e0) * 8);
    for (int t = 0; t < n_traversals; t++) {
        const int trav = (first_traverser + t) & 1;   // train(): for i in range(num_players) (:108-110)
        for (int s = 0; s < n_steps; s++) {
            // One visit (vanilla_cfr.py:87-97) per QUAD of lanes, lane i of the quad = action slot i: the slot's table cells, its child's
            // value, its product, its regret update and its division are the lane's own; the two ordered sums (np.sum over <= 4 slots,
            // left to right) are done by every lane on quad-broadcast operands.  Branch-free: slots >= n and quads without an event
            // are SELECTED away (never added as zeros: x + 0.0 would turn -0.0 into +0.0), stores are predicated.  With one lane per
            // node a step cost ~500 instructions on wavefronts that sit alone on their SIMD (1.2 us); a quad per node issues ~150.
            const int ne = n0 + quad < n1 ? n0 + quad : n0;                   // the next step's record (of this traversal or the next one: same schedule)
            const uint32_t nxt_ev = s_ev[ne];
            const uint4 nxt_pw = *reinterpret_cast<const uint4 *>(s_path + ne * 8);
            const int sm = s + 2 < n_steps ? s + 2 : s + 2 - n_steps;         // ... and the offsets of the one after
            const int m0 = s_st[sm], m1 = s_st[sm + 1];
            for (int eb = e0; eb < e1; eb += kSchedThreads / 4) {
                const int e = eb + quad;
                const bool live = e < e1;
                uint32_t ev = cur_ev;
                uint4 pw = cur_pw;
                if (eb != e0) {                                              // a step of more than 64 events: the later passes read their records here
                    ev = s_ev[live ? e : e0];
                    pw = *reinterpret_cast<const uint4 *>(s_path + (live ? e : e0) * 8);
                }
                const int node = (int)(ev & 2047u), d = (int)((ev >> 11) & 7u), n = 4 - (d >> 1), cbase = (int)(ev >> 16);
                const bool on = slot < n;
                // reach probabilities: the product, root first, of the ancestors' local_strategy entries along the path (:79-85)
                const uint32_t pcw[4] = {pw.x, pw.y, pw.z, pw.w};             // the 7 path cells and the infoset id
                double pl[kPlies - 1], r0 = 1.0, r1 = 1.0;
#pragma unroll
                for (int k = 0; k < kPlies - 1; k++) pl[k] = L[(pcw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];   // all seven reads in flight (cells beyond the ply: the 1.0 cell)
                const bool leafp = d == kPlies - 1;
                const int off = on ? slot : 0, I = (int)(pw.w >> 16);
                const int p0 = s_pay[leafp ? cbase + off : 0];
                const double child = val[leafp ? 0 : cbase + off];
                const double ls = L[I * 4 + slot], Rc = R[I * 4 + slot], Sc = S[I * 4 + slot];   // rows are zero-padded beyond n
#pragma unroll
                for (int k = 0; k < kPlies - 1; k++) {   // x * 1.0 is x: the cells of plies beyond the node's read 1.0 and leave the product untouched
                    if ((k & 1) == 0) r0 = r0 * pl[k]; else r1 = r1 * pl[k];
                }
                const bool is_trav = (d & 1) == trav;
                const double reach = trav == 0 ? r0 : r1, opp = trav == 0 ? r1 : r0;
                const double au = leafp ? 0.5 * (double)(trav == 0 ? p0 : -p0) : child;   // terminal (:58-59) or the child's value
                const double prod = ls * au;
                double v = quad_bcast<0>(prod);  // np.sum(local_strategy * action_utils) (:87)
                { const double t1 = v + quad_bcast<1>(prod); v = n > 1 ? t1 : v; }
                { const double t2 = v + quad_bcast<2>(prod); v = n > 2 ? t2 : v; }
                { const double t3 = v + quad_bcast<3>(prod); v = n > 3 ? t3 : v; }
                const bool upd = is_trav && on;   // (:89-95)
                const double Rn = upd ? Rc + opp * (au - v) : Rc, Sn = upd ? Sc + reach * ls : Sc;
                // local_strategy refresh on EVERY visit (:97) = InfoNode.get_strategy (:23-30)
                const double pos = (on && Rn > 0.0) ? Rn : 0.0;
                double sum = quad_bcast<0>(pos);
                { const double t1 = sum + quad_bcast<1>(pos); sum = n > 1 ? t1 : sum; }
                { const double t2 = sum + quad_bcast<2>(pos); sum = n > 2 ? t2 : sum; }
                { const double t3 = sum + quad_bcast<3>(pos); sum = n > 3 ? t3 : sum; }
                const double uni = n == 4 ? 0.25 : n == 3 ? 1.0 / 3.0 : n == 2 ? 0.5 : 1.0;   // 1.0 / n, correctly rounded either way
                const double q = pos / sum;
                if (live) {
                    L[I * 4 + slot] = on ? (sum > 0.0 ? q : uni) : 0.0;
                    if (upd) { R[I * 4 + slot] = Rn; S[I * 4 + slot] = Sn; }
                    if (slot == 0) val[node] = v;
                }
            }
            e0 = n0; e1 = n1; cur_ev = nxt_ev; cur_pw = nxt_pw; n0 = m0; n1 = m1;
            __syncthreads();
        }
        if (tid == 0 && root_values) root_values[t] = val[0];
    }
    __syncthreads();
    for (int i = tid; i < cells; i += kSchedThreads) { g_regret[i] = R[i]; g_strat[i] = S[i]; g_local[i] = L[i]; }
}

// as-soon-as-possible levelling of the EXIT events of one deal's tree under the per-infoset order (see the head of this file)
constexpr int kSchedSteps = 2 * kDecision, kSchedPaths = (3 * kDecision + 2 + 7) & ~7;   // offsets into d_sched, in uint16 units

// index, in float64 cells from the local_strategy table's base, of the cell k_cfr_exact_sched keeps at 1.0: the first 8 bytes of the (otherwise unused) node ->
// infoset area behind the tables, the node values and the event words
static int sched_one_cell(int n_infosets) { return n_infosets * 4 + kDecision + (kDecision + 1) / 2; }

static int32_t build_exact_schedule(scopa_ctx *ctx) {
    if (ctx->sched_valid) return SCOPA_OK;
    std::vector<uint16_t> inf(kDecision);
    SC_HIP(ctx, hipMemcpyAsync(inf.data(), ctx->d_infoset, kDecision * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> last_exit(kDecision, 0), exit_t(kDecision, 0);
    // iterative DFS in the reference's order (children in legal-action order); a node's "open" time = max(parent's, previous node of its infoset's exit)
    struct Frame { int d, idx, open, child, mx; };
    std::vector<Frame> st;
    st.push_back({0, 0, 0, 0, 0});
    st.back().open = last_exit[inf[0]]; st.back().mx = st.back().open;
    int T = 0;
    while (!st.empty()) {
        Frame &f = st.back();
        const int n = nlegal_at(f.d);
        if (f.d == kPlies - 1 || f.child == n) {   // children done (ply 7's children are terminals): this node exits
            const int node = level_offset(f.d) + f.idx, tx = f.mx + 1;
            exit_t[node] = tx; last_exit[inf[node]] = tx;
            if (tx > T) T = tx;
            st.pop_back();
            if (!st.empty() && tx > st.back().mx) st.back().mx = tx;
            continue;
        }
        const int cd = f.d + 1, cidx = f.idx * n + f.child;
        f.child++;
        const int popen = f.open;
        Frame c{cd, cidx, 0, 0, 0};
        const int cI = inf[level_offset(cd) + cidx];
        c.open = popen > last_exit[cI] ? popen : last_exit[cI];
        c.mx = c.open;
        st.push_back(c);
    }
    std::vector<uint16_t> steps(T + 1, 0);
    std::vector<uint32_t> events(kDecision);
    std::vector<int> fill(T + 1, 0);
    for (int node = 0; node < kDecision; node++) fill[exit_t[node]]++;          // exit times are 1..T
    int off = 0;
    for (int s = 1; s <= T; s++) { steps[s - 1] = (uint16_t)off; off += fill[s]; fill[s] = steps[s - 1]; }
    steps[T] = (uint16_t)off;
    for (int d = 0; d < kPlies; d++)
        for (int node = level_offset(d); node < level_offset(d + 1); node++) {
            const int idx = node - level_offset(d), first_child = (d == kPlies - 1 ? 0 : level_offset(d + 1)) + idx * nlegal_at(d);
            events[fill[exit_t[node]]++] = (uint32_t)node | ((uint32_t)d << 11) | ((uint32_t)first_child << 16);
        }
    // the local_strategy cells along every node's path (static per deal)
    std::vector<uint16_t> paths((size_t)kDecision * 8, 0);
    for (int d = 1; d < kPlies; d++)
        for (int idx = 0; idx < level_offset(d + 1) - level_offset(d); idx++) {
            int x = idx;
            for (int k = d; k > 0; k--) {
                const int pn = nlegal_at(k - 1), act = x % pn;
                x /= pn;
                paths[(size_t)(level_offset(d) + idx) * 8 + (k - 1)] = (uint16_t)(inf[level_offset(k - 1) + x] * 4 + act);
            }
        }
    // ... stored per EVENT (the order the kernel walks them in: the address of the next step's record never depends on the record itself), with the
    // node's infoset id in the unused eighth slot -- one 16-byte read per event
    // -- and the cells BEYOND the node's ply point at a cell that holds 1.0 (sched_one_cell: x * 1.0 is x, so the reach products need no selects)
    std::vector<uint16_t> paths_e((size_t)kDecision * 8, 0);
    const uint16_t one = (uint16_t)sched_one_cell(ctx->n_infosets);
    for (int e = 0; e < kDecision; e++) {
        const int node = (int)(events[e] & 2047u), d = (int)((events[e] >> 11) & 7u);
        for (int k = 0; k < 7; k++) paths_e[(size_t)e * 8 + k] = k < d ? paths[(size_t)node * 8 + k] : one;
        paths_e[(size_t)e * 8 + 7] = inf[node];
    }
    // one buffer (uint16 units): [2 * kDecision] events as uint32 | [kDecision + 2] step offsets | [kDecision][8] path cells (16-byte aligned)
    if (!ctx->d_sched) SC_HIP(ctx, hipMalloc(&ctx->d_sched, (size_t)(kSchedPaths + kDecision * 8) * sizeof(uint16_t)));
    SC_HIP(ctx, hipMemcpyAsync(ctx->d_sched + kSchedPaths, paths_e.data(), paths_e.size() * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->d_sched, events.data(), kDecision * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->d_sched + kSchedSteps, steps.data(), (size_t)(T + 1) * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sched_steps = T;
    ctx->sched_valid = true;
    return SCOPA_OK;
}

static size_t sched_lds_bytes(int n_infosets, int n_steps) {
    return (size_t)n_infosets * 4 * 8 * 3 + (size_t)kDecision * 8 + (size_t)(kDecision + 1) * 4 + (1656 + ((n_steps + 2 + 7) & ~7) + kDecision * 8) * sizeof(uint16_t) + kTerminal + 32;
}

static int32_t run_exact(scopa_ctx *ctx, int n_traversals, int first_traverser, double *h_values, int start_depth = 0,
                         int start_idx = 0, double r0 = 1.0, double r1 = 1.0) {
    SC_HIP(ctx, hipSetDevice(ctx->device));
    { const int32_t rc = ensure_scratch(ctx, (size_t)(n_traversals > 0 ? n_traversals : 1) * sizeof(double)); if (rc != SCOPA_OK) return rc; }
    if (start_depth == 0 && r0 == 1.0 && r1 == 1.0 && !ctx->exact_sequential) {   // whole-tree traversals: the scheduled form, if the deal's tables fit beside it
        const int32_t rc = build_exact_schedule(ctx);
        if (rc != SCOPA_OK) return rc;
        const size_t need = sched_lds_bytes(ctx->n_infosets, ctx->sched_steps);
        if (need <= (size_t)ctx->lds_limit) {
            SC_LDS_ATTR(ctx, scopa::kLdsCfrSched, k_cfr_exact_sched, ctx->lds_limit);
            hipLaunchKernelGGL(k_cfr_exact_sched, dim3(1), dim3(kSchedThreads), need, ctx->stream, ctx->d_infoset, ctx->d_payoff, ctx->d_regret,
                               ctx->d_strat, ctx->d_local, ctx->n_infosets, n_traversals, first_traverser, ctx->d_scratch, ctx->d_counters,
                               ctx->d_visit, ctx->d_meta, reinterpret_cast<const uint32_t *>(ctx->d_sched), ctx->d_sched + kSchedSteps, ctx->sched_steps,
                               ctx->d_sched + kSchedPaths);
            SC_HIP(ctx, hipGetLastError());
            if (h_values && n_traversals > 0)
                SC_HIP(ctx, hipMemcpyAsync(h_values, ctx->d_scratch, (size_t)n_traversals * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->sigcdf_valid = false;
            return SCOPA_OK;
        }
    }
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

int32_t scopa_cfr_exact_mode(scopa_ctx *ctx, int32_t sequential) {
    if (!ctx || (sequential != 0 && sequential != 1)) return SCOPA_EINVAL;
    ctx->exact_sequential = sequential != 0;
    return SCOPA_OK;
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
