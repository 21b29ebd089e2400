// scopa_mccfr.hip -- external-sampling MCCFR on gfx950.
//
// Reference behaviour: MCCFRTrainer._sample / .iteration (src/algorithms/mc_cfr.py:37-92).  One traversal of the
// reference is a recursion tree: at a traverser node the sampled child is followed (:55-67) and then EVERY legal
// action is re-expanded by a fresh sampled sub-traversal (:69-78); at an opponent node only the sampled child is
// followed.  With the 4,4,3,3,2,2,1,1 legal profile that recursion tree is always the same shape: per ply
// 1,5,5,20,20,60,60,120 nodes for traverser 0 and 1,1,5,5,20,20,60,60 for traverser 1 (291 / 172 decision visits,
// 120 leaves).  A node is (traverser, ntl, j): ntl traverser plies above it, j its branch index at that level (child of a traverser
// node with n legal actions: j*(n+1) for the sampled child, j*(n+1) + i + 1 for the re-expansion of action i).
//
// Kernel design (k_mccfr_traverse): LEVEL-SYNCHRONOUS over the recursion tree, one lane per UNIQUE node, a
// WAVEFRONT per traversal pair (two pairs in flight when it has two left).  Ply d of a pair is one step over its 2, 6, 10, 25, 40,
// 80 nodes (plies 0..5), each lane deriving its node from its parent's record word, fetched from the parent's lane with a cross-lane
// read (index arithmetic: the game tree is regular), sampling its action from the frozen sigma|threshold row and keeping its own
// record in a register for the next ply and in LDS for the update step.  What is static per lane (parent lane, forced action, draw
// word, the update lane's ancestor records, leaf values and forced actions) comes from a table built once per context (k_lane_table), staged
// into LDS by the prologue and read 16 bytes at a time by the stage that needs it.  The 16 wavefronts of a workgroup TAKE their pairs
// from a counter in LDS and run independently -- only wave-level LDS ordering between stages, no workgroup barrier in the main loop
// -- so they hide each other's latencies (v3 had workgroup-wide plies and was latency-bound).
// Random draws are Philox4x32-10 blocks keyed by the node: block (ntl, j >> 1) of the (global traversal id, iteration, traverser)
// stream serves the four nodes (j even | odd) x (opponent node | traverser node below it), one 32-bit word each.  All 58 blocks of a
// pair are computed in ONE dense pre-pass (draw_pairs: one 16-byte LDS store per lane) and kept as 31-bit integers that are compared with integer thresholds
// ceil(cdf * 2^31) -- so WHICH traversals are sampled does not depend on launch geometry, pass size or GPU count (the sums of their
// increments do, at rounding level: float64 atomics add in arrival order -- LDS within a workgroup, memory-side into a group table -- so two
// runs agree to ~1e-15 relative per iteration, not bit for bit, and since the rounded regrets feed the next iteration's integer thresholds a
// one-ulp difference can eventually flip a draw: batched MCCFR is reproducible to rounding per run and per grid; the reference-order
// kernels, k_mccfr_replay and k_cfr_exact*, are bit-exact).  Plies 6-7 have one legal action: they
// only resolve the 60 leaf payoffs per task and the visit counts.  The update step then gives one lane per traverser node (26 per
// task): reach and sampling probability rebuilt from the <= 5 ancestor records (the reference's products in the reference's order),
// v as the reference's fma chain over <= 4 leaf payoffs, <= 4 LDS ds_add_f64 into the workgroup's delta table.  Every stage
// gathers what it reads before it stores anything (branch-free loads, slots beyond a node's action count selected away).
// Everything a pair touches is LDS resident (sigma|threshold rows 35 KB, delta 24 KB, 16 x 2 x 1.8 KB wave scratch, tree maps 4 KB,
// lane table 15 KB at 738 infosets; one 1024-thread workgroup per CU).  A workgroup finally adds its non-zero cells to one of 16 GROUP TABLES in HBM
// with memory-side float64 atomics; k_mccfr_apply_groups (one small launch; k_mccfr_exchange_apply for N > 1, which exchanges the
// rows with the peers first, scopa_p2p.h; k_mccfr_fold + k_mccfr_apply on the split path) sums the group tables in table order and
// applies them.  Strategy sums are integer visit counts (sigma is frozen, so strategy_sum += count * sigma).
//
// History (rocprofv3, B = 4096 per traverser, profiles/; full table in profiles/HISTORY.md section 4): v1 one lane per leaf path + global f64
// atomics + one global counter atomic per wavefront: 122 us (100 us of it 8192 same-address atomics); v2 per-workgroup slabs +
// a reduce kernel + traverser-specialised walk: 20-23 us; v3 unique nodes with workgroup-wide plies: same time (latency-bound);
// v4-v7 (unique nodes per wavefront, dense Philox pre-pass, integer thresholds, node records in registers): 13.6 us + 8.3 us of
// slab reduce; v8-v11 (round 2: slabs -> sparse atomics into group tables, lane-per-cell apply kernel, one Philox pass of 31-bit
// draws, gather-then-store stages): 12.2 us + 4.9 us, 15.6 us per iteration; v12-v14 (one record word handed down, update step
// rebuilds its products, lane table, pairs taken from a workgroup counter, two pairs in flight): 15.4 us, and 60.7 us instead of
// 86.4 us at B = 65536; round 4 (lane table read per stage instead of unpacked per wavefront, one draw store per lane, identity record,
// first-visit tracking off once every infoset is marked, a lane per row in the epilogue): 680 -> 520 vector instructions per wavefront
// at one pair per wavefront, 13.3-13.6 -> 12.6-12.9 us per iteration (DESIGN.md section 4 has the attribution by phase).
#include <hip/hip_ext.h>

#include "scopa_ctx.h"
#include "scopa_p2p.h"
#include "scopa_philox.h"

using namespace scopa;

namespace {

// InfoNode.current_strategy, mc_cfr.py:20-24  (np.maximum, ndarray.sum left-to-right, elementwise divide)
// (all loops over the 4 slots are unrolled with a predicate on n: indexing by a run-time n would put the arrays in scratch memory)
__device__ __forceinline__ void mc_sigma(const double *R, int n, double *sigma) {
    double pos[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pos[i] = (i < n && R[i] > 0.0) ? R[i] : 0.0;
    double s = pos[0];
#pragma unroll
    for (int i = 1; i < 4; i++) if (i < n) s += pos[i];
#pragma unroll
    for (int i = 0; i < 4; i++) sigma[i] = i < n ? (s == 0.0 ? 1.0 / (double)n : pos[i] / s) : 0.0;
}

// np.random.choice(legal, p=sigma): cdf = p.cumsum(); cdf /= cdf[-1]; index = cdf.searchsorted(u, 'right') = #{cdf_i <= u}.
// Every u here is k * 2^-31 with an integer k < 2^31 (the top 31 bits of one Philox word), and cdf_i * 2^31 is exact in float64, so
// cdf_i <= u  <=>  ceil(cdf_i * 2^31) <= k: the row stores those integer thresholds (0 .. 2^31 fit a uint32; 2^31 = never) and the
// walk compares 32-bit integers -- the same answer as numpy's float64 compare bit for bit, without conversions per draw.
__device__ __forceinline__ void choice_cdf(const double *sigma, int n, uint32_t *thr) {
    double cdf[4];
    double c = sigma[0];
    cdf[0] = c;
#pragma unroll
    for (int i = 1; i < 4; i++) { if (i < n) c += sigma[i]; cdf[i] = c; }
    const double last = c;   // = cdf[n - 1]
#pragma unroll
    for (int i = 0; i < 4; i++) thr[i] = i < n ? (uint32_t)ceil((cdf[i] / last) * 2147483648.0) : 0xFFFFFFFFu;  // padding never counts
}
// a frozen row as it is kept in HBM and in LDS, [kRow = 6] float64: sigma[4] | thresholds as 4 x uint32 (48 bytes: the traversal's
// prologue copies the rows to LDS as they are; until round 2 the HBM rows were 64 bytes with 16 unused, a quarter of the prologue's row bytes)
__device__ __forceinline__ void row_store(double *__restrict__ row, const double *sg, const uint32_t *thr) {
    double2 *out = reinterpret_cast<double2 *>(row);
    out[0] = make_double2(sg[0], sg[1]);
    out[1] = make_double2(sg[2], sg[3]);
    out[2] = make_double2(__longlong_as_double((long long)(((unsigned long long)thr[1] << 32) | thr[0])),
                          __longlong_as_double((long long)(((unsigned long long)thr[3] << 32) | thr[2])));
}
// the plain float64 form, for the replay kernel whose uniforms come from the host
__device__ __forceinline__ void choice_cdf_f64(const double *sigma, int n, double *cdf) {
    double c = sigma[0];
    cdf[0] = c;
    for (int i = 1; i < n; i++) { c += sigma[i]; cdf[i] = c; }
    const double last = cdf[n - 1];
    for (int i = 0; i < 4; i++) cdf[i] = i < n ? cdf[i] / last : 2.0;  // 2.0 > any u: padding never counts
}

// traverser plies strictly above ply d, and nodes of ply d (d <= 6) in one task's recursion tree
__host__ __device__ constexpr int ntl_at(int trav, int d) { return trav == 0 ? (d + 1) >> 1 : d >> 1; }
__host__ __device__ constexpr int task_nodes(int trav, int d) {
    return ntl_at(trav, d) == 0 ? 1 : ntl_at(trav, d) == 1 ? 5 : ntl_at(trav, d) == 2 ? 20 : 60;
}

constexpr int kRow = 6;   // doubles per frozen row in LDS: sigma[4] | 4 x uint32 thresholds (the last one of a row never counts: cdf[n-1] = 1.0
                           // -> 2^31, padding = ~0).  48-byte rows start at 16 different bank alignments (64-byte rows could only start
                           // at 4: every random-row gather was >= 4-way bank-conflicted, SQ_LDS_BANK_CONFLICT = 47 % of LDS cycles)
constexpr int kUpd = 26;  // traverser nodes with > 1 legal action per task: 1 + 5 + 20
// visits of one traversal pair: the recursion tree has one shape (nodes per ply 1,5,5,20,20,60,60,120 for traverser 0 and
// 1,1,5,5,20,20,60,60 for traverser 1; every ply-7 node has one leaf below it)
constexpr unsigned int kPairDecisionVisits = (1 + 5 + 5 + 20 + 20 + 60 + 60 + 120) + (1 + 1 + 5 + 5 + 20 + 20 + 60 + 60), kPairTerminalVisits = 120 + 120;
static_assert(kPairDecisionVisits == 463, "291 + 172");

}  // namespace

// sigma | cdf rows of the frozen regret table, [n_infosets][kRow] float64, computed once per iteration
__global__ void __launch_bounds__(256)
k_mccfr_prepare(const uint64_t *__restrict__ g_key, const double *__restrict__ g_regret, double *__restrict__ g_sigcdf, int n_infosets) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_infosets) return;
    const int n = (int)((g_key[r] >> 1) & 7);
    double R[4], sg[4];
    uint32_t thr[4];
    for (int c = 0; c < 4; c++) R[c] = g_regret[r * 4 + c];
    mc_sigma(R, n, sg);
    choice_cdf(sg, n, thr);
    row_store(g_sigcdf + r * kRow, sg, thr);
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-wavefront scratch in LDS: the records of one traversal pair (2 tasks) + what the update step needs.
constexpr int kIdentSlot = 166;   // ws.npk[kIdentSlot]: the IDENTITY record (infoset = the all-ones row behind the table, action 0): what an update lane reads
                                  // for the plies at or below its own node -- the factor 1.0 comes out of the same loads as the real factors, no select
struct WaveScratch {
    uint32_t npk[168];       // one record per node of plies 0..5 (2 + 6 + 10 + 25 + 40 nodes, then ply 5's 60 + 20 from slot 86): idx | infoset << 10
                             // | sampled action << 21.  The leaf stage reads ply 5's; the update step rebuilds a traverser node's reach and
                             // sampling probability from its ancestors' records.  [166] = the identity record, [167] unused
    int8_t p6[128];          // resolved leaf payoffs x2 (traverser's sign) per task, 2 x 60 (+ 8 unused)
    uint32_t kdraw[64][4];   // the pair's random draws: lane l's Philox block as it came out of draw_pairs (31-bit integers) -- ONE 16-byte store per lane;
                             // which word a node consumes is the lane table's business (kword)
};
static_assert(sizeof(WaveScratch) % 16 == 0, "WaveScratch must keep 16-byte alignment");

// LDS traffic between lanes of ONE wavefront: DS operations of a wave execute in issue order, so only the compiler
// has to be kept from reordering, and outstanding LDS returns drained, before other lanes' records are read.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS addresses as 32-bit integers: what the lane table below holds (a generic pointer laundered through a register would
// come back as a flat pointer).
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t a) { return *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)a; }
__device__ __forceinline__ int lds_read_i8(uint32_t a) { return *(__attribute__((address_space(3))) const int8_t *)(uintptr_t)a; }
__device__ __forceinline__ double lds_read_f64(uint32_t a) { return *(__attribute__((address_space(3))) const double *)(uintptr_t)a; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a plain vector: HIP's uint4 class has no copy across address spaces
__device__ __forceinline__ uint4 lds_read_u4(uint32_t a) {
    const u32x4 v = *(__attribute__((address_space(3))) const u32x4 *)(uintptr_t)a;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_write_u4(uint32_t a, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)a = u32x4{x, y, z, w};
}

// All random draws of one traversal pair in ONE dense pass: 44 + 14 Philox blocks on 58 lanes (20 64-bit multiplies per wavefront;
// the first form drew one block per ply round at 3-60 % lane use, the second 112 blocks of which half the words went unused).
// Lane < 44: traverser 0's block `lane`; lanes 44..57: traverser 1's block `lane - 44` (lanes 58..63 compute a block nobody reads: cheaper
// than masking them).  Block p of a traverser covers ntl = 0 (p = 0), 1 (p = 1..3), 2 (p = 4..13), 3 (p = 14..43) and the node pair
// j0 = 2 * (p - first block of the level), j0 + 1; its words are x(j0) | y(j0) | x(j0 + 1) | y(j0 + 1), x for the opponent node, y for the
// traverser node below it.  Philox counter = (p, global traversal id, iteration, traverser), key = seed.
constexpr int kLaneSlotVecs = 15;                         // the lane table: [15][64] uint4 (below)
constexpr int kStaticLds = 64 + kLaneSlotVecs * 1024;     // k_mccfr_traverse / k_mccfr_multi: s_vis, s_next, s_slice (+ alignment) and the staged lane table, beside the dynamic LDS
constexpr int kStaticLdsMulti = kStaticLds;

template <int NP>
__device__ __forceinline__ void draw_pairs(uint32_t wsb, int lane, const uint32_t (&b)[NP], uint32_t iteration, uint32_t seed_lo, uint32_t seed_hi) {
    const int trav = lane < 44 ? 0 : 1, p = trav ? lane - 44 : lane;
    philox_out x[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) x[i] = philox4x32_10((uint32_t)p, b[i], iteration, (uint32_t)trav, seed_lo, seed_hi);   // independent chains
#pragma unroll
    for (int i = 0; i < NP; i++)
        lds_write_u4(wsb + (uint32_t)(i * sizeof(WaveScratch) + offsetof(WaveScratch, kdraw)) + 16u * (uint32_t)lane, x[i].x0 >> 1, x[i].x1 >> 1, x[i].x2 >> 1, x[i].x3 >> 1);
    wave_lds_sync();
}

// A node hands ONE word down to its children, kept in the REGISTER of the lane that computed it and fetched by the child with a
// cross-lane read (ds_bpermute: the LDS crossbar, no memory, no bank conflicts): pk = idx | infoset << 10 | sampled action << 21.
// The same word goes to ws.npk for the update step.  (Until round 2 a node also handed down the two float64 products reach and
// sampling probability: four more cross-lane reads, a sigma read and two multiplications on EVERY node of every ply, although only
// the 52 traverser nodes of the update step use them -- 15 LDS operations per ply; now 7.)
struct NodeRegs {
    uint32_t pk;
};

// where ply d's node records start in ws.npk, and the slot of node (traverser, j) of ply d
__host__ __device__ constexpr int npk_offset(int d) { return d == 0 ? 0 : d == 1 ? 2 : d == 2 ? 8 : d == 3 ? 18 : d == 4 ? 43 : 86; }
__host__ __device__ constexpr int npk_slot(int trav, int d, int j) { return npk_offset(d) + (trav ? task_nodes(0, d) : 0) + j; }
static_assert(npk_offset(1) == task_nodes(0, 0) + task_nodes(1, 0) && npk_offset(2) == npk_offset(1) + task_nodes(0, 1) + task_nodes(1, 1) &&
              npk_offset(3) == npk_offset(2) + task_nodes(0, 2) + task_nodes(1, 2) && npk_offset(4) == npk_offset(3) + task_nodes(0, 3) + task_nodes(1, 3) &&
              npk_offset(4) + task_nodes(0, 4) + task_nodes(1, 4) <= npk_offset(5) && npk_offset(5) + task_nodes(0, 5) + task_nodes(1, 5) == kIdentSlot,
              "record slots of plies 0..5 tile WaveScratch::npk");

// THE LANE TABLE.  Everything about a lane's node SLOT that does not depend on the pair being walked -- which lane holds the parent, whether the
// action towards the node is forced by a re-expansion, where its draw word lies, which records an update lane's ancestors left -- is the same for
// every wavefront of every launch: k_lane_table computes it ONCE per context into a 15 KB table, a workgroup stages it into LDS with its prologue,
// and a stage READS ITS SLOT'S 16 BYTES WHEN IT RUNS (one ds_read_b128 per ply round).  History: derived in the kernel it was ~300 VALU instructions
// per wavefront before the first pair; round 2's table was unpacked into 38 registers per lane once per wavefront (round 3: 156 VALU + 66 SALU
// instructions behind the prologue's barrier, 17 scalar registers spilled) -- at the headline batch a wavefront walks ONE pair, so "once per wavefront"
// was once per pair, and the update step still decoded its ancestor row with ~45 instructions per pair.  Offsets are relative to the wave's scratch.
//   vec S = 0..6   ply D = S (S = 6: ply 5's second round, 80 nodes): { 4 x parent lane (ds_bpermute address), amask, aforce, draw word offset }
//                  action towards this node = (parent's sampled action & amask) | aforce; lanes without a node read lane 0 / word 0 and are masked off
//   vec 7, 8       leaves lane, lane + 64 of the 120: { offset of the ply-5 ancestor's record, amask, aforce, - }
//   vec 9..14      the update lane x (0..51: traverser x / 26, level m, node j):
//                  9: record offsets of the ancestors at plies 0..3 | 10: ply 4's, the node's own record, its action count nX, - | 11: p6 offsets of the
//                  leaf values of actions 0..3 (beyond nX: the last one's) | 12: am24 of plies 0..3 | 13: am24 of ply 4, af8 of plies 0..2 | 14: af8 of plies 3, 4
//                  factor towards the node at ply q = sigma[record's infoset][((record >> 18) & am24) | af8 : 8 x the action]; plies at or below the node's
//                  own read the identity record (infoset = the all-ones row, action 0): factor 1.0
constexpr int kLvLeaf = 7, kLvUpd = 9;
struct LaneTabRow { uint4 v[kLaneSlotVecs]; };

template <int D, int ROUND>
__device__ __forceinline__ void lane_slot_build(LaneTabRow &row, int lane) {
    constexpr int S = D + ROUND;
    constexpr int c0 = task_nodes(0, D), c1 = task_nodes(1, D);
    const int t = lane + 64 * ROUND;
    const bool valid = t < c0 + c1;
    const int trav = t < c0 ? 0 : 1;
    const int j = valid ? (trav ? t - c0 : t) : 0;
    const bool is_trav = (D & 1) == trav;
    const int ntl = trav == 0 ? (D + 1) >> 1 : D >> 1;
    uint32_t plane4 = 0u, amask = valid ? ~0u : 0u, aforce = 0u;
    if constexpr (D > 0) {
        constexpr int pd = D > 0 ? D - 1 : 0, pn = 4 - (pd >> 1);
        const bool p_trav = (pd & 1) == trav;
        int pj = j, br = 0;
        if (p_trav) { pj = j / (pn + 1); br = j - pj * (pn + 1); }
        if (valid) plane4 = 4u * (uint32_t)((trav ? task_nodes(0, pd) : 0) + pj);
        if (valid && p_trav && br > 0) { amask = 0u; aforce = (uint32_t)(br - 1); }
    }
    // this node's draw: word (j & 1) * 2 + (traverser node ? 1 : 0) of block first(ntl) + j / 2 of its traverser, i.e. of lane (traverser ? 44 : 0) + block
    const int blk = (ntl == 0 ? 0 : ntl == 1 ? 1 : ntl == 2 ? 4 : 14) + (j >> 1);
    const uint32_t kword = valid ? (uint32_t)(offsetof(WaveScratch, kdraw) + 16 * ((trav ? 44 : 0) + blk) + 4 * ((j & 1) * 2 + (is_trav ? 1 : 0)))
                                 : (uint32_t)offsetof(WaveScratch, kdraw);
    row.v[S] = make_uint4(plane4, amask, aforce, kword);
}

// the update lane x's row (vecs 9..14)
__device__ __forceinline__ void upd_build(LaneTabRow &row, int x) {
    const uint32_t ident = (uint32_t)(offsetof(WaveScratch, npk) + 4 * kIdentSlot);
    uint32_t rec[5] = {ident, ident, ident, ident, ident}, am[5] = {24u, 24u, 24u, 24u, 24u}, af[5] = {0u, 0u, 0u, 0u, 0u};
    uint32_t own = ident, nX = 1u, p6o[4] = {(uint32_t)offsetof(WaveScratch, p6), (uint32_t)offsetof(WaveScratch, p6), (uint32_t)offsetof(WaveScratch, p6), (uint32_t)offsetof(WaveScratch, p6)};
    if (x < 2 * kUpd) {
        const int trav = x < kUpd ? 0 : 1, xx = trav ? x - kUpd : x;
        const int m = xx == 0 ? 0 : xx < 6 ? 1 : 2;
        int j = xx - (m == 0 ? 0 : m == 1 ? 1 : 6);
        const int d = 2 * m + trav;
        nX = (uint32_t)(4 - m);
        const int stride = m == 0 ? 12 : m == 1 ? 3 : 1;   // leaf group of (prefix, i + 1, 0, ...) = base + (i + 1) * stride
        const int base = m == 0 ? 0 : m == 1 ? j * 12 : j * 3;
        for (int c = 0; c < 4; c++) p6o[c] = (uint32_t)(offsetof(WaveScratch, p6) + trav * 60 + base + ((c < (int)nX ? c : (int)nX - 1) + 1) * stride);
        own = (uint32_t)(offsetof(WaveScratch, npk) + 4 * npk_slot(trav, d, j));
        for (int c = d; c > 0; c--) {                       // climb: node j of ply c -> its parent at ply c - 1
            const int pd = c - 1, pn = 4 - (pd >> 1);
            int forced = 0;
            if ((pd & 1) == trav) { const int pj = j / (pn + 1), br = j - pj * (pn + 1); forced = br; j = pj; }   // br = 0: the sampled child
            rec[pd] = (uint32_t)(offsetof(WaveScratch, npk) + 4 * npk_slot(trav, pd, j));
            if (forced) { am[pd] = 0u; af[pd] = 8u * (uint32_t)(forced - 1); }
        }
    }
    row.v[kLvUpd + 0] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    row.v[kLvUpd + 1] = make_uint4(rec[4], own, nX, 0u);
    row.v[kLvUpd + 2] = make_uint4(p6o[0], p6o[1], p6o[2], p6o[3]);
    row.v[kLvUpd + 3] = make_uint4(am[0], am[1], am[2], am[3]);
    row.v[kLvUpd + 4] = make_uint4(am[4], af[0], af[1], af[2]);
    row.v[kLvUpd + 5] = make_uint4(af[3], af[4], 0u, 0u);
}

// what the walk needs beside the pair: LDS base addresses (32-bit) and the tree maps
struct WalkEnv {
    uint32_t wsb;          // this wavefront's scratch (the first of its one or two WaveScratch)
    uint32_t tab;          // the staged lane table + 16 x lane: vec S of this lane at tab + 1024 S
    const uint16_t *s_inf;
    const int8_t *s_pay;
    const double *s_sigcdf;
    double *s_dR;
    uint8_t *s_seen;       // first-visit tracking (the dict views' insertion of keys): nullptr once every infoset of the deal has been seen
    unsigned int *s_cnt;
};

// One ply of NP traversal pairs (one or two: the pairs' dependent chains -- parent record, infoset, thresholds -- interleave).
template <int D, int NP>
__device__ __forceinline__ void ply_step(const WalkEnv &e, int lane, NodeRegs (&st)[NP]) {
    constexpr int n = 4 - (D >> 1);
    constexpr int c0 = task_nodes(0, D), c1 = task_nodes(1, D);
    constexpr int NR = c0 + c1 > 64 ? 2 : 1;   // ply 5: 80 nodes, two rounds
    // what a lane gathers for its node before anything is stored: every LDS READ of the ply (all rounds, all pairs) is issued before
    // the first LDS write or atomic, and none sits under a branch -- lanes without a node read slot 0 and are ignored afterwards
    struct Pre {
        int idx, In;
        uint32_t k, thr0, thr1, thr2;
    };
    Pre g[NP][NR];
    uint4 sl[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) sl[r] = lds_read_u4(e.tab + 1024u * (uint32_t)(D + r));
#pragma unroll
    for (int i = 0; i < NP; i++)
#pragma unroll
        for (int r = 0; r < NR; r++) {
            Pre &q = g[i][r];
            q.idx = 0;
            if constexpr (D > 0) {  // the parent's hand-down: executed by ALL lanes (a cross-lane read wants its source lane enabled)
                constexpr int pn = 4 - ((D > 0 ? D - 1 : 0) >> 1);
                const uint32_t ppk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)sl[r].x, (int)st[i].pk);
                const uint32_t act = ((ppk >> 21) & sl[r].y) | sl[r].z;
                q.idx = (int)((ppk & 1023u) * pn + act);
            }
            q.In = e.s_inf[level_offset(D) + q.idx];
            q.k = lds_read_u32(e.wsb + sl[r].w + (uint32_t)(i * sizeof(WaveScratch)));
            // only the first n-1 thresholds can count (those from n-1 on are >= 2^31 > k): one LDS read at the n = 2 plies, where
            // most nodes are, instead of three
            const uint32_t *thr = reinterpret_cast<const uint32_t *>(e.s_sigcdf + q.In * kRow + 4);
            q.thr0 = thr[0];
            q.thr1 = n > 2 ? thr[1] : 0xFFFFFFFFu;
            q.thr2 = n > 3 ? thr[2] : 0xFFFFFFFFu;
        }
#pragma unroll
    for (int i = 0; i < NP; i++)
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const Pre &q = g[i][r];
            const int t = 64 * r + lane;
            if (t < c0 + c1) {
                const int a = (q.thr0 <= q.k) + (q.thr1 <= q.k) + (q.thr2 <= q.k);
                if ((D & 1) == (t < c0 ? 0 : 1)) atomicAdd(&e.s_cnt[q.In], 1u);   // strategy_sum += sigma per traverser visit (mc_cfr.py:84); also marks the infoset seen
                else if (e.s_seen) e.s_seen[q.In] = 1;                          // benign race: every writer stores 1
                const uint32_t pk = (uint32_t)q.idx | ((uint32_t)q.In << 10) | ((uint32_t)a << 21);
                *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(e.wsb + (uint32_t)(i * sizeof(WaveScratch) + offsetof(WaveScratch, npk) + 4 * npk_offset(D)) + 4u * (uint32_t)t) = pk;
                if constexpr (D < 5) st[i].pk = pk;
            }
        }
    if constexpr (D == 5) wave_lds_sync();
}

#ifdef SCOPA_WALK_STAMPS   // development build only: shader-clock stamps of wavefront 0's stages, summed over its pairs (tests/tools/walk_stamps.py)
__device__ unsigned long long g_walk_stamps[16];
__device__ unsigned long long g_wave_clocks[3 * 16];   // whole pair loop per wavefront, workgroups 0, 100, 255; [2*16..] = wave's hardware id
__device__ unsigned long long g_wave_phases[4 * 16];   // workgroup 0, per wavefront: entry -> prologue barrier | -> first pair | pair loop | -> kernel end
#define WALK_STAMP(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) { g_walk_stamps[i] += now_ - t_prev_; } t_prev_ = now_; } while (0)
#else
#define WALK_STAMP(i) do { } while (0)
#endif

// NP traversal pairs (both traversers of global traversal ids b[0..NP)) on one wavefront: plies 0..5 level by level, plies 6-7, then
// the regret update.  Regret increments go to `s_dR` with LDS float64 atomics, traverser visits to `s_cnt`.
// A pair is ONE dependent chain -- about 26 LDS round trips and 300 vector instructions, 7 400 shader clocks measured, whatever the
// load on the CU (tests/tools/walk_stamps.py): with 16 wavefronts per CU neither the LDS array nor the SIMDs were more than 70 % busy.
// A wavefront that has two or more pairs to walk therefore takes them two at a time: every stage gathers for both, then stores for both.
template <int NP>
__device__ __forceinline__ void walk_pairs(const WalkEnv &e, int lane, const uint32_t (&b)[NP], uint32_t iteration, uint32_t seed_lo,
                                           uint32_t seed_hi, unsigned int &my_pairs) {
#ifdef SCOPA_WALK_STAMPS
    const unsigned long long w_start_ = wall_clock64();
    unsigned long long t_prev_ = clock64();
#endif
    draw_pairs<NP>(e.wsb, lane, b, iteration, seed_lo, seed_hi);
    WALK_STAMP(0);
    // plies 0..5: one lane per unique node of a pair's two recursion trees (ply constants are compile-time); a ply's nodes
    // stay in their lanes' registers for the next ply to fetch
    NodeRegs st[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) st[i].pk = 0u;
    ply_step<0, NP>(e, lane, st); WALK_STAMP(1);
    ply_step<1, NP>(e, lane, st); WALK_STAMP(2);
    ply_step<2, NP>(e, lane, st); WALK_STAMP(3);
    ply_step<3, NP>(e, lane, st); WALK_STAMP(4);
    ply_step<4, NP>(e, lane, st); WALK_STAMP(5);
    ply_step<5, NP>(e, lane, st); WALK_STAMP(6);
    my_pairs += NP;   // wave-uniform; the recursion tree has one shape: kPairDecisionVisits + kPairTerminalVisits per pair
    // plies 6-7 (one legal action each): leaf payoffs, seen flags, visit counts.  120 leaves on 64 lanes: all of a lane's items are
    // loaded before any is used (two dependent LDS round trips for the stage)
    {
        uint4 lf[2];
#pragma unroll
        for (int q = 0; q < 2; q++) lf[q] = lds_read_u4(e.tab + 1024u * (uint32_t)(kLvLeaf + q));
        uint32_t ppk[NP][2];
#pragma unroll
        for (int i = 0; i < NP; i++)
#pragma unroll
            for (int q = 0; q < 2; q++) ppk[i][q] = lds_read_u32(e.wsb + lf[q].x + (uint32_t)(i * sizeof(WaveScratch)));
        int I6[NP][2], I7[NP][2], pay[NP][2];
#pragma unroll
        for (int i = 0; i < NP; i++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t act = ((ppk[i][q] >> 21) & lf[q].y) | lf[q].z;
                const int idx6 = (int)((ppk[i][q] & 1023u) * 2 + act);   // = index of the ply-7 node and of the leaf as well
                I6[i][q] = e.s_inf[level_offset(6) + idx6];
                I7[i][q] = e.s_inf[level_offset(7) + idx6];
                pay[i][q] = e.s_pay[idx6];
            }
#pragma unroll
        for (int i = 0; i < NP; i++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int t = lane + 64 * q;
                if (t < 2 * 60) {
                    const int trv = t < 60 ? 0 : 1;
                    if (e.s_seen) e.s_seen[trv == 0 ? I7[i][q] : I6[i][q]] = 1;  // the opponent's node of the two (the traverser's is marked by its count)
                    atomicAdd(&e.s_cnt[trv == 0 ? I6[i][q] : I7[i][q]], 1u);     // the traverser's single-action node: strategy_sum += [1.0]
                    *(__attribute__((address_space(3))) int8_t *)(uintptr_t)(e.wsb + (uint32_t)(i * sizeof(WaveScratch) + offsetof(WaveScratch, p6)) + (uint32_t)t) =
                        (int8_t)(trv == 0 ? pay[i][q] : -pay[i][q]);
                }
            }
    }
    wave_lds_sync();
    WALK_STAMP(7);
    // update: one lane per traverser node with > 1 action (mc_cfr.py:79-84).  The node's opponent reach and own sampling probability
    // are rebuilt here from its ancestors' records -- the product, root first, of sigma[ancestor infoset][action towards the node] over
    // the opponent's / the traverser's plies above it: the same factors in the same order as the reference's top-down updates
    // (:58-65, :75-76).  Everything is loaded before anything is used; which records, which leaf values and which forced actions come from
    // the lane table (vecs 9..14).
    if (lane < 2 * kUpd) {
        const bool trav1 = lane >= kUpd;
        uint4 u[6];
#pragma unroll
        for (int q = 0; q < 6; q++) u[q] = lds_read_u4(e.tab + 1024u * (uint32_t)(kLvUpd + q));
        const uint32_t ro[5] = {u[0].x, u[0].y, u[0].z, u[0].w, u[1].x}, own = u[1].y;
        const int nX = (int)u[1].z;
        const uint32_t p6o[4] = {u[2].x, u[2].y, u[2].z, u[2].w};
        const uint32_t am[5] = {u[3].x, u[3].y, u[3].z, u[3].w, u[4].x}, af[5] = {u[4].y, u[4].z, u[4].w, u[5].x, u[5].y};
        uint32_t rec[NP][5];
        int IX[NP], pv[NP][4];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint32_t wi = e.wsb + (uint32_t)(i * sizeof(WaveScratch));
#pragma unroll
            for (int q = 0; q < 5; q++) rec[i][q] = lds_read_u32(wi + ro[q]);
            IX[i] = (int)((lds_read_u32(wi + own) >> 10) & 2047u);
#pragma unroll
            for (int c = 0; c < 4; c++) pv[i][c] = lds_read_i8(wi + p6o[c]);
        }
        double fq[NP][5], sg[NP][4];
        const uint32_t sig = lds_addr(e.s_sigcdf);
#pragma unroll
        for (int i = 0; i < NP; i++) {
#pragma unroll
            for (int q = 0; q < 5; q++)
                fq[i][q] = lds_read_f64(sig + ((rec[i][q] >> 10) & 2047u) * (uint32_t)(kRow * 8) + (((rec[i][q] >> 18) & am[q]) | af[q]));
            const double2 s01 = *reinterpret_cast<const double2 *>(e.s_sigcdf + IX[i] * kRow), s23 = *reinterpret_cast<const double2 *>(e.s_sigcdf + IX[i] * kRow + 2);
            sg[i][0] = s01.x; sg[i][1] = s01.y; sg[i][2] = s23.x; sg[i][3] = s23.y;
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            // even plies are traverser 0's, odd plies traverser 1's: two chains, root first (1.0 * x = x, x * 1.0 = x: the reference's products)
            const double even = (fq[i][0] * fq[i][2]) * fq[i][4], odd = fq[i][1] * fq[i][3];
            const double sX = trav1 ? odd : even, rX = trav1 ? even : odd;
            const double w = sX > 0.0 ? rX / sX : 0.0;       // weight = opp_reach / sampling_probs[player] if > 0 else 0
            double cfv[4], v = 0.0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                cfv[c] = 0.5 * (double)pv[i][c];
                const double t = fma(sg[i][c], cfv[c], v);       // np.dot on this numpy build: an fma chain (see oracle)
                v = c < nX ? t : v;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double delta = w * (cfv[c] - v);
                if (c < nX && delta != 0.0) atomicAdd(&e.s_dR[IX[i] * 4 + c], delta);
            }
        }
    }
    wave_lds_sync();  // the next pairs overwrite this wave's scratch
    WALK_STAMP(8);
#ifdef SCOPA_WALK_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_walk_stamps[15] += NP; g_walk_stamps[14] += wall_clock64() - w_start_; }
#endif
}

// the lane table of a context (one launch of 64 threads, at the first traversal launch): [kLaneSlotVecs][64] uint4
__global__ void __launch_bounds__(64) k_lane_table(uint4 *__restrict__ g_tab) {
    const int lane = threadIdx.x;
    LaneTabRow row;
    for (int i = 0; i < kLaneSlotVecs; i++) row.v[i] = make_uint4(0u, 0u, 0u, 0u);
    lane_slot_build<0, 0>(row, lane);
    lane_slot_build<1, 0>(row, lane);
    lane_slot_build<2, 0>(row, lane);
    lane_slot_build<3, 0>(row, lane);
    lane_slot_build<4, 0>(row, lane);
    lane_slot_build<5, 0>(row, lane);
    lane_slot_build<5, 1>(row, lane);
    for (int q = 0; q < 2; q++) {
        const int t = lane + 64 * q;
        const bool on = t < 2 * 60;
        const int tt = on ? t : 0, trv = tt < 60 ? 0 : 1, j = trv ? tt - 60 : tt;
        // ply-5 ancestor: traverser 0 -> opponent node (one child); traverser 1 -> traverser node (3 branches)
        int pj = j, kk = 0;
        if (trv == 1) { pj = j / 3; kk = j - pj * 3; }
        row.v[kLvLeaf + q] = make_uint4((uint32_t)(offsetof(WaveScratch, npk) + 4 * (npk_offset(5) + (trv ? 60 : 0) + pj)),
                                        (trv == 1 && kk > 0) ? 0u : ~0u, (trv == 1 && kk > 0) ? (uint32_t)(kk - 1) : 0u, 0u);
    }
    upd_build(row, lane);
    for (int i = 0; i < kLaneSlotVecs; i++) g_tab[i * 64 + lane] = row.v[i];
}

// a wavefront's part of the walk's set-up: its scratch's identity record(s) (lane 0; made visible to the wave's other lanes by the first
// wave_lds_sync of walk_pairs, long before the update step reads it)
__device__ __forceinline__ void wave_scratch_init(WaveScratch *ws, int n_scratch, int lane, int n_infosets) {
    if (lane == 0)
        for (int i = 0; i < n_scratch; i++) ws[i].npk[kIdentSlot] = (uint32_t)n_infosets << 10;
}

// which group table a workgroup adds into: neighbours in dispatch order -- eight consecutive workgroups, one per XCD -- share a table
// (b / 8 mod 16: 13.02 us per iteration at B = 4096; b mod 16, which gives every table to ONE XCD: 13.16 -- the adds execute at the
// memory side either way)
__device__ __forceinline__ int group_of(unsigned int b) { return (int)((b >> 3) % kDeltaGroups); }

// Where a launch leaves its result.  A workgroup's partial delta table (regret increments + traverser-visit counts, LDS) is added
// to a GROUP table in HBM with memory-side float64 atomics: workgroup b adds into table group_of(b).  A group table is
// [5][kGroupRows] float64 -- cell-major: dR0 of every infoset, dR1, dR2, dR3, counts -- so that a lane per cell adds, and a lane per
// cell later reads, with consecutive lanes on consecutive addresses.  Only non-zero cells are added.  Why groups: such an atomic
// executes at the memory side, one 64-byte line request at a time per line, and with peaked strategies every workgroup touches the
// same ~100 rows -- 256 workgroups on one table would queue 256 deep on every hot line (+2.5 .. 5 us at the end of the launch,
// benchmarks/micro/atomic_flush.hip).  How many tables: measured on the kernel itself, 4 / 8 / 16 / 24 / 32 / 64 tables -> 16.85 /
// 14.65 / 13.5 / 13.3 / 14.1 / 15.2 us per iteration at B = 4096 (more tables: shorter queues, but the apply launch reads them all);
// permuting rows so that hot rows fall on different lines is slower (a wavefront's adds then touch more lines: requests are what
// costs).  k_mccfr_apply_groups (one small launch) then sums the tables in table order, applies the sum and prepares the next
// iteration's rows.  This replaced per-workgroup SLABS (256 x 29.5 KB written per launch, read back by a reduce kernel that took
// 8.3 us of a 22 us iteration).
__global__ void __launch_bounds__(1024)
k_mccfr_traverse(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff,
                 const double *__restrict__ g_sigcdf, double *__restrict__ g_groups,
                 int n_infosets, uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t b0, uint32_t nb,
                 unsigned long long *__restrict__ g_wg_counts, uint32_t *__restrict__ g_visit, unsigned long long *__restrict__ g_clock,
                 const uint4 *__restrict__ g_lane_tab, const uint32_t *__restrict__ g_iter, uint32_t track_seen) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned int s_vis[2];
    if (g_iter) iteration = *g_iter;   // graph-captured iteration loops (scopa_mccfr_graph_mode): the iteration number lives in a device word that
                                       // the apply kernel advances, so that one captured graph serves every replay; draws stay keyed by it
    __shared__ uint32_t s_next[1];   // next pair of this workgroup not taken yet
    __shared__ uint32_t s_slice[1];  // wavefronts that have left the pair loop
    __shared__ uint4 s_lane_tab[kLaneSlotVecs * 64];   // the lane table, staged once per workgroup; a stage reads its slot's 16 bytes when it runs
    const unsigned long long t_start = wall_clock64();   // 100 MHz device-wide clock: this workgroup's phase stamps (sampled launches only)
#ifdef SCOPA_WALK_STAMPS
    const unsigned long long c_entry_ = clock64();
#endif
    const int I = n_infosets;
    double *s_sigcdf = reinterpret_cast<double *>(smem);                         // [I + 1][kRow]: sigma[4] | 4 x uint32 thresholds (48-byte rows); row I = all ones (the identity record's)
    double *s_dR = s_sigcdf + (size_t)(I + 1) * kRow;                            // [I][4] (16-byte aligned)
    WaveScratch *s_wave = reinterpret_cast<WaveScratch *>(s_dR + (size_t)I * 4); // [wavefronts of this workgroup][2 pairs in flight]
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_wave + 2 * (blockDim.x >> 6));  // [I] traverser visits
    uint16_t *s_inf = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(s_cnt) + (((size_t)I * 4 + 15) & ~(size_t)15));  // [1653] (+pad)
    int8_t *s_pay = reinterpret_cast<int8_t *>(s_inf + 1656);                    // [576]
    uint8_t *s_seen = reinterpret_cast<uint8_t *>(s_pay + kTerminal);            // [I]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6, nthr = blockDim.x;
    if (tid < 2) s_vis[tid] = 0u;
    const uint4 lane_piece = tid < kLaneSlotVecs * 64 ? g_lane_tab[tid] : make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) {
        s_slice[0] = 0u;
        const uint32_t per_wg = (nb + gridDim.x - 1) / gridDim.x, W = blockDim.x >> 6;
        const uint32_t first = blockIdx.x * per_wg, count = first < nb ? (nb - first < per_wg ? nb - first : per_wg) : 0u;
        s_next[0] = count >= 2 * W ? 2 * W : W;   // behind the static first takes (main loop)
    }
    // ---- prologue: this iteration's frozen strategy rows and the tree maps into LDS ----------------------------------
    // All global loads are issued before anything waits on one of them (one memory round trip for the whole prologue),
    // the LDS zeroing runs underneath them, then the loaded pieces are stored.
    const uint32_t *gi = reinterpret_cast<const uint32_t *>(g_infoset), *gp = reinterpret_cast<const uint32_t *>(g_payoff);
    {
        constexpr int kSig = 5;                                                   // 16-byte pieces per thread held in registers: covers I <= 1653 at 1024 threads
        const double2 *g2 = reinterpret_cast<const double2 *>(g_sigcdf);          // the rows have the LDS layout: a straight copy of 3 I pieces
        double2 *s2 = reinterpret_cast<double2 *>(s_sigcdf);
        double2 v[kSig];
#pragma unroll
        for (int j = 0; j < kSig; j++) {
            const int idx = tid + j * nthr;
            v[j] = idx < I * 3 ? g2[idx] : make_double2(0.0, 0.0);
        }
        // node -> infoset map: 1653 u16 = 826 words + one half word; leaf payoffs: 576 bytes = 144 words (both arrays 4-byte aligned)
        const uint32_t wi = tid < kDecision / 2 ? gi[tid] : 0u;
        const uint32_t wp = tid < kTerminal / 4 ? gp[tid] : 0u;
        const uint16_t last_inf = g_infoset[kDecision - 1];
        for (int i = tid; i < I * 2; i += nthr) reinterpret_cast<double2 *>(s_dR)[i] = make_double2(0.0, 0.0);
        for (int r = tid; r < I; r += nthr) s_cnt[r] = 0u;
        if (track_seen) for (int r = tid; r < I; r += nthr) s_seen[r] = 0;
        if (tid < 4) s_sigcdf[(size_t)I * kRow + tid] = 1.0;                      // the identity row
#pragma unroll
        for (int j = 0; j < kSig; j++) {
            const int idx = tid + j * nthr;
            if (idx < I * 3) s2[idx] = v[j];
        }
        for (int idx = tid + kSig * nthr; idx < I * 3; idx += nthr) s2[idx] = g2[idx];  // narrower workgroups (many infosets): the rest, plainly
        if (tid < kLaneSlotVecs * 64) s_lane_tab[tid] = lane_piece;
        if (tid < kDecision / 2) reinterpret_cast<uint32_t *>(s_inf)[tid] = wi;
        if (tid == 0) s_inf[kDecision - 1] = last_inf;
        if (tid < kTerminal / 4) reinterpret_cast<uint32_t *>(s_pay)[tid] = wp;
    }
    for (int i = tid + nthr; i < kLaneSlotVecs * 64; i += nthr) s_lane_tab[i] = g_lane_tab[i];   // workgroups narrower than 960 threads (deals with many infosets): the rest of the table
    for (int i = tid + nthr; i < kDecision / 2; i += nthr) reinterpret_cast<uint32_t *>(s_inf)[i] = gi[i];
    for (int i = tid + nthr; i < kTerminal / 4; i += nthr) reinterpret_cast<uint32_t *>(s_pay)[i] = gp[i];
    __syncthreads();
    const unsigned long long t_pro = wall_clock64();
#ifdef SCOPA_WALK_STAMPS
    const unsigned long long c_barrier_ = clock64();
    unsigned long long c_loop0_ = 0, c_loop1_ = 0;
#endif

    WaveScratch *ws = s_wave + 2 * wave;
    wave_scratch_init(ws, 2, lane, I);
    WalkEnv env;
    env.wsb = lds_addr(ws); env.tab = lds_addr(s_lane_tab) + 16u * (uint32_t)lane;
    env.s_inf = s_inf; env.s_pay = s_pay; env.s_sigcdf = s_sigcdf; env.s_dR = s_dR; env.s_seen = track_seen ? s_seen : nullptr; env.s_cnt = s_cnt;
    unsigned int my_pairs = 0;   // pairs this wavefront walked
    // ---- main loop: every WAVEFRONT walks its own traversal pairs, no workgroup barrier inside --------------------------
    // The workgroup owns pairs [first, first + count); its wavefronts TAKE them from a counter in LDS instead of owning a fixed share:
    // the SIMD's issue arbiter favours its oldest wavefront, so with equal shares the first wave of each SIMD finished its 16 pairs
    // (B = 65 536) in 79 k clocks and the fourth in 130 k, and the workgroup waited for the fourth (tests/tools/walk_stamps.py).
    // Which wavefront walks a pair does not matter: its draws are keyed by the pair's global id.
    {
#ifdef SCOPA_WALK_STAMPS
        const unsigned long long loop_start_ = clock64();
        c_loop0_ = loop_start_;
#endif
        const uint32_t per_wg = (nb + gridDim.x - 1) / gridDim.x;
        const uint32_t first = blockIdx.x * per_wg, count = first < nb ? (nb - first < per_wg ? nb - first : per_wg) : 0u;
        const uint32_t W = (uint32_t)n_waves;
        // the first take needs no counter: two pairs per wavefront if the workgroup has that many, else one (s_next starts behind them)
        const bool start_two = count >= 2 * W;
        uint32_t c = start_two ? 2 * (uint32_t)wave : (uint32_t)wave, g = start_two ? 2u : 1u;
        for (;;) {
            if (g == 2 && c + 1 < count) {      // two pairs in flight
                const uint32_t two[2] = {b0 + first + c, b0 + first + c + 1};
                walk_pairs<2>(env, lane, two, iteration, seed_lo, seed_hi, my_pairs);
            } else if (c < count) {
                const uint32_t one[1] = {b0 + first + c};
                walk_pairs<1>(env, lane, one, iteration, seed_lo, seed_hi, my_pairs);
            }
            if (count <= W || c + g >= count) break;   // nothing was left behind the static takes / the counter has run out
            g = count - c > 4 * W ? 2u : 1u;             // single pairs towards the end: the last take bounds the imbalance
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(s_next, g);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            if (c >= count) break;
        }
#ifdef SCOPA_WALK_STAMPS
        c_loop1_ = clock64();
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) g_wave_clocks[(blockIdx.x == 0 ? 0 : 16) + wave] += c_loop1_ - loop_start_;
        if (lane == 0 && blockIdx.x == 0) { unsigned int hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_wave_clocks[32 + wave] = hw; }
#endif
    }
    // A wavefront that has left the pair loop would wait at the barrier for the slowest one (up to 2 us at one pair per wavefront, where
    // the SIMD's arbiter decides who finishes when): meanwhile it moves a slice of the delta table to the group table -- exchange with
    // zero in LDS, add what was there; later additions to those rows by wavefronts still walking go out with the final pass.  Coldest rows
    // (highest ids) first; the last quarter of the finishers skips it (their slices would only delay the barrier).  Measured at B = 4096:
    // 13.5 -> 13.2 us per iteration (all 16 wavefronts pre-flushing: 13.9; the first 8 only: 13.7; hottest rows first: 13.3).
    {
        uint32_t sl = 0;
        if (lane == 0) sl = atomicAdd(s_slice, 1u);
        sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl);
        const int W = n_waves;
        if ((int)sl < W - W / 4) {
            const int per = (I + W - 1) / W, r0 = (W - 1 - (int)sl) * per, r1 = r0 + per < I ? r0 + per : I;
            double *tab = g_groups + (size_t)group_of(blockIdx.x) * kDeltaTable;
            for (int r = r0 + lane; r < r1; r += 64) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double v = __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long *>(&s_dR[r * 4 + k]), 0ull));
                    if (v != 0.0) atomicAdd(&tab[group_cell(k, r)], v);
                }
                const unsigned int c = atomicExch(&s_cnt[r], 0u);
                if (c != 0u) { atomicAdd(&tab[group_cell(4, r)], (double)c); if (track_seen) s_seen[r] = 1; }
            }
        }
    }
    __syncthreads();

    // ---- epilogue: this workgroup's non-zero cells -> its group table (one lane per cell: a wavefront covers 512 contiguous
    // bytes; float64 atomics execute at the memory side and return nothing) ------------------------------------------------------
    const unsigned long long t_walk = wall_clock64();
    {
        // infosets first seen by this launch (only while the deal still has unseen ones: track_seen): the loads go out first, their answers are used
        // after the atomics have been issued
        const bool first0 = track_seen && tid < I && (s_seen[tid] || s_cnt[tid]) && g_visit[tid] == 0u;
        const bool first1 = track_seen && tid + nthr < I && (s_seen[tid + nthr] || s_cnt[tid + nthr]) && g_visit[tid + nthr] == 0u;
        double *tab = g_groups + (size_t)group_of(blockIdx.x) * kDeltaTable;
        // a lane per ROW: its four increments in two 16-byte LDS reads, its count in a third; the five cells of consecutive rows are consecutive
        // addresses of the cell-major group table either way (rounds 2-3 made five passes with a lane per cell: the same step time within the noise
        // at 4096 traversals, 0.3 us slower at 65 536, 30 more vector instructions)
        for (int r = tid; r < I; r += nthr) {
            const double2 a = reinterpret_cast<const double2 *>(s_dR)[r * 2], b = reinterpret_cast<const double2 *>(s_dR)[r * 2 + 1];
            const unsigned int c = s_cnt[r];
            if (a.x != 0.0) atomicAdd(&tab[group_cell(0, r)], a.x);
            if (a.y != 0.0) atomicAdd(&tab[group_cell(1, r)], a.y);
            if (b.x != 0.0) atomicAdd(&tab[group_cell(2, r)], b.x);
            if (b.y != 0.0) atomicAdd(&tab[group_cell(3, r)], b.y);
            if (c != 0u) atomicAdd(&tab[group_cell(4, r)], (double)c);
        }
        if (first0) g_visit[tid] = 0x40000000u + (uint32_t)tid;              // racing writers store the same value
        if (first1) g_visit[tid + nthr] = 0x40000000u + (uint32_t)(tid + nthr);
        if (track_seen)
            for (int r = tid + 2 * nthr; r < I; r += nthr)                   // narrow workgroups (many infosets): the rest, plainly
                if ((s_seen[r] || s_cnt[r]) && g_visit[r] == 0u) g_visit[r] = 0x40000000u + (uint32_t)r;
    }
    // exact visit counters: pairs walked per wavefront x the visits of a pair -> LDS -> this workgroup's own slot (a no-return atomic on a word nobody else adds to;
    // scopa_counters() adds the slots up
    if (lane == 0) { atomicAdd(&s_vis[0], my_pairs * kPairDecisionVisits); atomicAdd(&s_vis[1], my_pairs * kPairTerminalVisits); }
    __syncthreads();
    if (tid < 2) atomicAdd(&g_wg_counts[blockIdx.x * 2 + tid], (unsigned long long)s_vis[tid]);   // its own slot: no contention, nothing to wait for
#ifdef SCOPA_WALK_STAMPS
    if (blockIdx.x == 0 && (tid & 63) == 0) {
        g_wave_phases[(tid >> 6)] += c_barrier_ - c_entry_; g_wave_phases[16 + (tid >> 6)] += c_loop0_ - c_barrier_;
        g_wave_phases[32 + (tid >> 6)] += c_loop1_ - c_loop0_; g_wave_phases[48 + (tid >> 6)] += clock64() - c_loop1_;
    }
#endif
    if (g_clock && tid == 0) {   // sampled launches: this workgroup's phase stamps (start | prologue done | walks done | end)
        g_clock[blockIdx.x * 4] = t_start; g_clock[blockIdx.x * 4 + 1] = t_pro; g_clock[blockIdx.x * 4 + 2] = t_walk; g_clock[blockIdx.x * 4 + 3] = wall_clock64();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-deal mode: ONE WORKGROUP PER DEAL runs n_iters whole iterations without leaving the kernel.  The deal's regret table
// lives in LDS for the whole solve; each iteration freezes sigma|cdf rows from it, lets the 16 wavefronts walk `batch` traversal
// pairs (LDS float64 atomics straight into the live regret table -- the frozen rows are what the walks read), then adds
// count * sigma to the strategy sums in HBM and refreezes.  No slabs, no reduce kernel, no per-iteration launch.
__global__ void __launch_bounds__(1024)
k_mccfr_multi(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, const uint64_t *__restrict__ g_key,
              double *__restrict__ g_regret, double *__restrict__ g_strat, const int32_t *__restrict__ g_meta, uint32_t *__restrict__ g_visit,
              unsigned long long *__restrict__ g_counters, uint32_t seed_lo, uint32_t seed_hi, uint32_t iter0, uint32_t n_iters,
              uint32_t batch, const uint4 *__restrict__ g_lane_tab) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned int s_vis[2];
    __shared__ uint4 s_lane_tab[kLaneSlotVecs * 64];
    for (int i = threadIdx.x; i < kLaneSlotVecs * 64; i += blockDim.x) s_lane_tab[i] = g_lane_tab[i];
    {
        const size_t deal = blockIdx.x;
        g_infoset += deal * kDecision; g_payoff += deal * kTerminal; g_key += deal * kDecision; g_regret += deal * kDecision * 4;
        g_strat += deal * kDecision * 4; g_meta += deal * 8; g_visit += deal * kDecision; g_counters += deal * 8;
    }
    const int I = g_meta[0];
    double *s_sigcdf = reinterpret_cast<double *>(smem);                         // [I + 1][kRow] frozen rows; row I = all ones (the identity record's)
    double *s_R = s_sigcdf + (size_t)(I + 1) * kRow;                             // [I][4] live regret table
    WaveScratch *s_wave = reinterpret_cast<WaveScratch *>(s_R + (size_t)I * 4);
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_wave + (blockDim.x >> 6));
    uint16_t *s_inf = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(s_cnt) + (((size_t)I * 4 + 15) & ~(size_t)15));
    int8_t *s_pay = reinterpret_cast<int8_t *>(s_inf + 1656);
    uint8_t *s_seen = reinterpret_cast<uint8_t *>(s_pay + kTerminal);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    if (tid < 2) s_vis[tid] = 0u;
    for (int i = tid; i < I * 4; i += blockDim.x) s_R[i] = g_regret[i];
    for (int r = tid; r < I; r += blockDim.x) { s_cnt[r] = 0u; s_seen[r] = 0; }
    for (int i = tid; i < kDecision; i += blockDim.x) s_inf[i] = g_infoset[i];
    for (int i = tid; i < kTerminal; i += blockDim.x) s_pay[i] = g_payoff[i];
    if (tid < 4) s_sigcdf[(size_t)I * kRow + tid] = 1.0;
    __syncthreads();
    WaveScratch *ws = s_wave + wave;
    wave_scratch_init(ws, 1, lane, I);
    WalkEnv env;
    env.wsb = lds_addr(ws); env.tab = lds_addr(s_lane_tab) + 16u * (uint32_t)lane;
    env.s_inf = s_inf; env.s_pay = s_pay; env.s_sigcdf = s_sigcdf; env.s_dR = s_R; env.s_seen = s_seen; env.s_cnt = s_cnt;
    unsigned int my_pairs = 0;   // pairs this wavefront walked
    for (uint32_t it = 0; it < n_iters; it++) {
        for (int r = tid; r < I; r += blockDim.x) {  // freeze this iteration's strategy
            const int n = (int)((g_key[r] >> 1) & 7);
            double sg[4];
            uint32_t thr[4];
            mc_sigma(s_R + r * 4, n, sg);
            choice_cdf(sg, n, thr);
            for (int c = 0; c < 4; c++) { s_sigcdf[r * kRow + c] = sg[c]; reinterpret_cast<uint32_t *>(s_sigcdf + r * kRow + 4)[c] = thr[c]; }
        }
        __syncthreads();
        for (uint32_t pg = (uint32_t)wave; pg < batch; pg += (uint32_t)n_waves) {
            const uint32_t one[1] = {pg};
            walk_pairs<1>(env, lane, one, iter0 + it, seed_lo, seed_hi, my_pairs);
        }
        __syncthreads();
        for (int r = tid; r < I; r += blockDim.x) {  // strategy_sum += count * sigma(frozen)
            const unsigned int c = s_cnt[r];
            if (c) {
                s_seen[r] = 1;   // the walks mark opponent nodes only: a counted traverser visit marks the row here
                const int n = (int)((g_key[r] >> 1) & 7);
                for (int k = 0; k < n; k++) g_strat[r * 4 + k] += (double)c * s_sigcdf[r * kRow + k];
                s_cnt[r] = 0u;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < I * 4; i += blockDim.x) g_regret[i] = s_R[i];
    for (int r = tid; r < I; r += blockDim.x) if (s_seen[r] && g_visit[r] == 0u) g_visit[r] = 0x40000000u + (uint32_t)r;
    if (lane == 0) { atomicAdd(&s_vis[0], my_pairs * kPairDecisionVisits); atomicAdd(&s_vis[1], my_pairs * kPairTerminalVisits); }
    __syncthreads();
    if (tid < 2) g_counters[tid] += s_vis[tid];
}

// Split path, after a traversal launch: delta[r][0..4] += the group tables in table order; the group tables are cleared.
// (The all-reduce payload stays the compact [n_infosets][5] table the caller may have bound.)
__global__ void __launch_bounds__(256)
k_mccfr_fold(double *__restrict__ g_groups, double *__restrict__ g_delta, int n_infosets) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_infosets) return;
    double d[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < kDeltaGroups; g++)
        for (int k = 0; k < 5; k++) {
            double *q = g_groups + (size_t)g * kDeltaTable + group_cell(k, r);
            d[k] += *q;
            *q = 0.0;
        }
    for (int k = 0; k < 5; k++) g_delta[r * 5 + k] += d[k];
}

// One infoset row of the apply step: regret += delta; strategy_sum += count * sigma(frozen regret) (mc_cfr.py:83-84 with the
// iteration's frozen sigma); the next iteration's sigma | cdf row (what k_mccfr_prepare would compute).  The row's table values are
// passed in: the callers load them together with the delta so that the kernel pays ONE memory round trip, not three.
struct ApplyRow { double R[4], S[4], sg[4]; int n; };   // regret, strategy_sum, the frozen sigma the launch sampled with, legal actions

__device__ __forceinline__ ApplyRow apply_row_load(int r, const uint64_t *__restrict__ g_key, const double *__restrict__ g_regret,
                                                   const double *__restrict__ g_strat, const double *__restrict__ g_sigcdf) {
    ApplyRow a;
    const double2 r0 = *reinterpret_cast<const double2 *>(g_regret + r * 4), r1 = *reinterpret_cast<const double2 *>(g_regret + r * 4 + 2);
    const double2 s0 = *reinterpret_cast<const double2 *>(g_strat + r * 4), s1 = *reinterpret_cast<const double2 *>(g_strat + r * 4 + 2);
    const double2 g0 = *reinterpret_cast<const double2 *>(g_sigcdf + r * kRow), g1 = *reinterpret_cast<const double2 *>(g_sigcdf + r * kRow + 2);
    a.n = (int)((g_key[r] >> 1) & 7);
    a.R[0] = r0.x; a.R[1] = r0.y; a.R[2] = r1.x; a.R[3] = r1.y;
    a.S[0] = s0.x; a.S[1] = s0.y; a.S[2] = s1.x; a.S[3] = s1.y;
    a.sg[0] = g0.x; a.sg[1] = g0.y; a.sg[2] = g1.x; a.sg[3] = g1.y;   // = mc_sigma(R): written by the previous apply / prepare
    return a;
}

__device__ __forceinline__ void apply_row_store(int r, ApplyRow &a, const double (&d)[5], double *__restrict__ g_regret,
                                                double *__restrict__ g_strat, double *__restrict__ g_sigcdf) {
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (c < a.n) { a.R[c] += d[c]; a.S[c] += d[4] * a.sg[c]; }
    *reinterpret_cast<double2 *>(g_regret + r * 4) = make_double2(a.R[0], a.R[1]);
    *reinterpret_cast<double2 *>(g_regret + r * 4 + 2) = make_double2(a.R[2], a.R[3]);
    *reinterpret_cast<double2 *>(g_strat + r * 4) = make_double2(a.S[0], a.S[1]);
    *reinterpret_cast<double2 *>(g_strat + r * 4 + 2) = make_double2(a.S[2], a.S[3]);
    double sg[4];
    uint32_t thr[4];
    mc_sigma(a.R, a.n, sg);
    choice_cdf(sg, a.n, thr);
    row_store(g_sigcdf + r * kRow, sg, thr);
}

// one cell (k = 0..3 regret deltas, 4 = visit count) of row r's delta: its group tables summed in table order; non-zero cells are cleared
__device__ __forceinline__ double groups_cell_take(double *__restrict__ g_groups, int r, int k) {
    double v[kDeltaGroups];
#pragma unroll
    for (int g = 0; g < kDeltaGroups; g++) v[g] = g_groups[(size_t)g * kDeltaTable + group_cell(k, r)];   // all loads in flight
    double d = 0.0;
#pragma unroll
    for (int g = 0; g < kDeltaGroups; g++) {
        d += v[g];                                                                                               // table order
        if (v[g] != 0.0) g_groups[(size_t)g * kDeltaTable + group_cell(k, r)] = 0.0;
    }
    return d;
}

// k_mccfr_apply_groups geometry: lanes per infoset row (5 of them fetch a delta cell each) and threads per workgroup.  Measured at
// B = 4096 (us per iteration): 8 lanes x 64 threads 15.50 | 16 lanes 15.51 | 32 lanes 15.95 | 128 threads 15.88 | 256 threads 16.68
// -- many one-wavefront workgroups spread over the compute units beat fewer, fuller ones for this latency-bound launch.
constexpr int kApplyLanes = 8, kApplyThreads = 64;

// The apply step after a single-GPU traversal launch: delta = the group tables summed in table order, cleared on the way.  A
// wavefront (= a workgroup: the rows spread over as many compute units as possible) takes 8 rows, lane = 8 * row + cell: five lanes
// of a row fetch one cell of its delta each, lane 0 of the row collects them and applies the row -- one memory round trip, few
// loads per lane.  The launch must find d_sigcdf current (it holds the sigma the traversal sampled with).
__global__ void __launch_bounds__(kApplyThreads)
k_mccfr_apply_groups(const uint64_t *__restrict__ g_key, double *__restrict__ g_regret, double *__restrict__ g_strat,
                     double *__restrict__ g_groups, int n_infosets, double *__restrict__ g_sigcdf, uint32_t *__restrict__ g_iter) {
    if (g_iter && blockIdx.x == 0 && threadIdx.x == 0) *g_iter += 1u;   // (graph mode) the next traversal launch reads it behind the kernel boundary
    const int lane = threadIdx.x & 63, k = lane & (kApplyLanes - 1);
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x) / kApplyLanes;
    const bool valid = r < n_infosets;
    ApplyRow a{};
    if (valid && k == 0) a = apply_row_load(r, g_key, g_regret, g_strat, g_sigcdf);
    const double mine = (valid && k < 5) ? groups_cell_take(g_groups, r, k) : 0.0;
    double d[5];
#pragma unroll
    for (int j = 0; j < 5; j++) d[j] = __shfl(mine, (lane & ~(kApplyLanes - 1)) + j);
    if (valid && k == 0) apply_row_store(r, a, d, g_regret, g_strat, g_sigcdf);
}

// Split path (traverse | all-reduce | apply): delta = the [I][5] buffer the caller all-reduced, cleared.  One thread per infoset row.
__global__ void __launch_bounds__(64)
k_mccfr_apply(const uint64_t *__restrict__ g_key, double *__restrict__ g_regret, double *__restrict__ g_strat,
              double *__restrict__ g_delta, int n_infosets, double *__restrict__ g_sigcdf) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_infosets) return;
    double d[5];
    for (int k = 0; k < 5; k++) d[k] = g_delta[r * 5 + k];
    ApplyRow a = apply_row_load(r, g_key, g_regret, g_strat, g_sigcdf);
    for (int k = 0; k < 5; k++) g_delta[r * 5 + k] = 0.0;
    apply_row_store(r, a, d, g_regret, g_strat, g_sigcdf);
}

// N > 1, after a traversal launch: this rank's delta of an infoset row (its group tables summed in table order, then cleared)
// is exchanged with the peers (scopa_p2p.h), becomes the rank-ordered sum over ranks (identical bits on every rank) and is applied
// -- the whole multi-GPU iteration stays two launches.  One wavefront per 4 rows, lane = 16 * row + peer; lane 0 of a row applies it.
__global__ void __launch_bounds__(64)
k_mccfr_exchange_apply(scopa::P2PArgs xa, double *__restrict__ g_groups, const uint64_t *__restrict__ g_key, double *__restrict__ g_regret,
                       double *__restrict__ g_strat, double *__restrict__ g_sigcdf, int n_infosets) {
    __shared__ double xch[4][scopa::kP2PMaxWorld][5];
    const int rl = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int r = blockIdx.x * 4 + rl;
    const bool valid = r < n_infosets;
    const double mine = (valid && q < 5) ? groups_cell_take(g_groups, r, q) : 0.0;       // lanes 0..4 of a row: one cell each (and clear it)
    double d[5];
#pragma unroll
    for (int j = 0; j < 5; j++) d[j] = __shfl(mine, (int)(threadIdx.x & ~15u) + j);      // every lane of the row holds the row's delta
    ApplyRow a{};
    if (valid && q == 0) a = apply_row_load(r, g_key, g_regret, g_strat, g_sigcdf);   // under the exchange's waits
    scopa::p2p_exchange_wave4(xa, r, valid, q, d, xch[rl]);
    if (valid && q == 0) apply_row_store(r, a, d, g_regret, g_strat, g_sigcdf);
}

// ---------------------------------------------------------------------------------------------------------------------
// Replay mode: the reference's own sequential semantics (tables are live: every update is seen by the next visit),
// driven by the uniforms np.random.choice would have drawn.  One lane walks, the others stage the tables in and out of
// LDS; this is the bit-exactness anchor, not the throughput path.
namespace {
struct ReplayWalk {
    double *R, *S;              // live tables (LDS)
    const uint16_t *inf;        // node -> infoset (LDS)
    const int8_t *pay;          // leaf payoffs x2 (LDS)
    uint32_t *visit;            // first-visit sequence numbers (LDS)
    const double *uniforms;     // what np.random.choice would have drawn, in DFS entry order (HBM, read sequentially)
    long long upos, n_uniforms;
    uint32_t seq;
    unsigned long long dvis, tvis;
};

// MCCFRTrainer._sample (mc_cfr.py:37-86) as a compile-time recursion over the 8 plies: frames in registers, tables and
// maps in LDS.  sigma is computed once on entry from the live regret row -- the row cannot change before this visit ends
// (its other nodes are on the same ply) -- and used for the sampling, the child reaches and the update.
template <int D, int TRAV>
__device__ __forceinline__ double replay_rec(ReplayWalk &w, int idx, double reach_opp, double samp_trav) {
    if constexpr (D == kPlies) {  // terminal: state.rewards()[traversing_player] (:38-39)
        w.tvis++;
        const int p0 = w.pay[idx];
        return 0.5 * (double)(TRAV == 0 ? p0 : -p0);
    } else {
        constexpr int n = 4 - (D >> 1);
        constexpr bool is_trav = (D & 1) == TRAV;
        w.dvis++;  // node entry (:49-55)
        const int I = w.inf[level_offset(D) + idx];
        if (w.visit[I] == 0u) w.visit[I] = ++w.seq;  // _get_node inserts on first visit (:32-35)
        double R[4], sigma[4], cdf[4];
        for (int c = 0; c < 4; c++) R[c] = w.R[I * 4 + c];
        mc_sigma(R, n, sigma);
        choice_cdf_f64(sigma, n, cdf);
        const double u = w.upos < w.n_uniforms ? w.uniforms[w.upos] : 0.0;
        w.upos++;
        int a = (cdf[0] <= u) + (cdf[1] <= u) + (cdf[2] <= u) + (cdf[3] <= u);
        a = a < n - 1 ? a : n - 1;
        double sa = sigma[0];
        for (int i = 1; i < n; i++) sa = a == i ? sigma[i] : sa;
        if constexpr (!is_trav) {  // opponent node: pass the sampled child's value up (:86)
            return replay_rec<D + 1, TRAV>(w, idx * n + a, reach_opp * sa, samp_trav);
        } else {
            const double util = replay_rec<D + 1, TRAV>(w, idx * n + a, reach_opp, samp_trav * sa);
            double cfv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int i = 0; i < n; i++)  // re-expand every legal action (:72-78)
                cfv[i] = replay_rec<D + 1, TRAV>(w, idx * n + i, reach_opp, samp_trav * sigma[i]);
            double v = 0.0;  // (:79-84)
            for (int i = 0; i < n; i++) v = fma(sigma[i], cfv[i], v);
            const double wt = samp_trav > 0.0 ? reach_opp / samp_trav : 0.0;
            for (int i = 0; i < n; i++) {
                w.R[I * 4 + i] += wt * (cfv[i] - v);
                w.S[I * 4 + i] += 1.0 * sigma[i];  // reach_probs[traverser] stays 1.0 (:61-65)
            }
            return util;
        }
    }
}
}  // namespace

__global__ void __launch_bounds__(256)
k_mccfr_replay(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, const uint64_t *__restrict__ g_key,
               double *__restrict__ g_regret, double *__restrict__ g_strat, const double *__restrict__ uniforms,
               long long n_uniforms, int n_iters, unsigned long long *__restrict__ g_counters, long long *__restrict__ consumed,
               uint32_t *__restrict__ g_visit, int32_t *__restrict__ g_meta) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint16_t s_inf[1656];
    __shared__ int8_t s_pay[kTerminal];
    __shared__ uint32_t s_visit[kDecision];
    const int tid = threadIdx.x, I = g_meta[0], cells = I * 4;
    double *R = reinterpret_cast<double *>(smem), *S = R + cells;
    for (int i = tid; i < cells; i += blockDim.x) { R[i] = g_regret[i]; S[i] = g_strat[i]; }
    for (int i = tid; i < kDecision; i += blockDim.x) s_inf[i] = g_infoset[i];
    for (int i = tid; i < kTerminal; i += blockDim.x) s_pay[i] = g_payoff[i];
    for (int i = tid; i < I; i += blockDim.x) s_visit[i] = g_visit[i];
    __syncthreads();
    if (tid == 0) {
        ReplayWalk w;
        w.R = R; w.S = S; w.inf = s_inf; w.pay = s_pay; w.visit = s_visit; w.uniforms = uniforms; w.upos = 0; w.n_uniforms = n_uniforms;
        w.seq = (uint32_t)g_meta[1]; w.dvis = 0; w.tvis = 0;
#pragma unroll 1
        for (int it = 0; it < n_iters; it++) {  // iteration(), mc_cfr.py:88-92
            replay_rec<0, 0>(w, 0, 1.0, 1.0);
            replay_rec<0, 1>(w, 0, 1.0, 1.0);
        }
        g_counters[0] += w.dvis;
        g_counters[1] += w.tvis;
        g_meta[1] = (int32_t)w.seq;
        *consumed = w.upos;
    }
    __syncthreads();
    for (int i = tid; i < cells; i += blockDim.x) { g_regret[i] = R[i]; g_strat[i] = S[i]; }
    for (int i = tid; i < I; i += blockDim.x) g_visit[i] = s_visit[i];
}

// ---------------------------------------------------------------------------------------------------------------------

static size_t traverse_lds_bytes(int n_infosets, int waves) {
    size_t b = ((size_t)(n_infosets + 1) * kRow + (size_t)n_infosets * 4) * sizeof(double);  // sigma|threshold rows (+ the identity row), delta table
    b += (size_t)waves * 2 * sizeof(WaveScratch);              // per-wavefront records, two pairs in flight
    b += (((size_t)n_infosets * 4 + 15) & ~(size_t)15);        // visit counts
    b += 1656 * 2 + 576;                                       // node -> infoset, leaf payoffs
    b += (size_t)n_infosets;                                   // seen flags
    return (b + 15) & ~(size_t)15;
}

// the context's lane table (k_lane_table), built on its stream before the first launch that reads it
static int32_t ensure_lane_table(scopa_ctx *ctx) {
    if (ctx->d_lane_tab) return SCOPA_OK;
    SC_HIP(ctx, hipMalloc(&ctx->d_lane_tab, (size_t)kLaneSlotVecs * 64 * sizeof(uint4)));
    hipLaunchKernelGGL(k_lane_table, dim3(1), dim3(64), 0, ctx->stream, (uint4 *)ctx->d_lane_tab);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

// One traversal launch of `nb` traversal pairs [b0, b0 + nb) of iteration `iteration` against the rows in d_sigcdf; the launch
// adds its deltas into the context's group tables (all-zero whenever no launch's result is pending).
struct TraverseGeom { int threads; size_t lds; uint32_t grid; };

// First-visit marks (d_visit: which keys the reference's dict would hold, mc_cfr.py:32-35) only ever go from 0 to non-zero, and once every infoset
// of the deal has one a traversal launch has nothing left to record: its walks then skip the `seen` flags and its epilogue the d_visit scan.  The
// marks are counted on the host -- one 6.6 KB read-back, a stream synchronisation -- at most every 16th launch, and only until they are complete.
static int32_t refresh_all_seen(scopa_ctx *ctx, bool now) {
    if (ctx->mccfr_all_seen) return SCOPA_OK;
    if (!now && ++ctx->mccfr_seen_wait < 16u) return SCOPA_OK;
    ctx->mccfr_seen_wait = 0;
    uint32_t h[kDecision];
    SC_HIP(ctx, hipMemcpyAsync(h, ctx->d_visit, (size_t)ctx->n_infosets * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool all = true;
    for (int r = 0; r < ctx->n_infosets; r++) all = all && h[r] != 0u;
    ctx->mccfr_all_seen = all;
    return SCOPA_OK;
}

// everything a traversal launch needs that is not a launch: the kernel's LDS cap, the lane table, current sigma | threshold rows
static int32_t prepare_traverse(scopa_ctx *ctx, uint32_t nb, TraverseGeom *g) {
    // 16 wavefronts per workgroup when the tables leave room for 16 x 2 scratch areas (<= ~960 infosets), fewer for deals
    // with more infosets, down to ONE at the maximum of 1653 (85 bytes per infoset + 23 KB must fit the 160 KB)
    int waves = 16;
    while (waves > 1 && traverse_lds_bytes(ctx->n_infosets, waves) + kStaticLds > (size_t)ctx->lds_limit) waves = waves > 2 ? waves - 2 : 1;
    g->threads = waves * 64;
    g->lds = traverse_lds_bytes(ctx->n_infosets, waves);
    SC_REQUIRE(ctx, g->lds + kStaticLds <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "mccfr traverse: infoset tables do not fit in LDS");
    SC_LDS_ATTR(ctx, scopa::kLdsTraverse, k_mccfr_traverse, ctx->lds_limit - kStaticLds);
    const uint32_t n_passes = (nb + waves - 1) / waves;  // one traversal pair per wavefront pass
    g->grid = n_passes < (uint32_t)ctx->n_cus ? n_passes : (uint32_t)ctx->n_cus;
    SC_REQUIRE(ctx, g->grid <= 1024u, SCOPA_ELIMIT, "mccfr traverse: more than 1024 compute units");
    static_assert(kClockStride >= 4 * 512, "clock sample stride");
    if (int32_t rc = ensure_lane_table(ctx)) return rc;
    if (int32_t rc = refresh_all_seen(ctx, false)) return rc;
    if (!ctx->sigcdf_valid) {  // tables were changed by another entry point since the last apply
        hipLaunchKernelGGL(k_mccfr_prepare, dim3((ctx->n_infosets + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_key,
                           ctx->d_regret, ctx->d_sigcdf, ctx->n_infosets);
        SC_HIP(ctx, hipGetLastError());
        ctx->sigcdf_valid = true;
    }
    return SCOPA_OK;
}

// One traversal launch of `nb` traversal pairs [b0, b0 + nb) of iteration `iteration` against the rows in d_sigcdf; the launch
// adds its deltas into the context's group tables (all-zero whenever no launch's result is pending).
static int32_t launch_traverse(scopa_ctx *ctx, uint32_t iteration, uint32_t b0, uint32_t nb) {
    TraverseGeom g;
    if (int32_t rc = prepare_traverse(ctx, nb, &g)) return rc;
    const int threads = g.threads;
    const size_t lds = g.lds;
    const uint32_t grid = g.grid;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool sampled = prof_events(ctx, &ev0, &ev1);
    unsigned long long *clock = sampled && ctx->d_clock && grid <= 512u ? ctx->d_clock + (size_t)((ctx->prof_launches - 1) % kClockSamples) * kClockStride : nullptr;
    if (sampled) ctx->clock_grid[(ctx->prof_launches - 1) % kClockSamples] = clock ? (uint16_t)grid : (uint16_t)0;
    if (sampled)
        hipExtLaunchKernelGGL(k_mccfr_traverse, dim3(grid), dim3(threads), lds, ctx->stream, ev0, ev1, 0, ctx->d_infoset, ctx->d_payoff,
                              ctx->d_sigcdf, ctx->d_groups, ctx->n_infosets, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32),
                              iteration, b0, nb, ctx->d_counters + 8, ctx->d_visit, clock, (const uint4 *)ctx->d_lane_tab, (const uint32_t *)nullptr,
                              ctx->mccfr_all_seen ? 0u : 1u);
    else
        hipLaunchKernelGGL(k_mccfr_traverse, dim3(grid), dim3(threads), lds, ctx->stream, ctx->d_infoset, ctx->d_payoff,
                           ctx->d_sigcdf, ctx->d_groups, ctx->n_infosets, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32),
                           iteration, b0, nb, ctx->d_counters + 8, ctx->d_visit, clock, (const uint4 *)ctx->d_lane_tab, (const uint32_t *)nullptr,
                           ctx->mccfr_all_seen ? 0u : 1u);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

// ---- graph mode: K iterations (traverse + apply each) captured ONCE into a HIP graph and replayed ------------------------------------
// The iteration number cannot be a kernel argument then: it lives in a device word (d_meta[2]) that the apply kernel advances and the
// traversal kernel reads, so results stay keyed by the same iteration ids as the eager loop's.
void scopa::mccfr_graphs_clear(scopa_ctx *ctx) {
    for (auto &e : ctx->mccfr_graphs) (void)hipGraphExecDestroy((hipGraphExec_t)e.exec);
    ctx->mccfr_graphs.clear();
}

static int32_t graph_for(scopa_ctx *ctx, uint32_t batch, uint32_t k, hipGraphExec_t *out) {
    const uint32_t track = ctx->mccfr_all_seen ? 0u : 1u;   // a launch argument of the captured traversals: part of the key
    for (auto &e : ctx->mccfr_graphs)
        if (e.batch == batch && e.k == k && e.track == track) { *out = (hipGraphExec_t)e.exec; return SCOPA_OK; }
    TraverseGeom g;
    if (int32_t rc = prepare_traverse(ctx, batch, &g)) return rc;
    uint32_t *d_iter = reinterpret_cast<uint32_t *>(ctx->d_meta + 2);
    hipGraph_t graph = nullptr;
    SC_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    for (uint32_t i = 0; i < k; i++) {
        hipLaunchKernelGGL(k_mccfr_traverse, dim3(g.grid), dim3(g.threads), g.lds, ctx->stream, ctx->d_infoset, ctx->d_payoff,
                           ctx->d_sigcdf, ctx->d_groups, ctx->n_infosets, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32),
                           0u, 0u, batch, ctx->d_counters + 8, ctx->d_visit, (unsigned long long *)nullptr, (const uint4 *)ctx->d_lane_tab, (const uint32_t *)d_iter, track);
        hipLaunchKernelGGL(k_mccfr_apply_groups, dim3((ctx->n_infosets * kApplyLanes + kApplyThreads - 1) / kApplyThreads), dim3(kApplyThreads), 0, ctx->stream, ctx->d_key,
                           ctx->d_regret, ctx->d_strat, ctx->d_groups, ctx->n_infosets, ctx->d_sigcdf, d_iter);
    }
    const hipError_t e_end = hipStreamEndCapture(ctx->stream, &graph);
    if (e_end != hipSuccess || !graph) return scopa::fail(ctx, SCOPA_EHIP, "mccfr graph: stream capture", e_end);
    hipGraphExec_t exec = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e_inst != hipSuccess) return scopa::fail(ctx, SCOPA_EHIP, "mccfr graph: instantiate", e_inst);
    if (ctx->mccfr_graphs.size() >= 16) { (void)hipGraphExecDestroy((hipGraphExec_t)ctx->mccfr_graphs.front().exec); ctx->mccfr_graphs.erase(ctx->mccfr_graphs.begin()); }
    ctx->mccfr_graphs.push_back({batch, k, track, (void *)exec});
    *out = exec;
    return SCOPA_OK;
}

namespace scopa {
int32_t launch_mccfr_multi(scopa_ctx *ctx, int n_deals, int max_infosets, const uint16_t *d_infoset, const int8_t *d_payoff,
                           const uint64_t *d_key, double *d_regret, double *d_strat, const int32_t *d_meta, uint32_t *d_visit,
                           unsigned long long *d_counters, uint64_t seed, uint32_t iter0, uint32_t n_iters, uint32_t batch) {
    int waves = 16;
    auto need = [&](int w) {
        size_t b = ((size_t)(max_infosets + 1) * kRow + (size_t)max_infosets * 4) * sizeof(double) + (size_t)w * sizeof(WaveScratch);
        b += (((size_t)max_infosets * 4 + 15) & ~(size_t)15) + 1656 * 2 + 576 + (size_t)max_infosets;
        return (b + 15) & ~(size_t)15;
    };
    while (waves > 1 && need(waves) + kStaticLdsMulti > (size_t)ctx->lds_limit) waves -= 2;
    SC_REQUIRE(ctx, need(waves) + kStaticLdsMulti <= (size_t)ctx->lds_limit, SCOPA_ELIMIT, "mccfr multi: infoset tables do not fit in LDS");
    SC_LDS_ATTR(ctx, scopa::kLdsMulti, k_mccfr_multi, ctx->lds_limit - kStaticLdsMulti);
    if (int32_t rc = ensure_lane_table(ctx)) return rc;
    hipLaunchKernelGGL(k_mccfr_multi, dim3(n_deals), dim3(waves * 64), need(waves), ctx->stream, d_infoset, d_payoff, d_key, d_regret, d_strat,
                       d_meta, d_visit, d_counters, (uint32_t)seed, (uint32_t)(seed >> 32), iter0, n_iters, batch, (const uint4 *)ctx->d_lane_tab);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}
}  // namespace scopa

extern "C" {

int32_t scopa_mccfr_seed(scopa_ctx *ctx, uint64_t seed) {
    if (!ctx) return SCOPA_EINVAL;
    if (seed != ctx->seed) scopa::mccfr_graphs_clear(ctx);   // captured launches carry the seed as an argument
    ctx->seed = seed;
    return SCOPA_OK;
}

int32_t scopa_mccfr_traverse(scopa_ctx *ctx, uint32_t iteration, uint32_t b0, uint32_t nb) {
    // split path: the launch's delta is folded into the (caller-bound) [n_infosets][5] buffer: traverse | all-reduce | apply
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_traverse: no deal set");
    SC_REQUIRE(ctx, nb <= (1u << 30), SCOPA_EINVAL, "scopa_mccfr_traverse: batch too large");
    if (nb == 0) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    scopa::Range r("scopa mccfr traverse (split path)");
    const int32_t rc = launch_traverse(ctx, iteration, b0, nb);
    if (rc != SCOPA_OK) return rc;
    hipLaunchKernelGGL(k_mccfr_fold, dim3((ctx->n_infosets + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_groups, ctx->d_delta, ctx->n_infosets);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
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
    scopa::Range r("scopa mccfr apply (split path)");
    if (!ctx->sigcdf_valid) {  // no traversal since another entry point changed the tables: the rows the apply reads sigma from are stale
        hipLaunchKernelGGL(k_mccfr_prepare, dim3((ctx->n_infosets + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_key,
                           ctx->d_regret, ctx->d_sigcdf, ctx->n_infosets);
        SC_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(k_mccfr_apply, dim3((ctx->n_infosets + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_key,
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
    // single GPU: two launches per iteration -- the traversal, and the lane-per-cell apply over its group tables
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_iterate: no deal set");
    SC_REQUIRE(ctx, batch > 0 && batch <= (1u << 30), SCOPA_EINVAL, "scopa_mccfr_iterate: bad batch");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->mccfr_graph_mode && !ctx->prof_on && n_iters > 1) {
        // graph mode: chunks of <= 64 iterations, each one replay of a captured (traverse, apply) x k chain; launches that carry
        // profiling events cannot be captured, so a profiled run takes the eager loop
        TraverseGeom g;
        if (int32_t rc = prepare_traverse(ctx, batch, &g)) return rc;
        SC_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_meta + 2), (int)ctx->iteration, 1, ctx->stream));
        for (uint32_t left = n_iters; left > 0;) {
            const uint32_t k = left >= 64u ? 64u : left;
            if (int32_t rc = refresh_all_seen(ctx, true)) return rc;   // between chunks, until the marks are complete
            hipGraphExec_t exec = nullptr;
            if (int32_t rc = graph_for(ctx, batch, k, &exec)) return rc;
            SC_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
            ctx->iteration += k;
            left -= k;
        }
        ctx->sigcdf_valid = true;
        return SCOPA_OK;
    }
    for (uint32_t it = 0; it < n_iters; it++) {
        int32_t rc;
        { scopa::Range r("scopa mccfr traverse"); rc = launch_traverse(ctx, ctx->iteration, 0, batch); }
        if (rc != SCOPA_OK) return rc;
        scopa::Range r("scopa mccfr apply");
        hipLaunchKernelGGL(k_mccfr_apply_groups, dim3((ctx->n_infosets * kApplyLanes + kApplyThreads - 1) / kApplyThreads), dim3(kApplyThreads), 0, ctx->stream, ctx->d_key,
                           ctx->d_regret, ctx->d_strat, ctx->d_groups, ctx->n_infosets, ctx->d_sigcdf, (uint32_t *)nullptr);
        SC_HIP(ctx, hipGetLastError());
        ctx->sigcdf_valid = true;
        ctx->iteration++;
    }
    return SCOPA_OK;
}

int32_t scopa_mccfr_graph_mode(scopa_ctx *ctx, int32_t on) {
    if (!ctx) return SCOPA_EINVAL;
    ctx->mccfr_graph_mode = on != 0;
    if (!on) scopa::mccfr_graphs_clear(ctx);
    return SCOPA_OK;
}

int32_t scopa_mccfr_iterate_sharded(scopa_ctx *ctx, uint32_t b0, uint32_t nb, uint32_t n_iters) {
    // N > 1: this rank's slice [b0, b0+nb) of every iteration's global traversal ids; the per-row exchange with the peers
    // (scopa_p2p_create / _connect first) sits inside the apply kernel.  Every rank must call it with the same n_iters.
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_mccfr_iterate_sharded: no deal set");
    SC_REQUIRE(ctx, nb > 0 && nb <= (1u << 30), SCOPA_EINVAL, "scopa_mccfr_iterate_sharded: bad slice");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t it = 0; it < n_iters; it++) {
        scopa::P2PArgs xa{};
        SC_REQUIRE(ctx, scopa::p2p_next_args(ctx, &xa), SCOPA_ESTATE, "mccfr sharded iteration: peer exchange not connected");
        int32_t rc;
        { scopa::Range r("scopa mccfr traverse"); rc = launch_traverse(ctx, ctx->iteration, b0, nb); }
        if (rc != SCOPA_OK) return rc;
        scopa::Range r("scopa mccfr exchange + apply");
        hipLaunchKernelGGL(k_mccfr_exchange_apply, dim3((ctx->n_infosets + 3) / 4), dim3(64), 0, ctx->stream, xa, ctx->d_groups, ctx->d_key,
                           ctx->d_regret, ctx->d_strat, ctx->d_sigcdf, ctx->n_infosets);
        SC_HIP(ctx, hipGetLastError());
        ctx->sigcdf_valid = true;
        ctx->iteration++;
    }
    // a peer that never answered must not go unnoticed: the launches above applied whatever their bounded waits were left with
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return scopa::p2p_check(ctx, "scopa_mccfr_iterate_sharded");
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
    SC_LDS_ATTR(ctx, scopa::kLdsReplay, k_mccfr_replay, ctx->lds_limit - 12 * 1024);
    hipLaunchKernelGGL(k_mccfr_replay, dim3(1), dim3(256), (size_t)ctx->n_infosets * 64, ctx->stream, ctx->d_infoset, ctx->d_payoff, ctx->d_key,
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

#ifdef SCOPA_WALK_STAMPS
extern "C" int scopa_debug_wave_phases(unsigned long long *out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_wave_phases), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
extern "C" int scopa_debug_wave_clocks(unsigned long long *out48) {
    return hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_wave_clocks), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -1;
}
extern "C" int scopa_debug_walk_stamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_walk_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_walk_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
