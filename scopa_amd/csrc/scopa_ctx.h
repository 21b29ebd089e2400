// scopa_ctx.h -- context object behind the C ABI (include/scopa.h): one HIP device, one stream, one deal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/scopa.h"
#include "scopa_rules.h"

// Device-resident data of one context.
//
//  tree (built on device by k_tree_build, BFS/level order; the tree's shape is deal-independent so children
//  are index arithmetic and only two small arrays are kept):
//    d_states   [2229]  scopa_state per node            (35.7 KB)
//    d_infoset  [1653]  uint16 dense infoset id per decision node; ids in reference DFS first-visit order
//    d_payoff   [576]   int8 rewards_x2 of player 0 at each terminal (player 1 = negation)
//    d_key      [1653]  uint64 infoset key per infoset id
//  tables, row-padded [kDecision][4] float64 (only the first n_infosets rows are live):
//    d_regret, d_strat, d_local
//  batched MCCFR:
//    d_delta    [kDecision][5] float64: 4 regret deltas + traverser-visit count (the all-reduce payload)
//    d_groups   [8][5][kDecision] float64: group tables the traversal launches add their partial deltas into (memory-side atomics)
struct scopa_p2p;  // scopa_p2p.hip

struct scopa_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char err[512] = {0};

    bool has_deal = false;
    uint8_t perm[16] = {0};
    int n_infosets = 0;

    scopa_state *d_states = nullptr;
    uint16_t *d_infoset = nullptr;
    int8_t *d_payoff = nullptr;
    uint64_t *d_key = nullptr;
    int32_t *d_meta = nullptr;  // [0] = n_infosets, [1] = first-visit sequence counter
    uint32_t *d_visit = nullptr;  // [kDecision] first-visit sequence number per infoset (0 = unvisited)

    double *d_regret = nullptr, *d_strat = nullptr, *d_local = nullptr;
    double *d_delta = nullptr;        // buffer in use (internal or caller-bound)
    double *d_delta_own = nullptr;    // the internal one
    double *d_scratch = nullptr;  // root values / uniforms staging
    double *d_sigcdf = nullptr;   // [kDecision][6] sigma | threshold rows of the frozen regret table (48 bytes: the layout the traversal keeps in LDS)
    bool sigcdf_valid = false;    // false whenever d_regret changed outside k_mccfr_apply
    bool mccfr_all_seen = false;  // every infoset of the deal has a first-visit mark in d_visit: traversal launches stop tracking first visits (scopa_mccfr.hip); false after scopa_tables_reset
    uint32_t mccfr_seen_wait = 0; // traversal launches since the marks were last counted
    double *d_groups = nullptr;   // [kDeltaGroups group tables][5][kGroupRows] float64: where traversal launches add their deltas (scopa_mccfr.hip); all-zero between launches' applies
    unsigned long long *d_clock = nullptr;   // [2048 sampled launches][512 workgroups][4] phase stamps on the 100 MHz device clock (allocated by scopa_prof_enable)
    uint16_t clock_grid[2048] = {0};         // workgroups of each sampled launch
    double prof_phase_us[3] = {0.0, 0.0, 0.0};   // mean (prologue, walks, epilogue) per workgroup of the samples last folded by scopa_prof_device
    double prof_spread_us[3] = {0.0, 0.0, 0.0};   // scopa_prof_spread: workgroup start offsets and the longest workgroup of the sampled launches
    size_t scratch_bytes = 0;

    unsigned long long *d_counters = nullptr;  // [0] decision visits, [1] terminal visits, [2] aux

    uint64_t seed = 0x5C09A;
    uint64_t sdcfr_visits = 0;  // decision-node visits of SDCFR traversals (one per frontier slot featurised)
    void *d_sdnode = nullptr;   // [kDecision] uint2: feature bits | mover's hand nibbles of every decision node (scopa_sdcfr.hip: k_sdcfr_nodeinfo), built at first use per deal
    bool sdnode_valid = false;  // false after scopa_set_deal
    void *d_eval_thr = nullptr;        // [kDecision][3] uint64: sampling thresholds of the tabular policy last given to scopa_eval_tabular_prepare
    bool eval_thr_valid = false;
    void *d_train_partial = nullptr;   // [32][13777] float: per-workgroup partial gradients (+ partial loss) of scopa_sdcfr_train_step
    void *d_sdpol = nullptr;    // [kDecision] float4: regret-matching policy of every decision node under the nets of the launch at hand (k_sdcfr_policy)
    int sdcfr_mode = 0;         // 0 = policy table per launch + walks (one deal: every node evaluated once), 1 = a forward pass per visit (k_sdcfr_traverse)
    int sdcfr_tile_t = 0, sdcfr_team_w = 0;   // k_sdcfr_traverse: traversals per task (2, 4) and wavefronts per task (1..3); 0 = the library's choice (scopa_sdcfr_tuning)
    uint32_t iteration = 0;

    // profiling of the dominant kernel
    bool prof_on = false;
    int prof_stride = 1;
    long long prof_tick = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;

    // exact vanilla CFR, scheduled form (scopa_cfr.hip): EXIT events of the deal's tree levelled under the per-infoset visit order
    uint32_t *d_lane_tab = nullptr;   // [15][64] uint4: what a lane's node slots are made of, for the traversal kernels (scopa_mccfr.hip: LaneSlots), built at first use
    uint16_t *d_sched = nullptr;   // events (uint32 each) | step offsets | [kDecision][8] path cells (layout: scopa_cfr.hip)
    int sched_steps = 0;
    bool sched_valid = false;      // false after scopa_set_deal
    bool exact_sequential = false; // scopa_cfr_exact_mode(ctx, 1): force the one-lane walk (the form the schedule is checked against)

    scopa_p2p *p2p = nullptr;  // peer-memory exchange of the N > 1 path

    // graph mode of scopa_mccfr_iterate (scopa_mccfr.hip): captured (traverse, apply) x k chains by (batch, k); the iteration number
    // the captured launches use lives in d_meta[2]
    struct GraphEntry { uint32_t batch, k, track; void *exec; };
    std::vector<GraphEntry> mccfr_graphs;
    bool mccfr_graph_mode = false;

    int lds_limit = 160 * 1024;
    int n_cus = 256;
    uint32_t lds_attr_done = 0;  // kernels whose dynamic-LDS cap was raised on THIS context's device (scopa::ensure_lds_attr)
};

namespace scopa {

// group tables of the batched-MCCFR delta (scopa_mccfr.hip): cell-major tables of [5][kGroupRows] float64 per context
#ifndef SCOPA_GROUPS
#define SCOPA_GROUPS 16
#endif
#ifndef SCOPA_GROUP_PERM
#define SCOPA_GROUP_PERM 0
#endif
constexpr int kDeltaGroups = SCOPA_GROUPS;
constexpr int kGroupRows = 1664;                                   // kDecision rounded up to 64 x 26
constexpr size_t kDeltaTable = (size_t)kGroupRows * 5;             // doubles per group table
// where cell k (0..3 regret deltas, 4 visit count) of infoset row r lies in a group table
__host__ __device__ inline size_t group_cell(int k, int r) {
#if SCOPA_GROUP_PERM == 1
    return (size_t)k * kGroupRows + (size_t)((r & 63) * 26 + (r >> 6));
#elif SCOPA_GROUP_PERM == 2
    return (size_t)k * kGroupRows + (size_t)((r & 7) * 208 + (r >> 3));
#else
    return (size_t)k * kGroupRows + (size_t)r;
#endif
}
constexpr int kClockSamples = 2048, kClockStride = 2048;           // sampled launches kept / uint64 per sample (512 workgroups x 4 phase stamps)

inline int32_t fail(scopa_ctx *ctx, int32_t code, const char *what, hipError_t e = hipSuccess) {
    if (ctx) {
        if (e != hipSuccess) snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, hipGetErrorString(e));
        else snprintf(ctx->err, sizeof ctx->err, "%s", what);
    }
    return code;
}

#define SC_HIP(ctx, call)                                                         \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) return scopa::fail((ctx), SCOPA_EHIP, #call, e__); \
    } while (0)

#define SC_REQUIRE(ctx, cond, code, msg)                        \
    do {                                                        \
        if (!(cond)) return scopa::fail((ctx), (code), (msg)); \
    } while (0)

int32_t ensure_scratch(scopa_ctx *ctx, size_t bytes);

// roctx ranges around the solver's phases (traverse / exchange + apply / train), for rocprofv3 --marker-trace timelines: on only with SCOPA_ROCTX=1 in the
// environment (resolved once, from librocprofiler-sdk-roctx.so / libroctx64.so by dlopen: no link-time dependency, nothing in the loop otherwise)
void range_push(const char *name);
void range_pop();
struct Range { explicit Range(const char *n) { range_push(n); } ~Range() { range_pop(); } };

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel, so the "already raised" flag lives in the
// context (one context = one device), not in a process-wide static: a second context on another device raises it again.
enum LdsAttrKernel : uint32_t { kLdsTraverse = 1u, kLdsReplay = 8u, kLdsCfrExact = 16u, kLdsCfrSched = 512u, kLdsExploit = 32u, kLdsCfrSync = 64u, kLdsSdcfr = 128u, kLdsMulti = 256u, kLdsSdcfr2 = 1024u, kLdsSdcfr3 = 2048u, kLdsSdcfr4 = 4096u, kLdsSdcfr5 = 8192u, kLdsSdcfr6 = 16384u, kLdsSdcfr7 = 32768u, kLdsSdPolicy = 65536u, kLdsSdWalk2 = 131072u, kLdsSdWalk4 = 262144u, kLdsSdWalk8 = 524288u, kLdsSdWalk1 = 1048576u, kLdsSdWalk8b = 1u << 21, kLdsSdWalk4b = 1u << 22, kLdsSdWalk2b = 1u << 23, kLdsSdWalk1b = 1u << 24 };
inline int32_t ensure_lds_attr(scopa_ctx *ctx, uint32_t kernel_bit, const void *fn, int bytes) {
    if (ctx->lds_attr_done & kernel_bit) return SCOPA_OK;
    SC_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ctx->lds_attr_done |= kernel_bit;
    return SCOPA_OK;
}
#define SC_LDS_ATTR(ctx, bit, kernel, bytes)                                                                     \
    do {                                                                                                         \
        const int32_t rc__ = scopa::ensure_lds_attr((ctx), (bit), reinterpret_cast<const void *>(kernel), (bytes)); \
        if (rc__ != SCOPA_OK) return rc__;                                                                       \
    } while (0)
// (start, stop) events to attach to a sampled launch of the dominant kernel when profiling is on
bool prof_events(scopa_ctx *ctx, hipEvent_t *start, hipEvent_t *stop);
void p2p_release(scopa_ctx *ctx);
void mccfr_graphs_clear(scopa_ctx *ctx);   // scopa_mccfr.hip: on a new deal, a new seed, context destruction

}  // namespace scopa
