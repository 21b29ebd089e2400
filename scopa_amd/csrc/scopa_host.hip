// scopa_host.hip -- context lifetime, host-side single-state glue, table I/O, counters, profiling.
#include <string.h>

#include <new>

#include "scopa_ctx.h"

using namespace scopa;

namespace scopa {

int32_t ensure_scratch(scopa_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return SCOPA_OK;
    if (ctx->d_scratch) SC_HIP(ctx, hipFree(ctx->d_scratch));
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    SC_HIP(ctx, hipMalloc(&ctx->d_scratch, bytes));
    ctx->scratch_bytes = bytes;
    return SCOPA_OK;
}

// A sampled launch of the dominant kernel gets a (start, stop) HIP event pair attached to the dispatch itself
// (hipExtLaunchKernelGGL): the events carry the kernel's own begin/end timestamps, i.e. what a profiler reports as its
// duration, without the dispatch gaps that hipEventRecord before/after a launch would add.  Returns false when this launch
// is not sampled (profiling off, or not the stride's turn).
// the event pool is created when profiling is switched on, not at the first sampled launch (that one launch took a millisecond)
static void prof_pool_fill(scopa_ctx *ctx) {
    while (ctx->ev_pool.size() < 4096) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        ctx->ev_pool.push_back(e);
    }
}

bool prof_events(scopa_ctx *ctx, hipEvent_t *start, hipEvent_t *stop) {
    if (!ctx->prof_on) return false;
    if ((ctx->prof_tick++ % ctx->prof_stride) != 0) return false;
    if (ctx->ev_used + 2 > ctx->ev_pool.size()) {
        // drain: fold finished pairs into the running sum, then reuse the pool
        if (ctx->ev_used) {
            (void)hipEventSynchronize(ctx->ev_pool[ctx->ev_used - 1]);
            for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]) == hipSuccess) ctx->prof_ms += ms;
            }
            ctx->ev_used = 0;
        }
        if (ctx->ev_pool.empty()) prof_pool_fill(ctx);
    }
    if (ctx->ev_used + 2 > ctx->ev_pool.size()) return false;
    *start = ctx->ev_pool[ctx->ev_used];
    *stop = ctx->ev_pool[ctx->ev_used + 1];
    ctx->ev_used += 2;
    ctx->prof_launches++;
    return true;
}

}  // namespace scopa

#include <dlfcn.h>
namespace scopa {
namespace {
int (*g_roctx_push)(const char *) = nullptr;
int (*g_roctx_pop)() = nullptr;
int g_roctx_state = 0;   // 0 = not looked up yet, 1 = on, -1 = off
void roctx_resolve() {
    g_roctx_state = -1;
    const char *on = getenv("SCOPA_ROCTX");
    if (!on || on[0] != '1') return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
        void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx_push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        g_roctx_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (g_roctx_push && g_roctx_pop) { g_roctx_state = 1; return; }
    }
}
}  // namespace
void range_push(const char *name) {
    if (g_roctx_state == 0) roctx_resolve();
    if (g_roctx_state == 1) (void)g_roctx_push(name);
}
void range_pop() {
    if (g_roctx_state == 1) (void)g_roctx_pop();
}
}  // namespace scopa

extern "C" {

int32_t scopa_abi_version(void) { return SCOPA_ABI_VERSION; }

const char *scopa_strerror(int32_t status) {
    switch (status) {
        case SCOPA_OK: return "ok";
        case SCOPA_EINVAL: return "invalid argument";
        case SCOPA_ENODEV: return "no usable HIP device (the solver path has no CPU fallback)";
        case SCOPA_EHIP: return "HIP runtime error";
        case SCOPA_ESTATE: return "call order violated";
        case SCOPA_ENOMEM: return "out of memory";
        case SCOPA_ELIMIT: return "problem exceeds a compiled-in capacity";
        case SCOPA_ETIMEOUT: return "a peer did not answer within the wait budget";
        default: return "unknown status";
    }
}

const char *scopa_last_error(const scopa_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int32_t scopa_ctx_create(int32_t device_id, void *hip_stream, scopa_ctx **out) {
    if (!out || device_id < 0) return SCOPA_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SCOPA_ENODEV;
    if (device_id >= count) return SCOPA_ENODEV;
    scopa_ctx *ctx = new (std::nothrow) scopa_ctx();
    if (!ctx) return SCOPA_ENOMEM;
    ctx->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e != hipSuccess) { delete ctx; return SCOPA_ENODEV; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        ctx->n_cus = prop.multiProcessorCount;
        ctx->lds_limit = (int)prop.sharedMemPerBlock;
    }
    if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
    else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return SCOPA_EHIP; }
        ctx->own_stream = true;
    }
    const size_t rows = (size_t)kDecision;
    bool ok = hipMalloc(&ctx->d_states, sizeof(scopa_state) * kNodes) == hipSuccess &&
              hipMalloc(&ctx->d_infoset, sizeof(uint16_t) * kDecision) == hipSuccess &&
              hipMalloc(&ctx->d_payoff, kTerminal) == hipSuccess &&
              hipMalloc(&ctx->d_key, sizeof(uint64_t) * kDecision) == hipSuccess &&
              hipMalloc(&ctx->d_meta, sizeof(int32_t) * 8) == hipSuccess &&
              hipMalloc(&ctx->d_visit, sizeof(uint32_t) * kDecision) == hipSuccess &&
              hipMalloc(&ctx->d_sigcdf, (size_t)kDecision * 6 * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->d_groups, scopa::kDeltaGroups * scopa::kDeltaTable * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->d_regret, rows * 4 * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->d_strat, rows * 4 * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->d_local, rows * 4 * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->d_delta_own, rows * 5 * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->d_counters, (8 + 2 * 1024) * sizeof(unsigned long long)) == hipSuccess;  // [0..7] totals, then per-workgroup pairs
    ctx->d_delta = ctx->d_delta_own;
    if (ok) ok = hipMemsetAsync(ctx->d_counters, 0, (8 + 2 * 1024) * sizeof(unsigned long long), ctx->stream) == hipSuccess &&
                 hipMemsetAsync(ctx->d_delta, 0, rows * 5 * sizeof(double), ctx->stream) == hipSuccess &&
                 hipMemsetAsync(ctx->d_groups, 0, scopa::kDeltaGroups * scopa::kDeltaTable * sizeof(double), ctx->stream) == hipSuccess;   // invariant: all-zero between calls
    if (!ok) { scopa_ctx_destroy(ctx); return SCOPA_ENOMEM; }
    *out = ctx;
    return SCOPA_OK;
}

int32_t scopa_ctx_destroy(scopa_ctx *ctx) {
    if (!ctx) return SCOPA_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    scopa::p2p_release(ctx);
    scopa::mccfr_graphs_clear(ctx);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    void *bufs[] = {ctx->d_states, ctx->d_infoset, ctx->d_payoff, ctx->d_key, ctx->d_meta, ctx->d_regret, ctx->d_strat,
                    ctx->d_local, ctx->d_delta_own, ctx->d_scratch, ctx->d_counters, ctx->d_visit, ctx->d_sigcdf, ctx->d_groups, ctx->d_clock, ctx->d_sched, ctx->d_lane_tab, ctx->d_sdnode, ctx->d_sdpol, ctx->d_train_partial, ctx->d_eval_thr};
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SCOPA_OK;
}

int32_t scopa_debug_lds_limit(scopa_ctx *ctx, int32_t bytes) {
    // test hook: pretend the device offers less LDS per workgroup, so that launch geometries that only very large deals reach
    // (narrow workgroups of the traversal kernel) can be exercised on the seed-42 deal; 0 restores the device's own limit
    if (!ctx || bytes < 0) return SCOPA_EINVAL;
    hipDeviceProp_t prop;
    SC_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    SC_REQUIRE(ctx, bytes == 0 || (bytes >= 64 * 1024 && (size_t)bytes <= prop.sharedMemPerBlock), SCOPA_EINVAL, "scopa_debug_lds_limit: 0 or 64 KB .. the device's limit");
    ctx->lds_limit = bytes ? bytes : (int)prop.sharedMemPerBlock;
    ctx->lds_attr_done = 0;                 // the kernels' dynamic-LDS caps are re-derived from the new limit
    scopa::mccfr_graphs_clear(ctx);         // captured launches carry the old geometry
    return SCOPA_OK;
}

int32_t scopa_ctx_synchronize(scopa_ctx *ctx) {
    if (!ctx) return SCOPA_EINVAL;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

// ---- CPython random.seed(int) + random.shuffle (MiniDeck.__init__, mini_scopa_game.py:25-28) ----------------
namespace {
struct Mt19937 {
    uint32_t mt[624];
    int at = 624;
    void seed_u32(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        at = 624;
    }
    void seed_array(const uint32_t *key, int n) {  // init_by_array, as CPython's random_seed() calls it
        seed_u32(19650218u);
        int i = 1, j = 0;
        for (int k = n > 624 ? n : 624; k > 0; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
            if (++j >= n) j = 0;
        }
        for (int k = 623; k > 0; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    void refill() {
        for (int k = 0; k < 624; k++) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        at = 0;
    }
    uint32_t next() {
        if (at >= 624) refill();
        uint32_t y = mt[at++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t below(uint32_t n) {  // Random._randbelow_with_getrandbits
        int bits = 32 - __builtin_clz(n);
        uint32_t r;
        do r = next() >> (32 - bits); while (r >= n);
        return r;
    }
};
}  // namespace

int32_t scopa_deal_py_seed(int64_t seed, uint8_t perm16[16]) {
    if (!perm16) return SCOPA_EINVAL;
    const uint64_t a = seed < 0 ? (uint64_t)0 - (uint64_t)seed : (uint64_t)seed;  // random.seed(int) takes abs()
    const uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
    Mt19937 g;
    g.seed_array(key, key[1] ? 2 : 1);
    for (int i = 0; i < 16; i++) perm16[i] = (uint8_t)i;  // deck order: suit-major, rank_idx minor = card id
    for (uint32_t i = 15; i >= 1; i--) {                  // random.shuffle
        const uint32_t j = g.below(i + 1);
        const uint8_t t = perm16[i]; perm16[i] = perm16[j]; perm16[j] = t;
    }
    return SCOPA_OK;
}

// ---- host-side single-state protocol ---------------------------------------------------------------------------
static bool perm_ok(const uint8_t *p) {
    uint32_t seen = 0;
    for (int i = 0; i < 16; i++) { if (p[i] > 15) return false; seen |= 1u << p[i]; }
    return seen == 0xFFFFu;
}

int32_t scopa_state_init(const uint8_t perm16[16], scopa_state *out) {
    if (!perm16 || !out || !perm_ok(perm16)) return SCOPA_EINVAL;
    state_init(*out, perm16);
    return SCOPA_OK;
}

int32_t scopa_state_step(scopa_state *s, int32_t action) {
    if (!s || action < 0 || action > 15) return SCOPA_EINVAL;
    step(*s, action);
    return SCOPA_OK;
}

int32_t scopa_state_clone(const scopa_state *s, scopa_state *out) {
    // MiniScopaState.clone (openspiel_mini_scopa.py:97-115): set_state(get_state()) copies the position; the new env's max_steps is 16 (:108)
    if (!s || !out) return SCOPA_EINVAL;
    // a terminal state's clone stays terminal (the terminations dict and _is_terminal are copied, mini_scopa_game.py:193, openspiel…:112):
    // terminal is recomputed from the fields here, so the limit that ended the game is kept
    scopa_state c = *s;
    if (!is_terminal(c)) c.step = (uint8_t)(c.step | SCOPA_STEP_CLONED);
    *out = c;
    return SCOPA_OK;
}

int32_t scopa_state_is_terminal(const scopa_state *s) { return s ? (is_terminal(*s) ? 1 : 0) : SCOPA_EINVAL; }

int32_t scopa_state_current_player(const scopa_state *s) { return s ? current_player(*s) : SCOPA_EINVAL; }

int32_t scopa_state_legal(const scopa_state *s, int32_t player, int32_t out[4], int32_t *n) {
    if (!s || !out || !n || player > 1) return SCOPA_EINVAL;
    int tmp[4];
    *n = legal(*s, player, tmp);
    for (int i = 0; i < *n; i++) out[i] = tmp[i];
    return SCOPA_OK;
}

int32_t scopa_state_rewards_x2(const scopa_state *s, int32_t r2[2]) {
    if (!s || !r2) return SCOPA_EINVAL;
    int a, b;
    rewards_x2(*s, a, b);
    r2[0] = a; r2[1] = b;
    return SCOPA_OK;
}

int32_t scopa_state_infoset_key(const scopa_state *s, int32_t player, uint64_t *key) {
    if (!s || !key || player > 1) return SCOPA_EINVAL;
    if (player < 0) player = s->step & 1;
    *key = infoset_key(*s, player);
    return SCOPA_OK;
}

int32_t scopa_key_to_string(uint64_t key, char *buf, int32_t cap) {
    // information_state_string, openspiel_mini_scopa.py:86-95:  P{p}:H[9f-6p]_T[7b]
    if (!buf || cap < 64) return SCOPA_EINVAL;
    static const char suit[4] = {'c', 'f', 'p', 'b'};
    const int player = (int)(key & 1), nh = (int)((key >> 1) & 7), nt = (int)((key >> 20) & 15);
    const uint32_t hand = (uint32_t)((key >> 4) & 0xFFFF), table = (uint32_t)(key >> 24);
    char *w = buf;
    w += sprintf(w, "P%d:H[", player);
    for (int i = 0; i < nh; i++) w += sprintf(w, "%s%d%c", i ? "-" : "", card_rank(nib(hand, i)), suit[nib(hand, i) >> 2]);
    w += sprintf(w, "]_T[");
    for (int i = 0; i < nt; i++) w += sprintf(w, "%s%d%c", i ? "-" : "", card_rank(nib(table, i)), suit[nib(table, i) >> 2]);
    w += sprintf(w, "]");
    return (int32_t)(w - buf);
}

int32_t scopa_state_infoset_string(const scopa_state *s, int32_t player, char *buf, int32_t cap) {
    if (!s || !buf || cap < 64 || player > 1) return SCOPA_EINVAL;
    if (player < 0) player = current_player(*s);
    if (is_terminal(*s) || player < 0) return (int32_t)sprintf(buf, "TERMINAL");
    return scopa_key_to_string(infoset_key(*s, player), buf, cap);
}

// ---- tables ----------------------------------------------------------------------------------------------------
int32_t scopa_tables_get(scopa_ctx *ctx, double *h_regret, double *h_strategy, double *h_local) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_tables_get: no deal set");
    const size_t bytes = (size_t)ctx->n_infosets * 4 * sizeof(double);
    if (h_regret) SC_HIP(ctx, hipMemcpyAsync(h_regret, ctx->d_regret, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (h_strategy) SC_HIP(ctx, hipMemcpyAsync(h_strategy, ctx->d_strat, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (h_local) SC_HIP(ctx, hipMemcpyAsync(h_local, ctx->d_local, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

int32_t scopa_tables_set(scopa_ctx *ctx, const double *h_regret, const double *h_strategy, const double *h_local) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_tables_set: no deal set");
    const size_t bytes = (size_t)ctx->n_infosets * 4 * sizeof(double);
    if (h_regret) SC_HIP(ctx, hipMemcpyAsync(ctx->d_regret, h_regret, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (h_strategy) SC_HIP(ctx, hipMemcpyAsync(ctx->d_strat, h_strategy, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (h_local) SC_HIP(ctx, hipMemcpyAsync(ctx->d_local, h_local, bytes, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sigcdf_valid = false;
    return SCOPA_OK;
}

int32_t scopa_visited_get(scopa_ctx *ctx, uint32_t *h_seq) {
    if (!ctx || !h_seq) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_visited_get: no deal set");
    SC_HIP(ctx, hipMemcpyAsync(h_seq, ctx->d_visit, (size_t)ctx->n_infosets * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCOPA_OK;
}

// ---- counters / profiling --------------------------------------------------------------------------------------
int32_t scopa_counters(scopa_ctx *ctx, uint64_t *decision_visits, uint64_t *terminal_visits) {
    if (!ctx) return SCOPA_EINVAL;
    // [0], [1]: totals the single-workgroup solvers add to; [8 + 2w], [9 + 2w]: the slot workgroup w of the batched traversal
    // kernel accumulates in (one plain read-modify-write per launch instead of 256 same-address atomics)
    static_assert(sizeof(unsigned long long) == 8, "counter layout");
    std::vector<unsigned long long> h(8 + 2 * 1024);
    SC_HIP(ctx, hipMemcpyAsync(h.data(), ctx->d_counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long d = h[0], t = h[1];
    for (int w = 0; w < 1024; w++) { d += h[8 + 2 * w]; t += h[9 + 2 * w]; }
    if (decision_visits) *decision_visits = d;
    if (terminal_visits) *terminal_visits = t;
    return SCOPA_OK;
}

int32_t scopa_prof_device(scopa_ctx *ctx, int64_t *launches, double *kernel_ms) {
    // the traversal kernel's own clock: per SAMPLED launch (scopa_prof_enable), first workgroup start -> last workgroup end on
    // the 100 MHz device-wide counter (s_memrealtime), summed over the samples still held (the last 2048): no events, no
    // dispatch latency, nothing folded on the device
    if (!ctx) return SCOPA_EINVAL;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n = ctx->d_clock ? (ctx->prof_launches < scopa::kClockSamples ? ctx->prof_launches : scopa::kClockSamples) : 0;
    unsigned long long ticks = 0;
    if (n > 0) {
        std::vector<unsigned long long> h((size_t)n * scopa::kClockStride);
        SC_HIP(ctx, hipMemcpy(h.data(), ctx->d_clock, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double ph[3] = {0.0, 0.0, 0.0}, sp[4] = {0.0, 0.0, 0.0, 0.0};
        long long wgs = 0, launches_seen = 0;
        for (int64_t s = 0; s < n; s++) {
            unsigned long long t0 = ~0ull, t1 = 0ull, late = 0ull, longest = 0ull;
            for (int w = 0; w < ctx->clock_grid[s]; w++) {
                const unsigned long long *q = &h[(size_t)s * scopa::kClockStride + 4 * w];
                t0 = q[0] < t0 ? q[0] : t0; t1 = q[3] > t1 ? q[3] : t1;
                for (int k = 0; k < 3; k++) ph[k] += (double)(q[k + 1] - q[k]);
                wgs++;
            }
            for (int w = 0; w < ctx->clock_grid[s]; w++) {     // where a launch's time goes beside its workgroups' own phases: how late they start, how long the longest runs
                const unsigned long long *q = &h[(size_t)s * scopa::kClockStride + 4 * w];
                sp[0] += (double)(q[0] - t0);
                late = q[0] - t0 > late ? q[0] - t0 : late;
                longest = q[3] - q[0] > longest ? q[3] - q[0] : longest;
            }
            if (ctx->clock_grid[s]) { sp[1] += (double)late; sp[2] += (double)longest; launches_seen++; }
            if (t1 > t0) ticks += t1 - t0;
        }
        for (int k = 0; k < 3; k++) ctx->prof_phase_us[k] = wgs ? ph[k] * 1e-2 / (double)wgs : 0.0;   // 10 ns ticks -> us, mean over workgroups
        ctx->prof_spread_us[0] = wgs ? sp[0] * 1e-2 / (double)wgs : 0.0;                                // mean start of a workgroup behind its launch's first
        ctx->prof_spread_us[1] = launches_seen ? sp[1] * 1e-2 / (double)launches_seen : 0.0;            // the last workgroup's start behind the first, mean over launches
        ctx->prof_spread_us[2] = launches_seen ? sp[2] * 1e-2 / (double)launches_seen : 0.0;            // the longest workgroup of a launch, mean over launches
    }
    if (launches) *launches = n;
    if (kernel_ms) *kernel_ms = (double)ticks * 1e-5;   // 10 ns ticks
    return SCOPA_OK;
}

int32_t scopa_prof_phases(scopa_ctx *ctx, double out_us[3]) {
    // mean over the sampled launches' workgroups of (prologue, walks, epilogue) on the device clock; call after scopa_prof_device
    if (!ctx || !out_us) return SCOPA_EINVAL;
    for (int k = 0; k < 3; k++) out_us[k] = ctx->prof_phase_us[k];
    return SCOPA_OK;
}

// development (tests/tools/wg_starts.py; not part of include/scopa.h): the raw per-workgroup stamps (start | prologue done | walks done | end, 10 ns ticks) of the
// last `n` sampled launches, [n][grid][4]; returns the grid of those launches or a negative status
int32_t scopa_debug_clock_dump(scopa_ctx *ctx, unsigned long long *out, int32_t n, int32_t max_grid) {
    if (!ctx || !out || n <= 0 || !ctx->d_clock) return SCOPA_EINVAL;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t have = ctx->prof_launches < scopa::kClockSamples ? ctx->prof_launches : scopa::kClockSamples;
    if (have < n) return SCOPA_ESTATE;
    int grid = 0;
    for (int32_t s = 0; s < n; s++) {
        const int64_t slot = (ctx->prof_launches - 1 - s) % scopa::kClockSamples;
        grid = ctx->clock_grid[slot];
        if (grid <= 0 || grid > max_grid) return SCOPA_ELIMIT;
        SC_HIP(ctx, hipMemcpy(out + (size_t)s * max_grid * 4, ctx->d_clock + (size_t)slot * scopa::kClockStride, (size_t)grid * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    return grid;
}

int32_t scopa_prof_spread(scopa_ctx *ctx, double out_us[3]) {
    // beside the phases: mean start of a workgroup behind the first of its launch | the LAST workgroup's start behind the first | the longest workgroup of a
    // launch (both means over the sampled launches); call after scopa_prof_device
    if (!ctx || !out_us) return SCOPA_EINVAL;
    for (int k = 0; k < 3; k++) out_us[k] = ctx->prof_spread_us[k];
    return SCOPA_OK;
}

int32_t scopa_prof_enable(scopa_ctx *ctx, int32_t stride) {
    if (!ctx || stride < 0) return SCOPA_EINVAL;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->prof_on = stride != 0;
    ctx->prof_stride = stride > 0 ? stride : 1;
    ctx->prof_tick = 0;
    ctx->ev_used = 0;
    ctx->prof_launches = 0;
    ctx->prof_ms = 0.0;
    if (ctx->prof_on) scopa::prof_pool_fill(ctx);
    if (ctx->prof_on && !ctx->d_clock)
        SC_HIP(ctx, hipMalloc(&ctx->d_clock, (size_t)scopa::kClockSamples * scopa::kClockStride * sizeof(unsigned long long)));
    return SCOPA_OK;
}

int32_t scopa_prof_read(scopa_ctx *ctx, int64_t *launches, double *kernel_ms) {
    if (!ctx) return SCOPA_EINVAL;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]) == hipSuccess) ctx->prof_ms += ms;
    }
    ctx->ev_used = 0;
    if (launches) *launches = ctx->prof_launches;
    if (kernel_ms) *kernel_ms = ctx->prof_ms;
    return SCOPA_OK;
}

}  // extern "C"
