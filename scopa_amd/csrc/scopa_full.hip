// scopa_full.hip -- FullScopa state engine: host-side protocol, batched device step, on-device random playouts.
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "scopa_ctx.h"
#include "scopa_full_rules.h"
#include "scopa_mt.h"
#include "scopa_philox.h"

using namespace scopa_full;
using scopa::fail;
using scopa::Mt;
static_assert(sizeof(scopa_full_state) == 64, "scopa_full_state must be 64 bytes");


// A lane per game, but the 64-byte states move between HBM and the lanes THROUGH LDS: a wavefront's 64 states are 4 KB of consecutive memory, which
// it loads and stores as 256 sixteen-byte pieces in memory order (four instructions of one contiguous kilobyte each) and transposes in LDS (state j at
// 80 j: the 16-byte pad keeps both the piece-order and the lane-order accesses free of bank conflicts).  A lane loading its own state directly reads
// 16 bytes out of every 64: four (or, member by member, ten) instructions that each touch all 64 cache lines of the wavefront.
constexpr int kFullLdsStride = 80;
__global__ void __launch_bounds__(256)
k_full_step_batch(scopa_full_state *__restrict__ states, const uint8_t *__restrict__ actions, const uint8_t *__restrict__ decks, long long n) {
    __shared__ __align__(16) unsigned char s_stage[4][64 * kFullLdsStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long w0 = ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63));   // first game of this wavefront
    if (w0 >= n) return;
    const long long left = n - w0;                                                       // games of this wavefront that exist (>= 1)
    unsigned char *st = s_stage[wave];
    int act = 0;                                                                         // (asked for with the states, not after them)
    const uint4 *src = reinterpret_cast<const uint4 *>(states + w0);
    if (left >= 64) {                                   // a whole wavefront (all but the last): the four loads are issued together, unguarded
        act = actions[w0 + lane];
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = src[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = k * 64 + lane;
            *reinterpret_cast<uint4 *>(st + (c >> 2) * kFullLdsStride + (c & 3) * 16) = v[k];
        }
    } else {
        if (lane < left) act = actions[w0 + lane];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = k * 64 + lane, j = c >> 2, part = c & 3;
            if (j < left) *reinterpret_cast<uint4 *>(st + j * kFullLdsStride + part * 16) = src[c];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < left) {
        uint4 raw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) raw[k] = *reinterpret_cast<const uint4 *>(st + lane * kFullLdsStride + k * 16);
        scopa_full_state s;
        memcpy(&s, raw, 64);
        step(s, decks + (size_t)s.game * 40, act % 40);
        memcpy(raw, &s, 64);
#pragma unroll
        for (int k = 0; k < 4; k++) *reinterpret_cast<uint4 *>(st + lane * kFullLdsStride + k * 16) = raw[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4 *dst = reinterpret_cast<uint4 *>(states + w0);
    if (left >= 64) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = k * 64 + lane;
            dst[c] = *reinterpret_cast<const uint4 *>(st + (c >> 2) * kFullLdsStride + (c & 3) * 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = k * 64 + lane, j = c >> 2, part = c & 3;
            if (j < left) dst[c] = *reinterpret_cast<const uint4 *>(st + j * kFullLdsStride + part * 16);
        }
    }
}

// One lane per game: deal from the seed, then uniform-random legal play to the end (36 plies without no-ops).
__global__ void __launch_bounds__(64)
k_full_random_playouts(const int64_t *__restrict__ seeds, long long n, int8_t *__restrict__ r2_p0, int16_t *__restrict__ plies,
                       uint32_t seed_lo, uint32_t seed_hi) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Mt g;
    const int64_t sd = seeds[i];
    g.seed(sd < 0 ? (uint64_t)0 - (uint64_t)sd : (uint64_t)sd);
    uint8_t deck[40];
    g.shuffle(deck, 40);
    scopa_full_state s;
    state_init(s, deck, (uint32_t)i);
    int ply = 0;
    while (!s.terminal && ply < 200) {
        const int p = s.step & 1, nl = nh_of(s, p);
        const scopa::philox_out x = scopa::philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)ply, 32u, seed_lo, seed_hi);
        int k = (int)(scopa::u53(x.x0, x.x1) * (double)(nl > 0 ? nl : 1));
        k = k < nl - 1 ? k : (nl > 0 ? nl - 1 : 0);
        step(s, deck, nl > 0 ? hand_get(s, p, k) : 0);
        ply++;
    }
    r2_p0[i] = s.r2_p0;
    plies[i] = (int16_t)ply;
}

extern "C" {

int32_t scopa_full_deal_py_seed(int64_t seed, uint8_t perm40[40]) {
    if (!perm40) return SCOPA_EINVAL;
    Mt g;
    g.seed(seed < 0 ? (uint64_t)0 - (uint64_t)seed : (uint64_t)seed);
    g.shuffle(perm40, 40);
    return SCOPA_OK;
}

static bool deck_ok(const uint8_t *d) {
    uint64_t seen = 0;
    for (int i = 0; i < 40; i++) { if (d[i] > 39) return false; seen |= 1ull << d[i]; }
    return seen == ((1ull << 40) - 1);
}

int32_t scopa_full_state_init(const uint8_t deck40[40], uint32_t game, scopa_full_state *out) {
    if (!deck40 || !out || !deck_ok(deck40)) return SCOPA_EINVAL;
    state_init(*out, deck40, game);
    return SCOPA_OK;
}

int32_t scopa_full_state_step(scopa_full_state *s, const uint8_t deck40[40], int32_t action) {
    if (!s || !deck40 || action < 0 || action > 39) return SCOPA_EINVAL;
    step(*s, deck40, action);
    return (s->flags & 1u) ? SCOPA_ELIMIT : SCOPA_OK;
}

int32_t scopa_full_state_legal(const scopa_full_state *s, int32_t player, int32_t out[3], int32_t *n) {
    if (!s || !out || !n || player > 1) return SCOPA_EINVAL;
    int tmp[3];
    *n = legal(*s, player, tmp);
    for (int i = 0; i < *n; i++) out[i] = tmp[i];
    return SCOPA_OK;
}

int32_t scopa_full_state_infoset_string(const scopa_full_state *s, int32_t player, char *buf, int32_t cap) {
    // information_state_string (openspiel_full_scopa.py:79-94): cards SORTED by (rank, suit name)
    if (!s || !buf || cap < 200 || player < 0 || player > 1) return SCOPA_EINVAL;
    static const char *suit_name[4] = {"denari", "coppe", "spade", "bastoni"};
    auto key = [&](int c) { return std::make_pair(rank_of(c), std::string(suit_name[c / 10])); };
    std::vector<int> h, t;
    for (int i = 0; i < s->nh[player]; i++) h.push_back(hand_get(*s, player, i));
    for (int i = 0; i < s->nt; i++) t.push_back(tab_get(*s, i));
    auto cmp = [&](int a, int b) { return key(a) < key(b); };
    std::sort(h.begin(), h.end(), cmp);
    std::sort(t.begin(), t.end(), cmp);
    char *w = buf;
    w += sprintf(w, "P%d:R%d:H[", player, (int)s->round);
    for (size_t i = 0; i < h.size(); i++) w += sprintf(w, "%s%d%c", i ? "-" : "", rank_of(h[i]), suit_name[h[i] / 10][0]);
    w += sprintf(w, "]:T[");
    for (size_t i = 0; i < t.size(); i++) w += sprintf(w, "%s%d%c", i ? "-" : "", rank_of(t[i]), suit_name[t[i] / 10][0]);
    w += sprintf(w, "]:C[%d,%d]:S[%d,%d]", popc64(s->cap[0]), popc64(s->cap[1]), (int)s->scopas[0], (int)s->scopas[1]);
    return (int32_t)(w - buf);
}

int32_t scopa_full_step_batch(scopa_ctx *ctx, scopa_full_state *d_states, const uint8_t *d_actions, const uint8_t *d_decks, int64_t n) {
    if (!ctx || n < 0 || (n && (!d_states || !d_actions || !d_decks))) return SCOPA_EINVAL;
    if (!n) return SCOPA_OK;
    SC_REQUIRE(ctx, ((uintptr_t)d_states & 15) == 0, SCOPA_EINVAL, "scopa_full_step_batch: states must be 16-byte aligned");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_full_step_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_states, d_actions, d_decks, (long long)n);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_full_step_batch_host(scopa_ctx *ctx, scopa_full_state *h_states, const uint8_t *h_actions, const uint8_t *h_decks,
                                   int64_t n_decks, int64_t n) {
    if (!ctx || n < 0 || n_decks <= 0 || (n && (!h_states || !h_actions || !h_decks))) return SCOPA_EINVAL;
    if (!n) return SCOPA_OK;
    for (int64_t i = 0; i < n; i++) if (h_states[i].game >= (uint64_t)n_decks) return fail(ctx, SCOPA_EINVAL, "scopa_full_step_batch_host: deck index out of range");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_s = nullptr, *d_a = nullptr, *d_d = nullptr;
    int32_t rc = SCOPA_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_s, (size_t)n * 64)) != hipSuccess || (e = hipMalloc(&d_a, (size_t)n)) != hipSuccess ||
        (e = hipMalloc(&d_d, (size_t)n_decks * 40)) != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipMalloc", e);
    if (rc == SCOPA_OK && ((e = hipMemcpyAsync(d_s, h_states, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
                           (e = hipMemcpyAsync(d_a, h_actions, (size_t)n, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
                           (e = hipMemcpyAsync(d_d, h_decks, (size_t)n_decks * 40, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess))
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(H2D)", e);
    if (rc == SCOPA_OK) rc = scopa_full_step_batch(ctx, (scopa_full_state *)d_s, (const uint8_t *)d_a, (const uint8_t *)d_d, n);
    if (rc == SCOPA_OK && (e = hipMemcpyAsync(h_states, d_s, (size_t)n * 64, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(D2H)", e);
    e = hipStreamSynchronize(ctx->stream);
    if (rc == SCOPA_OK && e != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipStreamSynchronize", e);
    if (d_s) (void)hipFree(d_s);
    if (d_a) (void)hipFree(d_a);
    if (d_d) (void)hipFree(d_d);
    return rc;
}

int32_t scopa_full_random_playouts(scopa_ctx *ctx, const int64_t *h_seeds, int64_t n_games, int8_t *h_r2_p0, int16_t *h_plies) {
    if (!ctx || n_games < 0 || (n_games && (!h_seeds || !h_r2_p0 || !h_plies))) return SCOPA_EINVAL;
    if (!n_games) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_seed = nullptr, *d_r = nullptr, *d_p = nullptr;
    int32_t rc = SCOPA_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_seed, (size_t)n_games * 8)) != hipSuccess || (e = hipMalloc(&d_r, (size_t)n_games)) != hipSuccess ||
        (e = hipMalloc(&d_p, (size_t)n_games * 2)) != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipMalloc", e);
    if (rc == SCOPA_OK && (e = hipMemcpyAsync(d_seed, h_seeds, (size_t)n_games * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(H2D)", e);
    if (rc == SCOPA_OK) {
        hipLaunchKernelGGL(k_full_random_playouts, dim3((unsigned)((n_games + 63) / 64)), dim3(64), 0, ctx->stream, (const int64_t *)d_seed,
                           (long long)n_games, (int8_t *)d_r, (int16_t *)d_p, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32));
        if ((e = hipGetLastError()) != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "k_full_random_playouts", e);
    }
    if (rc == SCOPA_OK && ((e = hipMemcpyAsync(h_r2_p0, d_r, (size_t)n_games, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
                           (e = hipMemcpyAsync(h_plies, d_p, (size_t)n_games * 2, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess))
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(D2H)", e);
    e = hipStreamSynchronize(ctx->stream);
    if (rc == SCOPA_OK && e != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipStreamSynchronize", e);
    if (d_seed) (void)hipFree(d_seed);
    if (d_r) (void)hipFree(d_r);
    if (d_p) (void)hipFree(d_p);
    return rc;
}

}  // extern "C"
