// scopa_team.hip -- Team MiniScopa TPI state engine: host-side protocol, batched device step, on-device random playouts.
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "scopa_ctx.h"
#include "scopa_mt.h"
#include "scopa_philox.h"
#include "scopa_team_rules.h"

using namespace scopa_team;
using scopa::fail;
static_assert(sizeof(scopa_team_state) == 40, "scopa_team_state must be 40 bytes");

// A lane per game; the 40-byte states cross between HBM and the lanes through LDS, as in k_full_step_batch: a wavefront's 64 states are 2 560 consecutive
// bytes (16-byte aligned: 64 x 40), loaded and stored as 160 sixteen-byte pieces in memory order and picked up by their lanes as five 8-byte words each.
__global__ void __launch_bounds__(256)
k_team_step_batch(scopa_team_state *__restrict__ states, const uint8_t *__restrict__ actions, long long n) {
    __shared__ __align__(16) unsigned char s_stage[4][64 * 40];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long w0 = (long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63);   // first game of this wavefront
    if (w0 >= n) return;
    const long long left = n - w0;
    const int pieces = left >= 64 ? 160 : (int)((left * 40 + 15) / 16);             // whole 16-byte pieces covering the wavefront's states ...
    const bool tail8 = left < 64 && ((left * 40) & 15);                               // ... of which the last is half a piece when an odd number of states is left
    unsigned char *st = s_stage[wave];
    int act = 0;                                                                         // (asked for with the states, not after them)
    const uint4 *src = reinterpret_cast<const uint4 *>(states + w0);
    if (left >= 64) {                                   // a whole wavefront (all but the last): the three loads are issued together, unguarded
        act = actions[w0 + lane];
        const uint4 v0 = src[lane], v1 = src[64 + lane], v2 = src[128 + (lane & 31)];   // (the upper half re-reads the lower half's pieces: same cache lines, no branch between the loads)
        *reinterpret_cast<uint4 *>(st + lane * 16) = v0;
        *reinterpret_cast<uint4 *>(st + (64 + lane) * 16) = v1;
        *reinterpret_cast<uint4 *>(st + (128 + (lane & 31)) * 16) = v2;                  // (twice the same bytes to the same place)
    } else {
        if (lane < left) act = actions[w0 + lane];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int c = k * 64 + lane;
            if (c < pieces) {
                if (tail8 && c == pieces - 1) *reinterpret_cast<uint2 *>(st + c * 16) = *reinterpret_cast<const uint2 *>(src + c);
                else *reinterpret_cast<uint4 *>(st + c * 16) = src[c];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < left) {
        uint2 raw[5];
#pragma unroll
        for (int k = 0; k < 5; k++) raw[k] = *reinterpret_cast<const uint2 *>(st + lane * 40 + k * 8);
        uint32_t w[10];
#pragma unroll
        for (int k = 0; k < 5; k++) { w[2 * k] = raw[k].x; w[2 * k + 1] = raw[k].y; }
        step_words(w, act);
#pragma unroll
        for (int k = 0; k < 5; k++) raw[k] = make_uint2(w[2 * k], w[2 * k + 1]);
#pragma unroll
        for (int k = 0; k < 5; k++) *reinterpret_cast<uint2 *>(st + lane * 40 + k * 8) = raw[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4 *dst = reinterpret_cast<uint4 *>(states + w0);
    if (left >= 64) {
        dst[lane] = *reinterpret_cast<const uint4 *>(st + lane * 16);
        dst[64 + lane] = *reinterpret_cast<const uint4 *>(st + (64 + lane) * 16);
        if (lane < 32) dst[128 + lane] = *reinterpret_cast<const uint4 *>(st + (128 + lane) * 16);
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int c = k * 64 + lane;
        if (c < pieces) {
            if (tail8 && c == pieces - 1) *reinterpret_cast<uint2 *>(dst + c) = *reinterpret_cast<const uint2 *>(st + c * 16);
            else dst[c] = *reinterpret_cast<const uint4 *>(st + c * 16);
        }
    }
}

// One lane per game: deal from the seed (MiniDeck(seed), team_mini_scopa_game.py:31-34), then uniform-random legal play.
__global__ void __launch_bounds__(64)
k_team_random_playouts(const int64_t *__restrict__ seeds, long long n, int8_t *__restrict__ r2_team0, uint8_t *__restrict__ scopas,
                       uint32_t seed_lo, uint32_t seed_hi) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    scopa::Mt g;
    const int64_t sd = seeds[i];
    g.seed(sd < 0 ? (uint64_t)0 - (uint64_t)sd : (uint64_t)sd);
    uint8_t perm[16];
    g.shuffle(perm, 16);
    scopa_team_state s;
    state_init(s, perm);
    uint32_t w[10];                                                   // the state as words (scopa_team_rules.h): no run-time index into the struct
    memcpy(w, &s, 40);
    for (int ply = 0; ply < kPlies && !(w[9] & ((uint32_t)kTerminal << 24)); ply++) {
        const uint32_t seat = (w[9] >> 8) & 3u;
        const int nl = (int)((w[7] >> (8u * seat)) & 255u);
        const uint32_t hand = (((seat & 2u) ? w[4] : w[3]) >> (16u * (seat & 1u))) & 0xFFFFu;
        const scopa::philox_out x = scopa::philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)ply, 48u, seed_lo, seed_hi);
        int k = (int)(scopa::u53(x.x0, x.x1) * (double)(nl > 0 ? nl : 1));
        k = k < nl - 1 ? k : (nl > 0 ? nl - 1 : 0);
        step_words(w, nl > 0 ? nib(hand, k) : 0);
    }
    memcpy(&s, w, 40);
    r2_team0[i] = (int8_t)r2_team0_of(s);
    *reinterpret_cast<uint32_t *>(scopas + i * 4) = w[8];
}

extern "C" {

int32_t scopa_team_state_init(const uint8_t perm16[16], scopa_team_state *out) {
    if (!perm16 || !out) return SCOPA_EINVAL;
    uint32_t seen = 0;
    for (int i = 0; i < 16; i++) { if (perm16[i] > 15) return SCOPA_EINVAL; seen |= 1u << perm16[i]; }
    if (seen != 0xFFFFu) return SCOPA_EINVAL;
    state_init(*out, perm16);
    return SCOPA_OK;
}

int32_t scopa_team_state_step(scopa_team_state *s, int32_t action) {
    if (!s || action < 0 || action > 15) return SCOPA_EINVAL;
    step(*s, action);
    return (s->flags & kTableOverflow) ? SCOPA_ELIMIT : SCOPA_OK;
}

int32_t scopa_team_state_legal(const scopa_team_state *s, int32_t out[4], int32_t *n) {
    if (!s || !out || !n) return SCOPA_EINVAL;
    int tmp[4];
    *n = legal(*s, tmp);
    for (int i = 0; i < *n; i++) out[i] = tmp[i];
    return SCOPA_OK;
}

int32_t scopa_team_state_rewards_x2(const scopa_team_state *s, int32_t r2_seat[4]) {
    if (!s || !r2_seat) return SCOPA_EINVAL;
    const int r = r2_team0_of(*s);
    r2_seat[0] = r2_seat[1] = r; r2_seat[2] = r2_seat[3] = -r;
    return SCOPA_OK;
}

int32_t scopa_team_state_infoset_string(const scopa_team_state *s, int32_t team, char *buf, int32_t cap) {
    // information_state_string (openspiel_team_mini_scopa.py:119-146): the seat to move if it is on `team`, else the team's
    // first seat; hand and table SORTED by (rank, suit name); the whole action history closes the key
    if (!s || !buf || cap < 160 || team < 0 || team > 1) return SCOPA_EINVAL;
    static const char *suit_name[4] = {"cuori", "fiori", "picche", "bello"};
    int seat = seat_to_move(*s);
    if ((seat >> 1) != team) seat = team * 2;
    auto key = [&](int c) { return std::make_pair(card_rank(c), std::string(suit_name[c / 4])); };
    std::vector<int> h, t;
    for (int i = 0; i < s->nh[seat]; i++) h.push_back(nib(s->hand[seat], i));
    for (int i = 0; i < s->nt; i++) t.push_back(nib(s->table, i));
    auto cmp = [&](int a, int b) { return key(a) < key(b); };
    std::sort(h.begin(), h.end(), cmp);
    std::sort(t.begin(), t.end(), cmp);
    char *w = buf;
    w += sprintf(w, "Team%d:P%d:H[", team, seat);
    for (size_t i = 0; i < h.size(); i++) w += sprintf(w, "%s%d%c", i ? "-" : "", card_rank(h[i]), suit_name[h[i] / 4][0]);
    w += sprintf(w, "]:T[");
    for (size_t i = 0; i < t.size(); i++) w += sprintf(w, "%s%d%c", i ? "-" : "", card_rank(t[i]), suit_name[t[i] / 4][0]);
    w += sprintf(w, "]:A[");
    for (int i = 0; i < s->step; i++) w += sprintf(w, "%s%d", i ? "-" : "", (int)((s->history >> (4 * i)) & 15u));
    w += sprintf(w, "]");
    return (int32_t)(w - buf);
}

int32_t scopa_team_step_batch(scopa_ctx *ctx, scopa_team_state *d_states, const uint8_t *d_actions, int64_t n) {
    if (!ctx || n < 0 || (n && (!d_states || !d_actions))) return SCOPA_EINVAL;
    if (!n) return SCOPA_OK;
    SC_REQUIRE(ctx, ((uintptr_t)d_states & 15) == 0, SCOPA_EINVAL, "scopa_team_step_batch: states must be 16-byte aligned");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_team_step_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_states, d_actions, (long long)n);
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_team_step_batch_host(scopa_ctx *ctx, scopa_team_state *h_states, const uint8_t *h_actions, int64_t n) {
    if (!ctx || n < 0 || (n && (!h_states || !h_actions))) return SCOPA_EINVAL;
    if (!n) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_s = nullptr, *d_a = nullptr;
    int32_t rc = SCOPA_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_s, (size_t)n * sizeof(scopa_team_state))) != hipSuccess || (e = hipMalloc(&d_a, (size_t)n)) != hipSuccess)
        rc = fail(ctx, SCOPA_EHIP, "hipMalloc", e);
    if (rc == SCOPA_OK && ((e = hipMemcpyAsync(d_s, h_states, (size_t)n * sizeof(scopa_team_state), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
                           (e = hipMemcpyAsync(d_a, h_actions, (size_t)n, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess))
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(H2D)", e);
    if (rc == SCOPA_OK) rc = scopa_team_step_batch(ctx, (scopa_team_state *)d_s, (const uint8_t *)d_a, n);
    if (rc == SCOPA_OK && (e = hipMemcpyAsync(h_states, d_s, (size_t)n * sizeof(scopa_team_state), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(D2H)", e);
    e = hipStreamSynchronize(ctx->stream);
    if (rc == SCOPA_OK && e != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipStreamSynchronize", e);
    if (d_s) (void)hipFree(d_s);
    if (d_a) (void)hipFree(d_a);
    return rc;
}

int32_t scopa_team_random_playouts(scopa_ctx *ctx, const int64_t *h_seeds, int64_t n_games, int8_t *h_r2_team0, uint8_t *h_scopas) {
    if (!ctx || n_games < 0 || (n_games && (!h_seeds || !h_r2_team0 || !h_scopas))) return SCOPA_EINVAL;
    if (!n_games) return SCOPA_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_seed = nullptr, *d_r = nullptr, *d_sc = nullptr;
    int32_t rc = SCOPA_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_seed, (size_t)n_games * 8)) != hipSuccess || (e = hipMalloc(&d_r, (size_t)n_games)) != hipSuccess ||
        (e = hipMalloc(&d_sc, (size_t)n_games * 4)) != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipMalloc", e);
    if (rc == SCOPA_OK && (e = hipMemcpyAsync(d_seed, h_seeds, (size_t)n_games * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(H2D)", e);
    if (rc == SCOPA_OK) {
        hipLaunchKernelGGL(k_team_random_playouts, dim3((unsigned)((n_games + 63) / 64)), dim3(64), 0, ctx->stream, (const int64_t *)d_seed,
                           (long long)n_games, (int8_t *)d_r, (uint8_t *)d_sc, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32));
        if ((e = hipGetLastError()) != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "k_team_random_playouts", e);
    }
    if (rc == SCOPA_OK && ((e = hipMemcpyAsync(h_r2_team0, d_r, (size_t)n_games, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
                           (e = hipMemcpyAsync(h_scopas, d_sc, (size_t)n_games * 4, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess))
        rc = fail(ctx, SCOPA_EHIP, "hipMemcpyAsync(D2H)", e);
    e = hipStreamSynchronize(ctx->stream);
    if (rc == SCOPA_OK && e != hipSuccess) rc = fail(ctx, SCOPA_EHIP, "hipStreamSynchronize", e);
    if (d_seed) (void)hipFree(d_seed);
    if (d_r) (void)hipFree(d_r);
    if (d_sc) (void)hipFree(d_sc);
    return rc;
}

}  // extern "C"
