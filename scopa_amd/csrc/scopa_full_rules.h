// scopa_full_rules.h -- FullScopa (40-card Italian deck, 2 players, 3-card hands with re-deals) on a packed 64-byte state.
//
// Reference behaviour: src/envs/full_scopa_game.py (FullDeck, FullScopaGame, FullScopaEnv.step), legal actions and
// infoset identity from src/envs/openspiel_full_scopa.py:22-94.  Card id = action id = suit_idx*10 + rank-1 with suits
// denari, coppe, spade, bastoni (full_scopa_game.py:20-21, 262-266).  The table is an ORDERED list (the capture rule
// depends on table order); captured cards only matter as sets, so they are 40-bit masks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scopa.h"

#define SCF_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define SCF_UNROLL _Pragma("unroll")     /* the table loops: slot shifts become constants on the device; the host keeps them as loops */
#else
#define SCF_UNROLL
#endif

namespace scopa_full {

constexpr int kMaxTable = 20;  // 2 x 10 six-bit slots

SCF_HD int rank_of(int c) { return c % 10 + 1; }
// true if the predicate holds in ANY lane of the wavefront (device) / for this state (host): the loops over the table's twenty slots stop, wave-uniformly,
// above the longest table of the wavefront's games (tables hold 4-6 cards in play, and games advanced in lockstep have tables of similar length)
SCF_HD bool any_lane(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p) != 0ull;
#else
    return p;
#endif
}
// The two-element arrays of the state are never indexed with a run-time value: a run-time index into a struct member sends the whole state
// to scratch memory on the device (84 scratch accesses per step in the first form of k_full_step_batch); a select between the two words keeps
// it in registers.
SCF_HD int tab_get(const scopa_full_state &s, int i) {
    const int w = i >= 10;
    return (int)(((w ? s.table[1] : s.table[0]) >> (6 * (i - 10 * w))) & 63u);
}
SCF_HD void tab_set(scopa_full_state &s, int i, int c) {
    const int w = i >= 10, sh = 6 * (i - 10 * w);
    const uint64_t keep = ~(63ull << sh), put = (uint64_t)c << sh;
    s.table[0] = w ? s.table[0] : (s.table[0] & keep) | put;
    s.table[1] = w ? (s.table[1] & keep) | put : s.table[1];
}
SCF_HD uint32_t hand_of(const scopa_full_state &s, int p) { return p ? s.hand[1] : s.hand[0]; }
SCF_HD int nh_of(const scopa_full_state &s, int p) { return p ? s.nh[1] : s.nh[0]; }
SCF_HD int hand_get(const scopa_full_state &s, int p, int i) { return (int)((hand_of(s, p) >> (6 * i)) & 63u); }
SCF_HD void cap_add(scopa_full_state &s, int p, uint64_t bits) { s.cap[0] |= p ? 0ull : bits; s.cap[1] |= p ? bits : 0ull; }   // (both words written: an if / else the compiler folds back into an indexed access)

SCF_HD void deal_round(scopa_full_state &s, const uint8_t *deck) {  // 3 cards to player 0, then 3 to player 1
#pragma unroll
    for (int p = 0; p < 2; p++) {
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) h |= (uint32_t)deck[s.deck_pos++] << (6 * i);
        s.hand[p] = h;
        s.nh[p] = 3;
    }
}

SCF_HD void state_init(scopa_full_state &s, const uint8_t *deck, uint32_t game) {  // FullScopaGame.reset (:60-75)
    s.table[0] = s.table[1] = 0; s.cap[0] = s.cap[1] = 0; s.hand[0] = s.hand[1] = 0;
    s.game = game; s.nh[0] = s.nh[1] = 0; s.nt = 0; s.deck_pos = 0; s.round = 0; s.last_capture = 0xFF;
    s.scopas[0] = s.scopas[1] = 0; s.step = 0; s.terminal = 0; s.r2_p0 = 0; s.flags = 0;
    for (int i = 0; i < 7; i++) s.pad[i] = 0;
    for (int i = 0; i < 4; i++) tab_set(s, s.nt++, deck[s.deck_pos++]);
    deal_round(s, deck);
}

// find_capture_combinations()[0] (:90-118): the first table card of the played rank if any; else the subset whose
// membership mask is the SMALLEST integer among all subsets summing to the rank (the reference enumerates masks
// 1, 2, 3, ... and play_card takes the first).
// One pass up the table: a subset that holds card i has a larger mask than any subset of the cards below it, so the smallest mask for a sum is
// the one found FIRST when the cards are added in table order (0/1 knapsack over the sums 0..10, kept as the 11-bit set `reach`), and it is the
// card that first reached the sum plus the smallest mask of the rest, which was fixed earlier.  The index of that card is kept as five bit planes
// over the sums (f[b] bit k = bit b of the index; the loop is unrolled, so which planes a card writes is a constant) and the subset is read back
// from the played rank down, a card per step.  (First forms: the reachable sets of every prefix in twenty registers and a second pass down the
// table; before that the set recomputed from card 0 at every step down.)  Slots above every lane's table are skipped wave-uniformly (any_lane).
SCF_HD uint32_t capture_mask(const scopa_full_state &s, int target) {
    const int nt = s.nt;
    if (nt == 0) return 0u;
    uint32_t reach = 1u, eq = 0u;                         // sums reachable so far; table cards of the played rank
    uint32_t f[5] = {0u, 0u, 0u, 0u, 0u};
    uint32_t rk[3] = {0u, 0u, 0u};                        // the ranks, a nibble per slot
SCF_UNROLL
    for (int i = 0; i < kMaxTable; i++) {
        if (!any_lane(i < nt)) break;
        const int c = tab_get(s, i);
        const uint32_t r = i < nt ? (uint32_t)(c - 10 * ((c * 26) >> 8) + 1) : 0u;   // rank_of for c < 64; a slot above the table counts as rank 0: no match, no new sum
        eq |= (uint32_t)(r == (uint32_t)target) << i;
        const uint32_t fresh = (reach << r) & ~reach & 0x7FFu;
        reach |= fresh;
SCF_UNROLL
        for (int b = 0; b < 5; b++) if ((i >> b) & 1) f[b] |= fresh;
        rk[i >> 3] |= r << (4 * (i & 7));
    }
    if (eq) return eq & (0u - eq);                        // the first of them
    if (!((reach >> target) & 1u)) return 0u;
    uint32_t mask = 0u;
    int rem = target;
    for (int k = 0; k < 10; k++) {                        // ranks are >= 1: ten cards at most
        if (!any_lane(rem > 0)) break;
        if (rem > 0) {
            const uint32_t idx = ((f[0] >> rem) & 1u) | (((f[1] >> rem) & 1u) << 1) | (((f[2] >> rem) & 1u) << 2) | (((f[3] >> rem) & 1u) << 3) | (((f[4] >> rem) & 1u) << 4);
            mask |= 1u << idx;
            const uint32_t w = idx >= 16u ? rk[2] : (idx >= 8u ? rk[1] : rk[0]);
            rem -= (int)((w >> (4u * (idx & 7u))) & 15u);
        }
    }
    return mask;
}

SCF_HD int primiera(uint64_t cap) {  // calculate_primiera_score (:152-164); values 7:21 6:18 1:16 5:15 4:14 3:13 2:12 8,9,10:10
    int total = 0;
    for (int su = 0; su < 4; su++) {
        const uint32_t b = (uint32_t)((cap >> (10 * su)) & 0x3FFu);
        int v = 0;
        if (b & (1u << 6)) v = 21; else if (b & (1u << 5)) v = 18; else if (b & 1u) v = 16; else if (b & (1u << 4)) v = 15;
        else if (b & (1u << 3)) v = 14; else if (b & (1u << 2)) v = 13; else if (b & (1u << 1)) v = 12; else if (b & 0x380u) v = 10;
        if (!v) return 0;  // a suit is missing: no primiera
        total += v;
    }
    return total;
}

SCF_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

SCF_HD void evaluate(scopa_full_state &s) {  // evaluate_game (:166-228)
    if (s.nt > 0 && s.last_capture != 0xFF)
    {
        uint64_t left = 0ull;                                  // the cards left on the table go to the last capturer (:170-173)
        for (int i = 0; i < s.nt; i++) left |= 1ull << tab_get(s, i);
        cap_add(s, s.last_capture, left);
    }
    int sc[2] = {0, 0};
    const int c0 = popc64(s.cap[0]), c1 = popc64(s.cap[1]);
    if (c0 != c1) sc[c0 > c1 ? 0 : 1]++;
    const int d0 = popc64(s.cap[0] & 0x3FFull), d1 = popc64(s.cap[1] & 0x3FFull);
    if (d0 != d1) sc[d0 > d1 ? 0 : 1]++;
    if (s.cap[0] & (1ull << 6)) sc[0]++; else if (s.cap[1] & (1ull << 6)) sc[1]++;   // sette bello
    const int p0 = primiera(s.cap[0]), p1 = primiera(s.cap[1]);
    if (p0 != p1) sc[p0 > p1 ? 0 : 1]++;                                             // both 0 or a tie: nobody
    sc[0] += s.scopas[0]; sc[1] += s.scopas[1];
    s.r2_p0 = (int8_t)(sc[0] - sc[1]);  // 2*(s0 - (s0+s1)/2); total == 0 gives 0 as the reference's special case
    s.terminal = 1;
}

SCF_HD void step(scopa_full_state &s, const uint8_t *deck, int action) {  // FullScopaEnv.step (:252-297)
    if (s.terminal) return;
    const int p = s.step & 1;
    int pos = -1;
#pragma unroll
    for (int i = 2; i >= 0; i--) if (i < nh_of(s, p) && hand_get(s, p, i) == action) pos = i;
    if (pos >= 0) {  // play_card (:120-150), capture_choice None
        const uint32_t cm = capture_mask(s, rank_of(action));
        if (cm) {
            uint64_t taken = 1ull << action, t0 = 0ull, t1 = 0ull;   // captured cards + the played one; the table that stays, in order
            int nk = 0;
SCF_UNROLL
            for (int i = 0; i < kMaxTable; i++) {
                if (!any_lane(i < s.nt)) break;
                if (i < s.nt) {
                    const int c = tab_get(s, i);
                    if ((cm >> i) & 1u) taken |= 1ull << c;
                    else {
                        const int w = nk >= 10;
                        const uint64_t put = (uint64_t)c << (6 * (nk - 10 * w));
                        t0 |= w ? 0ull : put; t1 |= w ? put : 0ull;
                        nk++;
                    }
                }
            }
            s.table[0] = t0; s.table[1] = t1;
            cap_add(s, p, taken);
            s.nt = (uint8_t)nk;
            s.last_capture = (uint8_t)p;
            if (nk == 0) { s.scopas[0] = (uint8_t)(s.scopas[0] + (p ? 0 : 1)); s.scopas[1] = (uint8_t)(s.scopas[1] + (p ? 1 : 0)); }
        } else if (s.nt < kMaxTable) {
            tab_set(s, s.nt, action);
            s.nt++;
        } else {
            s.flags |= 1u;  // table capacity exceeded (never seen in play: the reference's tables stay below 12)
        }
        const uint32_t h = hand_of(s, p), lo = h & ((1u << (6 * pos)) - 1u), hi = (h >> (6 * (pos + 1))) << (6 * pos);
        s.hand[0] = p ? s.hand[0] : (lo | hi); s.hand[1] = p ? (lo | hi) : s.hand[1];
        s.nh[0] = (uint8_t)(s.nh[0] - (p ? 0 : 1)); s.nh[1] = (uint8_t)(s.nh[1] - (p ? 1 : 0));
    }
    s.step++;
    if ((s.nh[0] | s.nh[1]) == 0) {
        if (40 - s.deck_pos >= 6) { deal_round(s, deck); s.round++; }   // deal_new_round (:81-88)
        else evaluate(s);
    }
    if (!s.terminal && s.step >= 200) evaluate(s);
}

SCF_HD int legal(const scopa_full_state &s, int player, int out[3]) {  // openspiel_full_scopa.py:22-42
    if (s.terminal) return 0;
    if (player < 0) player = s.step & 1;
    const int n = nh_of(s, player);
    for (int i = 0; i < n; i++) out[i] = hand_get(s, player, i);
    if (n == 0) { out[0] = 0; return 1; }
    return n;
}

}  // namespace scopa_full
