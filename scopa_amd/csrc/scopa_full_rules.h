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

namespace scopa_full {

constexpr int kMaxTable = 20;  // 2 x 10 six-bit slots

SCF_HD int rank_of(int c) { return c % 10 + 1; }
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
// 1, 2, 3, ... and play_card takes the first).  Smallest integer = decide bits from the top: leave card i out whenever
// the remaining sum is still reachable with the cards below it.
// pre[i] = sums reachable with table cards 0..i-1 (bit k = sum k, sums above 10 dropped), built ONCE in a pass up the table and read in
// the pass down (first form: the reachable set recomputed from card 0 at every step of the way down -- quadratic in the table, behind
// divisions by 10 for the slot of every card it touched).  Both loops are unrolled over the table's twenty slots, so slot shifts are
// constants and pre[] / rk[] live in registers; iterations above every lane's table length are skipped by the guard.
SCF_HD uint32_t capture_mask(const scopa_full_state &s, int target) {
    const int nt = s.nt;
    if (nt == 0) return 0u;
    uint32_t pre[kMaxTable + 1];
    int rk[kMaxTable];
    pre[0] = 1u;
    int match = -1;
#pragma unroll
    for (int i = 0; i < kMaxTable; i++) {
        rk[i] = 0;
        pre[i + 1] = pre[i];
        if (i < nt) {
            const int r = rank_of(tab_get(s, i));
            rk[i] = r;
            if (r == target && match < 0) match = i;
            pre[i + 1] = pre[i] | ((pre[i] << r) & 0x7FFu);
        }
    }
    if (match >= 0) return 1u << match;
    if (!((pre[kMaxTable] >> target) & 1u)) return 0u;    // (pre[kMaxTable] == pre[nt])
    uint32_t mask = 0u;
    int rem = target;
#pragma unroll
    for (int i = kMaxTable - 1; i >= 0; i--) {
        if (i < nt && rem > 0 && !((pre[i] >> rem) & 1u)) {   // not reachable without card i: it is in
            mask |= 1u << i;
            rem -= rk[i];
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
            scopa_full_state t = s;
            int nk = 0;
            t.table[0] = t.table[1] = 0;
#pragma unroll
            for (int i = 0; i < kMaxTable; i++) {
                if (i < s.nt) {
                    const int c = tab_get(s, i);
                    if ((cm >> i) & 1u) cap_add(t, p, 1ull << c); else tab_set(t, nk++, c);
                }
            }
            cap_add(t, p, 1ull << action);
            t.nt = (uint8_t)nk;
            t.last_capture = (uint8_t)p;
            if (nk == 0) { t.scopas[0] = (uint8_t)(t.scopas[0] + (p ? 0 : 1)); t.scopas[1] = (uint8_t)(t.scopas[1] + (p ? 1 : 0)); }
            s = t;
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
