// scopa_rules.h -- MiniScopa rules on the packed 16-byte state: one source for the host glue and the
// gfx950 kernels.
//
// Reference behaviour (paths relative to the reference repo): src/envs/mini_scopa_game.py (rules),
// src/envs/openspiel_mini_scopa.py:17-95 (legal actions, infoset identity).  Written for registers:
// ordered card lists are nibble strings, the capture rule's subset-sum table is a bitmask DP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scopa.h"

#define SC_HD __host__ __device__ __forceinline__

namespace scopa {

// ---- shape of every MiniScopa tree (2 players x 4 cards, mini_scopa_game.py:59,127) ----------------------
// Legal counts by ply are 4,4,3,3,2,2,1,1 whatever the deal, so the tree is regular: in BFS (level) order
// child a of the j-th node of ply d is node j*nlegal(d)+a of ply d+1.  No child array is ever stored.
constexpr int kPlies = 8;
constexpr int kNodes = SCOPA_N_NODES, kDecision = SCOPA_N_DECISION, kTerminal = SCOPA_N_TERMINAL;

SC_HD int nlegal_at(int d) { return 4 - (d >> 1); }
SC_HD int level_width(int d) {  // nodes at ply d; ply 8 = terminals
    constexpr int w[9] = {1, 4, 16, 48, 144, 288, 576, 576, 576};
    return w[d];
}
SC_HD int level_offset(int d) {  // BFS index of the first node of ply d; offset(8) = #decision nodes
    constexpr int o[10] = {0, 1, 5, 21, 69, 213, 501, 1077, 1653, 2229};
    return o[d];
}
SC_HD int subtree_size(int d) {  // nodes (incl. itself and terminals) below a node of ply d
    constexpr int s[9] = {2229, 557, 139, 46, 15, 7, 3, 2, 1};
    return s[d];
}

// ---- cards -----------------------------------------------------------------------------------------------
// rank of card id c (MiniDeck.ranks, mini_scopa_game.py:18-23): 2 5 8 10 | 2 5 7 9 | 3 6 8 9 | 3 6 7 10
constexpr uint64_t kRankLut = 0xA76398639752A852ull;
SC_HD int card_rank(int c) { return (int)((kRankLut >> (4 * c)) & 15u); }
SC_HD int nib(uint32_t list, int i) { return (int)((list >> (4 * i)) & 15u); }
// remove nibble i from a nibble list, keeping the order of the rest
SC_HD uint32_t nib_remove(uint32_t list, int i) {
    uint32_t lo = list & ((1u << (4 * i)) - 1u);
    uint32_t hi = i >= 7 ? 0u : (list >> (4 * (i + 1))) << (4 * i);
    return lo | hi;
}

// terminal iff all hands are empty or step_count >= env.max_steps (mini_scopa_game.py:160).  max_steps is num_players * 4 = 8 for a fresh
// env (:127) and 16 for the env of a MiniScopaState.clone() (openspiel_mini_scopa.py:108): the state carries that as SCOPA_STEP_CLONED in
// `step`.  Only illegal no-op actions (:155-159) can make the difference visible -- legal play empties both hands at step 8 either way --
// so every solver kernel, which plays hand cards only, sees the bit clear; parity (step & 1 = the mover) is unaffected by it.
SC_HD int step_count(const scopa_state &s) { return s.step & SCOPA_STEP_COUNT_MASK; }
SC_HD bool is_terminal(const scopa_state &s) {
    return ((s.nh[0] | s.nh[1]) == 0) || step_count(s) >= ((s.step & SCOPA_STEP_CLONED) ? 16 : 8);
}
SC_HD int current_player(const scopa_state &s) {  // openspiel_mini_scopa.py:17-20; PlayerId.TERMINAL = -4
    return is_terminal(s) ? -4 : (s.step & 1);
}

SC_HD void state_init(scopa_state &s, const uint8_t *perm16) {  // MiniScopaGame.reset, mini_scopa_game.py:56-64
    s.hand[0] = (uint16_t)(perm16[0] | (perm16[1] << 4) | (perm16[2] << 8) | (perm16[3] << 12));
    s.hand[1] = (uint16_t)(perm16[4] | (perm16[5] << 4) | (perm16[6] << 8) | (perm16[7] << 12));
    s.table = 0;
    s.nh[0] = s.nh[1] = 4;
    s.nt = 0; s.step = 0;
    s.ncap[0] = s.ncap[1] = 0;
    s.scopas[0] = s.scopas[1] = 0;
}

// ---- capture rule: card_in_table, mini_scopa_game.py:66-91 -----------------------------------------------
// Returns the bitmask of captured table positions (0 = no capture).
//  (1) any table card of the played rank -> the FIRST such card in table order, alone (:72-74);
//  (2) else 0/1 subset-sum over the table in index order; for each reachable sum the subset found FIRST is
//      kept (comb_sums[s] is written only while None, :81-85).  `valid` is the set of reachable sums, sub[s]
//      (8 bits each, packed) the subset that first reached s; one table card updates all sums at once from
//      the previous card's state, which is what the reference's descending-s inner loop computes.
SC_HD int sc_ctz32(uint32_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}
SC_HD uint32_t capture_mask(uint32_t table, int nt, int target) {
    if (nt == 0 || target <= 0) return 0u;
    uint32_t ranks = 0u;       // the table as a nibble list of RANKS (2..10), zero beyond nt
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (i < nt) ranks |= (uint32_t)card_rank(nib(table, i)) << (4 * i);
    {   // (1) the first table card of the played rank: the lowest zero nibble of ranks ^ (target in every nibble) -- nibbles beyond nt hold `target` != 0
        const uint32_t x = ranks ^ ((uint32_t)target * 0x11111111u);
        const uint32_t z = (x - 0x11111111u) & ~x & 0x88888888u;   // bit 4 i + 3 set for the lowest zero nibble i (higher ones may be false, the lowest never is)
        if (z) return 1u << (sc_ctz32(z) >> 2);
    }
    // (2) the reference's subset-sum: comb_sums[s] is written once, when card i first makes s reachable, as comb_sums[s - rank_i] (as it stood BEFORE card i) +
    // [i].  Kept here: the reachable set and, per sum, the index of the card that first reached it (a nibble each); the subset of `target` is then read
    // back along that chain -- card first[target], then the subset of target - rank, which was complete before that card -- instead of carrying an 8-bit
    // subset per sum through every card.
    uint32_t valid = 1u;       // bit s: sum s reachable
    uint64_t first = 0ull;     // nibble s: index of the card at which s became reachable
    const uint32_t upto = (2u << target) - 1u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i >= nt) break;
        const int r = (int)((ranks >> (4 * i)) & 15u);
        uint32_t fresh = (valid << r) & ~valid & upto;
        valid |= fresh;
        while (fresh) {        // one or two sums as a rule
            const int s = sc_ctz32(fresh);
            fresh &= fresh - 1u;
            first |= (uint64_t)i << (4 * s);
        }
    }
    if (!((valid >> target) & 1u)) return 0u;
    uint32_t mask = 0u;
    int rem = target;
    for (int step = 0; step < 5 && rem > 0; step++) {   // at most five cards (ranks >= 2, target <= 10)
        const int i = (int)((first >> (4 * rem)) & 15ull);
        mask |= 1u << i;
        rem -= (int)((ranks >> (4 * i)) & 15u);
    }
    return mask;
}

// ---- MiniScopaEnv.step + MiniScopaGame.play_card, mini_scopa_game.py:93-104,140-167 ------------------------
SC_HD void step(scopa_state &s, int action) {
    if (is_terminal(s)) return;  // _was_dead_step (:141-143)
    const int p = s.step & 1;
    const uint32_t hand = s.hand[p];
    const int nh = s.nh[p];
    int pos = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--)  // first card in hand order that is `action` (:155)
        if (i < nh && nib(hand, i) == action) pos = i;
    if (pos >= 0) {
        const uint32_t cap = capture_mask(s.table, s.nt, card_rank(action));
        if (cap) {
            uint32_t nt_new = 0, tab = 0;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < s.nt && !((cap >> i) & 1u)) { tab |= (uint32_t)nib(s.table, i) << (4 * nt_new); nt_new++; }
            s.ncap[p] = (uint8_t)(s.ncap[p] + (s.nt - nt_new) + 1);  // captured + [card] (:98)
            s.table = tab;
            s.nt = (uint8_t)nt_new;
            if (nt_new == 0) s.scopas[p]++;                            // (:100-101), last ply included
        } else if (s.nt < 8) {
            s.table |= (uint32_t)action << (4 * s.nt);                 // (:103)
            s.nt++;
        }   // a ninth table card cannot happen in play (table ranks are distinct and an equal rank always captures); a crafted
            // state that would need it keeps its eight cards -- the packed state has no ninth slot and no error channel

        s.hand[p] = (uint16_t)nib_remove(hand, pos);                   // (:104)
        s.nh[p] = (uint8_t)(nh - 1);
    }  // else: card not in hand -> silent no-op that still consumes the turn (:155-159)
    s.step++;
}

// evaluate_game (:106-114) times two: r2[i] = 2*r_i - total, an integer; zero until terminal.
SC_HD void rewards_x2(const scopa_state &s, int &r0x2, int &r1x2) {
    if (!is_terminal(s)) { r0x2 = r1x2 = 0; return; }
    const int r0 = s.ncap[0] + 2 * s.scopas[0], r1 = s.ncap[1] + 2 * s.scopas[1];
    r0x2 = r0 - r1;  // 2*r0 - (r0 + r1); total == 0 gives 0 as the reference's special case does
    r1x2 = r1 - r0;
}

// legal_actions (openspiel_mini_scopa.py:22-47): hand order; [0] if the hand is empty; none at terminal.
SC_HD int legal(const scopa_state &s, int player, int out[4]) {
    if (is_terminal(s)) return 0;
    if (player < 0) player = s.step & 1;
    const int n = s.nh[player];
    for (int i = 0; i < n; i++) out[i] = nib(s.hand[player], i);
    if (n == 0) { out[0] = 0; return 1; }
    return n;
}

// Infoset identity = (player, ORDERED hand, ORDERED table)  (information_state_string, openspiel…:86-95)
//   bit 0 player | bits 1-3 nh | bits 4-19 hand | bits 20-23 nt | bits 24-55 table
SC_HD uint64_t infoset_key(const scopa_state &s, int player) {
    return (uint64_t)(player & 1) | ((uint64_t)s.nh[player] << 1) | ((uint64_t)s.hand[player] << 4) |
           ((uint64_t)s.nt << 20) | ((uint64_t)s.table << 24);
}

}  // namespace scopa
