// scopa_rules.h -- MiniScopa rules on the packed 16-byte state: one source for the host glue and the
// gfx950 kernels.
//
// Reference behaviour (paths relative to the reference repo): src/envs/mini_scopa_game.py (rules),
// src/envs/openspiel_mini_scopa.py:17-95 (legal actions, infoset identity).  Written for registers:
// ordered card lists are nibble strings, the capture rule's subset-sum table is a bitmask DP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

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
// true if the predicate holds in ANY lane of the wavefront (device) / for this state (host): lets a loop over table positions stop, wave-uniformly, where
// no game of the wavefront has a card left -- games advanced in lockstep have tables of similar length, and a uniform branch costs one scalar instruction
SC_HD bool sc_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p) != 0ull;
#else
    return p;
#endif
}
SC_HD uint32_t capture_mask(uint32_t table, int nt, int target) {
    if (nt == 0 || target <= 0) return 0u;
    uint32_t ranks = 0u;       // the table as a nibble list of RANKS (2..10), zero beyond nt
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (!sc_any(i < nt)) break;
        ranks |= (uint32_t)card_rank(nib(table, i)) << (4 * i);
    }
    ranks &= nt >= 8 ? 0xFFFFFFFFu : ((1u << (4 * nt)) - 1u);   // (unused table nibbles are 0 = a card id: masked here instead of a test per card)
    {   // (1) the first table card of the played rank: the lowest zero nibble of ranks ^ (target in every nibble) -- nibbles beyond nt hold `target` != 0
        const uint32_t x = ranks ^ ((uint32_t)target * 0x11111111u);
        const uint32_t z = (x - 0x11111111u) & ~x & 0x88888888u;   // bit 4 i + 3 set for the lowest zero nibble i (higher ones may be false, the lowest never is)
        if (z) return 1u << (sc_ctz32(z) >> 2);
    }
    // (2) the reference's subset-sum: comb_sums[s] is written once, when card i first makes s reachable, as comb_sums[s - rank_i] (as it stood BEFORE card i) +
    // [i].  Kept here: the reachable set and, per sum, the INDEX of the card that first reached it, as three bit planes over the sums (bit s of F_b = bit b of
    // that index) -- a card updates all sums at once and ORs its fresh sums into the planes its (compile-time) index has set: no loop over the fresh sums, no
    // branch (a rank of 0 beyond the table's end makes `fresh` empty).  The subset of `target` is then read back along the chain -- card first[target], then the
    // subset of target - rank, which was complete before that card -- instead of carrying an 8-bit subset per sum through every card.
    uint32_t valid = 1u;       // bit s: sum s reachable
    uint32_t f0 = 0u, f1 = 0u, f2 = 0u;
    const uint32_t upto = (2u << target) - 1u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (!sc_any(i < nt)) break;
        const uint32_t r = (ranks >> (4 * i)) & 15u;
        const uint32_t fresh = (valid << r) & ~valid & upto;
        valid |= fresh;
        if (i & 1) f0 |= fresh;
        if (i & 2) f1 |= fresh;
        if (i & 4) f2 |= fresh;
    }
    if (!((valid >> target) & 1u)) return 0u;
    uint32_t mask = 0u;
    uint32_t rem = (uint32_t)target;
#pragma unroll
    for (int step = 0; step < 5; step++) {   // at most five cards (ranks >= 2, target <= 10); steps past the end of the chain (rem == 0) add nothing
        if (!sc_any(rem != 0u)) break;
        const uint32_t i = ((f0 >> rem) & 1u) | (((f1 >> rem) & 1u) << 1) | (((f2 >> rem) & 1u) << 2);
        const bool on = rem != 0u;
        mask |= on ? (1u << i) : 0u;
        rem -= on ? ((ranks >> (4 * i)) & 15u) : 0u;
    }
    return mask;
}

// ---- MiniScopaEnv.step + MiniScopaGame.play_card, mini_scopa_game.py:93-104,140-167 ------------------------
// On the state's four 32-bit words (x = hand[0] | hand[1] << 16, y = table, z = nh[0] | nh[1] << 8 | nt << 16 | step << 24, w = ncap[0] | ncap[1] << 8 |
// scopas[0] << 16 | scopas[1] << 24): the mover's fields are picked with shifts by 16 p / 8 p.  (Indexing the struct's arrays with the run-time mover sent the
// whole state to LDS in the kernels -- ten DS accesses per step; round 4.)
SC_HD void step_words(uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &w, int action) {
    const uint32_t stepb = z >> 24, nt = (z >> 16) & 255u;
    if (((z & 0xFFFFu) == 0u) || ((stepb & SCOPA_STEP_COUNT_MASK) >= ((stepb & SCOPA_STEP_CLONED) ? 16u : 8u))) return;  // terminal: _was_dead_step (:141-143)
    const uint32_t p = stepb & 1u;
    const uint32_t hand = (x >> (16u * p)) & 0xFFFFu;
    const int nh = (int)((z >> (8u * p)) & 255u);
    int pos = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--)  // first card in hand order that is `action` (:155)
        if (i < nh && nib(hand, i) == action) pos = i;
    if (pos >= 0) {
        const uint32_t cap = capture_mask(y, (int)nt, card_rank(action));
        if (cap) {
            uint32_t nt_new = 0, tab = 0;
            const uint32_t keep = ~cap & (nt >= 8u ? 0xFFu : ((1u << nt) - 1u));   // table positions that stay, in order
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (!sc_any(i < (int)nt)) break;
                const uint32_t k = (keep >> i) & 1u;
                tab |= (k ? (uint32_t)nib(y, i) : 0u) << (4 * nt_new);
                nt_new += k;
            }
            y = tab;
            const uint32_t mine = w >> (8u * p);                         // the mover's bytes: ncap at bits 0-7, scopas at bits 16-23
            const uint32_t ncap = (mine + (nt - nt_new) + 1u) & 255u;    // captured + [card] (:98); uint8 arithmetic, as the struct's fields
            const uint32_t scop = ((mine >> 16) + (nt_new == 0u ? 1u : 0u)) & 255u;   // (:100-101), last ply included
            w = (w & ~(0x00FF00FFu << (8u * p))) | ((ncap | (scop << 16)) << (8u * p));
            z = (z & ~(255u << 16)) | (nt_new << 16);
        } else if (nt < 8u) {
            y |= (uint32_t)action << (4u * nt);                         // (:103)
            z += 1u << 16;
        }   // a ninth table card cannot happen in play (table ranks are distinct and an equal rank always captures); a crafted
            // state that would need it keeps its eight cards -- the packed state has no ninth slot and no error channel
        x = (x & ~(0xFFFFu << (16u * p))) | (nib_remove(hand, pos) << (16u * p));   // (:104)
        z -= 1u << (8u * p);                                            // nh[p]--
    }  // else: card not in hand -> silent no-op that still consumes the turn (:155-159)
    z += 1u << 24;                                                      // step_count++ (the clone bit above it is never reached: counts stop at 16)
}

SC_HD void step(scopa_state &s, int action) {
    uint32_t v[4];
    memcpy(v, &s, 16);
    step_words(v[0], v[1], v[2], v[3], action);
    memcpy(&s, v, 16);
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
