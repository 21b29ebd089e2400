// scopa_team_rules.h -- Team MiniScopa TPI (2 teams x 2 seats, the 16-card deck, 16 plies) on a packed 40-byte state; one
// source for the host protocol and the gfx950 kernels.
//
// Reference behaviour: src/envs/team_mini_scopa_game.py (rules; the capture rule is MiniScopa's, :84-109) and
// src/envs/openspiel_team_mini_scopa.py:10-170 (the two-coordinator view: current player = TEAM of the seat to move, legal
// actions = that seat's hand in hand order, information-state string with sorted cards and the full action history).
// Seats 0,1 form team 0 and seats 2,3 team 1 (:60-65); seats move in order 0,1,2,3 (:201).
#pragma once
#include "scopa_rules.h"

namespace scopa_team {
using scopa::card_rank;
using scopa::nib;

constexpr int kSeats = 4, kPlies = 16;
constexpr uint8_t kTerminal = 1, kTableOverflow = 2;

SC_HD bool is_terminal(const scopa_team_state &s) { return (s.flags & kTerminal) != 0; }
SC_HD int seat_to_move(const scopa_team_state &s) { return s.step & 3; }
SC_HD int current_player(const scopa_team_state &s) { return is_terminal(s) ? -4 : (seat_to_move(s) >> 1); }  // openspiel_team…:23-29
SC_HD int popc16(uint32_t x) { x = x - ((x >> 1) & 0x5555u); x = (x & 0x3333u) + ((x >> 2) & 0x3333u); x = (x + (x >> 4)) & 0x0F0Fu; return (int)((x + (x >> 8)) & 0x1Fu); }

SC_HD void state_init(scopa_team_state &s, const uint8_t *perm16) {  // TeamMiniScopaGame.reset (:68-78)
    s.history = 0; s.table = 0;
    for (int p = 0; p < 4; p++) {
        s.hand[p] = (uint16_t)(perm16[4 * p] | (perm16[4 * p + 1] << 4) | (perm16[4 * p + 2] << 8) | (perm16[4 * p + 3] << 12));
        s.cap[p] = 0; s.nh[p] = 4; s.scopas[p] = 0;
    }
    s.nt = 0; s.step = 0; s.last_capture_team = 0xFF; s.flags = 0;
}

// legal_actions (openspiel_team…:52-86): the seat to move's hand in hand order; [0] if it is empty; none at terminal
SC_HD int legal(const scopa_team_state &s, int out[4]) {
    if (is_terminal(s)) return 0;
    const int seat = seat_to_move(s), n = s.nh[seat];
    for (int i = 0; i < n; i++) out[i] = nib(s.hand[seat], i);
    if (n == 0) { out[0] = 0; return 1; }
    return n;
}

// evaluate_game (:125-155) times two: team score = sum(len(captures) + 2*scopas); r2[team] = 2*score - total.
// (The leftover table cards were already added to cap[] of the last capturing team's first seat when the game ended.)
SC_HD int r2_team0_of(const scopa_team_state &s) {
    if (!is_terminal(s)) return 0;
    const int t0 = popc16(s.cap[0]) + popc16(s.cap[1]) + 2 * (s.scopas[0] + s.scopas[1]);
    const int t1 = popc16(s.cap[2]) + popc16(s.cap[3]) + 2 * (s.scopas[2] + s.scopas[3]);
    return t0 - t1;   // total == 0 gives 0 like the reference's special case
}

// TPIMiniScopaState.apply_action (:88-92) + TeamMiniScopaEnv.step (:173-201) + play_card (:111-123), on the state's ten 32-bit words:
//   w0, w1 history | w2 table | w3 = hand[0] | hand[1] << 16, w4 = hand[2] | hand[3] << 16 | w5, w6 = cap likewise | w7 = nh[0..3], a byte each |
//   w8 = scopas[0..3] | w9 = nt | step << 8 | last_capture_team << 16 | flags << 24.
// The seat's fields are picked with selects and shifts (indexing the struct's arrays with the run-time seat had sent the whole state to LDS in
// k_team_step_batch, a second staging buffer's worth; round 4, as scopa::step_words).
SC_HD void step_words(uint32_t (&w)[10], int action) {
    if (w[9] & ((uint32_t)kTerminal << 24)) return;  // _was_dead_step; the packed history holds the 16 plies of a game only
    action &= 15;
    const uint32_t stepc = (w[9] >> 8) & 255u, seat = stepc & 3u;
    uint32_t nt = w[9] & 255u;
    const uint32_t hbits = (uint32_t)action << (4u * (stepc & 7u));
    w[0] |= (stepc & 8u) ? 0u : hbits;
    w[1] |= (stepc & 8u) ? hbits : 0u;
    const bool hi = (seat & 2u) != 0u;
    const uint32_t sh16 = 16u * (seat & 1u), sh8 = 8u * seat;
    const uint32_t hand = ((hi ? w[4] : w[3]) >> sh16) & 0xFFFFu;
    const int nh = (int)((w[7] >> sh8) & 255u);
    int pos = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--)
        if (i < nh && nib(hand, i) == action) pos = i;
    if (pos >= 0) {
        // the table never holds two cards of one rank (an equal rank always captures), hence at most 8 cards: MiniScopa's
        // 8-slot capture rule applies unchanged
        const uint32_t cap = scopa::capture_mask(w[2], (int)nt, card_rank(action));
        if (cap) {
            uint32_t nt_new = 0, tab = 0, taken = 1u << action;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (!scopa::sc_any(i < (int)nt)) break;
                if (i < (int)nt) {
                    const uint32_t c = (uint32_t)nib(w[2], i);
                    if ((cap >> i) & 1u) taken |= 1u << c;
                    else { tab |= c << (4 * nt_new); nt_new++; }
                }
            }
            w[5] |= hi ? 0u : taken << sh16;
            w[6] |= hi ? taken << sh16 : 0u;
            w[2] = tab; nt = nt_new;
            w[9] = (w[9] & ~(255u << 16)) | ((seat >> 1) << 16);                                          // last_capture_team
            if (nt_new == 0) w[8] = (w[8] & ~(255u << sh8)) | ((((w[8] >> sh8) + 1u) & 255u) << sh8);     // scopas[seat]++ (uint8)
        } else if (nt < 8u) {
            w[2] |= (uint32_t)action << (4u * nt);
            nt++;
        } else w[9] |= (uint32_t)kTableOverflow << 24;  // unreachable by the argument above; kept loud rather than silent
        const uint32_t left = scopa::nib_remove(hand, pos) & 0xFFFFu;
        if (hi) w[4] = (w[4] & ~(0xFFFFu << sh16)) | (left << sh16);
        else    w[3] = (w[3] & ~(0xFFFFu << sh16)) | (left << sh16);
        w[7] -= 1u << sh8;                                                                                 // nh[seat]-- (nh >= 1 here)
    }  // else: card not in hand -> silent no-op that still consumes the turn (:184-186)
    const uint32_t step_new = (stepc + 1u) & 255u;
    w[9] = (w[9] & 0xFFFF0000u) | (step_new << 8) | (nt & 255u);
    if (w[7] == 0u || step_new >= (uint32_t)kPlies) {
        w[9] |= (uint32_t)kTerminal << 24;
        const uint32_t lct = (w[9] >> 16) & 255u;
        if (nt > 0u && lct != 0xFFu) {  // leftovers to the first seat of the last capturing team (:133-139)
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (!scopa::sc_any(i < (int)nt)) break;
                if (i < (int)nt) m |= 1u << nib(w[2], i);
            }
            w[5] |= (lct & 1u) ? 0u : m;
            w[6] |= (lct & 1u) ? m : 0u;
        }
    }
}

SC_HD void step(scopa_team_state &s, int action) {
    uint32_t w[10];
    memcpy(w, &s, 40);
    step_words(w, action);
    memcpy(&s, w, 40);
}

}  // namespace scopa_team
