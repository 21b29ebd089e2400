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

// TPIMiniScopaState.apply_action (:88-92) + TeamMiniScopaEnv.step (:173-201) + play_card (:111-123)
SC_HD void step(scopa_team_state &s, int action) {
    if (is_terminal(s)) return;  // _was_dead_step; the packed history holds the 16 plies of a game only
    action &= 15;
    s.history |= (uint64_t)action << (4 * s.step);
    const int seat = seat_to_move(s);
    const uint32_t hand = s.hand[seat];
    const int nh = s.nh[seat];
    int pos = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--)
        if (i < nh && nib(hand, i) == action) pos = i;
    if (pos >= 0) {
        // the table never holds two cards of one rank (an equal rank always captures), hence at most 8 cards: MiniScopa's
        // 8-slot capture rule applies unchanged
        const uint32_t cap = scopa::capture_mask(s.table, s.nt, card_rank(action));
        if (cap) {
            uint32_t nt_new = 0, tab = 0, taken = 1u << action;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < s.nt) {
                    const int c = nib(s.table, i);
                    if ((cap >> i) & 1u) taken |= 1u << c;
                    else { tab |= (uint32_t)c << (4 * nt_new); nt_new++; }
                }
            s.cap[seat] = (uint16_t)(s.cap[seat] | taken);
            s.table = tab; s.nt = (uint8_t)nt_new;
            s.last_capture_team = (uint8_t)(seat >> 1);
            if (nt_new == 0) s.scopas[seat]++;
        } else if (s.nt < 8) {
            s.table |= (uint32_t)action << (4 * s.nt);
            s.nt++;
        } else s.flags |= kTableOverflow;  // unreachable by the argument above; kept loud rather than silent
        s.hand[seat] = (uint16_t)scopa::nib_remove(hand, pos);
        s.nh[seat] = (uint8_t)(nh - 1);
    }  // else: card not in hand -> silent no-op that still consumes the turn (:184-186)
    s.step++;
    if ((s.nh[0] | s.nh[1] | s.nh[2] | s.nh[3]) == 0 || s.step >= kPlies) {
        s.flags |= kTerminal;
        if (s.nt > 0 && s.last_capture_team != 0xFF) {  // leftovers to the first seat of the last capturing team (:133-139)
            uint32_t m = 0;
            for (int i = 0; i < s.nt; i++) m |= 1u << nib(s.table, i);
            s.cap[s.last_capture_team * 2] = (uint16_t)(s.cap[s.last_capture_team * 2] | m);
        }
    }
}

}  // namespace scopa_team
