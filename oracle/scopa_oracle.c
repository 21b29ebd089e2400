/*
 * scopa_oracle.c -- CPU ORACLE (test infrastructure; see scopa_oracle.h for the rules
 * on who may call it).  Literal restatement of the reference's Python; every block cites
 * the reference lines it follows (paths relative to /root/reference/).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -lm   (no implicit FMA contraction: the
 * reference's float64 arithmetic is numpy's, one rounding per operation; the one place numpy
 * itself fuses -- np.dot through OpenBLAS -- is written as an explicit fma() chain below).
 */
#include "scopa_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ===== cards ======================================================================= */
/* MiniDeck.suits / .ranks, src/envs/mini_scopa_game.py:17-23 */
static const int  RANK_OF[16] = {2, 5, 8, 10, 2, 5, 7, 9, 3, 6, 8, 9, 3, 6, 7, 10};
static const char SUIT_CHAR[4] = {'c', 'f', 'p', 'b'}; /* cuori fiori picche bello */

int  og_card_rank(int card) { return RANK_OF[card & 15]; }
char og_card_suit_char(int card) { return SUIT_CHAR[(card >> 2) & 3]; }

/* ===== CPython random.seed(int) + random.shuffle =================================== */
/* MT19937 as in CPython's _randommodule.c: seed(int) -> init_by_array(32-bit words of |seed|) */
typedef struct { uint32_t mt[624]; int idx; } mt_t;

static void mt_init_genrand(mt_t *m, uint32_t s) {
    m->mt[0] = s;
    for (int i = 1; i < 624; i++)
        m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
static void mt_init_by_array(mt_t *m, const uint32_t *key, int klen) {
    mt_init_genrand(m, 19650218u);
    int i = 1, j = 0;
    int k = 624 > klen ? 624 : klen;
    for (; k; k--) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (k = 623; k; k--) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
    }
    m->mt[0] = 0x80000000u;
}
static uint32_t mt_u32(mt_t *m) {
    if (m->idx >= 624) {
        uint32_t *mt = m->mt;
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
/* Random._randbelow_with_getrandbits: k = n.bit_length(); r = getrandbits(k) until r < n */
static uint32_t py_randbelow(mt_t *m, uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++;
    uint32_t r;
    do { r = mt_u32(m) >> (32 - k); } while (r >= n);
    return r;
}

void og_deal_py_seed(int64_t seed, uint8_t perm[16]) {
    /* mini_scopa_game.py:25-28: cards in suit-major order, random.seed(seed), random.shuffle */
    uint64_t a = seed < 0 ? (uint64_t)(-seed) : (uint64_t)seed; /* seed(int) uses abs() */
    uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
    mt_t m;
    mt_init_by_array(&m, key, key[1] ? 2 : 1);
    for (int i = 0; i < 16; i++) perm[i] = (uint8_t)i;
    for (int i = 15; i >= 1; i--) { /* shuffle: for i in reversed(range(1, n)): j = randbelow(i+1) */
        uint32_t j = py_randbelow(&m, (uint32_t)i + 1);
        uint8_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

/* ===== game rules ================================================================== */
void og_reset(og_state *s, const uint8_t perm[16]) {
    /* MiniScopaGame.reset, mini_scopa_game.py:56-64; MiniScopaEnv.reset :131-138 */
    memset(s, 0, sizeof *s);
    for (int p = 0; p < 2; p++) {
        for (int i = 0; i < 4; i++) s->hand[p][i] = (int8_t)perm[p * 4 + i];
        s->nh[p] = 4;
    }
    memset(s->table, -1, sizeof s->table);
    s->max_steps = 8; /* MiniScopaEnv.__init__: max_steps = num_players * 4 (:127) */
}

void og_clone(const og_state *src, og_state *dst) {
    /* MiniScopaState.clone, src/envs/openspiel_mini_scopa.py:97-115: set_state(get_state()) copies hands, table, captures, scopas,
       agent_selection and step_count (mini_scopa_game.py:169-194); the new env's max_steps is the literal 16 (:108) */
    *dst = *src;
    /* A terminal state's clone stays terminal: the terminations dict and _is_terminal travel with it (mini_scopa_game.py:193,
       openspiel_mini_scopa.py:112), and step() on it is a dead step (:141-143).  og_state keeps no such flag -- terminal is recomputed
       from the fields -- so the limit that made it terminal is kept. */
    if (!og_is_terminal(src)) dst->max_steps = 16;
}

int og_is_terminal(const og_state *s) {
    /* mini_scopa_game.py:160 */
    return (s->nh[0] == 0 && s->nh[1] == 0) || s->step >= s->max_steps;
}

int og_current_player(const og_state *s) {
    /* agent_selection rotates every step (:167); TERMINAL = -4 (openspiel…:17-20) */
    return og_is_terminal(s) ? -4 : (s->step & 1);
}

int og_legal(const og_state *s, int player, int out[4]) {
    /* openspiel_mini_scopa.py:22-47 */
    if (og_is_terminal(s)) return 0;
    if (player < 0) player = og_current_player(s);
    int n = 0;
    for (int i = 0; i < s->nh[player]; i++) out[n++] = s->hand[player][i];
    if (n == 0) { out[0] = 0; n = 1; } /* "Fallback to avoid empty list" */
    return n;
}

int og_capture(const og_state *s, int card, int cap_idx[8]) {
    /* card_in_table, mini_scopa_game.py:66-91 */
    int target = RANK_OF[card];
    if (target <= 0 || s->nt == 0) return 0;
    for (int i = 0; i < s->nt; i++) /* :72-74 first exact match in table order */
        if (RANK_OF[s->table[i]] == target) { cap_idx[0] = i; return 1; }
    /* :76-85 0/1 subset-sum, first-found wins */
    int have[11];      /* comb_sums[s] is not None */
    int len[11];
    int comb[11][8];
    memset(have, 0, sizeof have);
    have[0] = 1; len[0] = 0;
    for (int idx = 0; idx < s->nt; idx++) {
        int r = RANK_OF[s->table[idx]];
        for (int sum = target; sum >= r; sum--) {
            if (!have[sum] && have[sum - r]) {
                have[sum] = 1;
                len[sum] = len[sum - r] + 1;
                memcpy(comb[sum], comb[sum - r], sizeof(int) * (size_t)len[sum - r]);
                comb[sum][len[sum] - 1] = idx;
            }
        }
    }
    if (!have[target]) return 0;
    for (int i = 0; i < len[target]; i++) cap_idx[i] = comb[target][i];
    return len[target];
}

static void play_card(og_state *s, int p, int hand_pos) {
    /* play_card, mini_scopa_game.py:93-104 */
    int card = s->hand[p][hand_pos];
    int cap[8];
    int nc = og_capture(s, card, cap);
    if (nc > 0) {
        int8_t keep[8]; int nk = 0;
        for (int i = 0; i < s->nt; i++) {
            int taken = 0;
            for (int j = 0; j < nc; j++) if (cap[j] == i) taken = 1;
            if (!taken) keep[nk++] = s->table[i];
        }
        memset(s->table, -1, sizeof s->table);
        memcpy(s->table, keep, (size_t)nk);
        s->nt = (int8_t)nk;
        s->ncap[p] = (int8_t)(s->ncap[p] + nc + 1);
        if (s->nt == 0) s->scopas[p]++;
    } else {
        s->table[s->nt++] = (int8_t)card;
    }
    for (int i = hand_pos; i + 1 < s->nh[p]; i++) s->hand[p][i] = s->hand[p][i + 1];
    s->nh[p]--;
    s->hand[p][s->nh[p]] = 0;
}

void og_step(og_state *s, int action) {
    /* MiniScopaEnv.step, mini_scopa_game.py:140-167 */
    if (og_is_terminal(s)) return; /* _was_dead_step */
    int p = s->step & 1;
    int pos = -1;
    for (int i = 0; i < s->nh[p]; i++) /* :155 first card in hand with that (rank, suit) */
        if (s->hand[p][i] == action) { pos = i; break; }
    if (pos >= 0) play_card(s, p, pos); /* else: silent no-op (:156) */
    s->step++;
}

void og_rewards_x2(const og_state *s, int r2[2]) {
    /* evaluate_game, mini_scopa_game.py:106-114; rewards stay 0 until terminal (openspiel…:78-81) */
    if (!og_is_terminal(s)) { r2[0] = r2[1] = 0; return; }
    int r0 = s->ncap[0] + 2 * s->scopas[0], r1 = s->ncap[1] + 2 * s->scopas[1];
    int total = r0 + r1;
    if (total == 0) { r2[0] = r2[1] = 0; return; }
    r2[0] = 2 * r0 - total; /* 2*(r - total/2) */
    r2[1] = 2 * r1 - total;
}

int og_infoset_string(const og_state *s, int player, char *buf) {
    /* information_state_string, openspiel_mini_scopa.py:86-95 */
    if (player < 0) player = og_current_player(s);
    if (og_is_terminal(s) || player < 0) return sprintf(buf, "TERMINAL");
    char *w = buf;
    w += sprintf(w, "P%d:H[", player);
    for (int i = 0; i < s->nh[player]; i++)
        w += sprintf(w, "%s%d%c", i ? "-" : "", RANK_OF[s->hand[player][i]], SUIT_CHAR[s->hand[player][i] >> 2]);
    w += sprintf(w, "]_T[");
    for (int i = 0; i < s->nt; i++)
        w += sprintf(w, "%s%d%c", i ? "-" : "", RANK_OF[s->table[i]], SUIT_CHAR[s->table[i] >> 2]);
    w += sprintf(w, "]");
    return (int)(w - buf);
}

/* ===== tree ======================================================================== */
#define MAXN 2229 /* 1653 decision + 576 terminal, deal-independent (4/4/3/3/2/2/1/1 legal profile) */

static int tree_rec(og_tree *t, const og_state *s, int depth) {
    int idx = t->n_nodes++;
    t->state[idx] = *s;
    t->depth[idx] = (int8_t)depth;
    int term = og_is_terminal(s);
    t->term[idx] = (int8_t)term;
    t->player[idx] = (int8_t)og_current_player(s);
    int r2[2];
    og_rewards_x2(s, r2);
    t->r2[idx * 2] = (int8_t)r2[0];
    t->r2[idx * 2 + 1] = (int8_t)r2[1];
    for (int i = 0; i < 4; i++) { t->legal[idx * 4 + i] = -1; t->child[idx * 4 + i] = -1; }
    if (term) { t->infoset[idx] = -1; t->nlegal[idx] = 0; return idx; }
    t->n_decision++;
    int legal[4];
    int n = og_legal(s, -1, legal);
    t->nlegal[idx] = (int8_t)n;
    char key[64];
    og_infoset_string(s, -1, key);
    int id = -1;
    for (int i = 0; i < t->n_infosets; i++)
        if (strcmp(t->infoset_str[i], key) == 0) { id = i; break; }
    if (id < 0) { /* dict insertion on first visit: vanilla_cfr.py:51-54 */
        id = t->n_infosets++;
        strcpy(t->infoset_str[id], key);
        t->infoset_nlegal[id] = (int8_t)n;
        t->infoset_player[id] = t->player[idx];
        for (int i = 0; i < 4; i++) t->infoset_legal[id * 4 + i] = (int8_t)(i < n ? legal[i] : -1);
    }
    t->infoset[idx] = (int16_t)id;
    for (int i = 0; i < n; i++) {
        t->legal[idx * 4 + i] = (int8_t)legal[i];
        og_state c; og_clone(s, &c);   /* clone() */
        og_step(&c, legal[i]);    /* apply_action */
        t->child[idx * 4 + i] = tree_rec(t, &c, depth + 1);
    }
    return idx;
}

og_tree *og_tree_build(const uint8_t perm[16]) {
    og_tree *t = (og_tree *)calloc(1, sizeof *t);
    t->term = (int8_t *)calloc(MAXN, 1); t->player = (int8_t *)calloc(MAXN, 1);
    t->nlegal = (int8_t *)calloc(MAXN, 1); t->depth = (int8_t *)calloc(MAXN, 1);
    t->infoset = (int16_t *)calloc(MAXN, 2); t->legal = (int8_t *)calloc(MAXN * 4, 1);
    t->child = (int32_t *)calloc(MAXN * 4, 4); t->r2 = (int8_t *)calloc(MAXN * 2, 1);
    t->state = (og_state *)calloc(MAXN, sizeof(og_state));
    t->infoset_str = (char(*)[64])calloc(MAXN, 64);
    t->infoset_nlegal = (int8_t *)calloc(MAXN, 1); t->infoset_legal = (int8_t *)calloc(MAXN * 4, 1);
    t->infoset_player = (int8_t *)calloc(MAXN, 1);
    og_state s;
    og_reset(&s, perm);
    tree_rec(t, &s, 0);
    return t;
}

void og_tree_free(og_tree *t) {
    if (!t) return;
    free(t->term); free(t->player); free(t->nlegal); free(t->depth); free(t->infoset); free(t->legal);
    free(t->child); free(t->r2); free(t->state); free(t->infoset_str); free(t->infoset_nlegal);
    free(t->infoset_legal); free(t->infoset_player); free(t);
}

int og_tree_counts(const og_tree *t, int *n_nodes, int *n_decision, int *n_infosets) {
    *n_nodes = t->n_nodes; *n_decision = t->n_decision; *n_infosets = t->n_infosets;
    return 0;
}

int og_tree_export(const og_tree *t, int8_t *term, int8_t *player, int8_t *nlegal, int8_t *depth,
                   int16_t *infoset, int8_t *legal, int32_t *child, int8_t *r2) {
    size_t n = (size_t)t->n_nodes;
    memcpy(term, t->term, n); memcpy(player, t->player, n); memcpy(nlegal, t->nlegal, n);
    memcpy(depth, t->depth, n); memcpy(infoset, t->infoset, n * 2); memcpy(legal, t->legal, n * 4);
    memcpy(child, t->child, n * 16); memcpy(r2, t->r2, n * 2);
    return 0;
}

int og_tree_states(const og_tree *t, int8_t *hands, int8_t *nh, int8_t *table, int8_t *nt, int8_t *ncap,
                   int8_t *scopas, int8_t *step) {
    for (int i = 0; i < t->n_nodes; i++) {
        const og_state *s = &t->state[i];
        for (int p = 0; p < 2; p++) {
            for (int k = 0; k < 4; k++) hands[(i * 2 + p) * 4 + k] = (int8_t)(k < s->nh[p] ? s->hand[p][k] : -1);
            nh[i * 2 + p] = s->nh[p]; ncap[i * 2 + p] = s->ncap[p]; scopas[i * 2 + p] = s->scopas[p];
        }
        for (int k = 0; k < 8; k++) table[i * 8 + k] = (int8_t)(k < s->nt ? s->table[k] : -1);
        nt[i] = s->nt; step[i] = s->step;
    }
    return 0;
}

int og_tree_infoset_string(const og_tree *t, int id, char *buf) {
    strcpy(buf, t->infoset_str[id]);
    return (int)strlen(buf);
}

int og_tree_infoset_meta(const og_tree *t, int8_t *nlegal, int8_t *legal, int8_t *player) {
    memcpy(nlegal, t->infoset_nlegal, (size_t)t->n_infosets);
    memcpy(legal, t->infoset_legal, (size_t)t->n_infosets * 4);
    memcpy(player, t->infoset_player, (size_t)t->n_infosets);
    return 0;
}

/* ===== numpy float64 reductions on <= 4 elements =================================== */
/* np.sum / ndarray.sum on a contiguous float64 vector of n < 8 elements: the add.reduce
 * inner loop is DOUBLE_pairwise_sum's small-n branch, a left-to-right running sum
 * (numpy/_core/src/umath/loops_utils.h.src).  Pinned bit-exactly by tests/golden/vanilla_cfr.npz. */
static double np_sum(const double *a, int n) {
    double r = a[0];
    for (int i = 1; i < n; i++) r += a[i];
    return r;
}
/* np.dot of two float64 vectors, n <= 4.  numpy hands this to cblas_ddot; with the numpy 2.2.6
 * wheel used to generate the fixtures (scipy-openblas 0.3.29, x86-64 with FMA3) the n < 32 tail loop
 * `dot += y[i]*x[i]` is FMA-contracted: a left-to-right chain r = fma(a[i], b[i], r) from r = 0.
 * Determined empirically against np.dot (3000 random vectors per n: 100 % fma chain, 81-91 % mul+add)
 * and pinned bit-exactly by tests/golden/mccfr.npz (9 runs up to 200 iterations). */
static double np_dot(const double *a, const double *b, int n) {
    double r = 0.0;
    for (int i = 0; i < n; i++) r = fma(a[i], b[i], r);
    return r;
}

/* ===== vanilla CFR, exact sequential semantics ===================================== */
void og_tables_init(const og_tree *t, double *regret, double *strat, double *local) {
    /* InfoNode.__post_init__, vanilla_cfr.py:15-21 */
    for (int i = 0; i < t->n_infosets; i++) {
        int n = t->infoset_nlegal[i];
        for (int k = 0; k < 4; k++) {
            if (regret) regret[i * 4 + k] = 0.0;
            if (strat) strat[i * 4 + k] = 0.0;
            if (local) local[i * 4 + k] = k < n ? 1.0 / (double)n : 0.0; /* np.ones(n)/n */
        }
    }
}

static void regret_match(const double *R, int n, double *out) {
    /* InfoNode.get_strategy, vanilla_cfr.py:23-30 */
    double pos[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) pos[i] = R[i] > 0.0 ? R[i] : 0.0; /* np.maximum(R, 0) */
    double s = np_sum(pos, n);
    if (s > 0.0) for (int i = 0; i < n; i++) out[i] = pos[i] / s;
    else         for (int i = 0; i < n; i++) out[i] = 1.0 / (double)n;
}

static double cfr_rec(const og_tree *t, double *R, double *S, double *L, int node, int trav, double r0, double r1) {
    /* CFRTrainer._cfr_recursive, vanilla_cfr.py:56-99 */
    if (t->term[node]) return (double)t->r2[node * 2 + trav] * 0.5;
    int p = t->player[node], I = t->infoset[node], n = t->nlegal[node];
    double *ls = L + I * 4;
    double au[4], prod[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) { /* :79-85 */
        int c = t->child[node * 4 + i];
        au[i] = p == 0 ? cfr_rec(t, R, S, L, c, trav, r0 * ls[i], r1)
                       : cfr_rec(t, R, S, L, c, trav, r0, r1 * ls[i]);
    }
    for (int i = 0; i < n; i++) prod[i] = ls[i] * au[i];
    double v = np_sum(prod, n); /* :87 */
    if (p == trav) { /* :89-95 */
        double reach = trav == 0 ? r0 : r1, opp = trav == 0 ? r1 : r0;
        for (int i = 0; i < n; i++) {
            double regret = au[i] - v;
            R[I * 4 + i] += opp * regret;
            S[I * 4 + i] += reach * ls[i];
        }
    }
    regret_match(R + I * 4, n, ls); /* :97, every visit */
    return v;
}

void og_cfr_exact(const og_tree *t, double *regret, double *strat, double *local, int n_iters, double *root_values) {
    /* CFRTrainer.train, vanilla_cfr.py:105-110 */
    for (int it = 0; it < n_iters; it++)
        for (int i = 0; i < 2; i++) {
            double v = cfr_rec(t, regret, strat, local, 0, i, 1.0, 1.0);
            if (root_values) root_values[it * 2 + i] = v;
        }
}

double og_cfr_exact_from(const og_tree *t, double *regret, double *strat, double *local, const int *path, int depth,
                         int trav, double r0, double r1) {
    /* CFRTrainer._cfr_recursive(state, trav, r0, r1) for the state reached by legal-action indices path[0..depth) */
    int node = 0;
    for (int d = 0; d < depth; d++) node = t->child[node * 4 + path[d]];
    return cfr_rec(t, regret, strat, local, node, trav, r0, r1);
}

/* ===== MCCFR replay (reference semantics, host-supplied uniforms) ================== */
static int np_choice(const double *p, int n, double u) {
    /* np.random.choice(a, p=p): cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(u, 'right') */
    double cdf[4];
    double c = 0.0;
    for (int i = 0; i < n; i++) { c = i ? c + p[i] : p[0]; cdf[i] = c; }
    double last = cdf[n - 1];
    int idx = 0;
    for (int i = 0; i < n; i++) { cdf[i] /= last; if (cdf[i] <= u) idx = i + 1; }
    return idx < n ? idx : n - 1;
}

static void mc_strategy(const double *R, int n, double *sigma) {
    /* InfoNode.current_strategy, mc_cfr.py:20-24 */
    double pos[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) pos[i] = R[i] > 0.0 ? R[i] : 0.0;
    double s = np_sum(pos, n);
    if (s == 0.0) for (int i = 0; i < n; i++) sigma[i] = 1.0 / (double)n;
    else          for (int i = 0; i < n; i++) sigma[i] = pos[i] / s;
}

typedef struct { const double *u; int64_t pos, n; } ustream;

static double mc_rec(const og_tree *t, double *R, double *S, int node, int trav, const double reach[2],
                     const double samp[2], ustream *us) {
    /* MCCFRTrainer._sample, mc_cfr.py:37-86 */
    if (t->term[node]) return (double)t->r2[node * 2 + trav] * 0.5;
    int p = t->player[node], I = t->infoset[node], n = t->nlegal[node];
    double sigma[4];
    mc_strategy(R + I * 4, n, sigma);
    double u = us->pos < us->n ? us->u[us->pos] : 0.0;
    us->pos++;
    int a = np_choice(sigma, n, u); /* :55 */
    double nreach[2] = {reach[0], reach[1]}, nsamp[2] = {samp[0], samp[1]};
    if (p == trav) nsamp[p] *= sigma[a];
    else { nreach[p] *= sigma[a]; nsamp[p] *= sigma[a]; }
    double util = mc_rec(t, R, S, t->child[node * 4 + a], trav, nreach, nsamp, us);
    if (p == trav) { /* :69-84 */
        double cfv[4];
        for (int i = 0; i < n; i++) {
            double ts[2] = {samp[0], samp[1]};
            ts[p] *= sigma[i];
            cfv[i] = mc_rec(t, R, S, t->child[node * 4 + i], trav, reach, ts, us);
        }
        double v = np_dot(sigma, cfv, n);
        double opp_reach = reach[1 - p];
        double weight = samp[p] > 0.0 ? opp_reach / samp[p] : 0.0;
        for (int i = 0; i < n; i++) {
            R[I * 4 + i] += weight * (cfv[i] - v);
            S[I * 4 + i] += reach[p] * sigma[i];
        }
    }
    return util;
}

int64_t og_mccfr_replay(const og_tree *t, double *regret, double *strat, int n_iters, const double *uniforms,
                        int64_t n_uniforms) {
    ustream us = {uniforms, 0, n_uniforms};
    const double one[2] = {1.0, 1.0};
    for (int it = 0; it < n_iters; it++)
        for (int p = 0; p < 2; p++) /* iteration(), mc_cfr.py:88-92 */
            mc_rec(t, regret, strat, 0, p, one, one, &us);
    return us.pos;
}

/* ===== Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants) ========= */
void og_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double og_philox_uniform(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t ctr[4] = {c0, c1, c2, c3}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, o[4];
    og_philox4x32_10(ctr, key, o);
    return ((double)(o[0] >> 5) * 67108864.0 + (double)(o[1] >> 6)) / 9007199254740992.0;
}

/* ===== batched MCCFR (frozen tables, path-keyed RNG) ================================ */
/* A node of one traversal's recursion tree is named by (ntl, j): ntl = how many traverser nodes lie above it, j = its branch index at
 * that level -- at a traverser node with n legal actions the sampled child (mc_cfr.py:55-67) is j*(n+1) and the re-expansion of legal
 * action i (:72-78) is j*(n+1) + i + 1.  An opponent node and the traverser node that follows it share (ntl, j).  Draws:
 *   block = Philox4x32-10(key = seed; ctr = (first block of level ntl + (j >> 1), traversal id, iteration, traverser)),
 *           first block of level 0,1,2,3 = 0,1,4,14
 *   word  = 2 * (j & 1) + (1 at the traverser node, 0 at the opponent node);   u = (word >> 1) * 2^-31                              */
typedef struct {
    const og_tree *t; const double *R; double *dR, *dS;
    uint64_t seed; uint32_t iter, b; int trav;
    uint64_t dvis, tvis;
    int32_t *tr_nodes; int8_t *tr_actions; int tr_n, tr_max;
} bctx;

static double u53(uint32_t a, uint32_t b) {
    /* 53-bit double in [0,1), the genrand_res53 construction numpy's random_sample uses */
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

static double mcb_rec(bctx *c, int node, uint32_t j, int ntl, double reach_opp, double samp_trav) {
    static const uint32_t first_block[4] = {0u, 1u, 4u, 14u};
    const og_tree *t = c->t;
    if (t->term[node]) { c->tvis++; return (double)t->r2[node * 2 + c->trav] * 0.5; }
    c->dvis++;
    int p = t->player[node], I = t->infoset[node], n = t->nlegal[node];
    double sigma[4];
    mc_strategy(c->R + I * 4, n, sigma);
    uint32_t ctr[4] = {first_block[ntl < 3 ? ntl : 3] + (j >> 1), c->b, c->iter, (uint32_t)c->trav};
    uint32_t key[2] = {(uint32_t)c->seed, (uint32_t)(c->seed >> 32)}, o[4];
    og_philox4x32_10(ctr, key, o);
    double u = (double)(o[2 * (j & 1u) + (p == c->trav ? 1u : 0u)] >> 1) / 2147483648.0;
    int a = np_choice(sigma, n, u);
    if (c->tr_nodes && c->tr_n < c->tr_max) { c->tr_nodes[c->tr_n] = node; c->tr_actions[c->tr_n] = (int8_t)a; }
    c->tr_n++;
    if (p != c->trav)
        return mcb_rec(c, t->child[node * 4 + a], j, ntl, reach_opp * sigma[a], samp_trav);
    double util = mcb_rec(c, t->child[node * 4 + a], j * (uint32_t)(n + 1), ntl + 1, reach_opp, samp_trav * sigma[a]);
    double cfv[4];
    for (int i = 0; i < n; i++)
        cfv[i] = mcb_rec(c, t->child[node * 4 + i], j * (uint32_t)(n + 1) + (uint32_t)(i + 1), ntl + 1, reach_opp, samp_trav * sigma[i]);
    double v = np_dot(sigma, cfv, n);
    double w = samp_trav > 0.0 ? reach_opp / samp_trav : 0.0;
    if (c->dR)
        for (int i = 0; i < n; i++) {
            c->dR[I * 4 + i] += w * (cfv[i] - v);
            c->dS[I * 4 + i] += sigma[i]; /* reach[traverser] is never multiplied: always 1.0 (mc_cfr.py:61-65) */
        }
    return util;
}

void og_mccfr_batched_delta(const og_tree *t, const double *regret, double *d_regret, double *d_strat, uint64_t seed,
                            uint32_t iteration, uint32_t b0, uint32_t nb, uint64_t *decision_visits,
                            uint64_t *terminal_visits) {
    bctx c;
    memset(&c, 0, sizeof c);
    c.t = t; c.R = regret; c.dR = d_regret; c.dS = d_strat; c.seed = seed; c.iter = iteration;
    for (uint32_t b = b0; b < b0 + nb; b++)
        for (int p = 0; p < 2; p++) {
            c.b = b; c.trav = p;
            mcb_rec(&c, 0, 0, 0, 1.0, 1.0);
        }
    if (decision_visits) *decision_visits += c.dvis;
    if (terminal_visits) *terminal_visits += c.tvis;
}

void og_mccfr_batched(const og_tree *t, double *regret, double *strat, uint64_t seed, uint32_t iter0, uint32_t n_iters,
                      uint32_t batch, uint64_t *decision_visits) {
    size_t cells = (size_t)t->n_infosets * 4;
    double *dR = (double *)malloc(cells * 8), *dS = (double *)malloc(cells * 8);
    for (uint32_t it = 0; it < n_iters; it++) {
        memset(dR, 0, cells * 8); memset(dS, 0, cells * 8);
        og_mccfr_batched_delta(t, regret, dR, dS, seed, iter0 + it, 0, batch, decision_visits, NULL);
        for (size_t k = 0; k < cells; k++) { regret[k] += dR[k]; strat[k] += dS[k]; }
    }
    free(dR); free(dS);
}

int og_mccfr_batched_trace(const og_tree *t, const double *regret, uint64_t seed, uint32_t iteration, uint32_t b,
                           int traverser, int32_t *nodes, int8_t *actions, int max_out) {
    bctx c;
    memset(&c, 0, sizeof c);
    c.t = t; c.R = regret; c.seed = seed; c.iter = iteration; c.b = b; c.trav = traverser;
    c.tr_nodes = nodes; c.tr_actions = actions; c.tr_max = max_out;
    mcb_rec(&c, 0, 0, 0, 1.0, 1.0);
    return c.tr_n;
}

/* ===== Single Deep CFR traversal (src/algorithms/deep_cfr/deep_cfr.py:284-365) ===== */
/* One net = torch state dict in its own order and layout: W1[128][34] b1[128] W2[64][128] b2[64] W3[16][64] b3[16]
 * (FlexibleNet(mode="mlp"), nets.py:296-331: Linear -> ReLU -> Linear -> ReLU -> Linear), float32. */
enum { SD_IN = 34, SD_H1 = 128, SD_H2 = 64, SD_OUT = 16, SD_NET = SD_H1 * SD_IN + SD_H1 + SD_H2 * SD_H1 + SD_H2 + SD_OUT * SD_H2 + SD_OUT };

static void sd_forward(const float *w, const float *x, float *out) {
    const float *W1 = w, *b1 = W1 + SD_H1 * SD_IN, *W2 = b1 + SD_H1, *b2 = W2 + SD_H2 * SD_H1, *W3 = b2 + SD_H2, *b3 = W3 + SD_OUT * SD_H2;
    float h1[SD_H1], h2[SD_H2];
    for (int o = 0; o < SD_H1; o++) { float a = b1[o]; for (int k = 0; k < SD_IN; k++) a += W1[o * SD_IN + k] * x[k]; h1[o] = a > 0.0f ? a : 0.0f; }
    for (int o = 0; o < SD_H2; o++) { float a = b2[o]; for (int k = 0; k < SD_H1; k++) a += W2[o * SD_H1 + k] * h1[k]; h2[o] = a > 0.0f ? a : 0.0f; }
    for (int o = 0; o < SD_OUT; o++) { float a = b3[o]; for (int k = 0; k < SD_H2; k++) a += W3[o * SD_H2 + k] * h2[k]; out[o] = a; }
}

typedef struct {
    const og_tree *t; const float *nets; /* [2][SD_NET] */
    int trav; uint64_t seed; uint32_t iter, b;
    const double *uniforms; int64_t upos, n_uniforms;   /* replay: one per opponent visit, DFS order; NULL: Philox */
    float *row_feat, *row_regret, *row_mask; int64_t rows; /* appended in the reference's order (DFS post-order) */
    uint64_t dvis;
} sdctx;

/* _state_to_features (:213-275) for the player to move, _get_legal_actions_mask (:277-282) */
static void sd_features(const og_state *s, int p, float *f, float *m) {
    for (int c = 0; c < 34; c++) f[c] = 0.0f;
    for (int c = 0; c < 16; c++) m[c] = 0.0f;
    for (int k = 0; k < s->nh[p]; k++) { f[s->hand[p][k]] = 1.0f; m[s->hand[p][k]] = 1.0f; }
    for (int k = 0; k < s->nt; k++) f[16 + s->table[k]] = 1.0f;
    f[32] = 1.0f; /* float(player == state.current_player()) */
}

static float sd_rec(sdctx *c, int node, int ply, uint32_t slot) {
    const og_tree *t = c->t;
    if (t->term[node]) return 0.5f * (float)t->r2[node * 2 + c->trav];           /* float(rewards[player]) (:286-293) */
    c->dvis++;
    const og_state *s = &t->state[node];
    const int p = t->player[node], nl = t->nlegal[node];
    float feat[34], mask[16], adv[16], pol[16];
    sd_features(s, p, feat, mask);
    sd_forward(c->nets + (size_t)p * SD_NET, feat, adv);                            /* get_advantages (:54-67) */
    /* positive_regret_policy (nets.py:93-101) on adv*mask - 1e6*(1-mask): relu(.)*mask / clamp_min(sum, 1e-8) */
    float z = 0.0f;
    for (int a = 0; a < 16; a++) { pol[a] = (mask[a] != 0.0f && adv[a] > 0.0f) ? adv[a] : 0.0f; z += pol[a]; }
    const float zc = z > 1e-8f ? z : 1e-8f;
    for (int a = 0; a < 16; a++) pol[a] = pol[a] / zc;
    if (p == c->trav) {                                                             /* (:320-346) */
        float value = 0.0f, cfv[16];
        for (int a = 0; a < 16; a++) cfv[a] = 0.0f;
        for (int k = 0; k < nl; k++) {
            const int a = t->legal[node * 4 + k];
            const float av = sd_rec(c, t->child[node * 4 + k], ply + 1, slot * (uint32_t)nl + (uint32_t)k);
            value += pol[a] * av;                                                    /* float32 under NEP 50 */
            cfv[a] = av;
        }
        float reg[16], mx = 0.0f;
        for (int a = 0; a < 16; a++) { reg[a] = cfv[a] - value; const float x = fabsf(reg[a]); if (x > mx) mx = x; }   /* illegal slots: -value */
        if (mx > 0.0f) { const float den = mx + 1e-8f; for (int a = 0; a < 16; a++) reg[a] = reg[a] / den; }          /* add_experience (:70-75) */
        if (c->row_feat) {
            memcpy(c->row_feat + c->rows * 34, feat, sizeof feat);
            memcpy(c->row_regret + c->rows * 16, reg, sizeof reg);
            memcpy(c->row_mask + c->rows * 16, mask, sizeof mask);
        }
        c->rows++;
        return value;
    }
    /* opponent: sample one action (:347-365) */
    float ap[4], sum = 0.0f;
    for (int k = 0; k < nl; k++) { ap[k] = pol[t->legal[node * 4 + k]]; sum = k ? sum + ap[k] : ap[0]; }
    double u;
    if (c->uniforms) { u = c->upos < c->n_uniforms ? c->uniforms[c->upos] : 0.0; c->upos++; }
    else {  /* the build's draw: Philox block keyed by (frontier slot + 1024*ply, traversal id, iteration, 4 + traverser) */
        uint32_t ctr[4] = {slot + 1024u * (uint32_t)ply, c->b, c->iter, 4u + (uint32_t)c->trav};
        uint32_t key[2] = {(uint32_t)c->seed, (uint32_t)(c->seed >> 32)}, o[4];
        og_philox4x32_10(ctr, key, o);
        u = u53(o[0], o[1]);
    }
    int a;
    if (sum == 0.0f) { a = (int)(u * (double)nl); if (a > nl - 1) a = nl - 1; }     /* np.random.choice(legal_actions): uniform */
    else {                                                                          /* p = float32 action_probs / sum; cdf in float64 */
        double pd[4];
        for (int k = 0; k < nl; k++) pd[k] = (double)(ap[k] / sum);
        a = np_choice(pd, nl, u);
    }
    return sd_rec(c, t->child[node * 4 + a], ply + 1, slot);
}

int64_t og_sdcfr_traverse(const og_tree *t, const float *nets, int traverser, uint64_t seed, uint32_t iteration, uint32_t b0,
                          uint32_t nb, const double *uniforms, int64_t n_uniforms, float *row_feat, float *row_regret,
                          float *row_mask, float *values, uint64_t *decision_visits) {
    sdctx c;
    memset(&c, 0, sizeof c);
    c.t = t; c.nets = nets; c.trav = traverser; c.seed = seed; c.iter = iteration;
    c.uniforms = uniforms; c.n_uniforms = n_uniforms;
    c.row_feat = row_feat; c.row_regret = row_regret; c.row_mask = row_mask;
    for (uint32_t b = 0; b < nb; b++) {
        c.b = b0 + b;
        const float v = sd_rec(&c, 0, 0, 0);
        if (values) values[b] = v;
    }
    if (decision_visits) *decision_visits += c.dvis;
    return c.rows;
}

/* ===== synchronous CFR (build-defined) ============================================= */
static double sync_rec(const og_tree *t, const double *sig, double *dR, double *dS, int node, double r0, double r1,
                       double *v1_out) {
    /* returns value for P0; P1's value is the negation (zero-sum, evaluate_game :106-114) */
    (void)v1_out;
    if (t->term[node]) return (double)t->r2[node * 2] * 0.5;
    int p = t->player[node], I = t->infoset[node], n = t->nlegal[node];
    const double *s = sig + I * 4;
    double au[4];
    for (int i = 0; i < n; i++)
        au[i] = p == 0 ? sync_rec(t, sig, dR, dS, t->child[node * 4 + i], r0 * s[i], r1, NULL)
                       : sync_rec(t, sig, dR, dS, t->child[node * 4 + i], r0, r1 * s[i], NULL);
    double v = 0.0;
    for (int i = 0; i < n; i++) v += s[i] * au[i];
    double reach = p == 0 ? r0 : r1, opp = p == 0 ? r1 : r0, sgn = p == 0 ? 1.0 : -1.0;
    for (int i = 0; i < n; i++) {
        dR[I * 4 + i] += opp * (sgn * (au[i] - v));
        dS[I * 4 + i] += reach * s[i];
    }
    return v;
}

void og_cfr_sync(const og_tree *t, double *regret, double *strat, int n_iters) {
    size_t cells = (size_t)t->n_infosets * 4;
    double *sig = (double *)calloc(cells, 8), *dR = (double *)malloc(cells * 8), *dS = (double *)malloc(cells * 8);
    for (int it = 0; it < n_iters; it++) {
        for (int i = 0; i < t->n_infosets; i++) regret_match(regret + i * 4, t->infoset_nlegal[i], sig + i * 4);
        memset(dR, 0, cells * 8); memset(dS, 0, cells * 8);
        sync_rec(t, sig, dR, dS, 0, 1.0, 1.0, NULL);
        for (size_t k = 0; k < cells; k++) { regret[k] += dR[k]; strat[k] += dS[k]; }
    }
    free(sig); free(dR); free(dS);
}

/* ===== policies, value, exploitability (build-defined; parity unpinned) ============= */
void og_average_policy(const og_tree *t, const double *strat, double *policy) {
    /* InfoNode.policy, vanilla_cfr.py:32-39 */
    for (int i = 0; i < t->n_infosets; i++) {
        int n = t->infoset_nlegal[i];
        double s = np_sum(strat + i * 4, n);
        for (int k = 0; k < 4; k++)
            policy[i * 4 + k] = k < n ? (s > 0.0 ? strat[i * 4 + k] / s : 1.0 / (double)n) : 0.0;
    }
}

static double value_rec(const og_tree *t, const double *pol, int node) {
    if (t->term[node]) return (double)t->r2[node * 2] * 0.5;
    int I = t->infoset[node], n = t->nlegal[node];
    double v = 0.0;
    for (int i = 0; i < n; i++) v += pol[I * 4 + i] * value_rec(t, pol, t->child[node * 4 + i]);
    return v;
}

double og_policy_value(const og_tree *t, const double *policy) { return value_rec(t, policy, 0); }

/* Best response of `br` against `policy`, OpenSpiel's procedural definition: per infoset (string key),
 * argmax_a sum_{h in I} opp_reach(h) * value(h.a), resolved deepest infosets first. */
static void br_fill_reach(const og_tree *t, const double *pol, int br, int node, double opp, double *reach) {
    reach[node] = opp;
    if (t->term[node]) return;
    int p = t->player[node], I = t->infoset[node], n = t->nlegal[node];
    for (int i = 0; i < n; i++)
        br_fill_reach(t, pol, br, t->child[node * 4 + i], p == br ? opp : opp * pol[I * 4 + i], reach);
}

static double best_response_value(const og_tree *t, const double *pol, int br) {
    int N = t->n_nodes;
    double *reach = (double *)malloc((size_t)N * 8), *val = (double *)malloc((size_t)N * 8);
    double *q = (double *)malloc((size_t)t->n_infosets * 4 * 8);
    int8_t *choice = (int8_t *)malloc((size_t)t->n_infosets);
    br_fill_reach(t, pol, br, 0, 1.0, reach);
    for (int d = 8; d >= 0; d--) {
        /* pass 1: accumulate q over the infosets of `br` at this depth (children are deeper: already valued) */
        for (int i = 0; i < t->n_infosets * 4; i++) q[i] = 0.0;
        for (int nd = 0; nd < N; nd++) {
            if (t->depth[nd] != d) continue;
            if (t->term[nd]) { val[nd] = (double)t->r2[nd * 2 + br] * 0.5; continue; }
            if (t->player[nd] == br)
                for (int i = 0; i < t->nlegal[nd]; i++) q[t->infoset[nd] * 4 + i] += reach[nd] * val[t->child[nd * 4 + i]];
        }
        for (int I = 0; I < t->n_infosets; I++) {
            int best = 0;
            for (int i = 1; i < t->infoset_nlegal[I]; i++) if (q[I * 4 + i] > q[I * 4 + best]) best = i;
            choice[I] = (int8_t)best; /* overwritten only for infosets at this depth that matter below */
        }
        /* pass 2: node values at this depth */
        for (int nd = 0; nd < N; nd++) {
            if (t->depth[nd] != d || t->term[nd]) continue;
            int I = t->infoset[nd], n = t->nlegal[nd];
            if (t->player[nd] == br) val[nd] = val[t->child[nd * 4 + choice[I]]];
            else {
                double v = 0.0;
                for (int i = 0; i < n; i++) v += pol[I * 4 + i] * val[t->child[nd * 4 + i]];
                val[nd] = v;
            }
        }
    }
    double r = val[0];
    free(reach); free(val); free(q); free(choice);
    return r;
}

double og_exploitability(const og_tree *t, const double *policy, double *br_values) {
    double b0 = best_response_value(t, policy, 0), b1 = best_response_value(t, policy, 1);
    if (br_values) { br_values[0] = b0; br_values[1] = b1; }
    return 0.5 * (b0 + b1); /* NashConv/2; v0(pi)+v1(pi)=0 */
}

/* ===================================================================================================================
 * FullScopa (40-card deck): src/envs/full_scopa_game.py, src/envs/openspiel_full_scopa.py.  Literal restatement. */
static const char *FSUIT_NAME[4] = {"denari", "coppe", "spade", "bastoni"}; /* FullDeck.suits (:20) */
static int f_rank(int c) { return c % 10 + 1; }
static int f_suit(int c) { return c / 10; }

void ogf_deal_py_seed(int64_t seed, uint8_t perm[40]) {
    /* FullDeck.__init__ (:29-32): 40 cards suit-major, random.seed(seed), random.shuffle */
    uint64_t a = seed < 0 ? (uint64_t)(-seed) : (uint64_t)seed;
    uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
    mt_t m;
    mt_init_by_array(&m, key, key[1] ? 2 : 1);
    for (int i = 0; i < 40; i++) perm[i] = (uint8_t)i;
    for (int i = 39; i >= 1; i--) {
        uint32_t j = py_randbelow(&m, (uint32_t)i + 1);
        uint8_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

void ogf_reset(ogf_state *s, const uint8_t perm[40]) {
    /* FullScopaGame.reset (:60-75): 4 cards to the table, then 3 to each player */
    memset(s, 0, sizeof *s);
    memcpy(s->deck, perm, 40);
    for (int i = 0; i < 4; i++) s->table[s->nt++] = (int8_t)perm[s->deck_pos++];
    for (int p = 0; p < 2; p++) { for (int i = 0; i < 3; i++) s->hand[p][i] = (int8_t)perm[s->deck_pos++]; s->nh[p] = 3; }
    s->last_capture = -1;
}

int ogf_legal(const ogf_state *s, int player, int out[3]) {
    /* FullScopaState.legal_actions (openspiel_full_scopa.py:22-42) */
    if (s->terminal) return 0;
    if (player < 0) player = s->step & 1;
    int n = 0;
    for (int i = 0; i < s->nh[player]; i++) out[n++] = s->hand[player][i];
    if (!n) { out[0] = 0; n = 1; }
    return n;
}

static void f_evaluate(ogf_state *s) {
    /* FullScopaGame.evaluate_game (:166-228) */
    int scores[2] = {0, 0};
    if (s->nt > 0 && s->last_capture >= 0) { /* remaining table cards to the last capturer */
        int p = s->last_capture;
        for (int i = 0; i < s->nt; i++) s->cap[p][s->ncap[p]++] = (uint8_t)s->table[i];
    }
    int cards[2] = {s->ncap[0], s->ncap[1]};
    if (cards[0] != cards[1]) scores[cards[0] > cards[1] ? 0 : 1] += 1;            /* carte */
    int den[2] = {0, 0};
    for (int p = 0; p < 2; p++) for (int i = 0; i < s->ncap[p]; i++) if (f_suit(s->cap[p][i]) == 0) den[p]++;
    if (den[0] != den[1]) scores[den[0] > den[1] ? 0 : 1] += 1;                    /* denari */
    for (int p = 0; p < 2; p++) {                                                  /* sette bello: first holder */
        int has = 0;
        for (int i = 0; i < s->ncap[p]; i++) if (s->cap[p][i] == 6) has = 1;
        if (has) { scores[p] += 1; break; }
    }
    static const int PV[11] = {0, 16, 12, 13, 14, 15, 18, 21, 10, 10, 10};         /* primiera_values (:24-27) */
    int prim[2];
    for (int p = 0; p < 2; p++) {
        int best[4] = {0, 0, 0, 0}, have[4] = {0, 0, 0, 0};
        for (int i = 0; i < s->ncap[p]; i++) {
            int c = s->cap[p][i], v = PV[f_rank(c)], su = f_suit(c);
            if (!have[su] || v > best[su]) { best[su] = v; have[su] = 1; }
        }
        prim[p] = (have[0] && have[1] && have[2] && have[3]) ? best[0] + best[1] + best[2] + best[3] : 0;
    }
    if (prim[0] > 0 || prim[1] > 0) {
        int mx = prim[0] > prim[1] ? prim[0] : prim[1];
        int w0 = prim[0] == mx && prim[0] > 0, w1 = prim[1] == mx && prim[1] > 0;
        if (w0 + w1 == 1) scores[w0 ? 0 : 1] += 1;
    }
    scores[0] += s->scopas[0]; scores[1] += s->scopas[1];
    int total = scores[0] + scores[1];
    if (total == 0) { s->r2[0] = s->r2[1] = 0; }
    else { s->r2[0] = 2 * scores[0] - total; s->r2[1] = 2 * scores[1] - total; }
    s->terminal = 1;
}

void ogf_step(ogf_state *s, int action) {
    /* FullScopaEnv.step (:252-297) */
    if (s->terminal) return;
    int p = s->step & 1, pos = -1;
    for (int i = 0; i < s->nh[p]; i++) if (s->hand[p][i] == action) { pos = i; break; }
    if (pos >= 0) { /* play_card (:120-150) with capture_choice None */
        int card = action, target = f_rank(card), capmask = 0;
        for (int i = 0; i < s->nt; i++) if (f_rank(s->table[i]) == target) { capmask = 1 << i; break; }   /* exact match first */
        if (!capmask && s->nt > 0)
            for (int mask = 1; mask < (1 << s->nt); mask++) {   /* find_capture_combinations (:104-118): first subset, masks ascending */
                int sum = 0;
                for (int i = 0; i < s->nt; i++) if (mask & (1 << i)) sum += f_rank(s->table[i]);
                if (sum == target) { capmask = mask; break; }
            }
        if (capmask) {
            int8_t keep[40]; int nk = 0;
            for (int i = 0; i < s->nt; i++) {
                if (capmask & (1 << i)) s->cap[p][s->ncap[p]++] = (uint8_t)s->table[i];
                else keep[nk++] = s->table[i];
            }
            s->cap[p][s->ncap[p]++] = (uint8_t)card;
            memcpy(s->table, keep, (size_t)nk); s->nt = (int8_t)nk;
            s->last_capture = (int8_t)p;
            if (s->nt == 0) s->scopas[p]++;
        } else s->table[s->nt++] = (int8_t)card;
        for (int i = pos; i + 1 < s->nh[p]; i++) s->hand[p][i] = s->hand[p][i + 1];
        s->nh[p]--;
    }
    s->step++;
    if (s->nh[0] == 0 && s->nh[1] == 0) {
        if (40 - s->deck_pos >= 6) { /* deal_new_round (:81-88) */
            for (int q = 0; q < 2; q++) { for (int i = 0; i < 3; i++) s->hand[q][i] = (int8_t)s->deck[s->deck_pos++]; s->nh[q] = 3; }
            s->round_number++;
        } else f_evaluate(s);
    }
    if (!s->terminal && s->step >= 200) f_evaluate(s);
}

static int f_cmp_cards(const void *a, const void *b) {
    /* sorted([(rank, suit_name)]) */
    int x = *(const int *)a, y = *(const int *)b;
    if (f_rank(x) != f_rank(y)) return f_rank(x) - f_rank(y);
    return strcmp(FSUIT_NAME[f_suit(x)], FSUIT_NAME[f_suit(y)]);
}

int ogf_infoset_string(const ogf_state *s, int player, char *buf) {
    /* FullScopaState.information_state_string (openspiel_full_scopa.py:79-94) */
    int h[3], t[40];
    for (int i = 0; i < s->nh[player]; i++) h[i] = s->hand[player][i];
    for (int i = 0; i < s->nt; i++) t[i] = s->table[i];
    qsort(h, (size_t)s->nh[player], sizeof(int), f_cmp_cards);
    qsort(t, (size_t)s->nt, sizeof(int), f_cmp_cards);
    char *w = buf;
    w += sprintf(w, "P%d:R%d:H[", player, s->round_number);
    for (int i = 0; i < s->nh[player]; i++) w += sprintf(w, "%s%d%c", i ? "-" : "", f_rank(h[i]), FSUIT_NAME[f_suit(h[i])][0]);
    w += sprintf(w, "]:T[");
    for (int i = 0; i < s->nt; i++) w += sprintf(w, "%s%d%c", i ? "-" : "", f_rank(t[i]), FSUIT_NAME[f_suit(t[i])][0]);
    w += sprintf(w, "]:C[%d,%d]:S[%d,%d]", s->ncap[0], s->ncap[1], s->scopas[0], s->scopas[1]);
    return (int)(w - buf);
}

/* ==== Team MiniScopa TPI ============================================================================================ */
void ogt_reset(ogt_state *s, const uint8_t perm[16]) {
    /* TeamMiniScopaGame.reset (team_mini_scopa_game.py:68-78): 4 cards to each of the 4 players, table empty */
    memset(s, 0, sizeof *s);
    for (int p = 0; p < 4; p++) { for (int i = 0; i < 4; i++) s->hand[p][i] = (int8_t)perm[p * 4 + i]; s->nh[p] = 4; }
    s->last_capture_team = -1;
}

int ogt_current_player(const ogt_state *s) {
    /* TPIMiniScopaState.current_player (openspiel_team_mini_scopa.py:23-29): the TEAM of the seat to move */
    return s->terminal ? -4 : ((s->step % 4) / 2);
}

int ogt_legal(const ogt_state *s, int out[4]) {
    /* legal_actions (:52-86): the hand of the seat to move, in hand order, whatever `player` says; [0] fallback; [] at terminal */
    if (s->terminal) return 0;
    int seat = s->step % 4, n = 0;
    for (int i = 0; i < s->nh[seat]; i++) out[n++] = s->hand[seat][i];
    if (!n) { out[0] = 0; n = 1; }
    return n;
}

static int t_capture(const ogt_state *s, int card, int cap_idx[16]) {
    /* TeamMiniScopaGame.card_in_table (:84-109): same rule as MiniScopa's */
    int target = RANK_OF[card];
    if (target <= 0 || s->nt == 0) return 0;
    for (int i = 0; i < s->nt; i++) if (RANK_OF[s->table[i]] == target) { cap_idx[0] = i; return 1; }
    int have[11], len[11], comb[11][16];
    memset(have, 0, sizeof have);
    have[0] = 1; len[0] = 0;
    for (int idx = 0; idx < s->nt; idx++) {
        int r = RANK_OF[s->table[idx]];
        for (int sum = target; sum >= r; sum--)
            if (!have[sum] && have[sum - r]) {
                have[sum] = 1; len[sum] = len[sum - r] + 1;
                memcpy(comb[sum], comb[sum - r], sizeof(int) * (size_t)len[sum - r]);
                comb[sum][len[sum] - 1] = idx;
            }
    }
    if (!have[target]) return 0;
    for (int i = 0; i < len[target]; i++) cap_idx[i] = comb[target][i];
    return len[target];
}

static void t_evaluate(ogt_state *s) {
    /* evaluate_game (:125-155): leftover table cards go to the FIRST player of the last capturing team (the table itself is
     * left as it is); team score = sum of len(captures) + 2*scopas; zero-sum around the mean */
    if (s->nt > 0 && s->last_capture_team >= 0) {
        int p = s->last_capture_team * 2;
        for (int i = 0; i < s->nt; i++) s->cap[p][s->ncap[p]++] = s->table[i];
    }
    int score[2] = {0, 0};
    for (int p = 0; p < 4; p++) score[p / 2] += s->ncap[p] + 2 * s->scopas[p];
    int total = score[0] + score[1];
    for (int p = 0; p < 4; p++) s->r2[p] = total == 0 ? 0 : 2 * score[p / 2] - total;
}

void ogt_step(ogt_state *s, int action) {
    /* TPIMiniScopaState.apply_action (:88-92) + TeamMiniScopaEnv.step (:173-201) + play_card (:111-123) */
    if (s->nhist < 16) s->history[s->nhist++] = (int8_t)action;   /* action_history grows even on a dead step */
    if (s->terminal) return;                                        /* _was_dead_step */
    int seat = s->step % 4, pos = -1;
    for (int i = 0; i < s->nh[seat]; i++) if (s->hand[seat][i] == action) { pos = i; break; }
    if (pos >= 0) {
        int cap[16];
        int nc = t_capture(s, action, cap);
        if (nc > 0) {
            int8_t keep[16]; int nk = 0;
            for (int i = 0; i < s->nt; i++) {
                int taken = 0;
                for (int j = 0; j < nc; j++) if (cap[j] == i) taken = 1;
                if (taken) s->cap[seat][s->ncap[seat]++] = s->table[i]; else keep[nk++] = s->table[i];
            }
            s->cap[seat][s->ncap[seat]++] = (int8_t)action;
            memcpy(s->table, keep, (size_t)nk); s->nt = (int8_t)nk;
            s->last_capture_team = (int8_t)(seat / 2);
            if (nk == 0) s->scopas[seat]++;
        } else s->table[s->nt++] = (int8_t)action;
        for (int i = pos; i + 1 < s->nh[seat]; i++) s->hand[seat][i] = s->hand[seat][i + 1];
        s->nh[seat]--;
    }
    s->step++;
    if ((s->nh[0] | s->nh[1] | s->nh[2] | s->nh[3]) == 0 || s->step >= 16) { s->terminal = 1; t_evaluate(s); }
}

static const char *TSUIT_NAME[4] = {"cuori", "fiori", "picche", "bello"};
static int t_cmp_cards(const void *a, const void *b) {
    /* sorted([(rank, suit_name)]) (:135-139) */
    int x = *(const int *)a, y = *(const int *)b;
    if (RANK_OF[x] != RANK_OF[y]) return RANK_OF[x] - RANK_OF[y];
    return strcmp(TSUIT_NAME[x / 4], TSUIT_NAME[y / 4]);
}

int ogt_infoset_string(const ogt_state *s, int team, char *buf) {
    /* information_state_string (:119-146): the seat to move if it belongs to `team`, else the team's first seat; hand and
     * table SORTED by (rank, suit name); the whole action history is part of the key */
    int seat = s->step % 4;
    if (seat / 2 != team) seat = team * 2;
    int h[4], t[16];
    for (int i = 0; i < s->nh[seat]; i++) h[i] = s->hand[seat][i];
    for (int i = 0; i < s->nt; i++) t[i] = s->table[i];
    qsort(h, (size_t)s->nh[seat], sizeof(int), t_cmp_cards);
    qsort(t, (size_t)s->nt, sizeof(int), t_cmp_cards);
    char *w = buf;
    w += sprintf(w, "Team%d:P%d:H[", team, seat);
    for (int i = 0; i < s->nh[seat]; i++) w += sprintf(w, "%s%d%c", i ? "-" : "", RANK_OF[h[i]], TSUIT_NAME[h[i] / 4][0]);
    w += sprintf(w, "]:T[");
    for (int i = 0; i < s->nt; i++) w += sprintf(w, "%s%d%c", i ? "-" : "", RANK_OF[t[i]], TSUIT_NAME[t[i] / 4][0]);
    w += sprintf(w, "]:A[");
    for (int i = 0; i < s->nhist; i++) w += sprintf(w, "%s%d", i ? "-" : "", s->history[i]);
    w += sprintf(w, "]");
    return (int)(w - buf);
}

