/*
 * scopa_oracle.h -- CPU ORACLE for the MiniScopa CFR traversal path.
 *
 * TEST INFRASTRUCTURE.  This is a plain-C restatement of the reference's algorithm
 * (rug-marl-group2/scopa, Python).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may call it -- as the checker / reported baseline,
 * never as the product.  The product (scopa_amd/, libscopa_hip.so) does not link,
 * import or fall back to anything in oracle/.
 *
 * Pinning: every function here is checked against fixtures produced by RUNNING the
 * reference's own Python in the build container (oracle/gen_golden.py ->
 * tests/golden/): deals, full game trees of 5 deals, 160 playouts with illegal
 * actions, vanilla-CFR tables after 1/2/5/50/200 iterations (bit-exact), MCCFR tables
 * under np.random.seed(k) (bit-exact).  Exploitability and the batched (frozen-table)
 * MCCFR have no reference counterpart: "parity unpinned" vs the reference, defined here.
 *
 * Citations are relative to /root/reference/.
 */
#ifndef SCOPA_ORACLE_H
#define SCOPA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- game state, kept the way the reference keeps it: ordered lists --------------- */
typedef struct {
    int8_t hand[2][4];   /* card ids in hand order (src/envs/mini_scopa_game.py:39,62) */
    int8_t nh[2];
    int8_t table[8];     /* card ids in table (insertion) order (:53,103)              */
    int8_t nt;
    int8_t ncap[2];      /* len(player.captures) (:98)                                 */
    int8_t scopas[2];    /* (:100-101)                                                 */
    int8_t step;         /* env.step_count (:138,159)                                  */
    int8_t max_steps;    /* env.max_steps: num_players * 4 = 8 for a fresh env (:127), 16 for the env of a
                            MiniScopaState.clone() (src/envs/openspiel_mini_scopa.py:108)                 */
} og_state;

/* card id = action id = suit_idx*4 + rank_idx (mini_scopa_game.py:17-23,149-153) */
int  og_card_rank(int card);
char og_card_suit_char(int card);

/* random.seed(seed); random.shuffle(16 cards) of CPython 3.x (mini_scopa_game.py:25-28) */
void og_deal_py_seed(int64_t seed, uint8_t perm[16]);
/* MiniScopaGame.reset (:56-64): first 4 of perm -> P0 hand, next 4 -> P1 hand, table empty */
void og_reset(og_state *s, const uint8_t perm[16]);
/* openspiel_mini_scopa.py:17-45: legal actions in HAND ORDER; [0] fallback; [] at terminal. player<0 = current */
int  og_legal(const og_state *s, int player, int out[4]);
int  og_is_terminal(const og_state *s);                 /* mini_scopa_game.py:160: all hands empty or step_count >= max_steps */
/* MiniScopaState.clone (src/envs/openspiel_mini_scopa.py:97-115): the copy's env is built by hand with max_steps = 16 (:108) and
 * filled through get_state / set_state, which carry neither max_steps nor the seed's deck */
void og_clone(const og_state *src, og_state *dst);
int  og_current_player(const og_state *s);              /* -4 at terminal (openspiel…:17-20) */
/* MiniScopaEnv.step (:140-167) incl. the silent no-op for a card not in hand */
void og_step(og_state *s, int action);
/* capture rule card_in_table (:66-91): returns number captured, indices into table in cap_idx */
int  og_capture(const og_state *s, int card, int cap_idx[8]);
/* evaluate_game (:106-114), times two so that it is an integer: r2[i] = 2*r_i - total */
void og_rewards_x2(const og_state *s, int r2[2]);
/* information_state_string (openspiel…:86-95); returns strlen */
int  og_infoset_string(const og_state *s, int player, char *buf);

/* ---- flat game tree in reference DFS order ---------------------------------------- */
typedef struct {
    int n_nodes, n_decision, n_infosets;
    int8_t  *term, *player, *nlegal, *depth;
    int16_t *infoset;          /* dense id in DFS first-visit order of the STRING key, -1 at terminals */
    int8_t  *legal;            /* [n][4] action ids  */
    int32_t *child;            /* [n][4] node ids    */
    int8_t  *r2;               /* [n][2] rewards x2  */
    og_state *state;           /* [n] */
    char   (*infoset_str)[64]; /* [n_infosets] */
    int8_t  *infoset_nlegal;   /* [n_infosets] */
    int8_t  *infoset_legal;    /* [n_infosets][4] */
    int8_t  *infoset_player;   /* [n_infosets] */
} og_tree;

og_tree *og_tree_build(const uint8_t perm[16]);
void     og_tree_free(og_tree *t);
/* accessors for ctypes */
int og_tree_counts(const og_tree *t, int *n_nodes, int *n_decision, int *n_infosets);
int og_tree_export(const og_tree *t, int8_t *term, int8_t *player, int8_t *nlegal, int8_t *depth,
                   int16_t *infoset, int8_t *legal, int32_t *child, int8_t *r2);
int og_tree_states(const og_tree *t, int8_t *hands /*[n][2][4]*/, int8_t *nh /*[n][2]*/, int8_t *table /*[n][8]*/,
                   int8_t *nt, int8_t *ncap /*[n][2]*/, int8_t *scopas /*[n][2]*/, int8_t *step);
int og_tree_infoset_string(const og_tree *t, int id, char *buf);
int og_tree_infoset_meta(const og_tree *t, int8_t *nlegal, int8_t *legal /*[I][4]*/, int8_t *player);

/* ---- solvers. Tables are [n_infosets][4] f64, rows padded with 0 ------------------- */
/* vanilla_cfr.py:56-120, sequential DFS semantics incl. the mid-traversal local_strategy
 * refresh (:97).  `local` must be initialised by og_tables_init. root_values: [n_iters][2] or NULL */
void og_tables_init(const og_tree *t, double *regret, double *strat, double *local);
void og_cfr_exact(const og_tree *t, double *regret, double *strat, double *local, int n_iters, double *root_values);
double og_cfr_exact_from(const og_tree *t, double *regret, double *strat, double *local, const int *path, int depth,
                         int trav, double r0, double r1);
/* mc_cfr.py:37-92 replayed from a host-supplied uniform stream (one double per decision
 * visit, DFS order; np.random.choice == searchsorted(cumsum(p)/sum, u, 'right')).
 * Returns the number of uniforms consumed. */
int64_t og_mccfr_replay(const og_tree *t, double *regret, double *strat, int n_iters, const double *uniforms, int64_t n_uniforms);

/* Batched external-sampling MCCFR (build-defined; tables FROZEN per iteration, both
 * traversers, traversal ids [b0, b0+nb) of a global batch; path-keyed Philox4x32-10).
 * Accumulates into delta tables (caller zeroes them / applies them). */
void og_mccfr_batched_delta(const og_tree *t, const double *regret, double *d_regret, double *d_strat,
                            uint64_t seed, uint32_t iteration, uint32_t b0, uint32_t nb,
                            uint64_t *decision_visits, uint64_t *terminal_visits);
/* full iterations: delta over [0,batch) then regret += d, strat += d */
void og_mccfr_batched(const og_tree *t, double *regret, double *strat, uint64_t seed, uint32_t iter0,
                      uint32_t n_iters, uint32_t batch, uint64_t *decision_visits);
/* records the sampled action of every decision visit of one traversal, in DFS order (for integer parity) */
int  og_mccfr_batched_trace(const og_tree *t, const double *regret, uint64_t seed, uint32_t iteration,
                            uint32_t b, int traverser, int32_t *nodes, int8_t *actions, int max_out);

void   og_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double og_philox_uniform(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3);

/* Single Deep CFR: nb external-sampling traversals for `traverser` (deep_cfr.py:284-365), one after the other as the reference
 * runs them, with the advantage nets given as two torch state dicts flattened in their own order (float32, [2][13776]).
 * Opponent draws: `uniforms` (one per opponent visit in DFS order -- what np.random.choice consumes) or, if NULL, the build's
 * Philox block keyed by (frontier slot + 1024*ply, traversal id, iteration, 4 + traverser).  Rows (features[34], normalised
 * regrets[16], mask[16]) are written in the reference's append order, 41 per traversal; returns the number of rows. */
int64_t og_sdcfr_traverse(const og_tree *t, const float *nets, int traverser, uint64_t seed, uint32_t iteration, uint32_t b0,
                          uint32_t nb, const double *uniforms, int64_t n_uniforms, float *row_feat, float *row_regret,
                          float *row_mask, float *values, uint64_t *decision_visits);

/* Synchronous ("frozen strategy") CFR, build-defined: sigma = RM(regret) frozen over the
 * whole iteration, both players updated from one sweep. */
void og_cfr_sync(const og_tree *t, double *regret, double *strat, int n_iters);

/* Average policy S/sum(S) (uniform if sum<=0) -> policy[I][4]; vanilla_cfr.py:32-39 */
void og_average_policy(const og_tree *t, const double *strat, double *policy);
/* Expected value for P0 of a behavioural policy pair, and best-response values (build-defined
 * exploitability = (BR0 + BR1)/2 with BRi = value of player i's best response per infoset STRING). */
double og_policy_value(const og_tree *t, const double *policy);
double og_exploitability(const og_tree *t, const double *policy, double *br_values /*[2] or NULL*/);

#ifdef __cplusplus
}
#endif
#endif

/* ==== FullScopa (40 cards), test-side restatement of src/envs/full_scopa_game.py + openspiel_full_scopa.py ========== */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    uint8_t deck[40];        /* FullDeck(seed).cards as card ids (suit_idx*10 + rank-1), deal order */
    int8_t deck_pos;         /* cards dealt so far */
    int8_t hand[2][3]; int8_t nh[2];
    int8_t table[40]; int8_t nt;
    uint8_t cap[2][40]; int8_t ncap[2];   /* captures as lists (order irrelevant for scoring) */
    int8_t scopas[2];
    int8_t round_number, last_capture /* -1 none */;
    int16_t step;
    int8_t terminal;
    int r2[2];               /* rewards x2 once terminal */
} ogf_state;
void ogf_deal_py_seed(int64_t seed, uint8_t perm[40]);
void ogf_reset(ogf_state *s, const uint8_t perm[40]);
int  ogf_legal(const ogf_state *s, int player, int out[3]);
void ogf_step(ogf_state *s, int action);
int  ogf_infoset_string(const ogf_state *s, int player, char *buf);
#ifdef __cplusplus
}
#endif

/* ==== Team MiniScopa TPI (4 seats, 16 cards, 16 plies), test-side restatement of src/envs/team_mini_scopa_game.py +
 * src/envs/openspiel_team_mini_scopa.py ============================================================================== */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    int8_t hand[4][4]; int8_t nh[4];       /* hands in deal order (team_mini_scopa_game.py:74-76) */
    int8_t table[16]; int8_t nt;           /* table in insertion order */
    int8_t cap[4][16]; int8_t ncap[4];     /* captures as lists */
    int8_t scopas[4];
    int8_t last_capture_team;              /* -1 = None */
    int8_t step;                           /* env.step_count; seat to move = step % 4, team = seat / 2 */
    int8_t terminal;
    int8_t history[16]; int8_t nhist;      /* TPIMiniScopaState.action_history */
    int r2[4];                             /* per-seat rewards x2 once terminal */
} ogt_state;
void ogt_reset(ogt_state *s, const uint8_t perm[16]);
int  ogt_legal(const ogt_state *s, int out[4]);
void ogt_step(ogt_state *s, int action);
int  ogt_current_player(const ogt_state *s);                         /* team id, -4 at terminal */
int  ogt_infoset_string(const ogt_state *s, int team, char *buf);
#ifdef __cplusplus
}
#endif
