/*
 * scopa.h -- C ABI of libscopa_hip.so, the MI355X (gfx950) MiniScopa CFR traversal engine.
 *
 * The reference (rug-marl-group2/scopa) has no FFI: its boundary is a Python object protocol.
 * Each entry point below therefore cites the reference Python interface it stands behind
 * (paths relative to the reference repo root); INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, every function returns int32 status: 0 = SCOPA_OK, negative = SCOPA_E*.
 *     No exception or abort crosses the boundary; scopa_last_error(ctx) gives the detail string.
 *   - opaque scopa_ctx: one context <-> one HIP device + one HIP stream (+ one deal / tree / table set).
 *     A context is not thread-safe; distinct contexts are independent.
 *   - bulk buffers are caller-owned and only borrowed for the duration of the call.  Arguments named
 *     d_* are DEVICE pointers (e.g. torch.Tensor.data_ptr()), h_* are host pointers.
 *   - solver entry points need a GPU: scopa_ctx_create fails with SCOPA_ENODEV when there is none.
 *     There is no CPU fallback.  The scopa_state_* / scopa_deal_* helpers are host-side glue for the
 *     single-state object protocol (State.apply_action etc.) and need no context.
 */
#ifndef SCOPA_H
#define SCOPA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCOPA_ABI_VERSION 1

enum {
    SCOPA_OK = 0,
    SCOPA_EINVAL = -1,   /* bad argument                                   */
    SCOPA_ENODEV = -2,   /* no usable HIP device                           */
    SCOPA_EHIP = -3,     /* a HIP runtime call failed (see last_error)     */
    SCOPA_ESTATE = -4,   /* call order violated (e.g. no deal set)         */
    SCOPA_ENOMEM = -5,
    SCOPA_ELIMIT = -6,   /* problem exceeds a compiled-in capacity         */
    SCOPA_ETIMEOUT = -7  /* a peer did not answer within the wait budget (N > 1 exchange): tables are not to be trusted */
};

/* Deal-independent shape of MiniScopa (src/envs/mini_scopa_game.py:59,127): 2 players x 4 cards,
 * 8 plies, legal-count profile 4,4,3,3,2,2,1,1. */
#define SCOPA_N_CARDS 16
#define SCOPA_N_PLIES 8
#define SCOPA_N_DECISION 1653
#define SCOPA_N_TERMINAL 576
#define SCOPA_N_NODES 2229
#define SCOPA_MAX_ACTIONS 4

/* Packed game state, 16 bytes, the unit the kernels keep in HBM.  Hands and table are ORDERED lists of
 * 4-bit card ids (card id = action id = suit*4 + rank_idx, mini_scopa_game.py:17-23,149-153); list order
 * is part of the infoset identity (openspiel_mini_scopa.py:86-95).  Unused nibbles are zero.
 * to_move = step & 1; terminal iff (nh[0]==0 && nh[1]==0) || step_count >= max_steps (mini_scopa_game.py:160), where
 * step_count = step & SCOPA_STEP_COUNT_MASK and max_steps = 8 (a fresh env, mini_scopa_game.py:127), or 16 when
 * SCOPA_STEP_CLONED is set: the env of a MiniScopaState.clone() (openspiel_mini_scopa.py:108).  Only illegal no-op actions make
 * the two differ; scopa_state_clone() sets the bit, every other entry point preserves it. */
#define SCOPA_STEP_CLONED 0x80u
#define SCOPA_STEP_COUNT_MASK 0x7Fu
typedef struct scopa_state {
    uint16_t hand[2];   /* nibble i = i-th card of the hand                  */
    uint32_t table;     /* nibble i = i-th card on the table                 */
    uint8_t  nh[2];     /* cards in hand                                     */
    uint8_t  nt;        /* cards on table                                    */
    uint8_t  step;      /* MiniScopaEnv.step_count | SCOPA_STEP_CLONED       */
    uint8_t  ncap[2];   /* len(player.captures)                              */
    uint8_t  scopas[2]; /* player.scopas                                     */
} scopa_state;

typedef struct scopa_ctx scopa_ctx;

int32_t     scopa_abi_version(void);
const char *scopa_strerror(int32_t status);
const char *scopa_last_error(const scopa_ctx *ctx);

/* device_id >= 0.  hip_stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream),
 * or NULL to let the context create its own. */
int32_t scopa_ctx_create(int32_t device_id, void *hip_stream, scopa_ctx **out);
int32_t scopa_ctx_destroy(scopa_ctx *ctx);
int32_t scopa_ctx_synchronize(scopa_ctx *ctx);

/* ---- host-side single-state protocol (no context, no device) ------------------------------------------
 * stands behind MiniScopaState / MiniScopaEnv: src/envs/openspiel_mini_scopa.py:17-115,
 * src/envs/mini_scopa_game.py:117-194 */
int32_t scopa_deal_py_seed(int64_t seed, uint8_t perm16[16]);              /* MiniDeck.__init__ :25-28        */
int32_t scopa_state_init(const uint8_t perm16[16], scopa_state *out);      /* MiniScopaGame.reset :56-64      */
int32_t scopa_state_step(scopa_state *s, int32_t action);                  /* MiniScopaEnv.step :140-167      */
int32_t scopa_state_clone(const scopa_state *s, scopa_state *out);         /* MiniScopaState.clone, openspiel_mini_scopa.py:97-115: a copy whose max_steps is 16 (:108) */
int32_t scopa_state_is_terminal(const scopa_state *s);                     /* 0/1                             */
int32_t scopa_state_current_player(const scopa_state *s);                  /* 0/1, -4 at terminal             */
int32_t scopa_state_legal(const scopa_state *s, int32_t player /* <0 = current */, int32_t out[4], int32_t *n);
int32_t scopa_state_rewards_x2(const scopa_state *s, int32_t r2[2]);       /* evaluate_game :106-114, x2      */
int32_t scopa_state_infoset_key(const scopa_state *s, int32_t player, uint64_t *key);
int32_t scopa_key_to_string(uint64_t key, char *buf, int32_t cap);         /* information_state_string :86-95 */
int32_t scopa_state_infoset_string(const scopa_state *s, int32_t player, char *buf, int32_t cap);

/* ---- batched game step (HIP kernel) --------------------------------------------------------------------
 * d_states[i] <- step(d_states[i], d_actions[i]); same semantics as scopa_state_step incl. the silent no-op. */
int32_t scopa_step_batch(scopa_ctx *ctx, scopa_state *d_states, const uint8_t *d_actions, int64_t n);
/* convenience for host buffers: H2D, kernel, D2H on the context's stream, synchronous */
int32_t scopa_step_batch_host(scopa_ctx *ctx, scopa_state *h_states, const uint8_t *h_actions, int64_t n);

/* ---- deal + game tree (built ON DEVICE by level-synchronous expansion with the step kernel) ------------
 * stands behind game.new_initial_state() + the clone()/apply_action() recursion of every solver */
int32_t scopa_set_deal(scopa_ctx *ctx, const uint8_t perm16[16]);          /* builds tree, zeroes tables      */
int32_t scopa_tree_counts(scopa_ctx *ctx, int32_t *n_nodes, int32_t *n_decision, int32_t *n_infosets);
/* Export in REFERENCE DFS ORDER (the order vanilla_cfr.py:79-85 visits nodes); any pointer may be NULL.
 * h_states[n_nodes], h_infoset[n_nodes] (-1 at terminals; ids in first-visit order = dict insertion order),
 * h_r2[n_nodes][2], h_infoset_key[n_infosets], h_infoset_nlegal[n_infosets], h_infoset_legal[n_infosets][4] */
int32_t scopa_tree_export(scopa_ctx *ctx, scopa_state *h_states, int32_t *h_infoset, int8_t *h_r2,
                          uint64_t *h_infoset_key, int8_t *h_infoset_nlegal, int8_t *h_infoset_legal);

/* ---- tables: [n_infosets][4] float64, rows padded with 0; any pointer may be NULL ----------------------
 * stands behind CFRTrainer.info_set_map / MCCFRTrainer.info_sets (InfoNode.regret_sum, .strategy_sum,
 * .local_strategy): src/algorithms/vanilla_cfr.py:8-39,47-54, src/algorithms/mc_cfr.py:9-35 */
int32_t scopa_tables_reset(scopa_ctx *ctx);
int32_t scopa_tables_get(scopa_ctx *ctx, double *h_regret, double *h_strategy, double *h_local);
int32_t scopa_tables_set(scopa_ctx *ctx, const double *h_regret, const double *h_strategy, const double *h_local);

/* First-visit order of every infoset, h_seq[n_infosets]: 0 = never visited by any solver since the last reset,
 * otherwise a strictly increasing sequence number (sequential solvers) or 0x40000000 + id (first seen by a batched
 * launch).  Mirrors which keys the reference's dicts hold, and in which insertion order
 * (CFRTrainer._get_or_create_node vanilla_cfr.py:51-54, MCCFRTrainer._get_node mc_cfr.py:32-35). */
int32_t scopa_visited_get(scopa_ctx *ctx, uint32_t *h_seq);

/* ---- vanilla CFR, exact sequential semantics (CFRTrainer._cfr_recursive / .train, vanilla_cfr.py:56-110)
 * n_iters iterations of "for i in (0,1): traverse(root, i, 1.0, 1.0)"; h_root_values[n_iters][2] or NULL.
 * Reproduces the reference bit-for-bit, incl. the mid-traversal local_strategy refresh (:97). */
int32_t scopa_cfr_exact_iterate(scopa_ctx *ctx, int32_t n_iters, double *h_root_values);
/* one traversal: CFRTrainer._cfr_recursive(new_initial_state(), player, 1.0, 1.0) -> value */
int32_t scopa_cfr_exact_traverse(scopa_ctx *ctx, int32_t traverser, double *h_value);
/* whole-tree traversals run as a schedule of ~75 parallel steps (same per-infoset visit order, bit-identical tables); 1 forces the
 * one-lane sequential walk, the form the schedule is checked against */
int32_t scopa_cfr_exact_mode(scopa_ctx *ctx, int32_t sequential);
/* CFRTrainer._cfr_recursive(state, player, reach_p0, reach_p1) for any state of the tree: the state reached from the
 * root by legal-action INDICES path[0..depth) (index into legal_actions(), i.e. hand position) */
int32_t scopa_cfr_exact_traverse_from(scopa_ctx *ctx, int32_t traverser, int32_t depth, const int32_t *path,
                                      double reach_p0, double reach_p1, double *h_value);

/* ---- synchronous CFR (build-defined): sigma = regret-match(regret) frozen per iteration, both players updated from
 * one level-parallel sweep.  Not the reference's visit-order-dependent algorithm: same fixed point, different trajectory. */
int32_t scopa_cfr_sync_iterate(scopa_ctx *ctx, int32_t n_iters);

/* ---- MCCFR replay: MCCFRTrainer.iteration() (mc_cfr.py:37-92) driven by a host-supplied uniform stream
 * (one float64 per decision visit in DFS order = what np.random.choice draws); bit-exact vs the reference. */
int32_t scopa_mccfr_replay(scopa_ctx *ctx, int32_t n_iters, const double *h_uniforms, int64_t n_uniforms,
                           int64_t *consumed);

/* ---- batched external-sampling MCCFR (the throughput path) ----------------------------------------------
 * `batch` traversals per traverser per iteration against tables frozen for the iteration; random draws from
 * Philox4x32-10 keyed by (seed; recursion-path code, global traversal id, iteration, traverser), so results do
 * not depend on how traversals are split over launches or GPUs. */
int32_t scopa_mccfr_seed(scopa_ctx *ctx, uint64_t seed);
/* n_iters x { traverse [0,batch) ; apply } on this device */
int32_t scopa_mccfr_iterate(scopa_ctx *ctx, uint32_t batch, uint32_t n_iters);
/* multi-GPU building blocks: accumulate the deltas of global traversal ids [b0, b0+nb) of `iteration` ... */
int32_t scopa_mccfr_traverse(scopa_ctx *ctx, uint32_t iteration, uint32_t b0, uint32_t nb);
/* ... expose the delta buffer ([n_infosets][5] float64: 4 regret deltas + traverser-visit count) for one
 * sum-all-reduce (RCCL via torch.distributed) ... */
int32_t scopa_mccfr_delta_buffer(scopa_ctx *ctx, void **d_delta, size_t *bytes);
/* ... or bind a CALLER-OWNED device buffer of at least n_infosets*5 float64 (e.g. a torch tensor handed to
 * torch.distributed.all_reduce) as the delta buffer; it is zeroed here.  d_buf = NULL returns to the internal one.
 * The binding lasts until the next scopa_set_deal / scopa_mccfr_bind_delta. */
int32_t scopa_mccfr_bind_delta(scopa_ctx *ctx, void *d_buf, size_t bytes);
/* host copy of the delta buffer, h_delta[n_infosets][5] (tests, non-RCCL transports) and its inverse */
int32_t scopa_mccfr_delta_get(scopa_ctx *ctx, double *h_delta);
int32_t scopa_mccfr_delta_set(scopa_ctx *ctx, const double *h_delta);
/* ... then regret += delta[:, :4]; strategy += count * sigma; delta <- 0; iteration counter += 1 */
int32_t scopa_mccfr_apply(scopa_ctx *ctx);
int32_t scopa_mccfr_iteration_counter(scopa_ctx *ctx, uint32_t *iteration);
/* Graph mode of scopa_mccfr_iterate: on = chunks of up to 64 iterations are replayed as ONE captured HIP graph of (traverse, apply)
 * launches each (captured once per (batch, chunk length) and deal), the iteration number read from a device word that the apply
 * launch advances -- the same iteration ids, hence the same draws and results, as the eager loop.  Off by default (measured:
 * DESIGN.md section 4). */
int32_t scopa_mccfr_graph_mode(scopa_ctx *ctx, int32_t on);
/* Test hook: treat the device as offering only `bytes` of LDS per workgroup (0 = its real limit again), so that the launch
 * geometries only deals with very many infosets reach -- narrow traversal workgroups -- run on any deal.  Results do not change. */
int32_t scopa_debug_lds_limit(scopa_ctx *ctx, int32_t bytes);

/* ---- SDCFR: level-synchronous external-sampling traversal (DeepCFR._external_sampling_cfr, deep_cfr.py:284-365) ---------
 * B traversals of one traverser advance ply by ply; the advantage MLP runs in PyTorch between the two calls of a ply.
 * Frontier slot s of ply d refers to tree node d_idx[s] (index within the ply); a traversal owns `width` consecutive slots
 * (scopa_sdcfr_frontier_width: 1,4,4,12,12,24,24,24,24 for traverser 0; 1,1,4,4,12,12,24,24,24 for traverser 1).
 *   features : DeepCFR._state_to_features + ._get_legal_actions_mask (:213-282) -> d_feats[n][34], d_mask[n][16] float32
 *   expand   : positive_regret_policy (nets.py:93-101) of d_adv[n][16] (raw net output); traverser ply -> all legal
 *              children d_child_idx[n*nlegal] (hand order), opponent ply -> ONE sampled child d_child_idx[n]
 *              (np.random.choice arithmetic; draws from d_uniforms[n] if given, else Philox keyed by the global
 *              traversal id b0 + s/width); d_pol[n][4] = policy of the legal actions in hand order
 *   terminal_values : float(rewards[player]) at ply 8
 *   backward : traverser ply: value = sum pol*child value (float32, hand order), regrets = cfv - value over all 16 slots
 *              (illegal slots = -value, as the reference), divided by max|.|+1e-8, and the memory row (feats, regrets, mask)
 *              written to ring position (write_base + traversal*41 + DFS-post-order rank) % capacity, i.e. in the
 *              reference's append order (AdvantageNetwork.add_experience :70-75); opponent ply: value = child's value */
int32_t scopa_sdcfr_frontier_width(int32_t traverser, int32_t ply);
int32_t scopa_sdcfr_features(scopa_ctx *ctx, int32_t ply, int64_t n, const int32_t *d_idx, float *d_feats, float *d_mask);
int32_t scopa_sdcfr_expand(scopa_ctx *ctx, int32_t ply, int32_t traverser, int64_t n, const int32_t *d_idx, const float *d_adv,
                           int32_t *d_child_idx, float *d_pol, const double *d_uniforms, uint32_t iteration, uint32_t b0);
int32_t scopa_sdcfr_terminal_values(scopa_ctx *ctx, int32_t traverser, int64_t n, const int32_t *d_idx, float *d_val);
int32_t scopa_sdcfr_backward(scopa_ctx *ctx, int32_t ply, int32_t traverser, int64_t n, const int32_t *d_idx, const float *d_pol,
                             const float *d_child_val, float *d_val, const float *d_feats, const float *d_mask, float *d_mem_feat,
                             float *d_mem_regret, float *d_mem_mask, int64_t capacity, int64_t write_base);
int32_t scopa_sdcfr_visits(scopa_ctx *ctx, uint64_t *decision_visits);
/* The same traversal as ONE launch: a wavefront walks four traversals together, both players' advantage MLPs (34-128-64-16
 * float32) resident in LDS and evaluated in-kernel on the matrix cores, sixteen frontier nodes per tile.
 * d_image[2][SCOPA_SDCFR_IMAGE_FLOATS]: per player the net as scopa_sdcfr_pack_weights lays it out (the operand layout of
 * v_mfma_f32_16x16x4_f32; the kernel copies it to LDS as it is).  d_uniforms (optional, tests): [batch][8][24] float64 draws
 * indexed (traversal, ply, slot).  d_mem_feat must be 8-byte, d_mem_regret / d_mem_mask 16-byte aligned.
 * d_mem_mask may be NULL (here and in scopa_sdcfr_backward): a traverser node's legal actions are the cards of the mover's hand
 * (openspiel_mini_scopa.py:36-45) and features[0..16) are that hand's one-hot (deep_cfr.py:213-275), so the mask row equals the first
 * sixteen floats of the feature row and need not be written a second time -- a memory row is then 200 bytes of HBM instead of 264.
 * Samples the same actions as the ply-by-ply path (same Philox keying); float32 sums run in a different order. */
#define SCOPA_SDCFR_IMAGE_FLOATS 13520
int32_t scopa_sdcfr_image_floats(void);
/* One advantage net (AdvantageNetwork.net = FlexibleNet(mode="mlp"), nets.py:296-331; torch tensors, row-major W[out][in]:
 * backbone.0.fc.weight [128][34] / .bias [128], backbone.1.fc.weight [64][128] / .bias [64], head.weight [16][64] / .bias [16],
 * all float32 device pointers) -> player's half of d_image.  One small launch on the context's stream; call it again
 * whenever the net changed (an optimiser step, load_state_dict). */
int32_t scopa_sdcfr_pack_weights(scopa_ctx *ctx, int32_t player, const float *d_w1, const float *d_b1, const float *d_w2,
                                 const float *d_b2, const float *d_w3, const float *d_b3, float *d_image);
int32_t scopa_sdcfr_traverse_fused(scopa_ctx *ctx, int32_t traverser, int32_t batch, const float *d_image, float *d_mem_feat,
                                   float *d_mem_regret, float *d_mem_mask, int64_t capacity, int64_t write_base,
                                   float *d_root_values, const double *d_uniforms, uint32_t iteration, uint32_t b0);
/* How scopa_sdcfr_traverse_fused evaluates the advantage nets.  0 (default): ONCE per decision node of the deal and launch -- the nets
 * are frozen while a launch runs and a node's features depend on the tree node alone, so its 1 653 policies are computed by one
 * small launch on the matrix cores and the traversals walk that table (two launches; the memory rows bound it).  1: a forward pass
 * per visit inside the traversal kernel (one launch; the form for batches that would not share a deal; also what d_uniforms takes).
 * Same rows, same values, same sampled actions either way. */
int32_t scopa_sdcfr_mode(scopa_ctx *ctx, int32_t forward_per_visit);
/* experiments: traversals per task of the traversal kernels (0 = the library's choice; 1, 2, 4 or 8 for the walk kernel, 2 or 4 for
 * the forward-per-visit kernel) and wavefronts that share a task's tiles in the latter (0 = the library's choice, 1..3).  Results
 * do not depend on either. */
int32_t scopa_sdcfr_tuning(scopa_ctx *ctx, int32_t traversals_per_task, int32_t wavefronts_per_task);
/* One optimiser step of an advantage net WITHOUT PyTorch kernels (AdvantageNetwork.train, deep_cfr.py:99-112): gather the n_rows ring rows d_rows[],
 * forward 34-128-64-16, MSE(pred * mask, target * mask) over n_rows x 16, backward, clip_grad_norm_(1.0), Adam(lr, betas 0.9 / 0.999, eps 1e-8).
 * OPT-IN (the default trains with PyTorch-ROCm): DeepCFR(train_backend="hip").  d_w1 .. d_b3 are the net's own tensors (torch layout W[out][in]),
 * updated in place; d_state = [2][13776] float (exp_avg, exp_avg_sq in net.parameters() order), zero before the first step; step = 1, 2, ...;
 * the step's loss is ADDED to d_loss[0].  n_rows: a multiple of 16.  d_mask NULL = every row's mask is its features[0..16) (rows written by the
 * traversal calls with d_mem_mask NULL).  Two launches (k_sdcfr_train_grad, k_sdcfr_train_adam), no host synchronisation. */
int32_t scopa_sdcfr_train_params(void);   /* 13 776 */
/* n_steps consecutive steps in one call: step e trains on d_rows[e * n_rows .. (e + 1) * n_rows) with step number first_step + e (the epochs of one train() call) */
int32_t scopa_sdcfr_train_steps(scopa_ctx *ctx, const int64_t *d_rows, int32_t n_rows, int32_t n_steps, const float *d_feat, const float *d_regret, const float *d_mask,
                                int64_t capacity, float *d_w1, float *d_b1, float *d_w2, float *d_b2, float *d_w3, float *d_b3, float *d_state,
                                int32_t first_step, float lr, float *d_loss);
int32_t scopa_sdcfr_train_step(scopa_ctx *ctx, const int64_t *d_rows, int32_t n_rows, const float *d_feat, const float *d_regret, const float *d_mask,
                               int64_t capacity, float *d_w1, float *d_b1, float *d_w2, float *d_b2, float *d_w3, float *d_b3, float *d_state,
                               int32_t step, float lr, float *d_loss);
/* features / masks of arbitrary device-resident states for the player to move (DeepCFR.get_policy, :497-504) */
int32_t scopa_features_from_states(scopa_ctx *ctx, const scopa_state *d_states, int64_t n, float *d_feats, float *d_mask);
/* batched evaluation episodes (evaluate_vs_random :367-429; evaluate_agent vanilla_cfr.py:157-216): n copies of the deal's
 * root state; one ply of play: the seat d_trained_seat[i] samples from d_probs[i][16] (uniform if it has no positive mass),
 * the other seat plays uniformly at random; finished episodes are left untouched.  Philox stream (seed, stream_id, ply_tag, i). */
int32_t scopa_eval_init_states(scopa_ctx *ctx, scopa_state *d_states, int64_t n);
int32_t scopa_eval_step(scopa_ctx *ctx, scopa_state *d_states, int64_t n, const float *d_probs, const int32_t *d_trained_seat,
                        uint32_t stream_id, uint32_t ply_tag);

/* one ply of n lockstep evaluation episodes of a TABULAR policy d_policy[n_infosets][4] (float64, hand order) vs uniform
 * random: evaluate_agent (vanilla_cfr.py:157-216).  d_node_idx[n] tracks each episode's tree node (start at 0). */
/* the sampling thresholds of a tabular policy ([n_infosets][4] float64, device), computed once per evaluation: scopa_eval_tabular_step with d_policy = NULL then
 * samples the trained seat by integer compares against them -- the same actions, bit for bit, as the float64 divisions of np.random.choice it performs when
 * given the policy itself.  Invalidated by scopa_set_deal. */
int32_t scopa_eval_tabular_prepare(scopa_ctx *ctx, const double *d_policy);
int32_t scopa_eval_tabular_step(scopa_ctx *ctx, scopa_state *d_states, int32_t *d_node_idx, int64_t n, int32_t ply,
                                const double *d_policy, const int32_t *d_trained_seat, uint32_t stream_id);
/* the whole match of the PREPARED tabular policy vs uniform random in one launch (evaluate_agent's loop, vanilla_cfr.py:173-216): episodes i < n_seat0 have the
 * policy in seat 0, the rest in seat 1; the same Philox draws and thresholds as eight scopa_eval_tabular_step calls, hence the same episodes bit for bit, walked
 * as node indices of the deal's tree with no per-ply state traffic.  h_stats[seat][5] = episodes, sum of the trained side's rewards x2, sum of their squares,
 * sum of its scopas, sum of the opponent's (integers: every reward is a multiple of 0.5).  d_states_out[n] / d_node_idx_out[n]: the final states / terminal
 * indices as the per-ply form leaves them, or NULL.  Synchronous. */
int32_t scopa_eval_tabular_match(scopa_ctx *ctx, int64_t n, int64_t n_seat0, uint32_t stream_id, scopa_state *d_states_out, int32_t *d_node_idx_out,
                                 int64_t h_stats[10]);

/* ---- policy value / exploitability (build-defined; the reference only calls OpenSpiel's, vanilla_cfr.py:112-118) --
 * h_policy[n_infosets][4] or NULL = the average policy of the strategy table (InfoNode.policy, vanilla_cfr.py:32-39).
 * h_out4 = {exploitability = (BR0+BR1)/2, BR0, BR1, value of the policy for player 0}; h_policy_out (optional)
 * receives the policy that was evaluated. */
int32_t scopa_exploitability(scopa_ctx *ctx, const double *h_policy, double *h_out4, double *h_policy_out);

/* ---- many deals at once ("replicas": one workgroup per independent solve) -------------------------------------------------
 * The reference solves one deal (seed 42) but its env takes a seed (MiniScopaEnv(seed=...), mini_scopa_game.py:120-132).
 * A scopa_multi keeps n deals resident in HBM and runs the per-deal kernels with one workgroup per deal.
 *   deal_py_seeds : MiniDeck(seed) for every deal ON DEVICE (CPython seed + shuffle, one lane per deal)
 *   build         : game trees + zeroed tables; h_n_infosets[n] (optional) receives the infoset counts
 *   cfr_exact / cfr_sync_iterate, exploitability (h_out4[n][4]) : as the single-deal entry points, per deal
 *   tables_get    : [n_infosets(deal)][4] tables and keys of one deal */
typedef struct scopa_multi scopa_multi;
int32_t scopa_multi_create(scopa_ctx *ctx, int32_t n_deals, scopa_multi **out);
int32_t scopa_multi_destroy(scopa_multi *m);
int32_t scopa_multi_deal_py_seeds(scopa_multi *m, const int64_t *h_seeds);
int32_t scopa_multi_set_perms(scopa_multi *m, const uint8_t *h_perms /*[n][16]*/);
int32_t scopa_multi_perms_get(scopa_multi *m, uint8_t *h_perms);
int32_t scopa_multi_build(scopa_multi *m, int32_t *h_n_infosets);
/* one workgroup per deal; from 8192 deals on it takes the lane form below (same bits) when that form's precondition holds */
int32_t scopa_multi_cfr_exact_iterate(scopa_multi *m, int32_t n_iters);
/* the same exact solve with one LANE per deal (64 deals share an instruction stream: the tree shape is deal-independent),
 * tables gathered from HBM: the throughput form for thousands of deals; bit-identical results */
int32_t scopa_multi_cfr_exact_iterate_lanes(scopa_multi *m, int32_t n_iters);
int32_t scopa_multi_cfr_sync_iterate(scopa_multi *m, int32_t n_iters);
/* batched MCCFR on every deal at once, persistent: one workgroup per deal keeps the deal's regret table in LDS and runs all
 * n_iters iterations of `batch` traversal pairs without leaving the kernel; same definition (frozen tables per iteration,
 * Philox keyed by seed / traversal id / iteration) as scopa_mccfr_iterate on that deal */
int32_t scopa_multi_mccfr_iterate(scopa_multi *m, uint32_t batch, uint32_t n_iters, uint64_t seed);
int32_t scopa_multi_exploitability(scopa_multi *m, double *h_out4);
int32_t scopa_multi_tables_get(scopa_multi *m, int32_t deal, double *h_regret, double *h_strategy, double *h_local, uint64_t *h_keys);
int32_t scopa_multi_counters(scopa_multi *m, uint64_t *decision_visits, uint64_t *terminal_visits);

/* ---- FullScopa: the 40-card game (src/envs/full_scopa_game.py, src/envs/openspiel_full_scopa.py) -- state engine -------------
 * No reference solver uses it; provided: deal, the state protocol, and the batched device step.
 * Card id = action id = suit_idx*10 + rank-1 (denari, coppe, spade, bastoni).  A state refers to its deck (the deal order,
 * needed for the re-deals) by index `game` into a caller-supplied array of 40-byte decks. */
typedef struct scopa_full_state {   /* 64 bytes */
    uint64_t table[2];     /* ordered table, 20 six-bit slots (10 per word)                                   */
    uint64_t cap[2];       /* captured cards per player, 40-bit masks                                          */
    uint32_t hand[2];      /* ordered hands, 3 six-bit slots                                                   */
    uint32_t game;         /* deck index                                                                       */
    uint8_t  nh[2], nt, deck_pos, round, last_capture /* 0xFF = none */, scopas[2];
    uint16_t step;
    uint8_t  terminal;
    int8_t   r2_p0;        /* rewards x2 of player 0 once terminal (player 1 = negation)                       */
    uint8_t  flags;        /* bit 0: table capacity (20) exceeded                                              */
    uint8_t  pad[7];
} scopa_full_state;
int32_t scopa_full_deal_py_seed(int64_t seed, uint8_t perm40[40]);                              /* FullDeck.__init__ :29-32 */
int32_t scopa_full_state_init(const uint8_t deck40[40], uint32_t game, scopa_full_state *out);  /* FullScopaGame.reset :60-75 */
int32_t scopa_full_state_step(scopa_full_state *s, const uint8_t deck40[40], int32_t action);   /* FullScopaEnv.step :252-297 */
int32_t scopa_full_state_legal(const scopa_full_state *s, int32_t player, int32_t out[3], int32_t *n);
int32_t scopa_full_state_infoset_string(const scopa_full_state *s, int32_t player, char *buf, int32_t cap);
/* d_states[i] <- step(d_states[i], d_decks[d_states[i].game], d_actions[i]) */
int32_t scopa_full_step_batch(scopa_ctx *ctx, scopa_full_state *d_states, const uint8_t *d_actions, const uint8_t *d_decks, int64_t n);
int32_t scopa_full_step_batch_host(scopa_ctx *ctx, scopa_full_state *h_states, const uint8_t *h_actions, const uint8_t *h_decks,
                                   int64_t n_decks, int64_t n);
/* n_games uniform-random playouts to the end, one lane per game, decks dealt ON DEVICE from seeds[i] (CPython shuffle);
 * h_r2_p0[n] = rewards x2 of player 0, h_plies[n] = game length.  Philox stream (ctx seed, game, ply). */
int32_t scopa_full_random_playouts(scopa_ctx *ctx, const int64_t *h_seeds, int64_t n_games, int8_t *h_r2_p0, int16_t *h_plies);

/* ---- Team MiniScopa TPI: 2 teams x 2 seats on the 16-card deck, 16 plies (src/envs/team_mini_scopa_game.py,
 * src/envs/openspiel_team_mini_scopa.py) -- state engine ------------------------------------------------------------------
 * No reference solver uses it; provided: the state protocol, the batched device step and device playouts.
 * The "player" of the TPI game is the TEAM (coordinator) of the seat to move; seats 0,1 = team 0, seats 2,3 = team 1. */
typedef struct scopa_team_state {   /* 40 bytes */
    uint64_t history;      /* nibble i = action of ply i (TPIMiniScopaState.action_history)                    */
    uint32_t table;        /* ordered table, nibble list (at most 8: table ranks are always distinct)          */
    uint16_t hand[4];      /* ordered hands                                                                    */
    uint16_t cap[4];       /* captured cards per seat, 16-bit masks                                            */
    uint8_t  nh[4];
    uint8_t  scopas[4];
    uint8_t  nt, step, last_capture_team /* 0xFF = None */, flags /* bit 0 terminal, bit 1 table overflow */;
} scopa_team_state;
int32_t scopa_team_state_init(const uint8_t perm16[16], scopa_team_state *out);               /* TeamMiniScopaGame.reset :68-78 */
int32_t scopa_team_state_step(scopa_team_state *s, int32_t action);                           /* apply_action / env.step :173-201 */
int32_t scopa_team_state_legal(const scopa_team_state *s, int32_t out[4], int32_t *n);        /* legal_actions, openspiel_team…:52-86 */
int32_t scopa_team_state_rewards_x2(const scopa_team_state *s, int32_t r2_seat[4]);           /* evaluate_game :125-155, x2 */
int32_t scopa_team_state_infoset_string(const scopa_team_state *s, int32_t team, char *buf, int32_t cap); /* :119-146 */
/* d_states[i] <- step(d_states[i], d_actions[i]) */
int32_t scopa_team_step_batch(scopa_ctx *ctx, scopa_team_state *d_states, const uint8_t *d_actions, int64_t n);
int32_t scopa_team_step_batch_host(scopa_ctx *ctx, scopa_team_state *h_states, const uint8_t *h_actions, int64_t n);
/* n_games uniform-random playouts to the end, one lane per game, dealt ON DEVICE from seeds[i] (CPython shuffle);
 * h_r2_team0[n] = reward x2 of team 0 (team 1 = negation), h_scopas[n][4] per seat.  Philox stream (ctx seed, game, ply). */
int32_t scopa_team_random_playouts(scopa_ctx *ctx, const int64_t *h_seeds, int64_t n_games, int8_t *h_r2_team0, uint8_t *h_scopas);

/* ---- N > 1: one-shot all-reduce of the delta buffer over peer (xGMI) memory -------------------------------------------------
 * One process per GPU on one node.  create: allocates this rank's inbox (fine-grained device memory) and returns its 64-byte
 * hipIpc handle; the caller all-gathers the handles (torch.distributed) and passes all `world` of them to connect.
 * allreduce_delta (on the context's stream): the delta buffer of every rank becomes the sum over ranks, added in rank order on
 * every rank (bit-identical replicas).  Waits are bounded (5 s, scopa_p2p_set_budget): a wait that gives up is counted, makes
 * every later wait fall through (a dead peer cannot hang a GPU) and makes allreduce_delta / iterate_sharded -- which synchronise
 * the stream before they return -- fail with SCOPA_ETIMEOUT, on that call and on every later one until the exchange is
 * re-created; status returns the count (0 = healthy) and the exchanges issued.  Replaces torch.distributed.all_reduce(delta)
 * between scopa_mccfr_traverse and scopa_mccfr_apply.
 * set_form: 0 (default) = plain stores + system-scope release fence / acquire fence (the textbook protocol); 1 = every access to an
 * inbox line is a system-scope (sc0 sc1) access ordered by s_waitcnt alone: 4.5 us less per iteration, no cache maintenance; use
 * it only after it has been validated against a collective on the topology at hand (scopa_amd/distributed.py does that). */
int32_t scopa_p2p_create(scopa_ctx *ctx, int32_t rank, int32_t world, uint8_t handle_out[64]);
int32_t scopa_p2p_connect(scopa_ctx *ctx, const uint8_t *handles /*[world][64]*/);
int32_t scopa_p2p_allreduce_delta(scopa_ctx *ctx);
int32_t scopa_p2p_set_form(scopa_ctx *ctx, int32_t light);
int32_t scopa_p2p_set_budget(scopa_ctx *ctx, double seconds);   /* wait budget of later exchanges, 1 ms .. 60 s */
/* n_iters whole iterations of this rank's slice [b0, b0+nb) of the global traversal ids, the exchange fused into the
 * reduce+apply kernel (two launches per iteration, as on one GPU); all ranks call it with the same n_iters */
int32_t scopa_mccfr_iterate_sharded(scopa_ctx *ctx, uint32_t b0, uint32_t nb, uint32_t n_iters);
int32_t scopa_p2p_status(scopa_ctx *ctx, int32_t *timeouts, uint64_t *exchanges);
int32_t scopa_p2p_destroy(scopa_ctx *ctx);

/* ---- counters / profiling -------------------------------------------------------------------------------
 * exact integer counts of decision-node visits ("infoset-traversals") and terminal visits since creation */
int32_t scopa_counters(scopa_ctx *ctx, uint64_t *decision_visits, uint64_t *terminal_visits);
/* stride > 0: every stride-th launch of the dominant traversal kernel carries a (start, stop) HIP event pair attached to
 * the dispatch itself (hipExtLaunchKernelGGL) on the context's stream (stride 1 = every launch); 0 = off.  scopa_prof_read
 * synchronises and returns the number of sampled launches and their summed kernel milliseconds since enable. */
int32_t scopa_prof_enable(scopa_ctx *ctx, int32_t stride);
int32_t scopa_prof_read(scopa_ctx *ctx, int64_t *launches, double *kernel_ms);
/* the same kernel timed by ITSELF: per sampled launch (the ones scopa_prof_enable's stride selects), first workgroup start ->
 * last workgroup end on the 100 MHz device-wide clock, summed over the samples held (the last 2048 at most) since scopa_prof_enable;
 * no events, no dispatch latency: launch ramp and end-of-kernel write-back are outside it */
int32_t scopa_prof_device(scopa_ctx *ctx, int64_t *launches, double *kernel_ms);
/* after scopa_prof_device: mean microseconds a workgroup of the sampled launches spent in (prologue, traversal walks, epilogue) */
int32_t scopa_prof_phases(scopa_ctx *ctx, double out_us[3]);
/* beside the phases: { mean start of a workgroup behind the first workgroup of its launch, the LAST workgroup's start behind the first, the longest
 * workgroup of a launch } in microseconds, means over the sampled launches -- what a launch's time is made of beside its workgroups' own phases */
int32_t scopa_prof_spread(scopa_ctx *ctx, double out_us[3]);

#ifdef __cplusplus
}
#endif
#endif /* SCOPA_H */
