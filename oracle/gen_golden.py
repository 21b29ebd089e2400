#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE (container-only).  Imports the reference's own Python from
/root/reference (read-only) under the base-class stand-ins of `oracle/refshim.py` and
dumps inputs + expected outputs as data (JSON / NPZ).  Nothing of the reference's source
text is written out; the reference never travels to the GPU box -- only these vectors do.

    python oracle/gen_golden.py            # all fixtures (~3 min on one core)
    python oracle/gen_golden.py tree cfr   # a subset

Fixtures (consumers: tests/, oracle pinning):
  deals.json          seed -> 16-card permutation of random.seed(seed)+shuffle   (mini_scopa_game.py:25-28)
  tree_seed<S>.npz    every node of the game tree in reference DFS order          (openspiel_mini_scopa.py:17-115)
  playouts.json       random action strings incl. ILLEGAL actions (silent no-op)  (mini_scopa_game.py:140-167)
  playouts_cloned.json the same through MiniScopaState.clone(): a clone plays to step 16  (openspiel_mini_scopa.py:97-115, line 108)
  vanilla_cfr.npz     CFRTrainer tables after 1,2,5,50,200 iterations             (vanilla_cfr.py:56-120)
  mccfr.npz           MCCFRTrainer tables under np.random.seed(k)                 (mc_cfr.py:37-99)
  MiniScopa_MCCFR_data.reference.json  the reference's committed 10-run MCCFR experiment output (experiment_tracker.py:82-158)
  vanilla_cfr_experiment.json  the reference's vanilla-CFR experiment runner, seeded and shortened (run_vanilla_cfr_experiment.py:59-131)
  mccfr_experiment_runs.json  24 seeded runs of the reference's published experiment (run_mccfr_experiment.py:64-137) + exact EV of each final policy
  mccfr_frozen.npz    MCCFRTrainer._sample driven with frozen strategies and path-keyed draws: batched-MCCFR deltas (mc_cfr.py:37-86)
  evaluate.json       evaluate_agent results under np.random.seed(k)              (vanilla_cfr.py:157-216, mc_cfr.py:146-206)
  sdcfr.npz           DeepCFR features/masks/traversal rows with saved weights    (deep_cfr.py:213-365)
  exploitability.json an independent best response over the reference's own state / policy objects for three reference-made policies
                      (cross-check of og_exploitability / k_exploitability; not a pin: the reference publishes no value)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
import refshim  # noqa: E402

SUITS = ["cuori", "fiori", "picche", "bello"]
RANKS = {"cuori": [2, 5, 8, 10], "fiori": [2, 5, 7, 9], "picche": [3, 6, 8, 9], "bello": [3, 6, 7, 10]}


def card_id(rank, suit):
    s = SUITS.index(suit)
    return s * 4 + RANKS[suit].index(rank)


def cid(card):
    return card_id(card.rank, card.suit)


# ----------------------------------------------------------------------------------
def gen_deals(ns):
    seeds = [42] + list(range(0, 64)) + [123, 2024, 99991, 2**31 - 1, 2**32 + 5, 10**12 + 7]
    out = {}
    for s in seeds:
        deck = ns.game.MiniDeck(s)
        out[str(s)] = [cid(c) for c in deck.cards]
    with open(os.path.join(OUT, "deals.json"), "w") as f:
        json.dump(out, f)
    print("deals:", len(out))


# ----------------------------------------------------------------------------------
def _new_state(ns, game, seed):
    """A MiniScopaState on deal `seed` (the reference always uses 42; other seeds go through
    MiniScopaEnv(seed=...), mini_scopa_game.py:120-132)."""
    if seed == 42:
        return game.new_initial_state()
    env = ns.game.MiniScopaEnv(seed=seed)
    return ns.spiel.MiniScopaState(game, env=env, skip_reset=True)


def _snap(st):
    g = st.env.game
    hands = [[cid(c) for c in p.hand] for p in g.players]
    table = [cid(c) for c in g.table]
    return hands, table, [len(p.captures) for p in g.players], [p.scopas for p in g.players], st.env.step_count


def gen_tree(ns, seeds=(42, 0, 1, 7, 123)):
    import pyspiel
    game = pyspiel.load_game("mini_scopa")
    for seed in seeds:
        rows = []
        infoset_ids = {}
        infoset_list = []
        hist_strs = []

        def rec(st, depth, parent, action):
            idx = len(rows)
            hands, table, ncap, scopas, step = _snap(st)
            term = st.is_terminal()
            cur = st.current_player()
            legal = st.legal_actions()
            istr = st.information_state_string() if not term else "TERMINAL"
            if not term:
                if istr not in infoset_ids:
                    infoset_ids[istr] = len(infoset_list)
                    infoset_list.append(istr)
                iid = infoset_ids[istr]
            else:
                iid = -1
            rw = st.rewards()
            rows.append(dict(depth=depth, parent=parent, action=action, term=int(term), player=cur if not term else -4,
                             hands=hands, table=table, ncap=ncap, scopas=scopas, step=step, legal=legal,
                             infoset=iid, r2=[int(round(2 * float(r))) for r in rw]))
            assert all(abs(2 * float(r) - round(2 * float(r))) < 1e-12 for r in rw)
            hist_strs.append(st.history_str())
            for a in legal:
                ch = st.clone()
                ch.apply_action(a)
                rec(ch, depth + 1, idx, a)

        rec(_new_state(ns, game, seed), 0, -1, -1)
        n = len(rows)
        arr = dict(
            depth=np.array([r["depth"] for r in rows], np.int8),
            parent=np.array([r["parent"] for r in rows], np.int32),
            action=np.array([r["action"] for r in rows], np.int8),
            term=np.array([r["term"] for r in rows], np.int8),
            player=np.array([r["player"] for r in rows], np.int8),
            step=np.array([r["step"] for r in rows], np.int8),
            infoset=np.array([r["infoset"] for r in rows], np.int32),
            ncap=np.array([r["ncap"] for r in rows], np.int8),
            scopas=np.array([r["scopas"] for r in rows], np.int8),
            r2=np.array([r["r2"] for r in rows], np.int8),
        )
        hands = -np.ones((n, 2, 4), np.int8)
        table = -np.ones((n, 8), np.int8)
        legal = -np.ones((n, 4), np.int8)
        nh = np.zeros((n, 2), np.int8)
        nt = np.zeros(n, np.int8)
        nl = np.zeros(n, np.int8)
        for i, r in enumerate(rows):
            for p in range(2):
                nh[i, p] = len(r["hands"][p])
                hands[i, p, :nh[i, p]] = r["hands"][p]
            nt[i] = len(r["table"])
            table[i, :nt[i]] = r["table"]
            nl[i] = len(r["legal"])
            legal[i, :nl[i]] = r["legal"]
        arr.update(hands=hands, table=table, legal=legal, nh=nh, nt=nt, nl=nl)
        arr["infoset_strings"] = np.array(infoset_list)
        arr["history_strings"] = np.array(hist_strs)
        np.savez_compressed(os.path.join(OUT, f"tree_seed{seed}.npz"), **arr)
        print(f"tree seed {seed}: nodes={n} decision={int((arr['term']==0).sum())} infosets={len(infoset_list)}")


# ----------------------------------------------------------------------------------
def gen_playouts(ns):
    """Random action strings over 0..15 (mostly illegal -> silent no-op) and legal-biased ones."""
    import pyspiel
    import random as pyrandom
    game = pyspiel.load_game("mini_scopa")
    rng = np.random.RandomState(20251205)
    cases = []
    for seed in [42, 0, 1, 2, 3, 5, 7, 11, 123, 2024]:
        for k in range(16):
            st = _new_state(ns, game, seed)
            acts, trail = [], []
            while not st.is_terminal():
                legal = st.legal_actions()
                if k % 2 == 0 or rng.rand() < 0.6:
                    a = int(legal[rng.randint(len(legal))])
                else:
                    a = int(rng.randint(16))
                acts.append(a)
                st.apply_action(a)
                hands, table, ncap, scopas, step = _snap(st)
                trail.append(dict(hands=hands, table=table, ncap=ncap, scopas=scopas, step=step,
                                  term=bool(st.is_terminal()),
                                  cur=int(st.current_player()) if not st.is_terminal() else -4,
                                  info0=st.information_state_string(0), info1=st.information_state_string(1),
                                  hist=st.history_str()))
            cases.append(dict(seed=seed, actions=acts, trail=trail, rewards=[float(r) for r in st.rewards()]))
    with open(os.path.join(OUT, "playouts.json"), "w") as f:
        json.dump(cases, f)
    print("playouts:", len(cases))


# ----------------------------------------------------------------------------------
def gen_playouts_cloned(ns):
    """Playouts that go through MiniScopaState.clone() (openspiel_mini_scopa.py:97-115).  A clone's env has max_steps = 16 (:108)
    where a fresh env has num_players * 4 = 8 (mini_scopa_game.py:127), so a cloned state that absorbs illegal no-op actions plays on
    past step 8 until both hands are empty or step 16 (terminal rule mini_scopa_game.py:160).  Every ply records what playouts.json
    records plus both players' legal actions, the rewards, and whether the live state is a clone (`cloned`); `clone_before[i]` says
    that the state was replaced by its clone() just before action i.  Cases with k % 4 == 0 play legal actions only (clone or not,
    the trail is the un-cloned one); the others draw 35 % of their actions from 0..15."""
    import pyspiel
    game = pyspiel.load_game("mini_scopa")
    rng = np.random.RandomState(20261005)
    cases = []
    for seed in [42, 0, 1, 2, 3, 5, 7, 11, 13, 123, 2024, 99991]:
        for k in range(20):
            st = _new_state(ns, game, seed)
            acts, trail, clone_before = [], [], []
            cloned = False
            p_clone = (0.35, 0.15, 0.35, 1.0)[k % 4]
            while not st.is_terminal():
                do_clone = bool(rng.rand() < p_clone)
                if do_clone:
                    st = st.clone()
                    cloned = True
                legal = st.legal_actions()
                if k % 4 == 0 or rng.rand() < 0.65:
                    a = int(legal[rng.randint(len(legal))])
                else:
                    a = int(rng.randint(16))
                acts.append(a)
                clone_before.append(do_clone)
                st.apply_action(a)
                hands, table, ncap, scopas, step = _snap(st)
                term = bool(st.is_terminal())
                trail.append(dict(hands=hands, table=table, ncap=ncap, scopas=scopas, step=step, term=term,
                                  cur=int(st.current_player()) if not term else -4,
                                  legal0=[int(x) for x in st.legal_actions(0)], legal1=[int(x) for x in st.legal_actions(1)],
                                  info0=st.information_state_string(0), info1=st.information_state_string(1),
                                  rewards=[float(r) for r in st.rewards()], hist=st.history_str(), cloned=cloned,
                                  max_steps=int(st.env.max_steps)))
            assert len(acts) <= 16
            # a clone of a TERMINAL state stays terminal whatever its max_steps: _is_terminal and the terminations dict are copied
            # (openspiel_mini_scopa.py:112-113, mini_scopa_game.py:193), so a further action is a dead step (:141-143)
            cl = st.clone()
            dead = int(rng.randint(16))
            cl.apply_action(dead)
            hands, table, ncap, scopas, step = _snap(cl)
            after = dict(action=dead, hands=hands, table=table, ncap=ncap, scopas=scopas, step=step, term=bool(cl.is_terminal()),
                         legal0=[int(x) for x in cl.legal_actions(0)], rewards=[float(r) for r in cl.rewards()], hist=cl.history_str(),
                         info0=cl.information_state_string(0), max_steps=int(cl.env.max_steps))
            cases.append(dict(seed=seed, actions=acts, clone_before=clone_before, trail=trail, rewards=[float(r) for r in st.rewards()],
                              terminal_clone=after))
    with open(os.path.join(OUT, "playouts_cloned.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    past8 = sum(1 for c in cases if len(c["actions"]) > 8)
    print("playouts_cloned:", len(cases), "cases,", past8, "play past step 8, longest", max(len(c["actions"]) for c in cases),
          "plies; hands left at the end in", sum(1 for c in cases if any(c["trail"][-1]["hands"])), "cases")


# ----------------------------------------------------------------------------------
def _dump_vanilla(trainer):
    keys = list(trainer.info_set_map.keys())
    n = len(keys)
    R = np.zeros((n, 4)); S = np.zeros((n, 4)); L = np.zeros((n, 4)); A = -np.ones((n, 4), np.int8); NL = np.zeros(n, np.int8)
    for i, k in enumerate(keys):
        nd = trainer.info_set_map[k]
        m = nd.legal_actions.size
        NL[i] = m
        A[i, :m] = nd.legal_actions
        R[i, :m] = nd.regret_sum
        S[i, :m] = nd.strategy_sum
        L[i, :m] = nd.local_strategy
    return keys, NL, A, R, S, L


def gen_cfr(ns):
    import pyspiel
    game = pyspiel.load_game("mini_scopa")
    tr = ns.vanilla.CFRTrainer(game)
    out = {}
    root_vals = []
    checkpoints = [1, 2, 5, 50, 200]
    for t in range(1, max(checkpoints) + 1):
        vals = []
        for i in range(game.num_players()):
            vals.append(tr._cfr_recursive(game.new_initial_state(), i, 1.0, 1.0))
        root_vals.append(vals)
        if t in checkpoints:
            keys, NL, A, R, S, L = _dump_vanilla(tr)
            out[f"it{t}_regret"] = R
            out[f"it{t}_strategy"] = S
            out[f"it{t}_local"] = L
            if t == checkpoints[0]:
                out["keys"] = np.array(keys)
                out["nlegal"] = NL
                out["legal"] = A
            else:
                assert keys == list(out["keys"])
            print("cfr it", t, "root", vals)
    out["root_values"] = np.array(root_vals, np.float64)
    out["checkpoints"] = np.array(checkpoints)
    np.savez_compressed(os.path.join(OUT, "vanilla_cfr.npz"), **out)


# ----------------------------------------------------------------------------------
def _dump_mc(trainer):
    keys = list(trainer.info_sets.keys())
    n = len(keys)
    R = np.zeros((n, 4)); S = np.zeros((n, 4)); A = -np.ones((n, 4), np.int8); NL = np.zeros(n, np.int8)
    for i, k in enumerate(keys):
        nd = trainer.info_sets[k]
        m = nd.legal_actions.size
        NL[i] = m
        A[i, :m] = nd.legal_actions
        R[i, :m] = nd.regret_sum
        S[i, :m] = nd.strategy_sum
    return [f"{p}|{s}" for p, s in keys], NL, A, R, S


def gen_mccfr(ns):
    import pyspiel
    game = pyspiel.load_game("mini_scopa")
    out = {}
    for seed in (0, 1, 2):
        for iters in (1, 10, 200):
            np.random.seed(seed)
            tr = ns.mc.MCCFRTrainer(game)
            for _ in range(iters):
                tr.iteration()
            keys, NL, A, R, S = _dump_mc(tr)
            tag = f"s{seed}_it{iters}"
            out[tag + "_keys"] = np.array(keys)
            out[tag + "_nlegal"] = NL
            out[tag + "_legal"] = A
            out[tag + "_regret"] = R
            out[tag + "_strategy"] = S
            # the next uniform after the run pins how many draws were consumed (463/iteration)
            out[tag + "_next_u"] = np.array([np.random.random_sample()])
            print("mccfr", tag, "infosets", len(keys))
    np.savez_compressed(os.path.join(OUT, "mccfr.npz"), **out)


# ----------------------------------------------------------------------------------
def _philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11; Random123 constants), plain Python integers."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    M = 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M, p1 & M, ((p0 >> 32) ^ c3 ^ k1) & M, p0 & M
        k0, k1 = (k0 + 0x9E3779B9) & M, (k1 + 0xBB67AE85) & M
    return c0, c1, c2, c3


def gen_mccfr_frozen(ns):
    """Batched MCCFR = B traversals of the reference's own MCCFRTrainer._sample (mc_cfr.py:37-86) against tables FROZEN for the
    iteration, with path-keyed draws.  The reference's recursion, its regret/strategy update lines and its InfoNode.current_strategy
    run unmodified; only two things are supplied from outside, by subclassing -- nothing of the reference is edited or restated:
      * nodes: current_strategy() is answered by a plain reference InfoNode holding the FROZEN regret row, while the node object
        that _sample updates (`node.regret_sum += ...`, `node.strategy_sum += ...`) starts at zero, i.e. accumulates the DELTA;
      * np.random.choice(legal, p=sigma) is replaced for the duration of a traversal by numpy's own inverse-cdf arithmetic
        (cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(u, 'right')) on u = the build's path-keyed Philox uniform of the node at
        hand.  A node is (ntl, j): ntl traverser nodes above it, j its branch index at that level (child of a traverser node with n
        legal actions: j*(n+1) for the sampled child, j*(n+1) + i + 1 for the re-expansion of legal action i; an opponent node shares
        (ntl, j) with the traverser node below it).  block = Philox4x32-10(key = seed; ctr = ((0, 1, 4, 14)[ntl] + (j >> 1), traversal
        id, iteration, traverser)); word 2*(j & 1) at the opponent node, 2*(j & 1) + 1 at the traverser node; u = (word >> 1) * 2^-31.
        The node's (ntl, j) is read off the recursion itself."""
    import pyspiel
    mc = ns.mc
    game = pyspiel.load_game("mini_scopa")

    class DeltaNode(mc.InfoNode):
        def current_strategy(self):
            return self.frozen_src.current_strategy()

    class FrozenTrainer(mc.MCCFRTrainer):
        def __init__(self, game, frozen):
            super().__init__(game)
            self.frozen, self.stack, self.trace = frozen, [], []

        def _get_node(self, key, legal_actions):
            if key not in self.info_sets:
                nd = DeltaNode(np.array(legal_actions))
                nd.frozen_src = mc.InfoNode(np.array(legal_actions))
                if key in self.frozen:
                    nd.frozen_src.regret_sum = self.frozen[key].copy()
                self.info_sets[key] = nd
            return self.info_sets[key]

        def _sample(self, state, traversing_player, reach_probs, sampling_probs):
            term = state.is_terminal()
            n = 0 if term else len(state.legal_actions(state.current_player()))
            if self.stack:
                par = self.stack[-1]
                if par["is_trav"]:       # child of a traverser node with n legal actions: j*(n+1) + (0 = sampled child, i+1 = re-expansion of action i)
                    ntl, j = par["ntl"] + 1, par["j"] * (par["n"] + 1) + par["calls"]
                    par["calls"] += 1
                else:
                    ntl, j = par["ntl"], par["j"]
            else:
                ntl, j = 0, 0
            self.stack.append(dict(ntl=ntl, j=j, n=n, calls=0, is_trav=(not term) and state.current_player() == traversing_player))
            try:
                return super()._sample(state, traversing_player, reach_probs, sampling_probs)
            finally:
                self.stack.pop()

    def run_case(frozen, seed, iteration, b0, nb):
        tr = FrozenTrainer(game, frozen)
        cur = dict(b=0, trav=0)

        def choice(legal, p=None):
            fr = tr.stack[-1]
            block = (0, 1, 4, 14)[min(fr["ntl"], 3)] + (fr["j"] >> 1)
            o = _philox4x32_10((block, cur["b"], iteration, cur["trav"]), (seed & 0xFFFFFFFF, seed >> 32))
            u = (o[2 * (fr["j"] & 1) + (1 if fr["is_trav"] else 0)] >> 1) / 2147483648.0
            cdf = np.asarray(p, np.float64).cumsum()
            cdf /= cdf[-1]
            idx = int(cdf.searchsorted(u, side="right"))
            tr.trace.append(int(legal[idx]))
            return legal[idx]

        real = np.random.choice
        np.random.choice = choice
        try:
            for b in range(b0, b0 + nb):
                for p in range(game.num_players()):
                    cur["b"], cur["trav"] = b, p
                    tr._sample(game.new_initial_state(), p, np.ones(2), np.ones(2))
        finally:
            np.random.choice = real
        keys = list(tr.info_sets.keys())
        dR, dS = np.zeros((len(keys), 4)), np.zeros((len(keys), 4))
        for i, k in enumerate(keys):
            nd = tr.info_sets[k]
            dR[i, :nd.regret_sum.size] = nd.regret_sum
            dS[i, :nd.strategy_sum.size] = nd.strategy_sum
        return [f"{p}|{s}" for p, s in keys], dR, dS, np.array(tr.trace, np.int8)

    # frozen table A: the reference's own sequential MCCFR after 50 iterations (mixed-sign regrets, some infosets unvisited = zero rows)
    np.random.seed(3)
    ref = mc.MCCFRTrainer(game)
    for _ in range(50):
        ref.iteration()
    frozenA = {k: nd.regret_sum.copy() for k, nd in ref.info_sets.items()}
    out = {}
    fk = list(frozenA.keys())
    FR = np.zeros((len(fk), 4))
    for i, k in enumerate(fk):
        FR[i, :frozenA[k].size] = frozenA[k]
    out["frozenA_keys"] = np.array([f"{p}|{s}" for p, s in fk])
    out["frozenA_regret"] = FR
    cases = [("A", frozenA, 0x5C09A, 0, 0, 24), ("A", frozenA, 12345678901234567, 7, 5 * 32768, 12), ("Z", {}, 1, 3, 5, 12)]
    meta = []
    for n, (tab, frozen, seed, it, b0, nb) in enumerate(cases):
        keys, dR, dS, trace = run_case(frozen, seed, it, b0, nb)
        out[f"c{n}_keys"], out[f"c{n}_dregret"], out[f"c{n}_dstrategy"], out[f"c{n}_actions"] = np.array(keys), dR, dS, trace
        meta.append(dict(table=tab, seed=str(seed), iteration=it, b0=b0, nb=nb))
        assert trace.size == 463 * nb, trace.size
        print("mccfr_frozen case", n, meta[-1], "infosets touched", len(keys))
    out["cases"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "mccfr_frozen.npz"), **out)


# ----------------------------------------------------------------------------------
def gen_evaluate(ns):
    import pyspiel
    game = pyspiel.load_game("mini_scopa")
    res = {}
    # vanilla: policy after 5 iterations
    tr = ns.vanilla.CFRTrainer(game)
    tr.train(steps=5)
    pol = tr.get_openspiel_policy()
    np.random.seed(7)
    avg, hist, stats = ns.vanilla.evaluate_agent(game, pol, ns.vanilla.RandomPolicy(game), num_episodes=200)
    res["vanilla_it5_seed7_ep200"] = dict(avg_reward=float(avg), hist_head=[float(x) for x in hist[:10]], hist_tail=[float(x) for x in hist[-5:]],
                                          trained_avg=float(stats["trained_avg"]), opponent_avg=float(stats["opponent_avg"]),
                                          difference=float(stats["difference"]), data_collected=bool(stats["data_collected"]))
    # a few policy lookups
    s = game.new_initial_state()
    probs = {}
    while not s.is_terminal():
        ap = pol.action_probabilities(s)
        probs[s.history_str()] = {str(k): float(v) for k, v in ap.items()}
        s.apply_action(s.legal_actions()[0])
    res["vanilla_it5_policy_first_legal_line"] = probs
    # mccfr: seed 3, 50 iterations then evaluate (continuing the same global stream)
    np.random.seed(3)
    mt = ns.mc.MCCFRTrainer(game)
    mt.train(iterations=50)
    mpol = mt.tabular_policy()
    avg, hist, stats = ns.mc.evaluate_agent(game, mpol, ns.mc.RandomPolicy(game), num_episodes=200)
    res["mccfr_seed3_it50_ep200"] = dict(avg_reward=float(avg), hist_head=[float(x) for x in hist[:10]], hist_tail=[float(x) for x in hist[-5:]],
                                         trained_avg=float(stats["trained_avg"]), opponent_avg=float(stats["opponent_avg"]),
                                         difference=float(stats["difference"]), n_infosets=len(mt.info_sets))
    # uniform-vs-uniform expected value for P0 by brute force over the tree
    def ev(st):
        if st.is_terminal():
            return st.rewards()[0]
        la = st.legal_actions()
        tot = 0.0
        for a in la:
            c = st.clone(); c.apply_action(a)
            tot += ev(c)
        return tot / len(la)
    res["uniform_ev_p0"] = float(ev(game.new_initial_state()))
    with open(os.path.join(OUT, "evaluate.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("evaluate:", {k: (v["avg_reward"] if isinstance(v, dict) and "avg_reward" in v else "...") for k, v in res.items()})


# ----------------------------------------------------------------------------------
def gen_sdcfr(ns):
    """DeepCFR (deep_cfr.py): feature encoder, one traversal per player with saved weights, one train() call.
    The freshly initialised nets give all-zero regret-matching policies (every value 0.0), which would pin
    nothing, so the head biases are shifted by a fixed ramp before the traversals; the weights used are saved."""
    import importlib
    import random as pyrandom
    import torch
    import pyspiel
    dc = importlib.import_module("deep_cfr")  # /root/reference/src/algorithms/deep_cfr on sys.path
    game = pyspiel.load_game("mini_scopa")
    torch.manual_seed(0)
    np.random.seed(0)
    d = dc.DeepCFR(game, num_players=2, device="cpu")
    out = {"input_dim": np.array([d.input_dim])}
    for p in range(2):
        with torch.no_grad():
            ramp = torch.linspace(0.15, 1.25, 16)
            d.advantage_nets[p].net.head.bias += ramp if p == 0 else ramp.flip(0)
        sd = d.advantage_nets[p].net.state_dict()
        out[f"net{p}_names"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[f"net{p}__{k}"] = v.detach().numpy().copy()
    # features / masks along the first-legal line for both players
    s = game.new_initial_state()
    feats, masks, players, hists = [], [], [], []
    while not s.is_terminal():
        for pl in (0, 1):
            feats.append(d._state_to_features(s, pl))
            masks.append(d._get_legal_actions_mask(s, pl))
            players.append(pl)
            hists.append(s.history_str())
        s.apply_action(s.legal_actions()[0])
    out["feat_line"] = np.array(feats, np.float32)
    out["mask_line"] = np.array(masks, np.float32)
    out["feat_line_player"] = np.array(players, np.int8)
    out["feat_line_hist"] = np.array(hists)
    # one traversal per player; log every node visit and every np.random.choice call
    orig_choice = np.random.choice
    for trav in (0, 1):
        log, draws = [], []
        for p in range(2):
            net = d.advantage_nets[p]
            def wrap(f, m, _orig=net.get_advantages, _p=p):
                adv = _orig(f, m)
                log.append((_p, np.array(f, np.float32).copy(), np.array(m, np.float32).copy(), np.array(adv, np.float32).reshape(-1).copy()))
                return adv
            net.get_advantages = wrap
        def choice(a, size=None, replace=True, p=None):
            r = orig_choice(a, size=size, replace=replace, p=p)
            draws.append((0 if p is None else 1, int(r)))
            return r
        np.random.choice = choice
        n_before = [len(d.advantage_nets[p].buffer) for p in range(2)]
        np.random.seed(100 + trav)
        try:
            val = d._external_sampling_cfr(game.new_initial_state(), trav)
        finally:
            np.random.choice = orig_choice
        for p in range(2):
            del d.advantage_nets[p].get_advantages
        rows = list(d.advantage_nets[trav].buffer)[n_before[trav]:]
        assert all(k == 1 for k, _ in draws), "a uniform-fallback draw occurred: the draw stream is no longer one random_sample per opponent visit"
        out[f"trav{trav}_value"] = np.array([float(val)])
        out[f"trav{trav}_value_is_f32"] = np.array([isinstance(val, np.float32)])
        out[f"trav{trav}_visit_player"] = np.array([l[0] for l in log], np.int8)
        out[f"trav{trav}_visit_feat"] = np.array([l[1] for l in log], np.float32)
        out[f"trav{trav}_visit_mask"] = np.array([l[2] for l in log], np.float32)
        out[f"trav{trav}_visit_adv"] = np.array([l[3] for l in log], np.float32)
        out[f"trav{trav}_draw_action"] = np.array([a for _, a in draws], np.int8)
        out[f"trav{trav}_row_feat"] = np.array([r[0] for r in rows], np.float32)
        out[f"trav{trav}_row_regret"] = np.array([r[1] for r in rows], np.float32)
        out[f"trav{trav}_row_mask"] = np.array([r[2] for r in rows], np.float32)
        print(f"sdcfr trav {trav}: visits={len(log)} rows={len(rows)} draws={len(draws)} value={float(val)!r} type={type(val).__name__}")
    # one train() call pins the loss pipeline.  In the reference the global `random` state at this point is
    # "seed(42) + shuffle of the 16-card deck" (every clone() re-seeds, mini_scopa_game.py:25-28); restore exactly that.
    pyrandom.seed(42)
    pyrandom.shuffle(list(range(16)))
    loss = d.advantage_nets[0].train(epochs=2)
    out["train_loss_p0_epochs2"] = np.array([loss])
    out["train_buffer_len"] = np.array([len(d.advantage_nets[0].buffer)])
    for k, v in d.advantage_nets[0].net.state_dict().items():
        out[f"net0_after__{k}"] = v.detach().numpy().copy()
    print("sdcfr train loss", loss, "buffer", len(d.advantage_nets[0].buffer))
    # get_policy with one strategy snapshot per player (weights = the saved nets, iteration weight 2)
    for p in range(2):
        snap = dc.FlexibleNet(mode="mlp", input_shape=(d.input_dim,), output_dim=16, mlp_hidden=dc.HIDDEN, mlp_act="relu", mlp_norm="none")
        snap.load_state_dict({k: torch.from_numpy(out[f"net{p}__{k}"]) for k in out[f"net{p}_names"]})
        d.strategy_buffers[p].add_strategy(snap, 1)
    s = game.new_initial_state()
    pols = []
    while not s.is_terminal():
        pols.append(d.get_policy(s, s.current_player()))
        s.apply_action(s.legal_actions()[0])
    out["policy_line"] = np.array(pols, np.float32)
    np.savez_compressed(os.path.join(OUT, "sdcfr.npz"), **out)


def gen_full(ns):
    """FullScopa (40 cards): deals and random playouts through the reference's FullScopaEnv / FullScopaState
    (src/envs/full_scopa_game.py, src/envs/openspiel_full_scopa.py).  Card id = action id = suit_idx*10 + rank-1."""
    import importlib
    fg = importlib.import_module("envs.full_scopa_game")
    fs = importlib.import_module("envs.openspiel_full_scopa")
    import pyspiel
    game = pyspiel.load_game("full_scopa")
    FS = fg.FullDeck.suits

    def fid(c):
        return FS.index(c.suit) * 10 + (c.rank - 1)

    deals = {str(s): [fid(c) for c in fg.FullDeck(s).cards] for s in [42, 0, 1, 2, 3, 7, 123, 2024, 2**32 + 5]}
    rng = np.random.RandomState(40)
    cases = []
    for seed in [42, 0, 1, 2, 3, 7, 123, 2024]:
        for k in range(6):
            env = fg.FullScopaEnv(seed=seed)
            st = fs.FullScopaState(game, env=env, skip_reset=True)
            acts, trail = [], []
            while not st.is_terminal():
                legal = st.legal_actions()
                a = int(legal[rng.randint(len(legal))]) if (k < 4 or rng.rand() < 0.7) else int(rng.randint(40))
                acts.append(a)
                st.apply_action(a)
                g = st.env.game
                trail.append(dict(hands=[[fid(c) for c in p.hand] for p in g.players], table=[fid(c) for c in g.table],
                                  caps=[sorted(fid(c) for c in p.captures) for p in g.players], scopas=[p.scopas for p in g.players],
                                  round=g.round_number, step=st.env.step_count, deck_remaining=g.deck.cards_remaining(),
                                  last=(g.players.index(g.last_capture) if g.last_capture else -1), term=bool(st.is_terminal()),
                                  cur=int(st.current_player()) if not st.is_terminal() else -4,
                                  legal=[int(x) for x in st.legal_actions()],
                                  info0=st.information_state_string(0), info1=st.information_state_string(1)))
            cases.append(dict(seed=seed, actions=acts, trail=trail[::3] + trail[-2:], trail_idx=list(range(0, len(trail), 3)) + [len(trail) - 2, len(trail) - 1],
                              rewards=[float(r) for r in st.rewards()], hist=st.history_str()))
    with open(os.path.join(OUT, "full_scopa.json"), "w") as f:
        json.dump(dict(deals=deals, playouts=cases), f)
    print("full scopa: deals", len(deals), "playouts", len(cases), "plies", sorted({len(c["actions"]) for c in cases}),
          "max table", max(len(t["table"]) for c in cases for t in c["trail"]))


def gen_team(ns):
    """Team MiniScopa TPI (4 seats, 16 plies): random playouts through the reference's TeamMiniScopaEnv / TPIMiniScopaState
    (src/envs/team_mini_scopa_game.py, src/envs/openspiel_team_mini_scopa.py).  Card id = action id = suit_idx*4 + rank_idx."""
    import importlib
    tg = importlib.import_module("envs.team_mini_scopa_game")
    ts = importlib.import_module("envs.openspiel_team_mini_scopa")
    import pyspiel
    game = pyspiel.load_game("team_mini_scopa_tpi")
    SU = tg.MiniDeck.suits

    def cid(c):
        return SU.index(c.suit) * 4 + tg.MiniDeck.ranks[c.suit].index(c.rank)

    def snap(st):
        g = st.env.game
        return dict(hands=[[cid(c) for c in p.hand] for p in g.players], table=[cid(c) for c in g.table],
                    caps=[sorted(cid(c) for c in p.captures) for p in g.players], scopas=[p.scopas for p in g.players],
                    last=-1 if g.last_capture_team is None else int(g.last_capture_team), step=st.env.step_count,
                    seat=st.env.agent_name_mapping[st.env.agent_selection], term=bool(st.is_terminal()),
                    cur=-4 if st.is_terminal() else int(st.current_player()), legal=[int(x) for x in st.legal_actions()],
                    info0=st.information_state_string(0), info1=st.information_state_string(1), hist=st.history_str(),
                    rewards=[float(r) for r in st.rewards()])

    rng = np.random.RandomState(16)
    cases = []
    for seed in [42, 0, 1, 2, 3, 7, 123, 2024]:
        for k in range(8):
            env = tg.TeamMiniScopaEnv(seed=seed) if seed else tg.TeamMiniScopaEnv(seed=0)
            if seed == 0:
                env.game.reset(0)      # `seed or self.seed` (team_mini_scopa_game.py:160) turns an explicit 0 into the default
            st = ts.TPIMiniScopaState(game, env=env, skip_reset=True)
            init = snap(st)
            acts, trail = [], []
            while not st.is_terminal():
                legal = st.legal_actions()
                a = int(legal[rng.randint(len(legal))]) if (k < 6 or rng.rand() < 0.75) else int(rng.randint(16))
                acts.append(a)
                if k == 7 and len(acts) == 5:
                    st = st.clone()                      # clone mid-game must not disturb anything
                st.apply_action(a)
                trail.append(snap(st))
            player_rewards = [float(st.env.rewards[f"player_{i}"]) for i in range(4)]
            # actions applied after the end: dead steps for the env (team_mini_scopa_game.py:174-176), yet action_history -- and with it history_str and the
            # A[...] part of the infoset strings, which a terminal state still hands out -- keeps growing (openspiel_team_mini_scopa.py:88-92)
            dead = [int(rng.randint(16)) for _ in range(2)]
            end = st.clone() if k % 2 else st
            for a in dead:
                end.apply_action(a)
            cases.append(dict(seed=seed, actions=acts, init=init, trail=trail, player_rewards=player_rewards, dead_actions=dead, after_end=snap(end)))
    # default construction path of the reference: TPIMiniScopaState(game) -> TeamMiniScopaEnv() -> reset() -> seed 42
    st = game.new_initial_state()
    default = snap(st)
    with open(os.path.join(OUT, "team_mini_scopa.json"), "w") as f:
        json.dump(dict(playouts=cases, default_initial=default), f)
    print("team mini scopa: playouts", len(cases), "plies", sorted({len(c["actions"]) for c in cases}),
          "max table", max(len(t["table"]) for c in cases for t in c["trail"]))


def _experiment_run(k):
    """One seeded run of the reference's published experiment, in a worker process."""
    import contextlib
    import importlib
    import io
    ns = refshim.import_reference()
    exp = importlib.import_module("experiments.run_mccfr_experiment")
    made = []

    class Recording(ns.mc.MCCFRTrainer):          # the reference's trainer, only remembered so that its final tables can be read
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            made.append(self)

    exp.MCCFRTrainer = Recording
    np.random.seed(7000 + k)
    with contextlib.redirect_stdout(io.StringIO()):
        m = exp.run_single_mccfr_experiment(k, iterations=500, eval_interval=5, final_eval_episodes=5000)
    tr = made[-1]
    # exact expected reward of the final average policy against uniform play, both seats (tree enumeration by the oracle)
    import oracle as O
    t = O.Tree(seed=42)
    S = np.zeros((t.n_infosets, 4))
    for (p, key), nd in tr.info_sets.items():
        S[t.infoset_strings.index(key), :nd.strategy_sum.size] = nd.strategy_sum
    P, U = t.average_policy(S), t.average_policy(np.zeros_like(S))
    is_p0 = (np.asarray(t.infoset_player) == 0)[:, None]
    ev = 0.5 * (t.policy_value(np.where(is_p0, P, U)) - t.policy_value(np.where(is_p0, U, P)))
    return dict(seed=7000 + k, final_reward=m.final_reward, final_scopa_trained=m.final_scopa_trained, final_scopa_random=m.final_scopa_random,
                num_info_sets=m.num_info_sets, exact_ev_vs_uniform=ev, eval_iterations=m.eval_iterations, eval_rewards=m.eval_rewards,
                eval_scopas_trained=m.eval_scopas_trained, eval_scopas_random=m.eval_scopas_random)


def gen_experiment(ns, n_runs=24):
    """The reference's only published experiment (run_mccfr_experiment.py:64-137: 500 MCCFR iterations, evaluate_policy_quick every 5,
    evaluate_agent with 5000 episodes at the end) re-run HERE by the reference's own code under np.random.seed(7000 + k), k < n_runs.
    Two uses: (1) run 0's evaluation curve is replayed bit for bit by the build (training AND evaluation share the global numpy
    stream); (2) the spread of the final reward over freshly seeded reference runs, beside the exact expected reward of each final
    policy, says whether the reference's committed 10-run mean (1.1545) and the build's 100-run mean (1.244) differ by more than
    the reference's own run-to-run noise."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(6) as pool:
        runs = pool.map(_experiment_run, range(n_runs))
    fr, ev = np.array([r["final_reward"] for r in runs]), np.array([r["exact_ev_vs_uniform"] for r in runs])
    summary = dict(n_runs=n_runs, final_reward_mean=float(fr.mean()), final_reward_std=float(fr.std()), final_reward_sem=float(fr.std(ddof=1) / np.sqrt(n_runs)),
                   exact_ev_mean=float(ev.mean()), exact_ev_std=float(ev.std()), exact_ev_sem=float(ev.std(ddof=1) / np.sqrt(n_runs)),
                   info_sets_min=int(min(r["num_info_sets"] for r in runs)), info_sets_max=int(max(r["num_info_sets"] for r in runs)))
    with open(os.path.join(OUT, "mccfr_experiment_runs.json"), "w") as f:
        json.dump(dict(summary=summary, runs=runs), f, separators=(",", ":"))
    print("experiment:", summary)


def gen_vanilla_experiment(ns):
    """The reference's vanilla-CFR experiment runner (run_vanilla_cfr_experiment.py:59-131: per iteration two direct _cfr_recursive
    calls, evaluate_policy_quick every 5 iterations, evaluate_agent at the end) run by the reference under np.random.seed(11), shortened
    to 30 iterations / 600 final episodes.  tests/test_gpu_boundary.py runs the same protocol on the build's classes and expects
    these numbers back exactly: training is deterministic and bit-pinned, the evaluators consume the numpy stream identically."""
    import contextlib
    import importlib
    import io
    exp = importlib.import_module("experiments.run_vanilla_cfr_experiment")
    np.random.seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        m = exp.run_vanilla_cfr_experiment(iterations=30, eval_interval=5, final_eval_episodes=600)
    out = dict(seed=11, iterations=30, eval_interval=5, final_eval_episodes=600, eval_iterations=m.eval_iterations, eval_rewards=m.eval_rewards,
               eval_scopas_trained=m.eval_scopas_trained, eval_scopas_random=m.eval_scopas_random, eval_scopa_diff=m.eval_scopa_diff,
               final_reward=m.final_reward, final_scopa_trained=m.final_scopa_trained, final_scopa_random=m.final_scopa_random,
               final_scopa_diff=m.final_scopa_diff, num_info_sets=m.num_info_sets, next_uniform=float(np.random.random_sample()))
    with open(os.path.join(OUT, "vanilla_cfr_experiment.json"), "w") as f:
        json.dump(out, f)
    print("vanilla experiment:", out["eval_rewards"], out["final_reward"], out["num_info_sets"])


def gen_tracker(ns):
    """The reference's committed experiment output (src/experiments/experiments/results/MiniScopa_MCCFR_data.json: 10 MCCFR runs x
    500 iterations, written by ExperimentTracker.save_data_for_plotting, experiment_tracker.py:82-158) as a fixture: a data file,
    re-dumped compactly.  tests/test_tracker.py feeds its `runs` to the build's tracker and expects the whole document back."""
    src = "/root/reference/src/experiments/experiments/results/MiniScopa_MCCFR_data.json"
    with open(src) as f:
        d = json.load(f)
    with open(os.path.join(OUT, "MiniScopa_MCCFR_data.reference.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))
    print("tracker: runs", d["num_runs"], "eval points", len(d["runs"][0]["eval_iterations"]))


def gen_exploitability(ns):
    """An INDEPENDENT exploitability, written here against the reference's own objects only -- states through clone / apply_action /
    information_state_string / returns (openspiel_mini_scopa.py:17-115), policies through the reference's own policy classes'
    action_probabilities (vanilla_cfr.py:122-155, mc_cfr.py:104-144) -- following the procedural definition OpenSpiel's
    exploitability(game, policy) uses (the call the reference makes at vanilla_cfr.py:112-118; OpenSpiel itself is not installed):
    best response of player i = per information-state string argmax_a sum_{h in I} cf_reach(h) * value(h.a) (first maximum in
    legal-action order), values of the other player's nodes = sum_a pi(a|h) value(h.a); exploitability = (BR_0 + BR_1) / 2 in this
    zero-sum game (NashConv / 2).  It shares no code with oracle/scopa_oracle.c or the kernels; agreement is a cross-check, not a pin:
    the reference publishes no exploitability value (parity unpinned)."""
    import pyspiel
    game = pyspiel.load_game("mini_scopa")

    def best_response_value(policy, br):
        infosets = {}                                        # information-state string -> [(state, counterfactual reach)] in DFS order

        def collect(state, reach):
            if state.is_terminal():
                return
            p = state.current_player()
            if p == br:
                infosets.setdefault(state.information_state_string(p), []).append((state, reach))
                for a in state.legal_actions():
                    c = state.clone(); c.apply_action(a)
                    collect(c, reach)
            else:
                for a, pr in policy.action_probabilities(state).items():
                    c = state.clone(); c.apply_action(a)
                    collect(c, reach * pr)

        collect(game.new_initial_state(), 1.0)
        choice, memo = {}, {}

        def value(state):
            key = state.history_str()
            if key in memo:
                return memo[key]
            if state.is_terminal():
                v = float(state.returns()[br])
            else:
                p = state.current_player()
                if p == br:
                    c = state.clone(); c.apply_action(best_action(state.information_state_string(p)))
                    v = value(c)
                else:
                    v = 0.0
                    for a, pr in policy.action_probabilities(state).items():
                        c = state.clone(); c.apply_action(a)
                        v += pr * value(c)
            memo[key] = v
            return v

        def best_action(istr):
            if istr not in choice:
                members = infosets[istr]
                best, best_q = None, None
                for a in members[0][0].legal_actions():
                    q = 0.0
                    for st, reach in members:
                        c = st.clone(); c.apply_action(a)
                        q += reach * value(c)
                    if best is None or q > best_q:
                        best, best_q = a, q
                choice[istr] = best
            return choice[istr]

        return value(game.new_initial_state())

    def on_policy_value(policy, state=None):
        state = game.new_initial_state() if state is None else state
        if state.is_terminal():
            return float(state.returns()[0])
        v = 0.0
        for a, pr in policy.action_probabilities(state).items():
            c = state.clone(); c.apply_action(a)
            v += pr * on_policy_value(policy, c)
        return v

    def table(policy):
        """{information-state string: probabilities in legal-action (hand) order} over every decision node of the tree"""
        out = {}

        def walk(state):
            if state.is_terminal():
                return
            p = state.current_player()
            pr = policy.action_probabilities(state)
            out.setdefault(state.information_state_string(p), [float(pr[a]) for a in state.legal_actions()])
            for a in state.legal_actions():
                c = state.clone(); c.apply_action(a)
                walk(c)
        walk(game.new_initial_state())
        return out

    cases = {}
    cases["uniform"] = ns.vanilla.RandomPolicy(game)
    tr = ns.vanilla.CFRTrainer(game)
    for _ in range(50):
        for i in range(game.num_players()):
            tr._cfr_recursive(game.new_initial_state(), i, 1.0, 1.0)
    cases["cfr50"] = tr.get_openspiel_policy()
    np.random.seed(0)
    mc = ns.mc.MCCFRTrainer(game)
    for _ in range(200):
        mc.iteration()
    cases["mccfr200"] = mc.tabular_policy()
    out = {}
    for name, pol in cases.items():
        b0, b1 = best_response_value(pol, 0), best_response_value(pol, 1)
        v0 = on_policy_value(pol)
        out[name] = {"policy": table(pol), "br": [b0, b1], "value_p0": v0, "exploitability": 0.5 * (b0 + b1)}
        print("exploitability", name, out[name]["exploitability"], "br", b0, b1, "value", v0, "infosets", len(out[name]["policy"]))
    with open(os.path.join(OUT, "exploitability.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


ALL = dict(exploitability=gen_exploitability, vanilla_experiment=gen_vanilla_experiment, experiment=gen_experiment, tracker=gen_tracker, mccfr_frozen=gen_mccfr_frozen, team=gen_team, full=gen_full, deals=gen_deals, tree=gen_tree, playouts=gen_playouts, playouts_cloned=gen_playouts_cloned, cfr=gen_cfr, mccfr=gen_mccfr,
           evaluate=gen_evaluate, sdcfr=gen_sdcfr)

if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        sys.exit("reference not present (GPU box?): fixtures are generated in the build container only")
    os.makedirs(OUT, exist_ok=True)
    ns = refshim.import_reference()
    for name in (sys.argv[1:] or list(ALL)):
        ALL[name](ns)
