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
  vanilla_cfr.npz     CFRTrainer tables after 1,2,5,50,200 iterations             (vanilla_cfr.py:56-120)
  mccfr.npz           MCCFRTrainer tables under np.random.seed(k)                 (mc_cfr.py:37-99)
  evaluate.json       evaluate_agent results under np.random.seed(k)              (vanilla_cfr.py:157-216, mc_cfr.py:146-206)
  sdcfr.npz           DeepCFR features/masks/traversal rows with saved weights    (deep_cfr.py:213-365)
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
            cases.append(dict(seed=seed, actions=acts, init=init, trail=trail, player_rewards=[float(st.env.rewards[f"player_{i}"]) for i in range(4)]))
    # default construction path of the reference: TPIMiniScopaState(game) -> TeamMiniScopaEnv() -> reset() -> seed 42
    st = game.new_initial_state()
    default = snap(st)
    with open(os.path.join(OUT, "team_mini_scopa.json"), "w") as f:
        json.dump(dict(playouts=cases, default_initial=default), f)
    print("team mini scopa: playouts", len(cases), "plies", sorted({len(c["actions"]) for c in cases}),
          "max table", max(len(t["table"]) for c in cases for t in c["trail"]))


ALL = dict(team=gen_team, full=gen_full, deals=gen_deals, tree=gen_tree, playouts=gen_playouts, cfr=gen_cfr, mccfr=gen_mccfr,
           evaluate=gen_evaluate, sdcfr=gen_sdcfr)

if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        sys.exit("reference not present (GPU box?): fixtures are generated in the build container only")
    os.makedirs(OUT, exist_ok=True)
    ns = refshim.import_reference()
    for name in (sys.argv[1:] or list(ALL)):
        ALL[name](ns)
