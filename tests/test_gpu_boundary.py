"""GPU tests of the drop-in Python boundary: the reference's class names / call signatures over the HIP engine,
checked against golden vectors produced by the reference itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def game(ctx):  # `ctx` only to skip on boxes without a GPU
    from scopa_amd.envs import load_game
    return load_game("mini_scopa")


def test_cfr_trainer_reproduces_reference_tables(game, golden):
    from scopa_amd.algorithms import CFRTrainer
    g = golden.npz("vanilla_cfr.npz")
    tr = CFRTrainer(game)
    assert tr.info_set_map == {}
    assert tr.train(steps=5, eval_interval=5, compute_exploitability=False) == []
    m = tr.info_set_map
    assert list(m.keys()) == list(g["keys"]) and len(m) == 738
    for i, (k, node) in enumerate(m.items()):
        n = int(g["nlegal"][i])
        assert list(node.legal_actions) == list(g["legal"][i, :n])
        assert np.array_equal(node.regret_sum, g["it5_regret"][i, :n])
        assert np.array_equal(node.strategy_sum, g["it5_strategy"][i, :n])
        assert np.array_equal(node.local_strategy, g["it5_local"][i, :n])
    root = m["P0:H[9f-6p-5f-7f]_T[]"]
    np.testing.assert_allclose(root.policy, [0.11066738, 0.65885247, 0.07278481, 0.15769534], atol=1e-8)  # SURVEY §4 KAT


def test_cfr_recursive_direct_calls(game, golden, oracle):
    """run_vanilla_cfr_experiment.py:89-91 drives _cfr_recursive(state, p, 1.0, 1.0) itself."""
    from scopa_amd.algorithms import CFRTrainer
    g = golden.npz("vanilla_cfr.npz")
    tr = CFRTrainer(game)
    for it in range(2):
        for p in range(game.num_players()):
            v = tr._cfr_recursive(game.new_initial_state(), p, 1.0, 1.0)
            assert v == g["root_values"][it, p]
    m = tr.info_set_map
    assert np.array_equal(np.array([m[k].regret_sum[0] for k in m]), g["it2_regret"][:, 0])
    # a non-root state with non-unit reaches, against the oracle
    t = oracle.Tree(seed=42)
    R, S, L = t.tables()
    t.cfr_exact(R, S, L, 2)
    s = game.new_initial_state()
    s.apply_action(s.legal_actions()[2])
    s.apply_action(s.legal_actions()[1])
    v = tr._cfr_recursive(s, 1, 0.3, 0.7)
    assert v == t.cfr_exact_from(R, S, L, [2, 1], 1, 0.3, 0.7)
    Rg, Sg, Lg = tr._engine.ctx.tables_get()
    assert np.array_equal(Rg, R) and np.array_equal(Sg, S) and np.array_equal(Lg, L)
    # terminal state: returns the reward, touches nothing
    while not s.is_terminal():
        s.apply_action(s.legal_actions()[0])
    assert tr._cfr_recursive(s, 0, 1.0, 1.0) == s.rewards()[0]


def test_evaluate_agent_reproduces_reference_numbers(game, golden):
    from scopa_amd.algorithms.vanilla_cfr import CFRTrainer, RandomPolicy, evaluate_agent
    ref = golden.json("evaluate.json")
    tr = CFRTrainer(game)
    tr.train(steps=5)
    pol = tr.get_openspiel_policy()
    np.random.seed(7)
    avg, hist, stats = evaluate_agent(game, pol, RandomPolicy(game), num_episodes=200)
    r = ref["vanilla_it5_seed7_ep200"]
    assert avg == r["avg_reward"] and hist[:10] == r["hist_head"] and hist[-5:] == r["hist_tail"]
    assert stats["trained_avg"] == r["trained_avg"] and stats["opponent_avg"] == r["opponent_avg"]
    assert stats["difference"] == r["difference"] and stats["data_collected"] is True
    s = game.new_initial_state()
    while not s.is_terminal():
        ap = pol.action_probabilities(s)
        want = ref["vanilla_it5_policy_first_legal_line"][s.history_str()]
        assert {str(k): float(v) for k, v in ap.items()} == want
        s.apply_action(s.legal_actions()[0])
    assert pol.action_probabilities(s) == {}


def test_mccfr_trainer_reference_mode(game, golden):
    """np.random.seed(k); MCCFRTrainer(game).train(n): same tables, same dict keys in the same order, same
    consumption of the global numpy stream as the reference."""
    from scopa_amd.algorithms import MCCFRTrainer
    m = golden.npz("mccfr.npz")
    for seed, iters in ((0, 10), (2, 200)):
        tag = f"s{seed}_it{iters}"
        np.random.seed(seed)
        tr = MCCFRTrainer(game)
        if iters == 10:
            for _ in range(iters):
                tr.iteration()
        else:
            assert tr.train(iterations=iters) == []
        assert np.random.random_sample() == m[tag + "_next_u"][0]
        keys = [f"{p}|{s}" for p, s in tr.info_sets.keys()]
        assert keys == list(m[tag + "_keys"])
        for i, node in enumerate(tr.info_sets.values()):
            n = int(m[tag + "_nlegal"][i])
            assert list(node.legal_actions) == list(m[tag + "_legal"][i, :n])
            assert np.array_equal(node.regret_sum, m[tag + "_regret"][i, :n])
            assert np.array_equal(node.strategy_sum, m[tag + "_strategy"][i, :n])


def test_mccfr_then_evaluate_reproduces_reference(game, golden):
    from scopa_amd.algorithms.mc_cfr import MCCFRTrainer, RandomPolicy, evaluate_agent
    r = golden.json("evaluate.json")["mccfr_seed3_it50_ep200"]
    np.random.seed(3)
    tr = MCCFRTrainer(game)
    tr.train(iterations=50)
    avg, hist, stats = evaluate_agent(game, tr.tabular_policy(), RandomPolicy(game), num_episodes=200)
    assert len(tr.info_sets) == r["n_infosets"]
    assert avg == r["avg_reward"] and hist[:10] == r["hist_head"] and hist[-5:] == r["hist_tail"]
    assert stats["trained_avg"] == r["trained_avg"] and stats["opponent_avg"] == r["opponent_avg"]


def test_reference_experiment_replayed_bit_for_bit(game, golden):
    """The reference's published experiment (run_mccfr_experiment.py:64-137) re-run by the reference itself under np.random.seed(7000)
    (tests/golden/mccfr_experiment_runs.json, run 0): training iterations and evaluate_policy_quick share ONE global numpy stream
    there, so reproducing its evaluation curve needs the trainer (463 draws per iteration, replayed on the GPU) AND the evaluator
    (one np.random.choice per ply) to consume the stream exactly as the reference does.  First 40 iterations = 8 evaluations."""
    from scopa_amd.algorithms.mc_cfr import MCCFRTrainer, RandomPolicy
    from scopa_amd.algorithms.evaluation import head_to_head
    r = golden.json("mccfr_experiment_runs.json")["runs"][0]
    np.random.seed(r["seed"])
    tr, rnd = MCCFRTrainer(game), RandomPolicy(game)
    for t in range(40):
        tr.iteration()
        if (t + 1) % 5 == 0:
            k = (t + 1) // 5 - 1
            rew, _, st = head_to_head(game, tr.tabular_policy(), rnd, 500)
            assert r["eval_iterations"][k] == t + 1
            assert (rew, st["trained_avg"], st["opponent_avg"]) == (r["eval_rewards"][k], r["eval_scopas_trained"][k], r["eval_scopas_random"][k])


def test_reference_vanilla_cfr_experiment_protocol_reproduced(game, golden):
    """The CALLER side of the path: the reference's vanilla-CFR experiment runner (run_vanilla_cfr_experiment.py:59-131 -- direct
    _cfr_recursive calls per player, evaluate_policy_quick every 5 iterations, evaluate_agent at the end), run by the reference under
    np.random.seed(11) (tests/golden/vanilla_cfr_experiment.json).  The build's runner on the build's classes returns the same
    ExperimentMetrics, number for number, and leaves the numpy stream at the same position."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("reproduce_vanilla", os.path.join(ROOT, "benchmarks", "reproduce_vanilla_cfr_experiment.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = golden.json("vanilla_cfr_experiment.json")
    np.random.seed(r["seed"])
    m = mod.run_vanilla_cfr_experiment(game, r["iterations"], r["eval_interval"], r["final_eval_episodes"])
    for k in ("eval_iterations", "eval_rewards", "eval_scopas_trained", "eval_scopas_random", "eval_scopa_diff", "final_reward",
              "final_scopa_trained", "final_scopa_random", "final_scopa_diff", "num_info_sets"):
        assert getattr(m, k) == r[k], k
    assert float(np.random.random_sample()) == r["next_uniform"]


def test_mccfr_trainer_batched_mode(game, oracle):
    from scopa_amd.algorithms import MCCFRTrainer
    tr = MCCFRTrainer(game, batch=512, seed=77)
    tr.train(iterations=20)
    t = oracle.Tree(seed=42)
    R, S, _ = t.tables()
    t.mccfr_batched(R, S, 77, 0, 20, 512)
    e_gpu = tr.exploitability()
    e_cpu, _ = t.exploitability(t.average_policy(S))
    assert abs(e_gpu - e_cpu) < 1e-9 and e_gpu < 1.0    # uniform policy: 2.26
    assert 600 < len(tr.info_sets) <= 738
    for (p, k), node in tr.info_sets.items():
        i = t.infoset_strings.index(k)
        np.testing.assert_allclose(node.regret_sum, R[i, :node.legal_actions.size], rtol=1e-9, atol=1e-9)


def test_exploitability_matches_oracle_bit_exact(ctx, sl, oracle, golden):
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    uni = ctx.exploitability(return_policy=True)            # zero strategy table -> uniform policy
    e, br = t.exploitability(t.average_policy(np.zeros((t.n_infosets, 4))))
    assert (uni["exploitability"], uni["br0"], uni["br1"]) == (e, br[0], br[1])
    assert abs(uni["value_p0"] - golden.json("evaluate.json")["uniform_ev_p0"]) < 1e-12
    assert uni["value_p0"] == t.policy_value(uni["policy"])
    ctx.cfr_exact_iterate(40)
    R, S, L = ctx.tables_get()
    got = ctx.exploitability(return_policy=True)
    P = t.average_policy(S)
    assert np.array_equal(got["policy"], P)
    e, br = t.exploitability(P)
    assert (got["exploitability"], got["br0"], got["br1"], got["value_p0"]) == (e, br[0], br[1], t.policy_value(P))
    assert 0 <= got["exploitability"] < uni["exploitability"]
    # an explicit policy argument
    rnd = np.random.RandomState(0).rand(t.n_infosets, 4)
    for i in range(t.n_infosets):
        rnd[i, t.infoset_nlegal[i]:] = 0
        rnd[i] /= rnd[i].sum()
    assert ctx.exploitability(rnd)["exploitability"] == t.exploitability(rnd)[0]


@pytest.mark.parametrize("name", ["uniform", "cfr50", "mccfr200"])
def test_exploitability_kernel_agrees_with_the_independent_best_response(ctx, sl, golden, name):
    """k_exploitability on the policy tables of tests/golden/exploitability.json -- best responses computed over the reference's own
    state / policy objects (oracle/gen_golden.py:gen_exploitability) -- gives the same BR values, policy value and exploitability."""
    e = golden.json("exploitability.json")[name]
    ctx.set_deal(sl.deal_py_seed(42))
    strings = [sl.key_to_string(k) for k in ctx.tree_export()["infoset_key"]]
    assert set(strings) == set(e["policy"])
    P = np.zeros((len(strings), 4))
    for i, s in enumerate(strings):
        P[i, :len(e["policy"][s])] = e["policy"][s]
    got = ctx.exploitability(P)
    assert abs(got["br0"] - e["br"][0]) < 1e-12 and abs(got["br1"] - e["br"][1]) < 1e-12
    assert abs(got["exploitability"] - e["exploitability"]) < 1e-12 and abs(got["value_p0"] - e["value_p0"]) < 1e-12


def test_exploitability_curve_falls(game):
    """'exploitability vs iters' (BASELINE metric, parity unpinned): monotone-ish decrease under vanilla CFR."""
    from scopa_amd.algorithms import CFRTrainer
    tr = CFRTrainer(game)
    hist = tr.train(steps=60, eval_interval=20, compute_exploitability=True)
    assert [t for t, _ in hist] == [20, 40, 60]
    assert hist[0][1] > hist[-1][1] >= 0


def test_entry_scripts_run(ctx):
    from scopa_amd import cfr_mini_scopa, mccfr_mini_scopa
    np.random.seed(0)
    assert cfr_mini_scopa.main(steps=20, num_episodes=100, do_plot=False) > 0          # beats the random agent
    assert isinstance(mccfr_mini_scopa.main(iterations=30, num_episodes=50), float)


def test_sync_cfr_matches_oracle_bit_exact(ctx, sl, oracle):
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    R, S, _ = t.tables()
    for k in (1, 4, 25):
        ctx.cfr_sync_iterate(k)
        t.cfr_sync(R, S, k)
        Rg, Sg, _ = ctx.tables_get()
        assert np.array_equal(Rg, R) and np.array_equal(Sg, S)
    e = ctx.exploitability()["exploitability"]
    assert e == t.exploitability(t.average_policy(S))[0] and e < 0.5
    assert ctx.counters() == (1653 * 30, 576 * 30)


def test_sync_mode_trainer_converges(game):
    from scopa_amd.algorithms import CFRTrainer
    tr = CFRTrainer(game, mode="sync")
    tr.train(steps=200)
    assert len(tr.info_set_map) == 738 and tr.exploitability() < 0.05


def test_device_evaluation_agrees_with_host_evaluation(game):
    """evaluate_agent_device (lockstep episodes on the device step kernel) vs the reference-shaped host evaluator."""
    from scopa_amd.algorithms import CFRTrainer, evaluate_agent_device
    from scopa_amd.algorithms.vanilla_cfr import RandomPolicy, evaluate_agent
    tr = CFRTrainer(game)
    tr.train(steps=30)
    avg_d, st_d = evaluate_agent_device(tr, num_episodes=200000)
    np.random.seed(1)
    avg_h, _, st_h = evaluate_agent(game, tr.get_openspiel_policy(), RandomPolicy(game), num_episodes=4000)
    se = (st_d["reward_std_error"] ** 2 + (1.6 / np.sqrt(4000)) ** 2) ** 0.5
    assert abs(avg_d - avg_h) < 5 * se and avg_d > 0.5                 # the trained policy beats random
    assert abs(st_d["trained_avg"] - st_h["trained_avg"]) < 0.06
    # exact expectation available: uniform "trained" policy vs uniform -> seat-swapped mean 0
    uni = tr._engine.ctx.exploitability(policy=None, return_policy=True)["policy"] * 0
    for i, n in enumerate(tr._engine.nlegal):
        uni[i, :n] = 1.0 / n
    avg_u, _ = evaluate_agent_device(tr, num_episodes=200000, policy=uni)
    assert abs(avg_u) < 0.03


def test_tabular_evaluator_thresholds_sample_like_the_float_divisions(game):
    """scopa_eval_tabular_prepare + scopa_eval_tabular_step(policy = NULL) -- integer thresholds per infoset, computed once -- against the step given the
    policy itself (np.random.choice's cumsum / normalise / searchsorted in float64 per visit): the SAME final states, episode for episode, for a trained
    policy, the uniform one, a one-hot one and one with all-zero rows (no compare holds there: action 0)."""
    import torch
    from scopa_amd.algorithms import CFRTrainer
    tr = CFRTrainer(game)
    tr.train(steps=40)
    ctx = tr._engine.ctx
    trained = ctx.exploitability(return_policy=True)["policy"]
    nl = np.asarray(tr._engine.nlegal)
    uni = np.zeros_like(trained)
    onehot = np.zeros_like(trained)
    for i, k in enumerate(nl):
        uni[i, :k] = 1.0 / k
        onehot[i, (7 * i) % k] = 1.0
    holes = trained.copy()
    holes[::3] = 0.0
    n = 300000
    seat = (torch.arange(n, device="cuda:0") >= n // 2).to(torch.int32)
    for name, P in (("trained", trained), ("uniform", uni), ("one-hot", onehot), ("zero rows", holes)):
        pol = torch.as_tensor(np.ascontiguousarray(P, np.float64), device="cuda:0")
        finals = []
        for prepared in (False, True):
            states = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")
            idx = torch.zeros(n, dtype=torch.int32, device="cuda:0")
            ctx.eval_init_states(states.data_ptr(), n)
            torch.cuda.synchronize()
            if prepared:
                ctx.eval_tabular_prepare(pol.data_ptr())
            for ply in range(8):
                ctx.eval_tabular_step(states.data_ptr(), idx.data_ptr(), n, ply, 0 if prepared else pol.data_ptr(), seat.data_ptr(), 77)
            ctx.synchronize()
            finals.append((states.clone(), idx.clone()))
        assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1]), name


def test_one_launch_match_is_the_per_ply_evaluator_episode_for_episode(game):
    """scopa_eval_tabular_match -- a whole seat-swapped match as walks over the deal's tree nodes, statistics summed in the kernel -- against eight
    scopa_eval_tabular_step launches on packed states: the same final state and terminal index for every episode (odd and small episode counts, another
    deal, a policy with all-zero rows), integer sums equal to what the final states give, and evaluate_agent_device's two forms agreeing."""
    import torch
    from scopa_amd import _lib
    from scopa_amd.algorithms import CFRTrainer, evaluate_agent_device
    tr = CFRTrainer(game)
    tr.train(steps=40)
    ctx = tr._engine.ctx
    trained = ctx.exploitability(return_policy=True)["policy"]
    holes = trained.copy()
    holes[::3] = 0.0
    for name, P, n, sid in (("trained", trained, 300001, 77), ("zero rows", holes, 70000, 5), ("few", trained, 3, 9), ("one", trained, 1, 2)):
        first = (n + 1) // 2
        pol = torch.as_tensor(np.ascontiguousarray(P, np.float64), device="cuda:0")
        seat = (torch.arange(n, device="cuda:0") >= first).to(torch.int32)
        states = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")
        idx = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ctx.eval_init_states(states.data_ptr(), n)
        torch.cuda.synchronize()
        ctx.eval_tabular_prepare(pol.data_ptr())
        for ply in range(8):
            ctx.eval_tabular_step(states.data_ptr(), idx.data_ptr(), n, ply, 0, seat.data_ptr(), sid)
        ctx.synchronize()
        w_states = torch.full((n, 4), -1, dtype=torch.int32, device="cuda:0")
        w_idx = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
        st = ctx.eval_tabular_match(n, first, sid, w_states.data_ptr(), w_idx.data_ptr())
        assert torch.equal(w_states, states) and torch.equal(w_idx, idx), name
        assert (ctx.eval_tabular_match(n, first, sid) == st).all(), name        # without the optional outputs: the same sums
        h = states.cpu().numpy().view(_lib.STATE_DTYPE).reshape(-1)
        r = h["ncap"].astype(np.int64) + 2 * h["scopas"].astype(np.int64)
        sv = seat.cpu().numpy().astype(np.int64)
        e = np.arange(n)
        mine = r[e, sv] - r[e, 1 - sv]
        for half in (0, 1):
            k = sv == half
            want = [int(k.sum()), int(mine[k].sum()), int((mine[k] ** 2).sum()), int(h["scopas"][e, sv][k].sum()), int(h["scopas"][e, 1 - sv][k].sum())]
            assert st[half].tolist() == want, (name, half)
    a1, s1 = evaluate_agent_device(tr, 200001, stream_id=31)
    a2, s2 = evaluate_agent_device(tr, 200001, stream_id=31, per_ply=True)
    close = lambda x, y: abs(x - y) <= 1e-13 * abs(y) + 1e-15      # the sums are the same integers; torch's mean and std round differently from sum / n
    assert close(a1, a2) and close(s1["trained_avg"], s2["trained_avg"]) and close(s1["opponent_avg"], s2["opponent_avg"])
    assert abs(s1["reward_std_error"] - s2["reward_std_error"]) < 1e-11 * s2["reward_std_error"] + 1e-15
    for b1, b2 in zip(s1["by_seat"], s2["by_seat"]):
        assert b1["episodes"] == b2["episodes"] and close(b1["reward"], b2["reward"]) and close(b1["trained_scopas"], b2["trained_scopas"]) and close(b1["opponent_scopas"], b2["opponent_scopas"])
    with pytest.raises(_lib.ScopaError):
        ctx.eval_tabular_match(10, 11, 1)                                       # more seat-0 episodes than episodes
    ctx.set_deal(_lib.deal_py_seed(7))                                          # a new deal invalidates the prepared thresholds
    with pytest.raises(_lib.ScopaError):
        ctx.eval_tabular_match(10, 5, 1)


def test_device_clock_profile_of_sampled_traversal_launches(ctx, sl):
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.mccfr_seed(5)
    assert ctx.prof_device() == (0, 0.0)                # nothing sampled yet
    ctx.prof_enable(2)                                   # every second launch
    ctx.mccfr_iterate(512, 14)
    n1, ms1 = ctx.prof_device()
    assert n1 == 7 == ctx.prof_read()[0]
    ctx.prof_enable(0)
    per_launch_us = 1e3 * ms1 / 7
    assert 1.0 < per_launch_us < 500.0      # a 100 MHz clock span of a kernel that takes ~10 us


# ---- exact expectations of "policy vs uniform random" by tree enumeration (test-side, over the oracle's flat tree) -------------------
def _exact_match(tree, policy_of_node, seat):
    """The trained agent sits in `seat` and plays policy_of_node(node) -> probabilities of the node's legal actions (hand order); the
    other seat plays uniformly.  Enumerates the oracle's tree: -> (E reward, Var reward, E scopas trained, E scopas opponent)."""
    st = tree.states()
    reach = np.zeros(tree.n_nodes)
    reach[0] = 1.0
    e_r = e_r2 = e_t = e_o = 0.0
    for n in range(tree.n_nodes):                      # DFS order: a parent comes before its children
        if reach[n] == 0.0:
            continue
        if tree.term[n]:
            r = tree.r2[n, seat] / 2.0
            e_r += reach[n] * r
            e_r2 += reach[n] * r * r
            e_t += reach[n] * st["scopas"][n, seat]
            e_o += reach[n] * st["scopas"][n, 1 - seat]
            continue
        k = int(tree.nlegal[n])
        p = policy_of_node(n) if tree.player[n] == seat else np.full(k, 1.0 / k)
        for a in range(k):
            reach[tree.child[n, a]] = reach[n] * p[a]
    return e_r, e_r2 - e_r * e_r, e_t, e_o


def _check_halves(by_seat, exact, what):
    for seat in (0, 1):
        got, (er, var, et, eo) = by_seat[seat], exact[seat]
        se = np.sqrt(var / got["episodes"])
        assert abs(got["reward"] - er) < 4 * se, (what, seat, got["reward"], er, se)
        assert abs(got["trained_scopas"] - et) < 4 * np.sqrt(max(et, 0.05) / got["episodes"]) + 1e-3, (what, seat, got["trained_scopas"], et)
        assert abs(got["opponent_scopas"] - eo) < 4 * np.sqrt(max(eo, 0.05) / got["episodes"]) + 1e-3, (what, seat, got["opponent_scopas"], eo)
        assert abs(got["reward_std_error"] - se) < 0.05 * se


def test_device_evaluator_against_exact_tree_enumeration(game, oracle):
    """evaluate_agent on the device (vanilla_cfr.py:157-216: trained vs uniform random, seats swapped at half time) against the EXACT
    expectation of every half, obtained by enumerating the tree: expected reward (within 4 standard errors of 10^6 episodes per seat,
    about 0.005) and expected scopas of either side, for three policies -- uniform, the average policy of 30 vanilla-CFR iterations,
    the average policy of batched MCCFR.  A policy row applied to the wrong infoset, seat or action slot moves these by tenths."""
    from scopa_amd.algorithms import CFRTrainer, MCCFRTrainer, evaluate_agent_device
    t = oracle.Tree(seed=42)
    tr = CFRTrainer(game)
    uni = np.zeros((t.n_infosets, 4))
    for i, n in enumerate(t.infoset_nlegal):
        uni[i, :n] = 1.0 / n
    tr.train(steps=30)
    cfr30 = tr._engine.ctx.exploitability(return_policy=True)["policy"]
    mc = MCCFRTrainer(game, batch=256, seed=11)
    mc.train(iterations=40)
    mcavg = mc._engine.ctx.exploitability(return_policy=True)["policy"]
    assert np.abs(cfr30 - uni).max() > 0.3 and np.abs(mcavg - cfr30).max() > 0.05         # three different policies
    for what, P in (("uniform", uni), ("cfr30", cfr30), ("mccfr", mcavg)):
        exact = [_exact_match(t, lambda n, P=P: P[t.infoset[n], :t.nlegal[n]], seat) for seat in (0, 1)]
        avg, st = evaluate_agent_device(tr, num_episodes=2_000_000, policy=P, stream_id=21)
        _check_halves(st["by_seat"], exact, what)
        assert abs(avg - 0.5 * (exact[0][0] + exact[1][0])) < 4 * np.sqrt(0.25 * (exact[0][1] + exact[1][1]) / 1e6)
    # the oracle's own policy value is the same enumeration: trained in seat 0 with uniform rows for player 1's infosets
    mixed = np.where((t.infoset_player == 0)[:, None], cfr30, uni)
    assert abs(t.policy_value(mixed) - _exact_match(t, lambda n: cfr30[t.infoset[n], :t.nlegal[n]], 0)[0]) < 1e-12


def test_deep_cfr_evaluate_vs_random_against_exact_tree_enumeration(game, oracle, golden):
    """DeepCFR.evaluate_vs_random (deep_cfr.py:367-429) with scripted nets -- the reference run's saved advantage nets as the only
    snapshot of either player -- against the exact expectation of each half: the policy of a node is
    positive_regret_policy(net(features), mask) (nets.py:93-101), uniform over the legal actions when nothing is positive, computed
    here on the CPU from the oracle's states; 10^6 episodes per seat."""
    import torch
    from scopa_amd.algorithms.deep_cfr import DeepCFR, FlexibleNet
    from scopa_amd.algorithms.deep_cfr.nets import positive_regret_policy
    g = golden.npz("sdcfr.npz")
    d = DeepCFR(game, num_players=2, device="cuda:0")
    cpu_nets = []
    for p in range(2):
        sd = {str(k): torch.from_numpy(g[f"net{p}__{k}"]) for k in g[f"net{p}_names"]}
        snap = FlexibleNet(mode="mlp", input_shape=(34,), output_dim=16, mlp_hidden=[128, 64]).to("cuda:0")
        snap.load_state_dict(sd)
        d.strategy_buffers[p].add_strategy(snap, 1)
        c = FlexibleNet(mode="mlp", input_shape=(34,), output_dim=16, mlp_hidden=[128, 64])
        c.load_state_dict(sd)
        cpu_nets.append(c)
    t = oracle.Tree(seed=42)
    st = t.states()
    pol = {}
    with torch.no_grad():
        for n in range(t.n_nodes):
            if t.term[n]:
                continue
            p, k = int(t.player[n]), int(t.nlegal[n])
            f, m = np.zeros(34, np.float32), np.zeros(16, np.float32)
            hand = [int(c) for c in st["hands"][n, p, :st["nh"][n, p]]]
            f[hand] = 1.0
            m[hand] = 1.0
            f[[16 + int(c) for c in st["table"][n, :st["nt"][n]]]] = 1.0
            f[32] = 1.0
            pr = positive_regret_policy(cpu_nets[p](torch.from_numpy(f)[None]), torch.from_numpy(m)[None])[0].numpy().astype(np.float64)
            w = np.maximum(pr[[int(a) for a in t.legal[n, :k]]], 0.0)
            pol[n] = w / w.sum() if w.sum() > 0 else np.full(k, 1.0 / k)       # k_eval_step: no positive mass -> uniform (:394-395)
    exact = [_exact_match(t, lambda n: pol[n], seat) for seat in (0, 1)]
    assert abs(exact[0][0] + 0.9201388888888888) + abs(exact[1][0] - 0.9201388888888888) > 0.05   # not the uniform policy's values (SURVEY 4 KAT)
    d.evaluate_vs_random(2_000_000)
    _check_halves(d.last_eval_by_seat, exact, "sdcfr nets")
