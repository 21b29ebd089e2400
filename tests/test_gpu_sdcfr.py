"""GPU tests of the SDCFR path against tests/golden/sdcfr.npz (produced by running the reference's DeepCFR).

Integer-valued outputs (features, masks, which rows are produced, their order) must match exactly; float32 values
within 1e-5, north_star's tolerance (the MLP forward runs through rocBLAS / the MFMA kernel here and through CPU BLAS in the reference)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ATOL = 1e-5


@pytest.fixture()
def dcfr(ctx, golden):
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    g = golden.npz("sdcfr.npz")
    torch.manual_seed(0)
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0")
    for p in range(2):
        sd = {str(k): torch.from_numpy(g[f"net{p}__{k}"]).to("cuda:0") for k in g[f"net{p}_names"]}
        d.advantage_nets[p].net.load_state_dict(sd)
    return d, g


def _opponent_draw_order(trav):
    """(ply, slot) of every opponent visit in the reference's DFS order (one np.random.choice each)."""
    order = []

    def rec(ply, slot):
        if ply == 8:
            return
        n = 4 - (ply >> 1)
        if (ply & 1) == trav:
            for k in range(n):
                rec(ply + 1, slot * n + k)
        else:
            order.append((ply, slot))
            rec(ply + 1, slot)
    rec(0, 0)
    return order


def test_constants_and_host_encoders(dcfr):
    d, g = dcfr
    assert d.input_dim == 34 == int(g["input_dim"][0])
    assert sum(p.numel() for p in d.advantage_nets[0].net.parameters()) == 13776
    s = d.game.new_initial_state()
    k = 0
    while not s.is_terminal():
        for pl in (0, 1):
            assert np.array_equal(d._state_to_features(s, pl), g["feat_line"][k]), (k, pl)
            assert np.array_equal(d._get_legal_actions_mask(s, pl), g["mask_line"][k])
            k += 1
        s.apply_action(s.legal_actions()[0])
    root = d._state_to_features(d.game.new_initial_state(), 0)
    assert list(np.nonzero(root)[0]) == [5, 6, 7, 9, 32]   # SURVEY §4 KAT


@pytest.mark.parametrize("trav", [0, 1])
def test_traversal_replay_matches_reference(dcfr, trav):
    import torch
    d, g = dcfr
    order = _opponent_draw_order(trav)
    u = np.random.RandomState(100 + trav).random_sample(len(order))
    assert len(order) == len(g[f"trav{trav}_draw_action"])
    uni = {}
    for ply in range(8):
        if (ply & 1) != trav:
            width = max(s for p, s in order if p == ply) + 1
            arr = np.zeros(width)
            for k, (p, s) in enumerate(order):
                if p == ply:
                    arr[s] = u[k]
            uni[ply] = torch.from_numpy(arr).to("cuda:0")
    seen = []

    def adv_fn(cur, feats, mask):
        seen.append((cur, feats.cpu().numpy().copy(), mask.cpu().numpy().copy()))
        return d.advantage_nets[cur].net(feats)

    v0 = d._engine.ctx.sdcfr_visits()
    val = d._traverse_batch(trav, 1, uniforms=uni, advantage_fn=adv_fn)
    assert d._engine.ctx.sdcfr_visits() - v0 == (105, 82)[trav]
    assert abs(float(val[0]) - float(g[f"trav{trav}_value"][0])) < ATOL
    # every node the reference visited was featurised identically (as multisets per ply: DFS vs level order)
    ref_f = g[f"trav{trav}_visit_feat"]
    mine = np.concatenate([f for _, f, _ in seen])
    assert mine.shape == ref_f.shape
    assert sorted(map(bytes, mine)) == sorted(map(bytes, ref_f))
    # the 41 memory rows, in the reference's append order
    mem = d.advantage_nets[trav].buffer
    assert len(mem) == 41
    f, r, m = mem.rows(torch.arange(41, device="cuda:0"))
    assert np.array_equal(f.cpu().numpy(), g[f"trav{trav}_row_feat"])
    assert np.array_equal(m.cpu().numpy(), g[f"trav{trav}_row_mask"])
    np.testing.assert_allclose(r.cpu().numpy(), g[f"trav{trav}_row_regret"], atol=ATOL, rtol=0)


def test_advantage_train_step_matches_reference(dcfr):
    import torch
    d, g = dcfr
    # put the reference's 41 rows into player 0's memory, then the same two Adam steps on the same 32-row samples
    mem = d.advantage_nets[0].buffer
    mem.feat[:41] = torch.from_numpy(g["trav0_row_feat"]).cuda()
    mem.regret[:41] = torch.from_numpy(g["trav0_row_regret"]).cuda()
    mem.mask[:41] = torch.from_numpy(g["trav0_row_mask"]).cuda()
    mem.advance(41)
    assert int(g["train_buffer_len"][0]) == 41
    loss = d.advantage_nets[0].train(epochs=2)
    assert abs(loss - float(g["train_loss_p0_epochs2"][0])) < 1e-5
    for k, v in d.advantage_nets[0].net.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g[f"net0_after__{k}"], atol=ATOL, rtol=0)


def test_get_policy_matches_reference(dcfr):
    import torch
    from scopa_amd.algorithms.deep_cfr import FlexibleNet
    d, g = dcfr
    for p in range(2):
        snap = FlexibleNet(mode="mlp", input_shape=(34,), output_dim=16, mlp_hidden=[128, 64]).to("cuda:0")
        snap.load_state_dict(d.advantage_nets[p].net.state_dict())
        d.strategy_buffers[p].add_strategy(snap, 1)
    s = d.game.new_initial_state()
    k = 0
    while not s.is_terminal():
        np.testing.assert_allclose(d.get_policy(s, s.current_player()), g["policy_line"][k], atol=ATOL, rtol=0)
        s.apply_action(s.legal_actions()[0])
        k += 1


def test_batched_traversal_invariants(dcfr):
    import torch
    d, _ = dcfr
    B = 512
    vals = d._traverse_batch(0, B)
    assert vals.shape == (B,) and float(vals.abs().max()) <= 4.0
    mem = d.advantage_nets[0].buffer
    assert len(mem) == B * 41
    f, r, m = mem.rows(torch.arange(B * 41, device="cuda:0"))
    f, r, m = f.cpu().numpy(), r.cpu().numpy(), m.cpu().numpy()
    assert np.array_equal(f[:, :16], m) and (f[:, 32] == 1).all() and (f[:, 33] == 0).all()
    nl = m.sum(1).reshape(B, 41)
    assert np.array_equal(np.sort(nl, axis=1), np.tile(np.sort([4] + [3] * 4 + [2] * 12 + [1] * 24), (B, 1)))
    assert (nl[:, -1] == 4).all()                          # post-order: the root's row is a traversal's last
    assert np.abs(r).max() <= 1.0 + 1e-6
    mx = np.abs(r).max(1)
    assert ((np.abs(mx - 1.0) < 1e-6) | (mx == 0)).all()   # max-abs normalisation
    # different traversals sample different opponent actions (Philox keyed by traversal id)
    assert len({bytes(x) for x in f.reshape(B, 41, 34)[:, 0]}) > 1


def test_train_loop_and_history(ctx):
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    torch.manual_seed(1)
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=8)
    d.train(iterations=3, advantage_epochs=2, eval_freq=1)
    h = d.training_history
    assert set(h) == {"losses", "values", "buffer_sizes", "eval_rewards", "eval_scopas"}
    assert h["buffer_sizes"][0] == [328, 656, 984] and len(h["losses"][1]) == 3 and len(h["eval_rewards"]) == 3
    assert len(d.strategy_buffers[0].strategies) == 2 and d.strategy_buffers[0].weights == [2, 3]
    reward, scopas = d.evaluate_vs_random(200)
    assert -4 <= reward <= 4 and len(scopas) == 2
    # a second train() call: the snapshot rule counts THIS call's iterations, as the reference's loop variable does (deep_cfr.py:431, 460-471: none at the
    # call's first iteration, weight = loop index + 1), while the traversal draws go on from the solver's running count; the host runs one iteration ahead
    # of the device and an evaluation drains that queue: the history stays in iteration order
    d.train(iterations=4, advantage_epochs=2, eval_freq=3)
    assert d.strategy_buffers[0].weights == [2, 3, 2, 3, 4] and d._iteration == 7
    assert h["buffer_sizes"][1] == [328 * k for k in range(1, 8)] and len(h["losses"][0]) == 7 and len(h["values"][1]) == 7 and len(h["eval_rewards"]) == 3 + 1 + 2
    assert all(np.isfinite(h["losses"][p]).all() and np.isfinite(h["values"][p]).all() for p in (0, 1))


def test_evaluate_vs_random_uniform_is_zero_mean(ctx):
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0")
    reward, scopas = d.evaluate_vs_random(40000)   # no snapshots: uniform vs uniform, seats swapped at half time
    assert abs(reward) < 0.06                       # +-0.92 per seat cancels; sigma/sqrt(n) ~ 0.01
    assert abs(scopas[0] - scopas[1]) < 0.03


def test_graphed_training_step_matches_eager(ctx, golden):
    """graph_training=True replays the optimiser step as a HIP graph: same two Adam steps as the reference fixture."""
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    g = golden.npz("sdcfr.npz")
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", graph_training=True)
    sd = {str(k): torch.from_numpy(g[f"net0__{k}"]).to("cuda:0") for k in g["net0_names"]}
    d.advantage_nets[0].net.load_state_dict(sd)
    mem = d.advantage_nets[0].buffer
    mem.feat[:41] = torch.from_numpy(g["trav0_row_feat"]).cuda()
    mem.regret[:41] = torch.from_numpy(g["trav0_row_regret"]).cuda()
    mem.mask[:41] = torch.from_numpy(g["trav0_row_mask"]).cuda()
    mem.advance(41)
    with torch.cuda.stream(d._stream):
        loss = d.advantage_nets[0].train(epochs=2)
    d._stream.synchronize()
    assert abs(loss - float(g["train_loss_p0_epochs2"][0])) < 1e-5
    for k, v in d.advantage_nets[0].net.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g[f"net0_after__{k}"], atol=ATOL, rtol=0)


@pytest.mark.parametrize("per_visit", [0, 1])
@pytest.mark.parametrize("trav", [0, 1])
def test_fused_traversal_matches_ply_by_ply_path(dcfr, trav, per_visit):
    """The one-call traversal in both its forms -- policy table per launch + walks (k_sdcfr_policy, k_sdcfr_walk; the default) and a
    forward pass per visit (k_sdcfr_traverse) -- vs the ply-by-ply path (PyTorch MLP): same Philox draws -> the same
    sampled actions, identical features/masks/row order, float32 values within 1e-5."""
    import torch
    d, _ = dcfr
    d._engine.ctx.sdcfr_mode(per_visit)
    B = 303                       # 75 whole tasks of four traversals and one of three
    v_ref = d._traverse_batch(trav, B, fused=False)
    mem = d.advantage_nets[trav].buffer
    fa, ra, ma = (t.clone() for t in mem.rows(torch.arange(B * 41, device="cuda:0")))
    mem.total = 0
    v0 = d._engine.ctx.sdcfr_visits()
    v_fused = d._traverse_batch(trav, B, fused=True)
    assert d._engine.ctx.sdcfr_visits() - v0 == (105, 82)[trav] * B
    fb, rb, mb = mem.rows(torch.arange(B * 41, device="cuda:0"))
    assert torch.equal(fa, fb) and torch.equal(ma, mb)
    np.testing.assert_allclose(rb.cpu().numpy(), ra.cpu().numpy(), atol=ATOL, rtol=0)
    np.testing.assert_allclose(v_fused.cpu().numpy(), v_ref.cpu().numpy(), atol=ATOL, rtol=0)


@pytest.mark.parametrize("trav", [0, 1])
def test_fused_traversal_replay_matches_reference(dcfr, trav):
    """The reference's own traversal (fixture) through the fused kernel, its np.random draws supplied as uniforms."""
    import torch
    d, g = dcfr
    order = _opponent_draw_order(trav)
    u = np.random.RandomState(100 + trav).random_sample(len(order))
    uni = {}
    for ply in range(8):
        if (ply & 1) != trav:
            width = max(s for p, s in order if p == ply) + 1
            arr = np.zeros(width)
            for k, (p, s) in enumerate(order):
                if p == ply:
                    arr[s] = u[k]
            uni[ply] = torch.from_numpy(arr).to("cuda:0")
    val = d._traverse_batch(trav, 1, uniforms=uni, fused=True)
    assert abs(float(val[0]) - float(g[f"trav{trav}_value"][0])) < ATOL
    mem = d.advantage_nets[trav].buffer
    f, r, m = mem.rows(torch.arange(41, device="cuda:0"))
    assert np.array_equal(f.cpu().numpy(), g[f"trav{trav}_row_feat"])
    assert np.array_equal(m.cpu().numpy(), g[f"trav{trav}_row_mask"])
    np.testing.assert_allclose(r.cpu().numpy(), g[f"trav{trav}_row_regret"], atol=ATOL, rtol=0)


def test_add_experience_normalises_and_appends_like_the_reference(dcfr):
    """AdvantageNetwork.add_experience (deep_cfr.py:70-75): advantages / (max|adv| + 1e-8) when the maximum is positive, the
    row appended at the deque's tail -- here the device ring the traversal kernels write, so kernel rows and caller rows interleave."""
    import torch
    d, g = dcfr
    a = d.advantage_nets[0]
    rng = np.random.RandomState(4)
    rows = []
    for k in range(5):
        f = rng.randint(0, 2, 34).astype(np.float32)
        adv = (rng.standard_normal(16) * 10.0 ** rng.randint(-3, 3)).astype(np.float32) if k != 2 else np.zeros(16, np.float32)
        m = rng.randint(0, 2, 16).astype(np.float32)
        a.add_experience(f, adv, m)
        want = adv / (np.max(np.abs(adv)) + 1e-8) if np.max(np.abs(adv)) > 0 else adv     # the reference's lines on float32 arrays
        rows.append((f, want.astype(np.float32), m))
    assert len(a.buffer) == 5
    for k, (f, r, m) in enumerate(rows):
        bf, br, bm = a.buffer[k]
        assert np.array_equal(bf, f) and np.array_equal(bm, m) and np.array_equal(br, r)
    d._traverse_batch(0, 2)                                              # kernel rows land behind the caller's rows
    assert len(a.buffer) == 5 + 2 * 41
    assert np.array_equal(a.buffer[4][0], rows[4][0])
    a.add_experience(rows[0][0], rows[0][1] * 3, rows[0][2])
    assert len(a.buffer) == 5 + 2 * 41 + 1 and np.array_equal(a.buffer[5 + 82][0], rows[0][0])
    assert a.train(epochs=1) >= 0.0                                      # and the memory trains
    # FIFO: a full ring drops the oldest row
    small = type(a)(34, 16, device="cuda:0", memory_size=3)
    for k in range(4):
        small.add_experience(np.full(34, k, np.float32), np.ones(16, np.float32), np.ones(16, np.float32))
    assert len(small.buffer) == 3 and [int(small.buffer[i][0][0]) for i in range(3)] == [1, 2, 3]


@pytest.mark.parametrize("trav", [0, 1])
def test_traversal_batch_with_device_draws_vs_oracle(dcfr, oracle, trav):
    """A batch of traversals with the product's own (Philox) draws, fused kernel and ply-by-ply kernels, against the oracle's
    restatement of _external_sampling_cfr (itself pinned to the reference run in tests/test_oracle_golden.py): the same 41 x B
    rows in the same order -- features / masks exact, normalised regrets and root values to 1e-5 -- at a non-zero iteration key."""
    import torch
    d, g = dcfr
    B = 96
    d._iteration = 3
    nets = np.stack([np.concatenate([v.cpu().numpy().reshape(-1) for v in d.advantage_nets[p].net.state_dict().values()]) for p in range(2)])
    t = oracle.Tree(seed=42)
    feat, reg, mask, ovals, visits = t.sdcfr_traverse(nets, trav, seed=0x5C09A, iteration=3, b0=0, nb=B)
    assert visits == (105, 82)[trav] * B
    for fused, per_visit in ((True, 0), (True, 1), (False, 0)):
        d._engine.ctx.sdcfr_mode(per_visit)
        mem = d.advantage_nets[trav].buffer
        base = len(mem)
        vals = d._traverse_batch(trav, B, fused=fused)
        f, r, m = mem.rows(torch.arange(base, base + 41 * B, device="cuda:0"))
        assert np.array_equal(f.cpu().numpy(), feat) and np.array_equal(m.cpu().numpy(), mask), (fused, per_visit)
        np.testing.assert_allclose(r.cpu().numpy(), reg, atol=ATOL, rtol=0)
        np.testing.assert_allclose(vals.cpu().numpy(), ovals, atol=ATOL, rtol=0)


def _weight_image(sd):
    """The fused kernel's LDS image of one net (include/scopa.h: scopa_sdcfr_pack_weights), built independently with numpy."""
    w1, b1 = sd["backbone.0.fc.weight"].cpu().numpy(), sd["backbone.0.fc.bias"].cpu().numpy()
    w2, b2 = sd["backbone.1.fc.weight"].cpu().numpy(), sd["backbone.1.fc.bias"].cpu().numpy()
    w3, b3 = sd["head.weight"].cpu().numpy(), sd["head.bias"].cpu().numpy()
    lane = np.arange(64)
    row, kq = lane & 15, lane >> 4
    i1 = np.empty((8, 2, 64, 4), np.float32)
    for mt in range(8):
        for g in range(2):
            for c in range(4):
                i1[mt, g, :, c] = w1[16 * mt + row, 4 * (4 * g + c) + kq]
    i2 = np.empty((4, 8, 64, 4), np.float32)
    for nt in range(4):
        for mt in range(8):
            for r in range(4):
                i2[nt, mt, :, r] = w2[16 * nt + row, 16 * mt + 4 * kq + r]
    i3 = np.empty((4, 64, 4), np.float32)
    for nt in range(4):
        for r in range(4):
            i3[nt, :, r] = w3[row, 16 * nt + 4 * kq + r]
    return np.concatenate([i1.reshape(-1), (b1 + w1[:, 32]).astype(np.float32), i2.reshape(-1), b2, i3.reshape(-1), b3])


def test_packed_weights_follow_the_nets(ctx):
    """The persistent weight image the fused kernel reads is rebuilt exactly when a net changed: after optimiser steps (plain and
    graph-REPLAYED -- a replay leaves the parameters' autograd version counters alone, so the second and later train() calls of a
    graph-trained net are the case that matters) and after load_state_dict it equals a fresh pack of the nets' tensors."""
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR

    def fresh(d):
        return np.stack([_weight_image(a.net.state_dict()) for a in d.advantage_nets])

    def packed(d):
        with torch.cuda.stream(d._stream):
            w = d._packed_weights()
        d._stream.synchronize()
        return w.cpu().numpy()

    for graph in (False, True, "hip"):                     # "hip": the opt-in hand-written step updates the tensors in place, outside ATen as well
        torch.manual_seed(3)
        d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=64, graph_training=(graph is True), train_backend="hip" if graph == "hip" else "torch")
        assert packed(d).shape == (2, 13520) and np.array_equal(packed(d), fresh(d))
        for p in (0, 1):
            d._traverse_batch(p, 64)
        for call in range(3):                              # call 0 captures the graph, calls 1 and 2 replay it
            before = packed(d).copy()
            with torch.cuda.stream(d._stream):
                d.advantage_nets[1].train(batch_size=128, epochs=2)
            d._stream.synchronize()
            now = packed(d)
            assert np.array_equal(now, fresh(d)), (graph, call)
            assert np.array_equal(now[0], before[0]) and not np.array_equal(now[1], before[1]), (graph, call)
        d.advantage_nets[0].net.load_state_dict(d.advantage_nets[1].net.state_dict())
        assert np.array_equal(packed(d), fresh(d)) and np.array_equal(packed(d)[0], packed(d)[1])
        with torch.no_grad():                              # an in-place edit made around the class
            d.advantage_nets[0].net.head.bias.add_(1.0)
        assert np.array_equal(packed(d), fresh(d))


def test_graph_training_and_eager_training_give_the_same_nets(ctx):
    """Three whole DeepCFR.train iterations with the optimiser step replayed as a HIP graph against the same three with eager
    steps: the traversals of iterations 1 and 2 read the nets the previous iteration trained (a stale weight image would
    freeze them at iteration 0's), so memory rows, losses and final weights agree."""
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    runs = []
    for graph in (False, True):
        torch.manual_seed(11)
        d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=64, graph_training=graph)
        d.train(iterations=3, advantage_epochs=3, eval_freq=100)
        mem = d.advantage_nets[0].buffer
        f, r, m = mem.rows(torch.arange(len(mem), device="cuda:0"))
        runs.append((d.training_history, [v.cpu().numpy() for a in d.advantage_nets for v in a.net.state_dict().values()], f.cpu().numpy(), r.cpu().numpy()))
    (h0, w0, f0, r0), (h1, w1, f1, r1) = runs
    same = (f0 == f1).all(1)                                               # the same actions were sampled in every iteration (a draw within
    assert same.mean() > 0.98                                              # float32 rounding of a cdf step may flip: a handful of rows at most)
    np.testing.assert_allclose(r0[same], r1[same], atol=2e-4, rtol=0)
    np.testing.assert_allclose(np.array(h0["losses"]), np.array(h1["losses"]), atol=1e-5, rtol=1e-4)
    for a, b in zip(w0, w1):
        np.testing.assert_allclose(a, b, atol=2e-5, rtol=0)                # fused Adam (graph mode) vs foreach Adam: same rule, float32 rounding
    assert not np.array_equal(r0[:64 * 41], r0[2 * 64 * 41:3 * 64 * 41])  # and iteration 2's regrets differ from iteration 0's: the nets moved


HIP_STEP_ATOL = 3e-6   # measured 2e-7 .. 1.1e-6 (differently ordered float32 sums) since Adam's bias corrections are computed on the host in double, as torch.optim.Adam does (round 3: 2e-5)


@pytest.mark.parametrize("batch_rows,epochs", [(128, 3), (32, 2), (4096, 2)])
def test_hip_training_step_matches_the_pytorch_step(ctx, batch_rows, epochs):
    """train_backend="hip" (scopa_sdcfr_train_step: forward, masked MSE, backward, clip_grad_norm_(1.0), Adam in two hand-written launches) against the
    default PyTorch step on the same memory rows and the same index batches: same loss, same weights after every train() call (float32 rounding of
    differently ordered sums: 3e-6), over three calls so that Adam's moments and step count carry over."""
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    ds = []
    for backend in ("torch", "hip"):
        torch.manual_seed(5)
        d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=128, train_backend=backend)
        ds.append(d)
    for it in range(3):
        ds[0]._iteration = it
        ds[0]._traverse_batch(0, 128)                                  # 5 248 fresh rows per call, written by the first solver ...
        src = ds[0].advantage_nets[0].buffer
        dst = ds[1].advantage_nets[0].buffer
        dst.feat.copy_(src.feat); dst.regret.copy_(src.regret); dst.mask.copy_(src.mask); dst.total = src.total   # ... and copied to the second: the SAME memory
        torch.cuda.synchronize()
        out = []
        for d in ds:
            a = d.advantage_nets[0]
            real = a.buffer.total
            if batch_rows == 32:
                a.buffer.total = 100                                   # fewer than 128 rows in memory: the reference's min(n, 32) batch
            with torch.cuda.stream(d._stream):
                loss = a.train(batch_size=128 if batch_rows == 32 else batch_rows, epochs=epochs)
            d._stream.synchronize()
            a.buffer.total = real
            out.append((loss, [v.detach().cpu().numpy().copy() for v in a.net.parameters()]))
        (l0, w0), (l1, w1) = out
        assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0)), (it, l0, l1)
        for x, y in zip(w0, w1):
            np.testing.assert_allclose(x, y, atol=HIP_STEP_ATOL, rtol=0)
    assert ds[1].advantage_nets[0]._hip_step == 3 * epochs and ds[0].advantage_nets[0]._hip_step == 0


def test_hip_training_step_orders_itself_against_a_foreign_stream(ctx):
    """AdvantageNetwork.train(train_backend="hip") called OUTSIDE the solver's stream (the public method, from torch's default stream): the launches go to the
    library context's stream, so the call must order itself behind what the caller's stream has queued and the caller's later reads behind its launches --
    same loss and bit-identical weights as the call made inside the solver's stream."""
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    outs = []
    for inside in (True, False):
        torch.manual_seed(9)
        d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=256, train_backend="hip")
        a = d.advantage_nets[0]
        for it in range(3):
            d._iteration = it
            d._traverse_batch(0, 256, sync=False)                     # rows still being written on the solver's stream when train() is called
            if inside:
                with torch.cuda.stream(d._stream):
                    loss = a.train(batch_size=128, epochs=4)
            else:
                loss = a.train(batch_size=128, epochs=4)              # torch's current (default) stream
                w_now = [p.detach().clone() for p in a.net.parameters()]   # a read on the caller's stream right after the call: must see the trained weights
            torch.cuda.synchronize()
            if not inside:
                for x, y in zip(w_now, a.net.parameters()):
                    assert torch.equal(x, y)
        outs.append((loss, [p.detach().cpu().numpy().copy() for p in a.net.parameters()]))
    (l0, w0), (l1, w1) = outs
    assert l0 == l1
    for x, y in zip(w0, w1):
        assert np.array_equal(x, y)


def test_hip_training_step_rejects_bad_arguments_and_trains(ctx):
    """scopa_sdcfr_train_steps: ragged batches, step 0 and misaligned weights are refused (SCOPA_EINVAL) -- AdvantageNetwork falls back to the PyTorch
    path for ragged batches by itself; and 200 iterations of DeepCFR.train on the hip backend bring the loss down and keep everything finite."""
    import torch
    from scopa_amd import _lib
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=64, train_backend="hip")
    a = d.advantage_nets[0]
    d._traverse_batch(0, 64)
    ptrs = tuple(p.data_ptr() for p in a.net.parameters())
    rows = torch.zeros(64, dtype=torch.long, device="cuda:0")
    state, loss = torch.zeros(2 * 13776, device="cuda:0"), torch.zeros(1, device="cuda:0")
    c = d._engine.ctx
    args = lambda n, step, p=ptrs: (rows.data_ptr(), n, 1, a.buffer.feat.data_ptr(), a.buffer.regret.data_ptr(), a.buffer.mask_ptr[0], a.buffer.capacity, p, state.data_ptr(), step, 5e-4, loss.data_ptr())
    for bad in (args(20, 1), args(0, 1), args(32, 0), args(32, 1, (ptrs[0] + 4,) + ptrs[1:])):
        with pytest.raises(_lib.ScopaError):
            c.sdcfr_train_steps(*bad)
    a.buffer.total = 20                                          # 20 rows in memory: the reference's min(n, 32) batch is 20 -- not whole tiles -> PyTorch path
    with torch.cuda.stream(d._stream), pytest.warns(RuntimeWarning, match="separate"):    # ... and says that its Adam state is not the hand-written step's
        a.train(epochs=1)
    d._stream.synchronize()
    assert a._hip_step == 0
    a.buffer.total = 41 * 64
    d.train(iterations=200, advantage_epochs=5, eval_freq=10 ** 9)
    L = np.array(d.training_history["losses"])
    assert np.isfinite(L).all() and L[:, -20:].mean() < 0.5 * L[:, :5].mean() and all(x._hip_step == 1000 for x in d.advantage_nets)
    assert all(bool(torch.isfinite(p).all()) for x in d.advantage_nets for p in x.net.parameters())


def _nets_flat(d):
    return np.stack([np.concatenate([v.cpu().numpy().reshape(-1) for v in d.advantage_nets[p].net.state_dict().values()]) for p in range(2)])


def _solver_with_reference_nets(golden, **kw):
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    g = golden.npz("sdcfr.npz")
    torch.manual_seed(0)
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", **kw)
    for p in range(2):
        d.advantage_nets[p].net.load_state_dict({str(k): torch.from_numpy(g[f"net{p}__{k}"]).to("cuda:0") for k in g[f"net{p}_names"]})
    return d, g


@pytest.mark.parametrize("per_visit", [0, 1])
@pytest.mark.parametrize("trav", [0, 1])
def test_traversal_at_the_stated_batch_vs_oracle(ctx, golden, oracle, trav, per_visit):
    """BASELINE configs[3]'s size (B = 4096 traversals per player per iteration) in ONE launch with the product's own
    draws: exact visit count, every row well-formed, and the 41 rows + root value of traversals 0, 1, 2047 and 4095 (first task,
    a middle one, the last lane of the last task) equal to the oracle's restatement of _external_sampling_cfr for those ids."""
    import torch
    B = 4096
    d, g = _solver_with_reference_nets(golden, batch=B)      # batch=B sizes the ring: 8 x 41 x B rows (one launch appends 167 936 rows, more than the
    d._iteration = 7                                         # reference's 100 000: a ring that small would only ever hold the tail of one launch)
    d._engine.ctx.sdcfr_mode(per_visit)
    mem = d.advantage_nets[trav].buffer
    assert mem.capacity == 8 * 41 * B
    v0 = d._engine.ctx.sdcfr_visits()
    vals = d._traverse_batch(trav, B).cpu().numpy()
    assert d._engine.ctx.sdcfr_visits() - v0 == (105, 82)[trav] * B
    assert len(mem) == 41 * B
    f, r, m = (x.cpu().numpy() for x in mem.rows(torch.arange(41 * B, device="cuda:0")))
    assert np.array_equal(f[:, :16], m) and (f[:, 32] == 1).all() and (f[:, 33] == 0).all()
    nl = m.sum(1).reshape(B, 41)
    assert np.array_equal(np.sort(nl, axis=1), np.tile(np.sort([4] + [3] * 4 + [2] * 12 + [1] * 24), (B, 1)))
    mx = np.abs(r).max(1)
    assert ((np.abs(mx - 1.0) < 1e-6) | (mx == 0)).all()
    nets, t = _nets_flat(d), oracle.Tree(seed=42)
    for tb in (0, 1, 2047, 4095):
        of, orr, om, ov, _ = t.sdcfr_traverse(nets, trav, seed=0x5C09A, iteration=7, b0=tb, nb=1)
        sl = slice(41 * tb, 41 * tb + 41)
        assert np.array_equal(f[sl], of) and np.array_equal(m[sl], om), tb
        np.testing.assert_allclose(r[sl], orr, atol=ATOL, rtol=0)
        assert abs(float(vals[tb]) - float(ov[0])) < ATOL


@pytest.mark.parametrize("trav", [0, 1])
def test_traversal_at_32768_both_forms_vs_each_other_and_oracle(ctx, golden, oracle, trav):
    """BASELINE configs[2]/[4]'s per-GPU shard size, 32 768 traversals in one call: here every wavefront takes SEVERAL tasks from its workgroup's counter
    (at 4 096 each takes at most one).  The table form and the forward-per-visit form write the same 1 343 488 rows and 32 768 root values bit for bit
    (compared on the device), every row is well-formed, and traversals 0, 1, 6143, 6144, 20 000 and 32 767 equal the oracle's, row for row."""
    import torch
    B = 32768
    d, g = _solver_with_reference_nets(golden, batch=B, memory_size=41 * B)
    d._iteration = 9
    mem = d.advantage_nets[trav].buffer
    got = {}
    for name, per_visit in (("table", 0), ("per-visit", 1)):
        d._engine.ctx.sdcfr_mode(per_visit)
        mem.total = 0
        v0 = d._engine.ctx.sdcfr_visits()
        vals = d._traverse_batch(trav, B)
        assert d._engine.ctx.sdcfr_visits() - v0 == (105, 82)[trav] * B
        got[name] = (mem.feat.clone(), mem.regret.clone(), mem.mask.clone(), vals.clone())
    d._engine.ctx.sdcfr_mode(0)
    for a, b in zip(got["table"], got["per-visit"]):
        assert torch.equal(a, b)
    f, r, m, vals = got["table"]
    assert torch.equal(f[:, :16], m) and bool((f[:, 32] == 1).all()) and bool((f[:, 33] == 0).all())
    nl = m.sum(1).reshape(B, 41).sort(dim=1).values
    assert torch.equal(nl, torch.tensor(sorted([4] + [3] * 4 + [2] * 12 + [1] * 24), dtype=nl.dtype, device=nl.device).expand(B, 41))
    mx = r.abs().amax(1)
    assert bool((((mx - 1.0).abs() < 1e-6) | (mx == 0)).all())
    nets, t = _nets_flat(d), oracle.Tree(seed=42)
    for tb in (0, 1, 6143, 6144, 20000, B - 1):
        of, orr, om, ov, _ = t.sdcfr_traverse(nets, trav, seed=0x5C09A, iteration=9, b0=tb, nb=1)
        sl = slice(41 * tb, 41 * tb + 41)
        assert np.array_equal(f[sl].cpu().numpy(), of) and np.array_equal(m[sl].cpu().numpy(), om), tb
        np.testing.assert_allclose(r[sl].cpu().numpy(), orr, atol=ATOL, rtol=0)
        assert abs(float(vals[tb]) - float(ov[0])) < ATOL


@pytest.mark.parametrize("fused", ["table", "per-visit", False])
def test_memory_ring_wraps_like_a_deque(ctx, oracle, golden, fused):
    """The advantage memory is a FIFO (deque(maxlen=...), deep_cfr.py:52): two launches of B traversals into a ring of 41 B + 100 rows --
    the second one STRADDLES the end of the ring -- must leave every row at its deque position: the kernel's rows (traversal, DFS
    post-order rank) against the oracle's rows of the same ids, indexed through a Python deque model; then sample_indices + rows()
    after the wrap (deep_cfr.py:88) fetch exactly the model's rows."""
    import collections
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    g = golden.npz("sdcfr.npz")
    B, cap = 96, 41 * 96 + 100
    torch.manual_seed(5)
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=B, memory_size=cap)
    for p in range(2):
        d.advantage_nets[p].net.load_state_dict({str(k): torch.from_numpy(g[f"net{p}__{k}"]).to("cuda:0") for k in g[f"net{p}_names"]})
    nets, t = _nets_flat(d), oracle.Tree(seed=42)
    trav = 0
    d._engine.ctx.sdcfr_mode(1 if fused == "per-visit" else 0)
    fused = bool(fused)
    mem = d.advantage_nets[trav].buffer
    model = collections.deque(maxlen=cap)                     # the reference's memory: (launch, row) tags in append order
    want = {}
    for launch in range(2):
        d._iteration = launch
        of, orr, om, _, _ = t.sdcfr_traverse(nets, trav, seed=0x5C09A, iteration=launch, b0=0, nb=B)
        for k in range(41 * B):
            model.append((launch, k))
            want[(launch, k)] = (of[k], orr[k], om[k])
        assert mem.write_base == (41 * B * launch) % cap
        d._traverse_batch(trav, B, fused=fused)
    assert len(mem) == cap == len(model) and mem.total == 2 * 41 * B
    assert mem.write_base == 2 * 41 * B - cap                 # the second launch ran over the end of the ring and continued at row 0
    f, r, m = (x.cpu().numpy() for x in mem.rows(torch.arange(cap, device="cuda:0")))   # deque order: index 0 = oldest surviving row
    for i, tag in enumerate(model):
        wf, wr, wm = want[tag]
        assert np.array_equal(f[i], wf) and np.array_equal(m[i], wm), (i, tag)
        np.testing.assert_allclose(r[i], wr, atol=ATOL, rtol=0)
    assert model[0] == (0, 41 * B - 100) and model[-1] == (1, 41 * B - 1)        # the 41 B - 100 oldest rows fell out
    # random.sample(self.buffer, batch) after the wrap: the same indices into the deque, the same rows
    a = d.advantage_nets[trav]
    a._rng.seed(42); a._rng.shuffle(list(range(16)))
    idx = a.sample_indices(len(mem), 128)
    import random
    ref = random.Random(); ref.seed(42); ref.shuffle(list(range(16)))
    assert idx == ref.sample(range(cap), 128)
    sf, sr, sm = (x.cpu().numpy() for x in mem.rows(torch.tensor(idx, device="cuda:0")))
    for k, i in enumerate(idx):
        wf, wr, wm = want[model[i]]
        assert np.array_equal(sf[k], wf) and np.array_equal(sm[k], wm)
        np.testing.assert_allclose(sr[k], wr, atol=ATOL, rtol=0)
    rows_dev = a._sample_rows(len(mem), 128, 2)               # what train() gathers: ring rows of the same deque indices
    a._rng.seed(42); a._rng.shuffle(list(range(16)))
    assert np.array_equal(rows_dev[0].cpu().numpy(), mem.logical_to_physical(torch.tensor(idx, device="cuda:0")).cpu().numpy())
    assert a.train(batch_size=128, epochs=1) >= 0.0


@pytest.mark.parametrize("seed", [7, 123])
def test_traversal_on_other_deals_vs_oracle(ctx, golden, oracle, seed):
    """The traversal kernels are deal-parametric (MiniScopaEnv(seed=k), mini_scopa_game.py:120-132; the reference's solvers only ever see
    seed 42): on two other deals -- other hands, other capture patterns, another node table in the fused kernel's LDS -- both paths
    give the oracle's rows for the same ids, and a second solver on the SAME context type sees its own deal."""
    import torch
    from scopa_amd.envs.openspiel_mini_scopa import MiniScopaGame
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    g = golden.npz("sdcfr.npz")
    torch.manual_seed(0)
    d = DeepCFR(MiniScopaGame(seed=seed), num_players=2, device="cuda:0", batch=64)
    for p in range(2):
        d.advantage_nets[p].net.load_state_dict({str(k): torch.from_numpy(g[f"net{p}__{k}"]).to("cuda:0") for k in g[f"net{p}_names"]})
    nets, t = _nets_flat(d), oracle.Tree(seed=seed)
    B = 64
    for trav in (0, 1):
        of, orr, om, ov, vis = t.sdcfr_traverse(nets, trav, seed=0x5C09A, iteration=0, b0=0, nb=B)
        assert vis == (105, 82)[trav] * B
        for fused, per_visit in ((True, 0), (True, 1), (False, 0)):
            d._engine.ctx.sdcfr_mode(per_visit)
            mem = d.advantage_nets[trav].buffer
            base = len(mem)
            vals = d._traverse_batch(trav, B, fused=fused)
            f, r, m = mem.rows(torch.arange(base, base + 41 * B, device="cuda:0"))
            assert np.array_equal(f.cpu().numpy(), of) and np.array_equal(m.cpu().numpy(), om), (seed, trav, fused, per_visit)
            np.testing.assert_allclose(r.cpu().numpy(), orr, atol=ATOL, rtol=0)
            np.testing.assert_allclose(vals.cpu().numpy(), ov, atol=ATOL, rtol=0)


@pytest.mark.parametrize("shift", [-10.0, -0.35])
def test_uniform_fallback_when_no_advantage_is_positive(ctx, golden, oracle, shift):
    """positive_regret_policy gives an ALL-ZERO row when no legal action has a positive advantage (nets.py:93-101), and the sampler then
    takes np.random.choice(legal_actions) -- uniform, numpy's float64 arithmetic on the same draw (deep_cfr.py:353-359); a traverser node
    there has value 0.  With the head bias lowered by 10 every node is such a node (the walk kernel's thr[0] = ~0 branch at every opponent
    visit); lowered by 0.35 about 30 % are (asserted: both branches run in one launch; with the fixture's nets as they are, none is).  All three forms against the oracle, and the walk
    against the per-visit kernel bit for bit."""
    import torch
    B = 67
    d, g = _solver_with_reference_nets(golden, batch=B)
    d._iteration = 5
    with torch.no_grad():
        for a in d.advantage_nets:
            a.net.head.bias.add_(shift)
            a._weights_changed()
    nets, t = _nets_flat(d), oracle.Tree(seed=42)
    for trav in (0, 1):
        feat, reg, mask, ovals, visits = t.sdcfr_traverse(nets, trav, seed=0x5C09A, iteration=5, b0=0, nb=B)
        with torch.no_grad():
            adv = d.advantage_nets[trav].net(torch.from_numpy(feat).to("cuda:0")).cpu().numpy()
        no_positive = ~((adv > 0) & (mask > 0)).any(1)    # traverser rows whose policy is the all-zero row (the opponent's nodes are the other traverser's rows)
        if shift == -10.0:
            assert no_positive.all() and (ovals == 0).all()   # every traverser node's value is 0 (:335 with an all-zero policy)
        else:
            assert 0.05 < no_positive.mean() < 0.95           # a mixture: both sampling branches run in one launch
        got = {}
        for name, fused, per_visit in (("table", True, 0), ("per-visit", True, 1), ("ply-by-ply", False, 0)):
            d._engine.ctx.sdcfr_mode(per_visit)
            mem = d.advantage_nets[trav].buffer
            mem.total = 0
            vals = d._traverse_batch(trav, B, fused=fused)
            f, r, m = mem.rows(torch.arange(41 * B, device="cuda:0"))
            assert np.array_equal(f.cpu().numpy(), feat) and np.array_equal(m.cpu().numpy(), mask), (name, trav)
            np.testing.assert_allclose(r.cpu().numpy(), reg, atol=ATOL, rtol=0)
            np.testing.assert_allclose(vals.cpu().numpy(), ovals, atol=ATOL, rtol=0)
            got[name] = (r.clone(), vals.clone())
        assert torch.equal(got["table"][0], got["per-visit"][0]) and torch.equal(got["table"][1], got["per-visit"][1])
    d._engine.ctx.sdcfr_mode(0)


def test_walk_and_per_visit_forms_are_bitwise_the_same(dcfr):
    """k_sdcfr_policy evaluates a node with the tile arithmetic of k_sdcfr_traverse (same MFMA sequence): the two forms of the one-call
    traversal produce the SAME BITS -- rows, regrets, root values -- at every task shape of the walk kernel (1, 2, 4, 8 traversals per
    wavefront) and at batches that leave partial tasks."""
    import torch
    d, _ = dcfr
    ctx = d._engine.ctx
    d._iteration = 2
    for trav in (0, 1):
        for B in (1, 5, 67):
            ref = None
            for per_visit, T in ((1, 0), (0, 1), (0, 2), (0, 4), (0, 8)):
                ctx.sdcfr_mode(per_visit)
                ctx.sdcfr_tuning(T, 0)
                mem = d.advantage_nets[trav].buffer
                mem.total = 0
                vals = d._traverse_batch(trav, B)
                got = [x.clone() for x in mem.rows(torch.arange(41 * B, device="cuda:0"))] + [vals.clone()]
                if ref is None:
                    ref = got
                else:
                    assert all(torch.equal(a, b) for a, b in zip(ref, got)), (trav, B, per_visit, T)
    ctx.sdcfr_mode(0)
    ctx.sdcfr_tuning(0, 0)
