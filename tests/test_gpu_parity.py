"""GPU parity: the HIP path (through the C ABI) against the oracle and the committed golden vectors.

Integer game state, infoset ids and visit counters must be bit-exact; the exact-semantics solvers must reproduce
the reference's float64 tables bit-for-bit; the batched (frozen-table) MCCFR matches the oracle's definition to
1e-12 relative (float64 atomic sums re-associate) with EXACT integer visit counts."""
import numpy as np
import pytest

from conftest import frozen_case, unpack_state

pytestmark = pytest.mark.gpu


def _random_states(sl, oracle, n, rng):
    """n states reached by random play (20 % illegal actions) on random deals, and the next action to apply."""
    states = np.zeros(n, sl.STATE_DTYPE)
    actions = np.zeros(n, np.uint8)
    expect = []
    for i in range(n):
        perm = rng.permutation(16).astype(np.uint8)
        o = oracle.State(perm=perm)
        for _ in range(rng.randint(0, 9)):
            legal = o.legal()
            if not legal:
                break
            o.step(int(legal[rng.randint(len(legal))]) if rng.rand() < 0.8 else int(rng.randint(16)))
        sn = o.snapshot()
        for p in range(2):
            states[i]["hand"][p] = sum(c << (4 * k) for k, c in enumerate(sn["hands"][p]))
            states[i]["nh"][p] = len(sn["hands"][p])
            states[i]["ncap"][p] = sn["ncap"][p]
            states[i]["scopas"][p] = sn["scopas"][p]
        states[i]["table"] = sum(c << (4 * k) for k, c in enumerate(sn["table"]))
        states[i]["nt"] = len(sn["table"])
        states[i]["step"] = sn["step"]
        legal = o.legal()
        a = int(legal[rng.randint(len(legal))]) if legal and rng.rand() < 0.8 else int(rng.randint(16))
        actions[i] = a
        o.step(a)
        expect.append(o.snapshot())
    return states, actions, expect


def test_step_batch_bit_exact(ctx, sl, oracle):
    rng = np.random.RandomState(11)
    states, actions, expect = _random_states(sl, oracle, 4096, rng)
    ctx.step_batch_host(states, actions)
    for i in range(states.size):
        assert unpack_state(states[i]) == expect[i], i


def test_step_batch_edge_cases(ctx, sl):
    ctx.step_batch_host(np.zeros(0, sl.STATE_DTYPE), np.zeros(0, np.uint8))  # empty batch is a no-op
    s = np.zeros(1, sl.STATE_DTYPE)
    s[0]["step"] = 8  # terminal: a dead step changes nothing
    before = s.copy()
    ctx.step_batch_host(s, np.array([3], np.uint8))
    assert s.tobytes() == before.tobytes()


def test_step_batch_cloned_states_play_to_step_16(ctx, sl, golden):
    """a MiniScopaState.clone() has max_steps 16 (openspiel_mini_scopa.py:108).  The 240 reference playouts that clone
    at random plies (tests/golden/playouts_cloned.json), all stepped in lockstep by k_step_batch through scopa_step_batch: every ply of
    every game bit-exact, incl. the plies past step 8 that an un-cloned state never reaches; finished games take dead steps."""
    cases = golden.json("playouts_cloned.json")
    n = len(cases)
    L = sl.lib()
    import ctypes as C
    states = np.zeros(n, sl.STATE_DTYPE)
    for i, c in enumerate(cases):
        s = sl.State16()
        L.scopa_state_init(sl.deal_py_seed(c["seed"]).ctypes.data_as(C.c_void_p), C.byref(s))
        states[i] = np.frombuffer(bytes(s), sl.STATE_DTYPE)[0]
    past8 = 0
    for ply in range(17):
        actions = np.zeros(n, np.uint8)
        for i, c in enumerate(cases):
            if ply < len(c["actions"]):
                actions[i] = c["actions"][ply]
                if c["clone_before"][ply]:
                    states[i]["step"] |= sl.STEP_CLONED        # what scopa_state_clone does (checked on the host in test_abi_host)
            else:
                actions[i] = (7 * i + ply) & 15                # game over: any action is a dead step
        before = states.copy()
        ctx.step_batch_host(states, actions)
        for i, c in enumerate(cases):
            if ply >= len(c["actions"]):
                assert states[i].tobytes() == before[i].tobytes(), (i, ply)
                continue
            tr = c["trail"][ply]
            sn = unpack_state(states[i])
            assert bool(sn["step"] & sl.STEP_CLONED) == tr["cloned"]
            sn["step"] &= sl.STEP_COUNT_MASK
            assert sn == {k: tr[k] for k in sn}, (i, ply)
            past8 += tr["step"] > 8
    assert past8 > 300
    r2 = np.array([[2 * r for r in c["rewards"]] for c in cases])
    got = np.stack([states["ncap"][:, 0] + 2.0 * states["scopas"][:, 0], states["ncap"][:, 1] + 2.0 * states["scopas"][:, 1]], 1)
    assert np.array_equal(got - got[:, ::-1], r2)          # evaluate_game x 2 = own points - the other's (mini_scopa_game.py:106-114)


def test_step_batch_random_cloned_states_vs_oracle(ctx, sl, oracle):
    """65 536 random games, each cloned at a random ply (or never), 35 % arbitrary actions, 16 lockstep plies on the GPU against the
    oracle's literal max_steps rule."""
    rng = np.random.RandomState(2026)
    n = 65536
    perms = np.stack([rng.permutation(16) for _ in range(n)]).astype(np.uint8)
    states = np.zeros(n, sl.STATE_DTYPE)
    states["hand"][:, 0] = (perms[:, 0] | (perms[:, 1] << 4)).astype(np.uint16) | ((perms[:, 2].astype(np.uint16) | (perms[:, 3].astype(np.uint16) << 4)) << 8)
    states["hand"][:, 1] = (perms[:, 4] | (perms[:, 5] << 4)).astype(np.uint16) | ((perms[:, 6].astype(np.uint16) | (perms[:, 7].astype(np.uint16) << 4)) << 8)
    states["nh"][:] = 4
    clone_at = rng.randint(0, 20, n)                           # >= 16: never cloned
    acts = rng.randint(0, 16, (16, n)).astype(np.uint8)
    play_legal = rng.rand(16, n) < 0.65
    ref = [oracle.State(perm=perms[i]) for i in range(2048)]   # the oracle follows the first 2048 games ply by ply
    for ply in range(16):
        hand = states["hand"][np.arange(n), ply & 1].astype(np.uint32)
        nh = states["nh"][np.arange(n), ply & 1].astype(np.uint32)
        pick = (hand >> (4 * (acts[ply] % np.maximum(nh, 1)))) & 15
        a = np.where(play_legal[ply] & (nh > 0), pick, acts[ply]).astype(np.uint8)
        over = ((states["nh"][:, 0] | states["nh"][:, 1]) == 0) | (states["step"] == 8)   # a terminal state's clone keeps its limit (scopa_state_clone)
        states["step"][(clone_at == ply) & ~over] |= sl.STEP_CLONED
        ctx.step_batch_host(states, a)
        for i in range(2048):
            if clone_at[i] == ply:
                ref[i] = ref[i].clone()
            ref[i].step(int(a[i]))
            sn = unpack_state(states[i])
            sn["step"] &= sl.STEP_COUNT_MASK
            assert sn == ref[i].snapshot(), (i, ply)
    steps = states["step"] & sl.STEP_COUNT_MASK
    cloned = (states["step"] & sl.STEP_CLONED) != 0
    empty = (states["nh"][:, 0] | states["nh"][:, 1]) == 0
    assert np.all(steps[~cloned] <= 8) and np.all(empty | (steps == np.where(cloned, 16, 8)))   # the terminal rule, on all 65 536
    assert (steps > 8).sum() > 10000


@pytest.mark.parametrize("seed", [42, 0, 1, 7, 123])
def test_tree_build_matches_reference(ctx, sl, golden, seed):
    g = golden.npz(f"tree_seed{seed}.npz")
    n_inf = ctx.set_deal(sl.deal_py_seed(seed))
    assert n_inf == len(g["infoset_strings"])
    t = ctx.tree_export()
    assert np.array_equal(t["infoset"], g["infoset"])
    assert np.array_equal(t["r2"], g["r2"])
    assert [sl.key_to_string(k) for k in t["infoset_key"]] == list(g["infoset_strings"])
    for i in range(0, 2229, 7):
        sn = unpack_state(t["states"][i])
        nh = g["nh"][i]
        assert sn["hands"] == [list(g["hands"][i, p, :nh[p]]) for p in range(2)]
        assert sn["table"] == list(g["table"][i, :g["nt"][i]])
        assert sn["ncap"] == list(g["ncap"][i]) and sn["scopas"] == list(g["scopas"][i]) and sn["step"] == g["step"][i]


def test_tree_build_random_deals_vs_oracle(ctx, sl, oracle):
    rng = np.random.RandomState(3)
    for _ in range(6):
        perm = rng.permutation(16).astype(np.uint8)
        o = oracle.Tree(perm=perm)
        assert ctx.set_deal(perm) == o.n_infosets
        t = ctx.tree_export()
        assert np.array_equal(t["infoset"], o.infoset.astype(np.int32))
        assert np.array_equal(t["r2"], o.r2)
        assert [sl.key_to_string(k) for k in t["infoset_key"]] == o.infoset_strings
        assert np.array_equal(t["infoset_legal"], o.infoset_legal)


def test_vanilla_cfr_exact_bit_exact_vs_reference(ctx, sl, golden):
    g = golden.npz("vanilla_cfr.npz")
    ctx.set_deal(sl.deal_py_seed(42))
    done, rvs = 0, []
    for cp in [int(c) for c in g["checkpoints"]]:
        rvs.append(ctx.cfr_exact_iterate(cp - done))
        done = cp
        R, S, L = ctx.tables_get()
        assert np.array_equal(R, g[f"it{cp}_regret"]), cp
        assert np.array_equal(S, g[f"it{cp}_strategy"]), cp
        assert np.array_equal(L, g[f"it{cp}_local"]), cp
    assert np.array_equal(np.concatenate(rvs), g["root_values"])
    assert ctx.counters() == (3306 * done, 1152 * done)


def test_vanilla_cfr_single_traversals(ctx, sl, golden):
    g = golden.npz("vanilla_cfr.npz")
    ctx.set_deal(sl.deal_py_seed(42))
    assert ctx.cfr_exact_traverse(0) == g["root_values"][0, 0]
    assert ctx.cfr_exact_traverse(1) == g["root_values"][0, 1]
    R, _, _ = ctx.tables_get()
    assert np.array_equal(R, g["it1_regret"])


@pytest.mark.parametrize("seed", [42, 0, 123, 282, 129])
def test_cfr_exact_schedule_is_bit_identical_to_the_sequential_walk(ctx, sl, seed):
    """Whole-tree vanilla-CFR traversals run as a schedule of parallel steps (31 .. 194 steps for these deals instead of 1653 visits)
    that keeps the reference's visit order per infoset; the one-lane sequential walk (itself bit-exact vs vanilla_cfr.npz) is the check:
    tables, root values, first-visit order and counters must be IDENTICAL, on deals with few / many infosets and from non-zero tables."""
    ctx.set_deal(sl.deal_py_seed(seed))
    out = []
    for sequential in (True, False):
        ctx.tables_reset()
        ctx.cfr_exact_mode(sequential)
        c0 = ctx.counters()
        rv = [ctx.cfr_exact_iterate(1), ctx.cfr_exact_iterate(6)]
        v1 = ctx.cfr_exact_traverse(1)                      # a single traversal of one player on top
        R, S, L = ctx.tables_get()
        c1 = ctx.counters()
        out.append((np.concatenate(rv), v1, R, S, L, ctx.visited_get(), (c1[0] - c0[0], c1[1] - c0[1])))
    ctx.cfr_exact_mode(False)
    a, b = out
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert np.array_equal(a[5], b[5]) and a[6] == b[6] == (1653 * 15, 576 * 15)


@pytest.mark.parametrize("seed,iters", [(0, 1), (0, 10), (1, 200), (2, 200)])
def test_mccfr_replay_bit_exact_vs_reference(ctx, sl, golden, seed, iters):
    m = golden.npz("mccfr.npz")
    tag = f"s{seed}_it{iters}"
    ctx.set_deal(sl.deal_py_seed(42))
    u = np.random.RandomState(seed).random_sample(463 * iters)
    assert ctx.mccfr_replay(iters, u) == 463 * iters
    R, S, _ = ctx.tables_get()
    keys = [sl.key_to_string(k) for k in ctx.tree_export()["infoset_key"]]
    idx = [keys.index(k.split("|", 1)[1]) for k in m[tag + "_keys"]]
    assert np.array_equal(R[idx], m[tag + "_regret"]) and np.array_equal(S[idx], m[tag + "_strategy"])
    assert ctx.counters() == (463 * iters, 240 * iters)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_mccfr_batched_delta_vs_reference_sample(ctx, sl, golden, case):
    """k_mccfr_traverse against the REFERENCE: tests/golden/mccfr_frozen.npz holds the deltas that the reference's own
    MCCFRTrainer._sample (mc_cfr.py:37-86) accumulates when driven with frozen strategies and the path-keyed Philox draws
    (oracle/gen_golden.py:gen_mccfr_frozen).  Visit counts exact, strategy deltas = count x sigma(frozen) and regret deltas to 1e-12
    (the kernel adds the pairs' increments in another order)."""
    ctx.set_deal(sl.deal_py_seed(42))
    keys = [sl.key_to_string(k) for k in ctx.tree_export()["infoset_key"]]
    R, seed, it, b0, nb, dR, dS, idx, _ = frozen_case(golden, keys, case)
    ctx.tables_set(regret=R, strategy=np.zeros_like(R))
    ctx.mccfr_seed(seed)
    ctx.mccfr_traverse(it, b0, nb)
    d = ctx.mccfr_delta_get()
    assert np.array_equal(d[:, 4], np.rint(dS.sum(1)))           # every traverser visit adds a probability vector: row sums = visit counts
    np.testing.assert_allclose(d[:, :4], dR, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(dR).max()))
    assert ctx.counters() == (463 * nb, 240 * nb)
    ctx.mccfr_apply()                                            # strategy_sum += count * sigma(frozen regret): the reference's `+= 1.0 * sigma` per visit
    Rn, Sn, _ = ctx.tables_get()
    np.testing.assert_allclose(Sn, dS, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(Rn, R + dR, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(dR).max()))
    seen = ctx.visited_get() != 0
    assert set(np.flatnonzero(seen)) == set(idx)                 # exactly the reference's dict keys exist afterwards


def test_mccfr_first_visit_tracking_switches_off_once_every_infoset_is_marked(ctx, sl, oracle):
    """A traversal launch records which infosets it sees for the first time (the keys the reference's dict would hold, mc_cfr.py:32-35).  Once every infoset
    of the deal is marked the host stops asking for it (scopa_mccfr.hip refresh_all_seen: checked at most every 16th launch): the walks skip their `seen`
    flags and the epilogue its scan.  The marks grow monotonically to all 738, stay what they are afterwards, and a launch on the tracking-off path gives the
    oracle's deltas and exact counts like a launch on a fresh context."""
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.mccfr_seed(0x5C09A)
    prev = np.zeros(t.n_infosets, bool)
    for chunk in range(12):
        ctx.mccfr_iterate(256, 20)                                  # 240 launches in all: the host re-counts the marks every 16th
        seen = ctx.visited_get() != 0
        assert np.all(seen[prev])                                   # monotone
        prev = seen
    assert prev.all()                                               # 61 440 traversal pairs reach every infoset of the deal
    marks = ctx.visited_get().copy()
    ctx.mccfr_iterate(256, 40)                                      # by now launches run with tracking off
    assert np.array_equal(ctx.visited_get(), marks)
    R, S, _ = ctx.tables_get()
    it = ctx.mccfr_iteration()
    c0 = ctx.counters()
    ctx.mccfr_delta_set(np.zeros((t.n_infosets, 5)))
    ctx.mccfr_traverse(it, 0, 3000)
    d = ctx.mccfr_delta_get()
    c1 = ctx.counters()
    dR, dS, dv, tv = t.mccfr_batched_delta(R, 0x5C09A, it, 0, 3000)
    assert (c1[0] - c0[0], c1[1] - c0[1]) == (dv, tv) == (463 * 3000, 240 * 3000)
    assert np.array_equal(d[:, 4], np.rint(dS.sum(1)))
    np.testing.assert_allclose(d[:, :4], dR, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(dR).max()))
    ctx.tables_reset()                                              # a reset clears the marks: tracking is on again
    ctx.mccfr_traverse(0, 0, 8)
    assert 0 < int((ctx.visited_get() != 0).sum()) < t.n_infosets


# batches beyond 4096 exercise how a workgroup's wavefronts take pairs (scopa_mccfr.hip, main loop): 4101 = 16 wavefronts x 256
# workgroups + a ragged last workgroup; 5000 = between one and two pairs per wavefront (single takes from the counter); 10240 = two pairs
# in flight, then single takes; 17923 = more than four pairs per wavefront (double takes, single ones towards the end, ragged tail)
@pytest.mark.parametrize("batch", [1, 7, 64, 1000, 4101, 5000, 10240, 17923])
def test_mccfr_batched_delta_vs_oracle(ctx, sl, oracle, batch):
    """One iteration's deltas from a non-trivial frozen table: regret deltas to 1e-12, visit counts EXACT."""
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    R, S, L = t.tables()
    t.cfr_exact(R, S, L, 3)  # a table with mixed-sign regrets
    ctx.tables_set(regret=R)
    ctx.mccfr_seed(0xABCDEF12345)
    ctx.mccfr_traverse(5, 10, batch)
    d = ctx.mccfr_delta_get()
    dR, dS, dv, tv = t.mccfr_batched_delta(R, 0xABCDEF12345, 5, 10, batch)
    assert np.array_equal(d[:, 4], np.rint(dS.sum(1)))          # traverser-visit counts per infoset: exact
    assert d[:, 4].sum() == 172 * batch                          # 86 + 86 traverser visits per traversal pair
    np.testing.assert_allclose(d[:, :4], dR, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(dR).max()))
    assert ctx.counters() == (dv, tv) == (463 * batch, 240 * batch)


def test_mccfr_batched_iterations_vs_oracle(ctx, sl, oracle):
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.mccfr_seed(99)
    ctx.mccfr_iterate(256, 6)
    R, S, _ = ctx.tables_get()
    Ro, So, _ = t.tables()
    t.mccfr_batched(Ro, So, 99, 0, 6, 256)
    np.testing.assert_allclose(R, Ro, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(S, So, rtol=1e-10, atol=1e-10)
    assert ctx.mccfr_iteration() == 6
    assert ctx.counters()[0] == 463 * 256 * 6


def test_mccfr_batched_split_invariance(ctx, sl):
    """Traversal ids split over several launches (as over several GPUs) give the same deltas: the RNG is keyed by
    the global traversal id, not by launch geometry."""
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.mccfr_seed(5)
    ctx.mccfr_traverse(0, 0, 600)
    whole = ctx.mccfr_delta_get()
    ctx.mccfr_delta_set(np.zeros_like(whole))
    for b0, nb in ((0, 100), (100, 371), (471, 129)):
        ctx.mccfr_traverse(0, b0, nb)
    parts = ctx.mccfr_delta_get()
    assert np.array_equal(parts[:, 4], whole[:, 4])
    np.testing.assert_allclose(parts[:, :4], whole[:, :4], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("b0,nb", [(5 * 32768, 32768), (0, 65536)])
def test_mccfr_config2_rank_shard_vs_oracle(ctx, sl, oracle, b0, nb):
    """BASELINE configs[2]'s per-rank shape: rank 5 of 8 owns traversal ids [5*32768, 6*32768) of the 262144-id iteration (and the
    65536-pair launch of bench.py's large-batch figure).  nb > 4096 means every wavefront walks SEVERAL pairs and reuses its scratch
    (the multi-pass loop of k_mccfr_traverse): the whole launch is compared with the oracle -- visit counts exact, regret deltas to
    1e-12 -- plus split invariance over the shard (the same ids in three launches of other geometry)."""
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    R, S, L = t.tables()
    t.cfr_exact(R, S, L, 3)
    ctx.tables_set(regret=R)
    seed, it = 0x5C09A, 7
    ctx.mccfr_seed(seed)
    ctx.mccfr_traverse(it, b0, nb)
    d = ctx.mccfr_delta_get()
    dR, dS, dv, tv = t.mccfr_batched_delta(R, seed, it, b0, nb)
    assert np.array_equal(d[:, 4], np.rint(dS.sum(1)))
    assert d[:, 4].sum() == 172 * nb and d[0, 4] == nb
    scale = max(1.0, np.abs(dR).max())
    np.testing.assert_allclose(d[:, :4], dR, rtol=1e-12, atol=1e-12 * scale)
    assert ctx.counters() == (dv, tv) == (463 * nb, 240 * nb)
    ctx.mccfr_delta_set(np.zeros_like(d))
    for o, n in ((0, 4097), (4097, 20000), (24097, nb - 24097)):
        ctx.mccfr_traverse(it, b0 + o, n)
    parts = ctx.mccfr_delta_get()
    assert np.array_equal(parts[:, 4], d[:, 4])
    np.testing.assert_allclose(parts[:, :4], d[:, :4], rtol=1e-12, atol=1e-12 * scale)
    # a sub-range on its own (ids b0 .. b0+255) is the oracle's delta of exactly those ids
    ctx.mccfr_delta_set(np.zeros_like(d))
    ctx.mccfr_traverse(it, b0, 256)
    sub = ctx.mccfr_delta_get()
    sR, sS, _, _ = t.mccfr_batched_delta(R, seed, it, b0, 256)
    assert np.array_equal(sub[:, 4], np.rint(sS.sum(1)))
    np.testing.assert_allclose(sub[:, :4], sR, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(sR).max()))


def test_mccfr_config2_whole_iteration_on_one_gpu(ctx, sl, oracle):
    """BASELINE configs[2] in one piece: the 262 144-traversal iteration as the eight ranks of the 8-GPU run would split it --
    eight launches mccfr_traverse(it, r * 32768, 32768), r = 0..7 (shard_range's partition) -- into one delta, ONE mccfr_apply,
    then a second iteration on the tables that left; against the oracle's og_mccfr_batched over all 262 144 pairs per iteration
    (MCCFRTrainer.iteration, mc_cfr.py:88-92, with tables frozen per iteration): visit counts exact, strategy tables to 1e-12,
    regret tables to 1e-11 relative.  What the 8-GPU run adds to this is only where the eight partial deltas are summed."""
    from scopa_amd.distributed import shard_range
    total, world = 262144, 8
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    R, S, L = t.tables()
    t.cfr_exact(R, S, L, 2)                       # a table with structure, not all zeros
    ctx.tables_set(regret=R, strategy=S)
    seed = 0x5C09A
    ctx.mccfr_seed(seed)
    Ro, So = R.copy(), S.copy()
    d0, t0 = ctx.counters()
    for it in range(2):
        assert ctx.mccfr_iteration() == it
        for r in range(world):
            b0, nb = shard_range(total, r, world)
            assert (b0, nb) == (r * 32768, 32768)
            ctx.mccfr_traverse(it, b0, nb)
        delta = ctx.mccfr_delta_get()
        assert delta[:, 4].sum() == 172 * total and delta[0, 4] == total   # traverser visits: 86 + 86 per pair; the root once per pair
        ctx.mccfr_apply()
        visits = t.mccfr_batched(Ro, So, seed, it, 1, total)
        assert visits == 463 * total
        Rg, Sg, _ = ctx.tables_get()
        # 262 144 pairs' increments added in another order (float64 atomics): the weights reach / sampling probability of the
        # reference's update span many orders of magnitude (|regret| up to 1e12 here), so sums cancel: 1e-11 relative
        np.testing.assert_allclose(Rg, Ro, rtol=1e-11, atol=1e-12 * max(1.0, np.abs(Ro).max()))
        np.testing.assert_allclose(Sg, So, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(So).max()))
        Ro, So = Rg.copy(), Sg.copy()             # the next iteration starts from the SAME tables on both sides (its integer sampling
                                                  # thresholds are a function of the rounded regrets)
    d1, t1 = ctx.counters()
    assert (d1 - d0, t1 - t0) == (2 * 463 * total, 2 * 240 * total)
    assert ctx.mccfr_iteration() == 2


def test_mccfr_graph_mode_replays_the_same_iterations(ctx, sl, oracle):
    """scopa_mccfr_graph_mode: 150 iterations as captured HIP graphs (64 + 64 + 22: two replays of one graph and a shorter one), the
    iteration number read from a device word the apply launch advances -- the same iteration ids as the eager loop, so the same
    draws: exact visit counters, the same first-visit order, tables equal to the eager run's up to the order of the float64
    atomics; then eager iterations continue from where the graphs stopped."""
    perm = sl.deal_py_seed(42)
    runs = []
    for graph in (False, True):
        ctx.set_deal(perm)
        ctx.mccfr_seed(77)
        ctx.mccfr_graph_mode(graph)
        c0 = ctx.counters()
        ctx.mccfr_iterate(512, 150)
        assert ctx.mccfr_iteration() == 150
        ctx.mccfr_graph_mode(False)
        ctx.mccfr_iterate(512, 3)                # eager launches take over at iteration 150
        assert ctx.mccfr_iteration() == 153
        R, S, _ = ctx.tables_get()
        c1 = ctx.counters()
        runs.append((R, S, ctx.visited_get(), (c1[0] - c0[0], c1[1] - c0[1])))
    (R0, S0, v0, n0), (R1, S1, v1, n1) = runs
    assert n0 == n1 == (463 * 512 * 153, 240 * 512 * 153)
    assert np.array_equal(v0 > 0, v1 > 0)
    np.testing.assert_allclose(R1, R0, rtol=1e-9, atol=1e-9 * np.abs(R0).max())
    np.testing.assert_allclose(S1, S0, rtol=1e-9, atol=1e-9 * np.abs(S0).max())
    # and against the oracle from scratch for a short run (graph mode only)
    ctx.set_deal(perm)
    ctx.mccfr_seed(5)
    ctx.mccfr_graph_mode(True)
    ctx.mccfr_iterate(256, 5)
    ctx.mccfr_graph_mode(False)
    t = oracle.Tree(seed=42)
    Ro, So, _ = t.tables()
    t.mccfr_batched(Ro, So, 5, 0, 5, 256)
    R, S, _ = ctx.tables_get()
    np.testing.assert_allclose(R, Ro, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(Ro).max()))
    np.testing.assert_allclose(S, So, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(So).max()))


def test_mccfr_narrow_workgroups_give_the_same_deltas(ctx, sl, oracle):
    """Deals with very many infosets leave room for fewer than 16 wavefronts per traversal workgroup (narrower than the 960 threads
    that stage the lane table in one step; ONE wavefront at the maximum of 1653 infosets).  The test hook scopa_debug_lds_limit makes
    the seed-42 deal run with 8, 4, 2 and 1 wavefronts per workgroup: visit counts exact, deltas as the 16-wavefront launch's and the
    oracle's."""
    t = oracle.Tree(seed=42)
    ctx.set_deal(sl.deal_py_seed(42))
    R, S, L = t.tables()
    t.cfr_exact(R, S, L, 3)
    ctx.tables_set(regret=R)
    ctx.mccfr_seed(0x5C09A)
    dR, dS, dv, tv = t.mccfr_batched_delta(R, 0x5C09A, 4, 100, 3000)
    try:
        for limit in (0, 112 * 1024, 98 * 1024, 90 * 1024, 86 * 1024):   # 16, 8, 4, 2, 1 wavefronts per workgroup at 738 infosets (82 098 + 3 648 per wavefront bytes)
            ctx.debug_lds_limit(limit)
            c0 = ctx.counters()
            ctx.mccfr_delta_set(np.zeros((t.n_infosets, 5)))
            ctx.mccfr_traverse(4, 100, 3000)
            d = ctx.mccfr_delta_get()
            c1 = ctx.counters()
            assert (c1[0] - c0[0], c1[1] - c0[1]) == (dv, tv) == (463 * 3000, 240 * 3000), limit
            assert np.array_equal(d[:, 4], np.rint(dS.sum(1))), limit
            np.testing.assert_allclose(d[:, :4], dR, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(dR).max()))
    finally:
        ctx.debug_lds_limit(0)


def test_two_contexts_are_independent(sl, oracle):
    """include/scopa.h: distinct contexts are independent.  Every entry point whose kernel needs more than the default 64 KB of
    dynamic LDS (the cap is raised per context = per device, not once per process) runs on a first AND on a second context."""
    try:
        a, b = sl.Context(0), sl.Context(0)
    except sl.ScopaError as e:
        if e.status == sl.SCOPA_ENODEV:
            pytest.skip("no GPU on this box")
        raise
    t = oracle.Tree(seed=42)
    try:
        out = []
        for c in (a, b):
            c.set_deal(sl.deal_py_seed(42))
            c.cfr_exact_iterate(2)                                   # k_cfr_exact
            Rc, Sc, Lc = c.tables_get()
            e1 = c.exploitability()["exploitability"]                # k_exploitability
            c.tables_reset()
            c.cfr_sync_iterate(2)                                    # k_cfr_sync
            c.tables_reset()
            u = np.random.RandomState(0).random_sample(463 * 2)
            assert c.mccfr_replay(2, u) == 463 * 2                   # k_mccfr_replay
            c.tables_reset()
            c.mccfr_seed(3)
            c.mccfr_iterate(96, 2)                                   # k_mccfr_traverse
            Rm, Sm, _ = c.tables_get()
            out.append((Rc, Sc, Lc, e1, Rm, Sm))
        Ro, So, Lo = t.tables()
        t.cfr_exact(Ro, So, Lo, 2)
        for Rc, Sc, Lc, e1, Rm, Sm in out:
            assert np.array_equal(Rc, Ro) and np.array_equal(Sc, So) and np.array_equal(Lc, Lo)
            assert e1 == t.exploitability(t.average_policy(So))[0]
        np.testing.assert_allclose(out[0][4], out[1][4], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(out[0][5], out[1][5], rtol=1e-12, atol=1e-12)
    finally:
        a.close()
        b.close()


def test_mccfr_full_size_properties(ctx, sl):
    """BASELINE configs[1] size (4096 traversals per traverser): size-independent invariants."""
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.mccfr_seed(1)
    ctx.mccfr_traverse(0, 0, 4096)
    d = ctx.mccfr_delta_get()
    assert d[:, 4].sum() == 172 * 4096 and d[0, 4] == 4096      # the root infoset is visited once per traversal
    assert ctx.counters() == (463 * 4096, 240 * 4096)
    ctx.mccfr_apply()
    R, S, _ = ctx.tables_get()
    n = ctx.tree_export()["infoset_nlegal"]
    rows = S.sum(1)
    np.testing.assert_allclose(rows, d[:, 4], rtol=1e-12)        # each visit adds a probability vector
    assert all((S[i, n[i]:] == 0).all() and (R[i, n[i]:] == 0).all() for i in range(len(n)))
    assert not ctx.mccfr_delta_get().any()                       # apply clears the delta buffer


def test_profiling_counts_launches(ctx, sl):
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.prof_enable(True)
    ctx.mccfr_iterate(512, 5)
    n, ms = ctx.prof_read()
    assert n == 5 and ms > 0
    ctx.prof_enable(False)


@pytest.mark.parametrize("seed,n_inf", [(282, 251), (129, 1144)])
def test_mccfr_batched_extreme_infoset_counts(ctx, sl, oracle, seed, n_inf):
    """Deals with the fewest / most infosets among seeds 0..399: the traversal kernel sizes its workgroup to what the
    LDS holds (1144 infosets leave room for fewer wave scratch areas) and still matches the oracle."""
    t = oracle.Tree(seed=seed)
    assert t.n_infosets == n_inf == ctx.set_deal(sl.deal_py_seed(seed))
    ctx.mccfr_seed(11)
    ctx.mccfr_iterate(300, 3)
    R, S, _ = ctx.tables_get()
    Ro, So, _ = t.tables()
    t.mccfr_batched(Ro, So, 11, 0, 3, 300)
    np.testing.assert_allclose(R, Ro, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(S, So, rtol=1e-10, atol=1e-10)
    assert ctx.counters() == (463 * 300 * 3, 240 * 300 * 3)
    # the exact-semantics solver and exploitability on the same deal
    ctx.tables_reset()
    ctx.cfr_exact_iterate(2)
    Ro, So, Lo = t.tables()
    t.cfr_exact(Ro, So, Lo, 2)
    Rg, Sg, Lg = ctx.tables_get()
    assert np.array_equal(Rg, Ro) and np.array_equal(Sg, So) and np.array_equal(Lg, Lo)
    assert ctx.exploitability()["exploitability"] == t.exploitability(t.average_policy(So))[0]
