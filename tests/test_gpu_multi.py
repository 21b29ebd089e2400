"""Multi-deal mode: n independent deals, one workgroup per deal -- deal on device, tree build, exact and synchronous
CFR, exploitability -- each deal checked against the oracle / the reference fixtures."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_deal_matches_cpython_shuffle(ctx, sl, golden):
    deals = golden.json("deals.json")
    seeds = [int(s) for s in deals]
    m = sl.MultiDeal(ctx, len(seeds))
    m.deal_py_seeds(seeds)
    perms = m.perms()
    for i, s in enumerate(seeds):
        assert list(perms[i]) == deals[str(s)], s
    m.close()


def test_multi_deal_solvers_match_oracle_per_deal(ctx, sl, oracle, golden):
    seeds = [42, 0, 1, 7, 123, 282, 129, 5, 6, 8, 9, 10]
    m = sl.MultiDeal(ctx, len(seeds))
    m.deal_py_seeds(seeds)
    ninf = m.build()
    trees = [oracle.Tree(seed=s) for s in seeds]
    assert list(ninf) == [t.n_infosets for t in trees]
    m.cfr_exact_iterate(3)
    expl = m.exploitability()
    for i, t in enumerate(trees):
        R, S, L = t.tables()
        t.cfr_exact(R, S, L, 3)
        Rg, Sg, Lg, K = m.tables_get(i)
        assert np.array_equal(Rg, R) and np.array_equal(Sg, S) and np.array_equal(Lg, L), seeds[i]
        assert [sl.key_to_string(k) for k in K] == t.infoset_strings
        e, br = t.exploitability(t.average_policy(S))
        assert (expl[i, 0], expl[i, 1], expl[i, 2], expl[i, 3]) == (e, br[0], br[1], t.policy_value(t.average_policy(S)))
    assert m.counters() == (3306 * 3 * len(seeds), 1152 * 3 * len(seeds))
    # deal 0 is the reference's deal: its tables are the reference's (golden fixture), too
    g = golden.npz("vanilla_cfr.npz")
    m2 = sl.MultiDeal(ctx, 3)
    m2.deal_py_seeds([42, 42, 42])
    m2.build()
    m2.cfr_exact_iterate(5)
    for i in range(3):
        R, S, L, _ = m2.tables_get(i)
        assert np.array_equal(R, g["it5_regret"]) and np.array_equal(S, g["it5_strategy"]) and np.array_equal(L, g["it5_local"])
    m.close(); m2.close()


def test_multi_deal_sync_cfr_many_deals(ctx, sl, oracle):
    n = 600                                   # more deals than CUs: the grid wraps
    rng = np.random.RandomState(2)
    perms = np.array([rng.permutation(16) for _ in range(n)], np.uint8)
    m = sl.MultiDeal(ctx, n)
    m.set_perms(perms)
    ninf = m.build()
    m.cfr_sync_iterate(20)
    expl = m.exploitability()
    assert (expl[:, 0] >= 0).all() and (expl[:, 0] < 1.5).all()
    for i in (0, 17, 599):
        t = oracle.Tree(perm=perms[i])
        assert ninf[i] == t.n_infosets
        R, S, _ = t.tables()
        t.cfr_sync(R, S, 20)
        Rg, Sg, _, _ = m.tables_get(i)
        assert np.array_equal(Rg, R) and np.array_equal(Sg, S)
        assert expl[i, 0] == t.exploitability(t.average_policy(S))[0]
    m.close()


def test_multi_deal_persistent_mccfr_matches_oracle(ctx, sl, oracle):
    """One workgroup per deal runs all iterations in-kernel (regret table in LDS): same definition as the single-deal
    batched path, so each deal matches the oracle's og_mccfr_batched and the single-deal API."""
    seeds = [42, 0, 1, 129, 282, 7]
    m = sl.MultiDeal(ctx, len(seeds))
    m.deal_py_seeds(seeds)
    m.build()
    m.mccfr_iterate(batch=48, n_iters=5, seed=321)
    for i, s in enumerate(seeds):
        t = oracle.Tree(seed=s)
        R, S, _ = t.tables()
        t.mccfr_batched(R, S, 321, 0, 5, 48)
        Rg, Sg, _, _ = m.tables_get(i)
        np.testing.assert_allclose(Rg, R, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(Sg, S, rtol=1e-10, atol=1e-10)
    assert m.counters() == (463 * 48 * 5 * len(seeds), 240 * 48 * 5 * len(seeds))
    # continuing is the same as one longer run; and deal 0 equals the single-deal entry point
    m.mccfr_iterate(batch=48, n_iters=3, seed=321)
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.mccfr_seed(321)
    ctx.mccfr_iterate(48, 8)
    R1, S1, _ = ctx.tables_get()
    Rg, Sg, _, _ = m.tables_get(0)
    np.testing.assert_allclose(Rg, R1, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(Sg, S1, rtol=1e-10, atol=1e-10)
    m.close()


def test_lane_per_deal_exact_cfr_is_bit_identical(ctx, sl, oracle, golden):
    """scopa_multi_cfr_exact_iterate_lanes (64 deals per wavefront, tables gathered from HBM) against the workgroup-per-deal
    kernel on every deal, against the oracle on a sample, and against the reference's own tables on the seed-42 deal."""
    n = 200                                   # 3 full wavefronts + a ragged one
    seeds = [42] + list(range(1, n))
    a = sl.MultiDeal(ctx, n); b = sl.MultiDeal(ctx, n)
    a.deal_py_seeds(seeds); b.deal_py_seeds(seeds)
    ninf = a.build(); assert np.array_equal(ninf, b.build())
    a.cfr_exact_iterate_lanes(2); a.cfr_exact_iterate_lanes(3)      # resumable: 2 + 3 == 5
    b.cfr_exact_iterate(5)
    for i in range(n):
        Ra, Sa, La, Ka = a.tables_get(i); Rb, Sb, Lb, Kb = b.tables_get(i)
        assert np.array_equal(Ra, Rb) and np.array_equal(Sa, Sb) and np.array_equal(La, Lb) and np.array_equal(Ka, Kb), seeds[i]
    assert a.counters() == b.counters() == (3306 * 5 * n, 1152 * 5 * n)
    assert np.array_equal(a.exploitability(), b.exploitability())
    g = golden.npz("vanilla_cfr.npz")
    R, S, L, _ = a.tables_get(0)
    assert np.array_equal(R, g["it5_regret"]) and np.array_equal(S, g["it5_strategy"]) and np.array_equal(L, g["it5_local"])
    for i in (1, 63, 64, 199):
        t = oracle.Tree(seed=seeds[i]); R, S, L = t.tables(); t.cfr_exact(R, S, L, 5)
        Rg, Sg, Lg, _ = a.tables_get(i)
        assert np.array_equal(Rg, R) and np.array_equal(Sg, S) and np.array_equal(Lg, L), seeds[i]
    a.close(); b.close()


def test_lane_per_deal_cfr_refuses_tables_another_solver_left_inconsistent(ctx, sl):
    """The lane-per-deal kernel relies on local_strategy == regret_matching(regret_sum) (true for every table the reference's
    own CFR produces); synchronous CFR moves regret_sum without touching local_strategy, so packing must refuse -- loudly, and
    leave the tables usable by the workgroup-per-deal kernel."""
    m = sl.MultiDeal(ctx, 8)
    m.deal_py_seeds(list(range(8)))
    m.build()
    m.cfr_exact_iterate_lanes(1)            # fine on fresh tables
    m.cfr_sync_iterate(3)                   # regret_sum changes, local_strategy does not
    with pytest.raises(sl.ScopaError):
        m.cfr_exact_iterate_lanes(1)
    m.cfr_exact_iterate(1)                  # the literal kernel still runs on them
    m.close()


def test_exact_cfr_on_many_deals_takes_the_lane_form_with_identical_results(ctx, sl, oracle):
    """scopa_multi_cfr_exact_iterate switches to one lane per deal from 8192 deals on; same bits as the oracle either way."""
    n = 8192
    m = sl.MultiDeal(ctx, n)
    m.deal_py_seeds(np.arange(n))
    m.build()
    m.cfr_exact_iterate(2)
    m.cfr_exact_iterate(1)
    assert m.counters() == (3306 * 3 * n, 1152 * 3 * n)
    for i in (0, 4097, 8191):
        t = oracle.Tree(seed=i); R, S, L = t.tables(); t.cfr_exact(R, S, L, 3)
        Rg, Sg, Lg, _ = m.tables_get(i)
        assert np.array_equal(Rg, R) and np.array_equal(Sg, S) and np.array_equal(Lg, L), i
    m.close()
