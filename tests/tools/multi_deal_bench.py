#!/usr/bin/env python3
"""TEST-SIDE TOOL (times the oracle beside the GPU, hence it lives under tests/).

Many deals at once (one workgroup per deal): deal on device, build trees, exact and synchronous CFR, exploitability.
    python tests/tools/multi_deal_bench.py --deals 4096
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--deals", type=int, default=4096)
    ap.add_argument("--exact-iters", type=int, default=10)
    ap.add_argument("--sync-iters", type=int, default=100)
    ap.add_argument("--lanes-deals", type=int, default=0, help="also time the lane-per-deal exact CFR kernel on this many deals")
    a = ap.parse_args()
    from scopa_amd import _lib
    import oracle as O
    ctx = _lib.Context(0)
    m = _lib.MultiDeal(ctx, a.deals)
    t0 = time.perf_counter(); m.deal_py_seeds(np.arange(a.deals)); ctx.synchronize(); t_deal = time.perf_counter() - t0
    t0 = time.perf_counter(); ninf = m.build(); t_build = time.perf_counter() - t0
    m.cfr_exact_iterate(1)  # warm
    t0 = time.perf_counter(); m.cfr_exact_iterate(a.exact_iters); t_exact = time.perf_counter() - t0
    t0 = time.perf_counter(); m.cfr_sync_iterate(a.sync_iters); t_sync = time.perf_counter() - t0
    t0 = time.perf_counter(); e = m.exploitability(); t_expl = time.perf_counter() - t0
    # CPU oracle, one core, a sample of deals
    k = min(a.deals, 32)
    trees = [O.Tree(seed=s) for s in range(k)]              # tree construction is not part of the solver's time
    c_exact = c_sync = 0.0
    for t in trees:
        R, S, L = t.tables()
        t0 = time.perf_counter(); t.cfr_exact(R, S, L, a.exact_iters * 20); c_exact += (time.perf_counter() - t0) / 20
        R, S, L = t.tables()
        t0 = time.perf_counter(); t.cfr_sync(R, S, a.sync_iters); c_sync += time.perf_counter() - t0
    c_exact /= k; c_sync /= k
    lanes = None
    if a.lanes_deals:
        n = a.lanes_deals
        m.close()
        m = _lib.MultiDeal(ctx, n); m.deal_py_seeds(np.arange(n)); m.build()
        m.cfr_exact_iterate_lanes(1)
        t0 = time.perf_counter(); m.cfr_exact_iterate_lanes(a.exact_iters); t_l = time.perf_counter() - t0
        # algorithmic bytes per deal-iteration: per decision visit one L row read + R row read + L row written (96 B) and, at the
        # traverser's nodes (half of the visits per traversal), R and S written + S read (96 B more); maps 2 B / 1 B per visit
        alg = 3306 * 96 + 1653 * 96 + 3306 * 2 + 1152
        lanes = {"deals": n, "iterations": a.exact_iters, "seconds": t_l, "deal_iterations_per_s": n * a.exact_iters / t_l,
                 "visits_per_s": n * a.exact_iters * 3306 / t_l, "tables_resident_GB": n * 1653 * 96 / 1e9,
                 "algorithmic_GBps": n * a.exact_iters * alg / t_l / 1e9}
    print(json.dumps({
        "lane_per_deal_exact_cfr": lanes,
        "deals": a.deals, "hbm_resident_MB": round(a.deals * (2229 * 16 + 1653 * (2 + 8 + 4 + 96) + 576) / 1e6, 1),
        "infosets_min_mean_max": [int(ninf.min()), float(ninf.mean()), int(ninf.max())],
        "deal_on_device_ms": 1e3 * t_deal, "tree_build_ms": 1e3 * t_build,
        "exact_cfr": {"iterations": a.exact_iters, "seconds": t_exact, "deal_iterations_per_s": a.deals * a.exact_iters / t_exact,
                      "visits_per_s": a.deals * a.exact_iters * 3306 / t_exact, "cpu_oracle_1core_deal_iterations_per_s": a.exact_iters / c_exact},
        "sync_cfr": {"iterations": a.sync_iters, "seconds": t_sync, "deal_iterations_per_s": a.deals * a.sync_iters / t_sync,
                     "visits_per_s": a.deals * a.sync_iters * 1653 / t_sync, "cpu_oracle_1core_deal_iterations_per_s": a.sync_iters / c_sync},
        "exploitability_ms": 1e3 * t_expl, "mean_exploitability_after_sync": float(e[:, 0].mean())}))


if __name__ == "__main__":
    main()
