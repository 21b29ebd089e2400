#!/usr/bin/env python3
"""Lane-per-deal exact CFR (scopa_multi_cfr_exact_iterate_lanes): many independent solves of the reference's vanilla CFR
(src/algorithms/vanilla_cfr.py:56-110), one deal per lane, tables resident in HBM.

    python benchmarks/multi_deal_lanes_bench.py --deals 131072 --iters 5
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# algorithmic bytes: SURVEY.md 8(d) prices vanilla CFR at 192 B per decision visit (state 32 B, regret row 32 B, local_strategy
# read + write 64 B, and at the traverser's half of the visits regret/strategy read-modify-write 128 B); 3306 visits per iteration.
ALG_BYTES = 3306 * 192


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--deals", type=int, default=131072)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from scopa_amd import _lib
    ctx = _lib.Context(0)
    m = _lib.MultiDeal(ctx, a.deals)
    m.deal_py_seeds(np.arange(a.deals))
    t0 = time.perf_counter(); m.build(); t_build = time.perf_counter() - t0
    m.cfr_exact_iterate_lanes(1)
    best = 1e30
    for _ in range(a.reps):
        t0 = time.perf_counter(); m.cfr_exact_iterate_lanes(a.iters); best = min(best, time.perf_counter() - t0)
    e = m.exploitability()
    print(json.dumps({"workload": "vanilla CFR, %d deals x %d iterations, one deal per lane" % (a.deals, a.iters), "deals": a.deals,
                      "iterations": a.iters, "seconds": best, "deal_iterations_per_s": a.deals * a.iters / best,
                      "visits_per_s": a.deals * a.iters * 3306 / best, "row_image_resident_GB": a.deals * 1653 * 64 / 1e9,
                      "tree_build_s": t_build, "algorithmic_bytes_per_deal_iteration": ALG_BYTES,
                      "algorithmic_GBps": a.deals * a.iters * ALG_BYTES / best / 1e9, "mean_exploitability": float(e[:, 0].mean())}))
    m.close()


if __name__ == "__main__":
    main()
