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
    from benchmarks.subrecords import many_deals          # the same measurement bench.py's `many_deals` sub-record reports (HIP events on the kernel's stream)
    print(json.dumps(many_deals(0, a.deals, a.iters, a.reps)))


if __name__ == "__main__":
    main()
