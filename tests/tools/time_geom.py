#!/usr/bin/env python3
"""scopa_mccfr_iterate at other workgroup widths (the test hook scopa_debug_lds_limit makes the launch code pick fewer wavefronts per workgroup):
us per iteration at 16 / 8 / 4 wavefronts per workgroup for the seed-42 deal.    python tests/tools/time_geom.py [batch] [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scopa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.Context(0); ctx.set_deal(_lib.deal_py_seed(42)); ctx.mccfr_seed(0x5C09A)
ctx.mccfr_iterate(B, 500); ctx.synchronize()
for rnd in range(3):
    for waves, limit in ((16, 0), (8, 112 * 1024), (4, 98 * 1024)):
        ctx.debug_lds_limit(limit)
        ctx.mccfr_iterate(B, 200); ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); ctx.mccfr_iterate(B, N); ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / N)
        print(f"B={B} {waves} wavefronts per workgroup: {best * 1e6:.2f} us/iteration", flush=True)
ctx.debug_lds_limit(0)
