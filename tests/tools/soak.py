#!/usr/bin/env python3
"""One long run of the throughput path (a soak, run once per round on the GPU box): 10^6 batched-MCCFR iterations at B = 4096 in
chunks, exact visit counters after every chunk, finite tables, exploitability of the average strategy along the way.
    python tests/tools/soak.py [iterations]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scopa_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = _lib.Context(0); ctx.set_deal(_lib.deal_py_seed(42)); ctx.mccfr_seed(0x5C09A)
done, pts, t0 = 0, [], time.perf_counter()
for chunk in [1000, 9000, 90000] + [100000] * ((N - 100000) // 100000):
    ctx.mccfr_iterate(4096, chunk); done += chunk
    d, t = ctx.counters()
    assert (d, t) == (463 * 4096 * done, 240 * 4096 * done), (d, t, done)
    R, S, _ = ctx.tables_get()
    assert np.isfinite(R).all() and np.isfinite(S).all()
    # every traverser visit adds a probability vector to strategy_sum: a lost or doubled visit count (wavefronts take pairs from a counter and
    # pre-flush the delta table concurrently) would show here
    assert abs(S.sum() - 172.0 * 4096 * done) <= 1e-9 * 172.0 * 4096 * done, (S.sum(), 172.0 * 4096 * done)
    pts.append({"iterations": done, "exploitability": float(ctx.exploitability()["exploitability"]), "seconds": time.perf_counter() - t0})
    print(pts[-1], file=sys.stderr, flush=True)
print(json.dumps({"batch": 4096, "points": pts, "visits": 463 * 4096 * done, "visits_per_s_incl_host_checks": 463 * 4096 * done / (time.perf_counter() - t0)}))
