import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from scopa_amd import _lib
import oracle as O
ctx = _lib.Context(0); ctx.set_deal(_lib.deal_py_seed(42))
res = {}
for seq in (True, False):
    ctx.tables_reset(); ctx.cfr_exact_mode(seq); ctx.cfr_exact_iterate(20)
    t0 = time.perf_counter(); ctx.cfr_exact_iterate(1000); dt = time.perf_counter() - t0
    res["sequential_walk" if seq else "scheduled"] = dt
    R, S, L = ctx.tables_get()
    res["hash_" + ("seq" if seq else "sched")] = hash(R.tobytes() + S.tobytes())
t = O.Tree(seed=42); R, S, L = t.tables(); t.cfr_exact(R, S, L, 20)
t0 = time.perf_counter(); t.cfr_exact(R, S, L, 1000); res["c_oracle_1core"] = time.perf_counter() - t0
res["note"] = "seconds per 1000 iterations of vanilla CFR on the seed-42 deal (BASELINE configs[0]); reference Python: ~276 s"
print(json.dumps(res))
