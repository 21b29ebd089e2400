#!/usr/bin/env python3
"""Time scopa_mccfr_iterate without any checking (for kernel experiments built with -DSCOPA_EXP_*; set SCOPA_HIP_LIBRARY to the variant).
    python tests/tools/time_iter.py [batch] [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scopa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = _lib.Context(0); ctx.set_deal(_lib.deal_py_seed(42)); ctx.mccfr_seed(0x5C09A)
ctx.mccfr_iterate(B, max(50, 20000000 // B // 20)); ctx.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); ctx.mccfr_iterate(B, N); ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / N)
print(f"{os.environ.get('SCOPA_HIP_LIBRARY', 'default')}: B={B} {best * 1e6:.2f} us/iteration")
