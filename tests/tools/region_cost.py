#!/usr/bin/env python3
"""What a SHORT timed region costs beside its iterations (the driver times 20-iteration regions): scopa_mccfr_iterate(4096, K) between two synchronisations for
K = 1 .. 1000, with torch.cuda.synchronize and with the context's own stream synchronisation, and the host time of the enqueue alone (us per iteration).
    python tests/tools/region_cost.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from scopa_amd import _lib
st = torch.cuda.Stream()
ctx = _lib.Context(0, stream=st.cuda_stream); ctx.set_deal(_lib.deal_py_seed(42)); ctx.mccfr_seed(0x5C09A)
ctx.mccfr_iterate(4096, 20000); torch.cuda.synchronize()
def region(K, sync):
    sync(); t0 = time.perf_counter(); ctx.mccfr_iterate(4096, K); sync(); return time.perf_counter() - t0
for name, sync in (("torch.cuda.synchronize", torch.cuda.synchronize), ("ctx.synchronize", ctx.synchronize)):
    for K in (1, 5, 20, 100, 1000):
        ts = sorted(region(K, sync) for _ in range(41))
        print(f"{name:24s} K={K:5d}: median region {ts[20]*1e6:9.1f} us = {ts[20]*1e6/K:7.2f} us/iteration   min {ts[0]*1e6:9.1f}")
# host time of the enqueue alone
t0 = time.perf_counter(); ctx.mccfr_iterate(4096, 1000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue of 1000 iterations returns after {(t1-t0)*1e6:.0f} us; device done after {(t2-t0)*1e6:.0f} us")
# host cost of the enqueue alone at queue depths that cannot fill the ring
for K in (10, 20, 50, 100, 200):
    ts = []
    for _ in range(21):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.mccfr_iterate(4096, K); t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append((t1 - t0) / K)
    ts.sort()
    print(f"enqueue K={K:4d}: host returns after {ts[10]*1e6:6.2f} us per iteration (min {ts[0]*1e6:.2f}, max {ts[-1]*1e6:.2f})")
