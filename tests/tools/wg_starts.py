#!/usr/bin/env python3
"""When do the workgroups of a k_mccfr_traverse launch start, by blockIdx?  Sampled launches carry per-workgroup stamps on the 100 MHz device clock
(start | prologue done | walks done | end); this prints, averaged over the last 32 sampled launches, the start offset behind the launch's first workgroup
and the end offset, for every 8th workgroup, and per residue blockIdx % 8 (the XCD a workgroup lands on, if dispatch is round-robin).
    python tests/tools/wg_starts.py [batch] [idle]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scopa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.Context(0); ctx.set_deal(_lib.deal_py_seed(42)); ctx.mccfr_seed(0x5C09A)
ctx.mccfr_iterate(B, 2000); ctx.synchronize()
IDLE = len(sys.argv) > 2 and sys.argv[2] == "idle"      # every sampled launch is the first after a stream synchronisation (an idle GPU) instead of one in a running loop
if IDLE:
    ctx.prof_enable(1)
    for _ in range(40):
        ctx.mccfr_iterate(B, 1); ctx.synchronize()
else:
    ctx.prof_enable(7)
    ctx.mccfr_iterate(B, 7 * 40); ctx.synchronize()
ctx.prof_read()
L = _lib.lib()
n, G = 32, 512
out = np.zeros((n, G, 4), np.uint64)
L.scopa_debug_clock_dump.restype = C.c_int32
grid = L.scopa_debug_clock_dump(ctx._h, out.ctypes.data_as(C.c_void_p), n, G)
assert grid > 0, grid
t = out[:, :grid, :].astype(np.int64)
t0 = t[:, :, 0].min(axis=1, keepdims=True)
start = (t[:, :, 0] - t0).mean(axis=0) * 0.01
end = (t[:, :, 3] - t0).mean(axis=0) * 0.01
dur = ((t[:, :, 3] - t[:, :, 0]).mean(axis=0)) * 0.01
print(f"grid {grid}; kernel (first start -> last end) {((t[:, :, 3].max(axis=1) - t0[:, 0]).mean()) * 0.01:.2f} us; mean start {start.mean():.2f}, max {start.max():.2f}; mean duration {dur.mean():.2f}")
print("blockIdx: start / duration / end (us)")
for b in range(0, grid, 8):
    print(f"  {b:4d}: {start[b]:5.2f} {dur[b]:5.2f} {end[b]:5.2f}")
print("by blockIdx % 8: mean start, mean end")
for x in range(8):
    print(f"  {x}: {start[x::8].mean():5.2f} {end[x::8].mean():5.2f}")
order = np.argsort(start)
print("correlation of start with blockIdx:", float(np.corrcoef(np.arange(grid), start)[0, 1]), " with blockIdx // 8:", float(np.corrcoef(np.arange(grid) // 8, start)[0, 1]))
print("workgroups in start order (first 32):", order[:32].tolist())
print("ends: min %.2f  median %.2f  max %.2f;  end of the 32 earliest starters: %.2f, of the 32 latest: %.2f" % (end.min(), np.median(end), end.max(), end[order[:32]].mean(), end[order[-32:]].mean()))
