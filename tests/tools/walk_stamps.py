#!/usr/bin/env python3
"""Where one wavefront's walk_pair spends its cycles (development build: SCOPA_EXTRA_HIPCC_FLAGS=-DSCOPA_WALK_STAMPS, library given by
SCOPA_HIP_LIBRARY).  Shader-clock stamps of workgroup 0 / wavefront 0 at the stage boundaries, averaged over its pairs.
    python tests/tools/walk_stamps.py [batch] [iterations]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scopa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx = _lib.Context(0); ctx.set_deal(_lib.deal_py_seed(42)); ctx.mccfr_seed(0x5C09A)
ctx.mccfr_iterate(B, 200); ctx.synchronize()
lib = ctypes.CDLL(os.environ["SCOPA_HIP_LIBRARY"])
out = np.zeros(16, np.uint64)
lib.scopa_debug_walk_stamps(out.ctypes.data_as(ctypes.c_void_p), 1)
ctx.mccfr_iterate(B, N); ctx.synchronize()
lib.scopa_debug_walk_stamps(out.ctypes.data_as(ctypes.c_void_p), 0)
n = float(out[15]); names = ["draws", "ply0", "ply1", "ply2", "ply3", "ply4", "ply5", "leaves", "update"]
tot = float(out[:9].sum()) / n
print(f"B={B}: {int(n)} pairs of wave 0; {tot:.0f} clocks per pair; shader clock = {float(out[:9].sum()) / (float(out[14]) * 10.0):.3f} GHz (clock64 / 100 MHz wall clock)")
for k, nm in enumerate(names):
    print(f"  {nm:7s} {float(out[k]) / n:8.1f}  {100 * float(out[k]) / n / tot:5.1f} %")
wc = np.zeros(48, np.uint64)
lib.scopa_debug_wave_clocks(wc.ctypes.data_as(ctypes.c_void_p))
its = N + 200
for g, nm in ((0, "workgroup 0"), (16, "workgroup 100")):
    print(f"  pair loop per wavefront, {nm} (clocks per launch):", " ".join(f"{float(x) / its:.0f}" for x in wc[g:g + 16]))
print("  HW_ID of workgroup 0's wavefronts (wave slot bits 3:0, SIMD bits 5:4, CU bits 11:8):", " ".join(f"simd{(int(x) >> 4) & 3}" for x in wc[32:48]))
ph = np.zeros(64, np.uint64)
lib.scopa_debug_wave_phases(ph.ctypes.data_as(ctypes.c_void_p))
for k, nm in enumerate(("kernel entry -> prologue barrier", "barrier -> first pair (lane table unpack, lane statics)", "pair loop", "pair loop end -> kernel end (barrier, atomics)")):
    print(f"  workgroup 0, {nm}:", " ".join(f"{float(x) / its:.0f}" for x in ph[16 * k:16 * k + 16]))
