#!/usr/bin/env python3
"""Where one wavefront of k_sdcfr_traverse spends its cycles (development build: library built with -DSCOPA_WALK_STAMPS, given by
SCOPA_HIP_LIBRARY): shader-clock stamps of workgroup 0 / wavefront 0 at the stage boundaries, per traversal, both traversers.
    python tests/tools/sdcfr_stamps.py [batch] [launches]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from scopa_amd.algorithms.deep_cfr.deep_cfr import DeepCFR
from scopa_amd.envs import load_game
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = DeepCFR(load_game("mini_scopa"), device="cuda:0", batch=B)
d._engine.ctx.sdcfr_mode(1)   # the stamps live in the forward-per-visit kernel (k_sdcfr_traverse)
if os.environ.get("SCOPA_SDCFR_T") or os.environ.get("SCOPA_SDCFR_W"):
    d._engine.ctx.sdcfr_tuning(int(os.environ.get("SCOPA_SDCFR_T", "0")), int(os.environ.get("SCOPA_SDCFR_W", "0")))
lib = ctypes.CDLL(os.environ["SCOPA_HIP_LIBRARY"])
names = ["frontier info", "layer 1", "layer 2", "layer 3 + policy", "expand / sample", "skipped plies", "leaves + backward", "take next"]
for p in (0, 1):
    d._traverse_batch(p, B); torch.cuda.synchronize()
    out = np.zeros(16, np.uint64)
    lib.scopa_debug_sdcfr_stamps(out.ctypes.data_as(ctypes.c_void_p), 1)
    for _ in range(N):
        d._traverse_batch(p, B)
    torch.cuda.synchronize()
    lib.scopa_debug_sdcfr_stamps(out.ctypes.data_as(ctypes.c_void_p), 1)
    n = float(out[15]); tot = float(out[:8].sum()) / n
    print(f"traverser {p}, B={B}: {int(n)} tasks (4 traversals each unless SCOPA_SDCFR_T=2) of wave 0; {tot:.0f} clocks per task")
    print(f"  shader clock while a task runs: {float(out[13]) / max(float(out[14]), 1.0) * 100.0:.0f} MHz ({float(out[13]) / n:.0f} clocks per task incl. stamping)")
    for k, nm in enumerate(names):
        print(f"  {nm:18s} {float(out[k]) / n:9.0f}  {100 * float(out[k]) / n / tot:5.1f} %")
