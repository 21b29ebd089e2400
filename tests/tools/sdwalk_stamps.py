#!/usr/bin/env python3
"""Where one wavefront of k_sdcfr_walk spends its cycles (development build: library built with -DSCOPA_WALK_STAMPS, given by
SCOPA_HIP_LIBRARY): shader-clock stamps of workgroup 0 / wavefront 0 at the stage boundaries, per task, both traversers.
    python tests/tools/sdwalk_stamps.py [batch] [launches]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from scopa_amd.algorithms.deep_cfr.deep_cfr import DeepCFR
from scopa_amd.envs import load_game
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = DeepCFR(load_game("mini_scopa"), device="cuda:0", batch=B)
if os.environ.get("SCOPA_SDCFR_T"):
    d._engine.ctx.sdcfr_tuning(int(os.environ.get("SCOPA_SDCFR_T", "0")), 0)
lib = ctypes.CDLL(os.environ["SCOPA_HIP_LIBRARY"])
names = ["staging (per launch)", "draws", "forward", "feature/mask sweep", "leaves + backward", "take next"]
for p in (0, 1):
    d._traverse_batch(p, B); torch.cuda.synchronize()
    out = np.zeros(16, np.uint64)
    lib.scopa_debug_sdwalk_stamps(out.ctypes.data_as(ctypes.c_void_p), 1)
    for _ in range(N):
        d._traverse_batch(p, B)
    torch.cuda.synchronize()
    lib.scopa_debug_sdwalk_stamps(out.ctypes.data_as(ctypes.c_void_p), 1)
    n = float(out[15]); tot = float(out[1:8].sum()) / n
    mhz = float(out[13]) / max(float(out[14]), 1.0) * 100.0
    print(f"traverser {p}, B={B}: {int(n)} tasks of wave 0 over {N} launches; {tot:.0f} clocks per task; shader clock {mhz:.0f} MHz")
    print(f"  {names[0]:22s} {float(out[0]) / N:9.0f} clocks per launch")
    for k in range(1, 6):
        print(f"  {names[k]:22s} {float(out[k]) / n:9.0f}  {100 * float(out[k]) / n / tot:5.1f} %")
