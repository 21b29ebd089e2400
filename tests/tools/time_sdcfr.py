#!/usr/bin/env python3
"""Time k_sdcfr_traverse alone (for kernel experiments; set SCOPA_HIP_LIBRARY to the variant).
    python tests/tools/time_sdcfr.py [batch] [launches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scopa_amd.algorithms.deep_cfr.deep_cfr import DeepCFR
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
from scopa_amd.envs import load_game
solver = DeepCFR(load_game("mini_scopa"), device="cuda:0", batch=B)
if os.environ.get("SCOPA_SDCFR_MODE"):
    solver._engine.ctx.sdcfr_mode(int(os.environ["SCOPA_SDCFR_MODE"]))   # 1 = a forward pass per visit (k_sdcfr_traverse), 0 = policy table + walks
if os.environ.get("SCOPA_SDCFR_T") or os.environ.get("SCOPA_SDCFR_W"):
    solver._engine.ctx.sdcfr_tuning(int(os.environ.get("SCOPA_SDCFR_T", "0")), int(os.environ.get("SCOPA_SDCFR_W", "0")))   # traversals / wavefronts per task
for p in (0, 1):
    solver._traverse_batch(p, B)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        solver._traverse_batch(0, B); solver._traverse_batch(1, B)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / (2 * N))
print(f"{os.environ.get('SCOPA_HIP_LIBRARY', 'default')} T={os.environ.get('SCOPA_SDCFR_T', 'auto')} W={os.environ.get('SCOPA_SDCFR_W', 'auto')}: B={B} {best * 1e6:.1f} us per traversal launch ({B * 93.5 / best:.3e} visits/s)")
