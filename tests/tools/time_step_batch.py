#!/usr/bin/env python3
"""k_step_batch alone: a whole game of 2^24 packed MiniScopa games, per-ply kernel time (HIP events) and HBM fraction at 33 B per game-step.
    python tests/tools/time_step_batch.py            (SCOPA_HIP_LIBRARY selects a variant library)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scopa_amd import _lib
from benchmarks import state_engines_bench as seb
stream = torch.cuda.Stream()
ctx = _lib.Context(0, stream=stream.cuda_stream)
best = None
for _ in range(3):
    r = seb.measure(ctx, 1 << 24, 0, 0, 1024, stream)["mini"]
    if best is None or r["whole_game"]["seconds"] < best["whole_game"]["seconds"]:
        best = r
print(os.environ.get("SCOPA_HIP_LIBRARY", "default").split("/")[-1], "whole game frac %.3f" % best["whole_game"]["frac"], "per ply us:", [round(1e6 * t) for t in best["seconds_per_ply"]],
      "frac per ply:", [round(best["games"] * 33 / t / 8e12, 2) for t in best["seconds_per_ply"]])
