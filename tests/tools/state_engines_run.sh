#!/usr/bin/env bash
# the three batched state engines (benchmarks/state_engines_bench.py): steps/s and achieved HBM GB/s per engine
cd $GRAFT_REPO_ROOT; [ -n "$LIB" ] && export SCOPA_HIP_LIBRARY="$PWD/$LIB"
timeout -k 10 600 python benchmarks/state_engines_bench.py > gpurun_out/state_engines${TAG}.json 2> gpurun_out/state_engines${TAG}.err || { tail -5 gpurun_out/state_engines${TAG}.err; exit 1; }
python - <<PY
import json; d=json.load(open("gpurun_out/state_engines${TAG}.json"))
for k in ("mini","team","full"):
    r=d[k]; print("${TAG}", k, r["games"], r["plies"], "best %.3g steps/s %.0f GB/s frac %.3f"%(r["best_ply"]["game_steps_per_s"], r["best_ply"]["achieved_GBps"], r["best_ply"]["frac"]), "median frac %.3f"%r["median_ply"]["frac"], "game frac %.3f"%r["whole_game"]["frac"], [round(1e3*t,2) for t in r["seconds_per_ply"]][:16])
PY
