#!/usr/bin/env bash
# SQ and TCC counter passes of the state-engine step kernels (benchmarks/state_engines_bench.py), per kernel: instructions per game-step, unit busy shares, HBM bytes per game-step
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_engines"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
ARGS="--mini ${MINI:-16777216} --team ${TEAM:-16777216} --full ${FULL:-8388608}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/a" -o p -- python3 "$ROOT/benchmarks/state_engines_bench.py" $ARGS > "$OUT/a.json" 2> "$OUT/a.err" || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -o p -- python3 "$ROOT/benchmarks/state_engines_bench.py" $ARGS > "$OUT/f.json" 2> "$OUT/f.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -o p -- python3 "$ROOT/benchmarks/state_engines_bench.py" $ARGS > "$OUT/w.json" 2> "$OUT/w.err" || exit 1
for P in a f w; do { head -1 "$OUT/$P/p_counter_collection.csv"; grep "step_batch" "$OUT/$P/p_counter_collection.csv"; } > "$OUT/$P/counters.csv"; rm -f "$OUT/$P/p_counter_collection.csv" "$OUT/$P"/*kernel_trace.csv; done
python3 - <<PY
import csv, collections, json
games = {"k_step_batch": ${MINI:-16777216}, "k_team_step_batch": ${TEAM:-16777216}, "k_full_step_batch": ${FULL:-8388608}}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for P in "afw":
    rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]), r) for r in csv.DictReader(open("$OUT/%s/counters.csv" % P))]
    full = collections.defaultdict(int)
    for k, g, r in rows: full[k] = max(full[k], g)
    for k, g, r in rows:
        if g < full[k]: continue                             # the bench's warm-up launch of each kernel (a pool's worth of games)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}; n = games[k]
    ws = m["SQ_WAVES"] * max(1.0, n / (64.0 * m["SQ_WAVES"]))      # wavefront-steps per launch (k_step_batch strides over the games: 32 steps per wavefront here)
    out[k] = {"games": n, "launches": len(c["SQ_WAVES"]), "steps_per_wave": ws / m["SQ_WAVES"], "valu_instr_per_wave_step": m["SQ_INSTS_VALU"] / ws, "salu_instr_per_wave_step": m["SQ_INSTS_SALU"] / ws,
              "vmem_rd_per_wave_step": m["SQ_INSTS_VMEM_RD"] / ws, "vmem_wr_per_wave_step": m["SQ_INSTS_VMEM_WR"] / ws,
              "valu_busy_share": 4.0 * m["SQ_ACTIVE_INST_VALU"] / 1024.0 / (m["SQ_BUSY_CYCLES"] / 32.0),
              "fetch_bytes_per_game_step_raw_x2": 2.0 * 1024.0 * m["FETCH_SIZE"] / n, "write_bytes_per_game_step": 1024.0 * m["WRITE_SIZE"] / n, "per_launch_mean": m}
    print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in out[k].items() if a != "per_launch_mean"})
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
