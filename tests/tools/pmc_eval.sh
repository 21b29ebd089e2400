#!/usr/bin/env bash
# SQ and TCC counter passes of the device evaluator's step kernel (benchmarks/eval_bench.py: k_eval_tabular_step, both sampling forms), full-size launches only:
# instructions per wavefront, VALU share of the busy time, HBM bytes per episode-ply -> gpurun_out/pmc_eval/summary.json (kept as profiles/pmc_eval.json)
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_eval"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
N="${EPISODES:-4194304}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/a" -o p -- python3 "$ROOT/benchmarks/eval_bench.py" --episodes "$N" > "$OUT/a.json" 2> "$OUT/a.err" || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -o p -- python3 "$ROOT/benchmarks/eval_bench.py" --episodes "$N" > "$OUT/f.json" 2> "$OUT/f.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -o p -- python3 "$ROOT/benchmarks/eval_bench.py" --episodes "$N" > "$OUT/w.json" 2> "$OUT/w.err" || exit 1
for P in a f w; do { head -1 "$OUT/$P/p_counter_collection.csv"; grep "k_eval_tabular_step" "$OUT/$P/p_counter_collection.csv"; } > "$OUT/$P/counters.csv"; rm -f "$OUT/$P/p_counter_collection.csv" "$OUT/$P"/*kernel_trace.csv; done
python3 - <<PY
import csv, collections, json
n = $N
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for P in "afw":
    rows = [(("thresholds" if "<true>" in r["Kernel_Name"] else "divisions"), int(r["Grid_Size"]), r) for r in csv.DictReader(open("$OUT/%s/counters.csv" % P))]
    for k, g, r in rows:
        if g < n: continue                                   # evaluate_agent_device's warm-up
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    m = {a: sum(v) / len(v) for a, v in c.items()}
    out[k] = {"kernel": "k_eval_tabular_step<%s>" % ("true" if k == "thresholds" else "false"), "episodes": n, "launches": len(c["SQ_WAVES"]),
              "valu_instr_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"], "salu_instr_per_wave": m["SQ_INSTS_SALU"] / m["SQ_WAVES"],
              "vmem_rd_per_wave": m["SQ_INSTS_VMEM_RD"] / m["SQ_WAVES"], "vmem_wr_per_wave": m["SQ_INSTS_VMEM_WR"] / m["SQ_WAVES"],
              "valu_busy_share": 4.0 * m["SQ_ACTIVE_INST_VALU"] / 1024.0 / (m["SQ_BUSY_CYCLES"] / 32.0),
              "fetch_bytes_per_episode_ply_raw_x2": 2.0 * 1024.0 * m["FETCH_SIZE"] / n, "write_bytes_per_episode_ply": 1024.0 * m["WRITE_SIZE"] / n, "per_launch_mean": m}
    print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in out[k].items() if a != "per_launch_mean"})
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
