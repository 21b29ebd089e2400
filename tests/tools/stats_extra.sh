#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the round's other kernels: the hand-written optimiser step (tests/tools/sdcfr_train_breakdown.py), the three batched step
# kernels (benchmarks/state_engines_bench.py) and the device evaluator (benchmarks/eval_bench.py): per-kernel durations under gpurun_out/stats_extra/
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
O="$ROOT/gpurun_out/stats_extra"; rm -rf "$O"; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train" -o s -- python3 "$ROOT/tests/tools/sdcfr_train_breakdown.py" > "$O/train.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/engines" -o s -- python3 "$ROOT/benchmarks/state_engines_bench.py" --no-warm-up --mini 16777216 --team 16777216 --full 8388608 > "$O/engines.json" 2> "$O/engines.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/eval" -o s -- python3 "$ROOT/benchmarks/eval_bench.py" --episodes 4194304 > "$O/eval.json" 2> "$O/eval.err" || exit 1
rm -f "$O"/*/*kernel_trace.csv
python3 - <<PY
import csv, glob
for part, keys in (("train", ("k_sdcfr_train",)), ("engines", ("step_batch",)), ("eval", ("k_eval",))):
    f = glob.glob("$O/%s/*kernel_stats.csv" % part)[0]
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in keys):
            print(part, r["Name"][:44], r["Calls"], "avg %.1f us min %.1f max %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
