#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of bench --workload sdcfr at BATCH: per-kernel durations of the traversal launches
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for B in ${BATCHES:-4096 32768}; do
OUT="$ROOT/gpurun_out/statsq_$B"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 "$ROOT/bench.py" --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch $B > "$OUT/line.json" 2> "$OUT/err.txt" || exit 1
rm -f "$OUT"/*kernel_trace.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/s_kernel_stats.csv")):
    if 'sdcfr' in r['Name']: print($B, r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
done
