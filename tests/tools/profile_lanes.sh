#!/usr/bin/env bash
# Round-4 profiler passes for k_cfr_exact_lanes (the kernel SURVEY 8(d)'s 192 B/visit HBM model binds) and the counter calibration for its access
# pattern (benchmarks/micro/row_gather).  Run ON THE GPU BOX through gpurun; outputs under gpurun_out/lanes/, folded into profiles/ by
# tests/tools/fold_lanes.py in the build container.  Counters are collected in runs of their own (--kernel-trace only beside --pmc); rocprofv3 is
# given the program itself, no wrapper.  Any failing step stops the script with a non-zero status.
#     gpurun --timeout 900 -- 'bash tests/tools/profile_lanes.sh'
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out/lanes"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
L="$ROOT/benchmarks/multi_deal_lanes_bench.py"
G="$ROOT/benchmarks/micro/row_gather"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
grep -o "TCC_[A-Z0-9_]*" "$OUT/counters_available.txt" | sort -u > "$OUT/tcc_counters.txt" || true
# 1. unprofiled figures (HIP-event / host timing of the programs themselves)
python3 "$L" --deals 131072 --iters 5 --reps 3 > "$OUT/lanes_bench.json" 2> "$OUT/lanes_bench.err"
"$G" 27 > "$OUT/row_gather.jsonl" 2> "$OUT/row_gather.err"
# 2. kernel durations
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 "$L" --deals 131072 --iters 5 --reps 3 > "$OUT/stats_lanes.json" 2> "$OUT/stats_lanes.err"
rm -f "$OUT"/stats/*kernel_trace.csv
# 3. HBM counters, one pass each (TCC slots do not hold FETCH_SIZE and WRITE_SIZE together), for the kernel and for the calibration patterns
# (the request-SIZE counters are what calibrates FETCH_SIZE = RDREQ x 64 B: bytes = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B)
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ TCC_EA0_WRREQ" "TCC_HIT TCC_MISS" "TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B TCC_EA0_WRREQ_64B" "TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ_DRAM"; do
  T=$(echo $C | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_lanes_$T" -o p -- python3 "$L" --deals 131072 --iters 5 --reps 1 > "$OUT/pmc_lanes_$T.json" 2> "$OUT/pmc_lanes_$T.err" \
    || { echo "pass $T (lanes) failed" >&2; case "$T" in FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ+TCC_EA0_WRREQ) exit 1;; esac; }
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_calib_$T" -o p -- "$G" 27 > "$OUT/pmc_calib_$T.jsonl" 2> "$OUT/pmc_calib_$T.err" \
    || { echo "pass $T (calibration) failed" >&2; case "$T" in FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ+TCC_EA0_WRREQ) exit 1;; esac; }
  rm -f "$OUT"/pmc_*_$T/*kernel_trace.csv
done
# keep what travels back small: only the rows of the kernels of interest
for f in "$OUT"/pmc_*/*counter_collection.csv; do
  { head -1 "$f"; grep -E "k_cfr_exact_lanes|k_stream16|k_gather32|k_gather64|k_scatter64|k_rmw64" "$f" || true; } > "$f.keep"; mv "$f.keep" "$f"
done
ls -la "$OUT"
