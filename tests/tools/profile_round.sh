#!/usr/bin/env bash
# Profiler passes behind bench.py's roofline block (run ON THE GPU BOX through gpurun; outputs under gpurun_out/prof/, folded into
# profiles/ by tests/tools/fold_profiles.py in the build container).  Counters are collected in runs of their own, never together
# with a trace domain other than --kernel-trace; rocprofv3 is given the program itself (python3 bench.py), no wrapper.
#     gpurun --timeout 900 -- 'bash tests/tools/profile_round.sh'
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
# 1. kernel durations of the bench command itself (its SDCFR sub-record included: k_sdcfr_traverse is in the same summary)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 "$B" --no-cpu-baseline --steps 1000 --regions 5 > "$OUT/stats_bench.json" 2> "$OUT/stats_bench.err" || exit 1
rm -f "$OUT"/stats/*kernel_trace.csv
# 2. HBM traffic: separate FETCH_SIZE / WRITE_SIZE passes (TCC slots do not hold both), few dispatches
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- python3 "$B" --no-cpu-baseline --no-sdcfr --no-subrecords --steps 40 --warmup 10 --regions 3 --pre-phase-s 0 > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" || exit 1
  rm -f "$OUT"/pmc_$C/*kernel_trace.csv
done
# 3. SQ counters of the traversal kernel, two passes of 8, at the headline batch and at a batch that fills the wavefronts' loops
for BATCH in 4096 65536; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq_a_$BATCH" -o p -- python3 "$B" --no-cpu-baseline --no-sdcfr --no-subrecords --batch $BATCH --steps 20 --warmup 5 --regions 3 --pre-phase-s 0 > "$OUT/sq_a_$BATCH.json" 2> "$OUT/sq_a_$BATCH.err" || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/sq_b_$BATCH" -o p -- python3 "$B" --no-cpu-baseline --no-sdcfr --no-subrecords --batch $BATCH --steps 20 --warmup 5 --regions 3 --pre-phase-s 0 > "$OUT/sq_b_$BATCH.json" 2> "$OUT/sq_b_$BATCH.err" || exit 1
  rm -f "$OUT"/sq_?_$BATCH/*kernel_trace.csv
done
ls "$OUT"
