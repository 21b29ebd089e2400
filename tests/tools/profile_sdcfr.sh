#!/usr/bin/env bash
# Profiler passes of the SDCFR traversal kernels (run ON THE GPU BOX through gpurun; outputs under gpurun_out/prof_sdcfr/).
#     gpurun --timeout 900 -- 'BATCH=4096 bash tests/tools/profile_sdcfr.sh'
# visit/: the forward-per-visit kernel k_sdcfr_traverse (scopa_sdcfr_mode 1 in the timed region: SCOPA_SDCFR_MODE=1): three SQ passes + stats
# table/: the default form, k_sdcfr_policy + k_sdcfr_walk: two SQ passes, FETCH_SIZE / WRITE_SIZE passes, stats
# Counters are collected in runs of their own (never with a trace domain other than --kernel-trace); rocprofv3 gets the program itself.
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_sdcfr"
rm -rf "$OUT"; mkdir -p "$OUT/visit" "$OUT/table"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
ARGS="--workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch ${BATCH:-4096}"
keep() { { head -1 "$1/p_counter_collection.csv"; grep k_sdcfr_ "$1/p_counter_collection.csv"; } > "$1/sdcfr_counters.csv"; rm -f "$1/p_counter_collection.csv" "$1"/*kernel_trace.csv; }   # the PyTorch optimiser kernels fill the collection (50 MB per pass)
SQ_A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
SQ_B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"
SQ_C="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_BRANCH"
export SCOPA_SDCFR_MODE=1
for P in a b c; do
  case $P in a) C="$SQ_A";; b) C="$SQ_B";; c) C="$SQ_C";; esac
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/visit/$P" -o p -- python3 "$B" $ARGS > "$OUT/visit/$P.json" 2> "$OUT/visit/$P.err" || exit 1
  keep "$OUT/visit/$P"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/visit/stats" -o s -- python3 "$B" --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch ${BATCH:-4096} > "$OUT/visit/stats.json" 2> "$OUT/visit/stats.err" || exit 1
rm -f "$OUT"/visit/stats/*kernel_trace.csv
export SCOPA_SDCFR_MODE=0
for P in a b; do
  case $P in a) C="$SQ_A";; b) C="$SQ_B";; esac
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/table/$P" -o p -- python3 "$B" $ARGS > "$OUT/table/$P.json" 2> "$OUT/table/$P.err" || exit 1
  keep "$OUT/table/$P"
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/table/pmc_$C" -o p -- python3 "$B" $ARGS > "$OUT/table/pmc_$C.json" 2> "$OUT/table/pmc_$C.err" || exit 1
  keep "$OUT/table/pmc_$C"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/table/stats" -o s -- python3 "$B" --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch ${BATCH:-4096} > "$OUT/table/stats.json" 2> "$OUT/table/stats.err" || exit 1
rm -f "$OUT"/table/stats/*kernel_trace.csv
ls -la "$OUT" "$OUT"/*/ | head -40
