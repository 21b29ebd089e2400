#!/usr/bin/env bash
# SQ counter passes of k_sdcfr_traverse (run ON THE GPU BOX through gpurun; outputs under gpurun_out/prof_sdcfr/).
#     gpurun --timeout 600 -- 'bash tests/tools/profile_sdcfr.sh'
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_sdcfr"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$OUT/a" -o p -- python3 "$B" --workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch ${BATCH:-4096} > "$OUT/a.json" 2> "$OUT/a.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/b" -o p -- python3 "$B" --workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch ${BATCH:-4096} > "$OUT/b.json" 2> "$OUT/b.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_BRANCH --output-format csv -d "$OUT/c" -o p -- python3 "$B" --workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch ${BATCH:-4096} > "$OUT/c.json" 2> "$OUT/c.err" || exit 1
# HBM traffic of the kernel: separate FETCH_SIZE / WRITE_SIZE passes
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- python3 "$B" --workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch ${BATCH:-4096} > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" || exit 1
  { head -1 "$OUT/pmc_$C/p_counter_collection.csv"; grep k_sdcfr_traverse "$OUT/pmc_$C/p_counter_collection.csv"; } > "$OUT/pmc_$C/sdcfr_counters.csv"; rm -f "$OUT/pmc_$C/p_counter_collection.csv" "$OUT"/pmc_$C/*kernel_trace.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 "$B" --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch ${BATCH:-4096} > "$OUT/stats.json" 2> "$OUT/stats.err" || exit 1
rm -f "$OUT"/?/*kernel_trace.csv "$OUT"/stats/*kernel_trace.csv
# the PyTorch optimiser kernels fill the collection (50 MB per pass): keep the traversal kernel's rows only
for d in a b c; do { head -1 "$OUT/$d/p_counter_collection.csv"; grep k_sdcfr_traverse "$OUT/$d/p_counter_collection.csv"; } > "$OUT/$d/sdcfr_counters.csv"; rm -f "$OUT/$d/p_counter_collection.csv"; done
ls -la "$OUT" "$OUT"/*/
