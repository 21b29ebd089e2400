#!/usr/bin/env bash
# one SQ pass of k_sdcfr_traverse (matrix pipe busy, wave states) at BATCH (default 32768)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out/pmcq"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/b" -o p -- python3 "$ROOT/bench.py" --workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch ${BATCH:-32768} > "$OUT/b.json" 2> "$OUT/b.err" || exit 1
{ head -1 "$OUT/b/p_counter_collection.csv"; grep k_sdcfr_traverse "$OUT/b/p_counter_collection.csv"; } > "$OUT/b/sdcfr_counters.csv"; rm -f "$OUT/b/p_counter_collection.csv" "$OUT"/b/*kernel_trace.csv
python3 - <<PY
import csv, collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open("$OUT/b/sdcfr_counters.csv")):
    tot[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
m={k:tot[k]/n[k] for k in tot}
waves=3072 if ${BATCH:-32768} >= 12288 else ${BATCH:-32768}//4
cyc=4*m['SQ_WAVE_CYCLES']/waves
print('cycles per working wave', cyc, 'mfma busy per simd', m['SQ_VALU_MFMA_BUSY_CYCLES']/1024, 'share', m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc)
print('issuing', m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES'], 'wait-issue', m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'], 'wait-any', m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'])
PY
