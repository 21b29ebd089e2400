#!/usr/bin/env bash
# two SQ passes of k_sdcfr_walk at BATCH (default 32768): instruction mix per traversal, unit busy shares, wave states.  LIB=build/x.so profiles a development variant.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
B=${BATCH:-32768}; TAG=${TAG:-product}
[ -n "$LIB" ] && export SCOPA_HIP_LIBRARY="$ROOT/$LIB"
OUT="$ROOT/gpurun_out/walkpmc_${TAG}_b$B"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/a" -o p -- python3 "$ROOT/bench.py" --workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch $B > "$OUT/a.json" 2> "$OUT/a.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/b" -o p -- python3 "$ROOT/bench.py" --workload sdcfr --no-cpu-baseline --steps 3 --warmup 1 --batch $B > "$OUT/b.json" 2> "$OUT/b.err" || exit 1
for P in a b; do { head -1 "$OUT/$P/p_counter_collection.csv"; grep k_sdcfr_walk "$OUT/$P/p_counter_collection.csv"; } > "$OUT/$P/walk_counters.csv"; rm -f "$OUT/$P/p_counter_collection.csv" "$OUT/$P"/*kernel_trace.csv; done
python3 - <<PY
import csv, collections
m={}
for P in 'ab':
    tot=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open("$OUT/%s/walk_counters.csv"%P)):
        tot[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    m.update({k:tot[k]/n[k] for k in tot})
B=$B
print('$TAG B=%d'%B, {k:round(v,1) for k,v in m.items()})
print('per traversal: VALU %.0f SALU %.0f LDS %.0f VMEM_WR %.1f SMEM %.1f (wave-instructions)' % tuple(m[k]/B for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_WR','SQ_INSTS_SMEM')))
print('waves', m['SQ_WAVES'], 'busy cycles (per SE?)', m['SQ_BUSY_CYCLES'], 'wave cycles', m['SQ_WAVE_CYCLES'])
wc=m['SQ_WAVE_CYCLES']
print('of wave time: issuing %.3f wait-issue %.3f wait-any %.3f' % (m['SQ_ACTIVE_INST_ANY']/wc, m['SQ_WAIT_INST_ANY']/wc, m['SQ_WAIT_ANY']/wc))
print('unit busy / SQ_BUSY_CYCLES: valu %.3f lds-idx %.3f lds-conflict %.3f lds-inst %.3f scalar %.3f' % tuple(m[k]/m['SQ_BUSY_CYCLES'] for k in ('SQ_ACTIVE_INST_VALU','SQ_LDS_IDX_ACTIVE','SQ_LDS_BANK_CONFLICT','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_SCA')))
PY
