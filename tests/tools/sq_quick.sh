#!/usr/bin/env bash
# SQ counters of k_mccfr_traverse for one or more library variants at one batch (GPU box): two passes of 8 counters over
# `python3 tests/tools/time_iter.py BATCH ITERS`, per-wavefront figures printed per library.
#     gpurun -- 'bash tests/tools/sq_quick.sh 4096 40 build/libscopa_base.so scopa_amd/libscopa_hip.so'
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
BATCH=$1; ITERS=$2; shift 2
OUT="$ROOT/gpurun_out/sq_quick"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  TAG=$(basename "$LIB" .so)
  export SCOPA_HIP_LIBRARY="$ROOT/$LIB"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$OUT/a_$TAG" -o p -- python3 "$ROOT/tests/tools/time_iter.py" "$BATCH" "$ITERS" > "$OUT/a_$TAG.txt" 2> "$OUT/a_$TAG.err"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/b_$TAG" -o p -- python3 "$ROOT/tests/tools/time_iter.py" "$BATCH" "$ITERS" > "$OUT/b_$TAG.txt" 2> "$OUT/b_$TAG.err"
  python3 - "$OUT" "$TAG" "$BATCH" <<'PY'
import csv, glob, sys, collections, json
out, tag, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(list)
for d in ("a_", "b_"):
    for r in csv.DictReader(open(glob.glob(f"{out}/{d}{tag}/*counter_collection.csv")[0])):
        if r["Kernel_Name"].startswith("k_mccfr_traverse"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in acc.items()}   # the later half of the dispatches (the timed ones)
w = m["SQ_WAVES"]
res = {"library": tag, "batch": batch, "dispatches": len(acc["SQ_WAVES"]), "waves": w, "valu_per_wave": m["SQ_INSTS_VALU"] / w, "salu_per_wave": m["SQ_INSTS_SALU"] / w,
       "lds_per_wave": m["SQ_INSTS_LDS"] / w, "valu_busy_cycles_per_wave": 4 * m["SQ_ACTIVE_INST_VALU"] / w, "lds_idx_cycles_per_wave": m["SQ_LDS_IDX_ACTIVE"] / w,
       "lds_bank_conflict_share": m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_ACTIVE_INST_LDS"], 1), "wave_quad_cycles": m["SQ_WAVE_CYCLES"] / w,
       "active_share": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], "wait_any_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
       "vmem_rd_per_wave": m["SQ_INSTS_VMEM_RD"] / w, "vmem_wr_per_wave": m["SQ_INSTS_VMEM_WR"] / w}
print(json.dumps(res))
json.dump(res, open(f"{out}/{tag}_b{batch}.json", "w"), indent=1)
PY
  rm -rf "$OUT/a_$TAG" "$OUT/b_$TAG"
done
