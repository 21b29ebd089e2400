#!/usr/bin/env bash
# SDCFR GPU tests, then traversal time by form (MODE 0 = policy table + walks, 1 = forward per visit) and task shape, both batches, by HIP events via bench --workload sdcfr
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout -k 10 400 python -m pytest tests/test_gpu_sdcfr.py -x -q -m gpu > gpurun_out/r3c/tests.log 2>&1 || { tail -30 gpurun_out/r3c/tests.log; exit 1; }
tail -1 gpurun_out/r3c/tests.log
for M in ${MODES:-0}; do for TT in ${TS:-0}; do for B in 4096 32768; do
  SCOPA_SDCFR_MODE=$M SCOPA_SDCFR_T=$TT timeout -k 10 200 python bench.py --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch $B > gpurun_out/r3c/m${M}_t${TT}_b${B}.json 2> gpurun_out/r3c/m${M}_t${TT}_b${B}.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r3c/m${M}_t${TT}_b${B}.json'))
print('mode=$M T=$TT B=$B', round(d['traversal_only']['kernel_avg_us'],1), 'us', {k:round(v['frac'],3) for k,v in d['roofline']['bounds'].items()}, 'per-visit', d['traversal_only']['forward_per_visit_kernel_avg_us'], 'ms/step', round(d['ms_per_step'],3))
PY
done; done; done
