#!/usr/bin/env bash
# SDCFR GPU tests, then kernel time of k_sdcfr_traverse by task shape (W wavefronts per task), both batches, by HIP events via bench --workload sdcfr
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout -k 10 300 python -m pytest tests/test_gpu_sdcfr.py -x -q -m gpu > gpurun_out/r3c/tests.log 2>&1 || { tail -30 gpurun_out/r3c/tests.log; exit 1; }
tail -1 gpurun_out/r3c/tests.log
for W in ${WS:-1}; do for B in 4096 32768; do
  SCOPA_SDCFR_W=$W timeout -k 10 200 python bench.py --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch $B > gpurun_out/r3c/w${W}_b${B}.json 2> gpurun_out/r3c/w${W}_b${B}.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r3c/w${W}_b${B}.json'))
print('W=$W B=$B', round(d['traversal_only']['kernel_avg_us'],1), 'us', round(d['roofline']['bounds']['mfma-f32']['frac'],3), 'ms/step', round(d['ms_per_step'],3))
PY
done; done
