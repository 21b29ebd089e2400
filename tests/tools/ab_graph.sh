#!/usr/bin/env bash
# A/B on ONE box, variants interleaved: eager in-library iteration loop vs captured HIP graphs (bench.py --graph), driver-shaped (--steps 20) and 2000-step regions
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
for rep in 1 2 3; do for v in eager graph; do for steps in 20 2000; do
  flag=""; [ $v = graph ] && flag="--graph"
  timeout -k 10 120 python bench.py --gpus 1 --steps $steps --warmup 5 --no-cpu-baseline --no-sdcfr $flag > gpurun_out/r3e/${v}_${steps}_${rep}.json 2> gpurun_out/r3e/${v}_${steps}_${rep}.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r3e/${v}_${steps}_${rep}.json'))
print('$v steps=$steps rep=$rep', round(1e3*d['ms_per_step'],2), 'us/step  min', round(1e3*d['timing']['ms_per_step_min'],2), ' value', '%.3e'%d['value'])
PY
done; done; done
