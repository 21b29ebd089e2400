#!/usr/bin/env bash
# traversal call time (policy + walk, HIP events) by batch: where the fixed cost ends and the per-traversal cost begins
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3curve; mkdir -p $O
[ -n "$LIB" ] && export SCOPA_HIP_LIBRARY="$PWD/$LIB"
for B in ${BATCHES:-64 512 1024 2048 4096 8192 16384 32768}; do
  timeout -k 10 200 python bench.py --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch $B > $O/b$B.json 2> $O/b$B.err || { tail -5 $O/b$B.err; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/b$B.json'))
print('B=$B', round(d['traversal_only']['kernel_avg_us'],1), 'us')
PY
done
