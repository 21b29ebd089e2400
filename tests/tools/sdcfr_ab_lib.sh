#!/usr/bin/env bash
# traversal time of the product library next to development variants of it (build/*.so made by tests/tools/build_variant.py), both batches:
#     LIBS="build/libscopa_rowmask.so ..." bash tests/tools/sdcfr_ab_lib.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ab; mkdir -p $O
for rep in 1 2; do for L in product ${LIBS}; do for B in ${BATCHES:-4096 32768}; do
  N=$(basename $L .so)
  if [ "$L" = product ]; then unset SCOPA_HIP_LIBRARY; else export SCOPA_HIP_LIBRARY=$PWD/$L; fi
  timeout -k 10 200 python bench.py --workload sdcfr --no-cpu-baseline --steps 10 --warmup 2 --batch $B > $O/${N}_b$B.json 2> $O/${N}_b$B.err || { tail -5 $O/${N}_b$B.err; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/${N}_b$B.json'))
print('$N B=$B', round(d['traversal_only']['kernel_avg_us'],1), 'us')
PY
done; done; done
