#!/usr/bin/env bash
# Development probe of k_sdcfr_traverse on the GPU box: the SDCFR GPU tests, stage stamps (needs build/libscopa_stamps.so =
# build_variant.py ... -DSCOPA_WALK_STAMPS), bench --workload sdcfr at two batches.  Outputs under gpurun_out/r3a/.
#     gpurun --timeout 900 -- bash tests/tools/sdcfr_probe.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 300 python -m pytest tests/test_gpu_sdcfr.py -x -q -m gpu > gpurun_out/r3a/tests.log 2>&1 || { tail -30 gpurun_out/r3a/tests.log; exit 1; }
tail -2 gpurun_out/r3a/tests.log
SCOPA_HIP_LIBRARY=$PWD/build/libscopa_stamps.so timeout -k 10 120 python tests/tools/sdcfr_stamps.py 4096 10 > gpurun_out/r3a/stamps_b4096.txt 2>&1
SCOPA_HIP_LIBRARY=$PWD/build/libscopa_stamps.so timeout -k 10 120 python tests/tools/sdcfr_stamps.py 32768 5 > gpurun_out/r3a/stamps_b32768.txt 2>&1
for W in 1 2 3; do for B in 4096 32768; do SCOPA_SDCFR_W=$W timeout -k 10 120 python tests/tools/time_sdcfr.py $B 20 2>&1 | tail -1; done; done
timeout -k 10 200 python bench.py --workload sdcfr --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r3a/bench_sdcfr_b4096.json 2> gpurun_out/r3a/bench_sdcfr_b4096.err
timeout -k 10 200 python bench.py --workload sdcfr --no-cpu-baseline --steps 20 --warmup 3 --batch 32768 > gpurun_out/r3a/bench_sdcfr_b32768.json 2> gpurun_out/r3a/bench_sdcfr_b32768.err
grep -A9 "traverser 0" gpurun_out/r3a/stamps_b4096.txt
python - <<'PY'
import json
for b in (4096, 32768):
    d=json.load(open(f'gpurun_out/r3a/bench_sdcfr_b{b}.json'))
    print(b, d['ms_per_step'], d['traversal_only'], d['roofline']['bounds']['mfma-f32']['frac'])
PY
