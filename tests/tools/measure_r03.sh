#!/usr/bin/env bash
# Round-3 measurement set (GPU box): the driver's invocation, the default invocation, large batches, SDCFR at both batches, stage stamps.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"; O=gpurun_out/r03m; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_shape_steps20.json 2> $O/driver.err || exit 1
python bench.py > $O/bench_n1.json 2> $O/n1.err || exit 1
python bench.py --batch 32768 --steps 500 --no-sdcfr > $O/bench_n1_b32768.json 2> $O/b32768.err || exit 1
python bench.py --batch 65536 --steps 300 --no-sdcfr > $O/bench_n1_b65536.json 2> $O/b65536.err || exit 1
python bench.py --workload sdcfr --steps 30 --warmup 3 > $O/bench_sdcfr_b4096.json 2> $O/sd4096.err || exit 1
python bench.py --workload sdcfr --steps 30 --warmup 3 --batch 32768 > $O/bench_sdcfr_b32768.json 2> $O/sd32768.err || exit 1
python bench.py --workload sdcfr --steps 30 --warmup 3 --sdcfr-train-batch 4096 > $O/bench_sdcfr_b4096_trainbatch4096.json 2> $O/sdtb.err || exit 1
python bench.py --gpus 1 --force-dist --exchange p2p --no-sdcfr > $O/bench_n1_dist_path_p2p.json 2> $O/p2p.err || exit 1
python bench.py --gpus 1 --force-dist --exchange rccl --no-sdcfr > $O/bench_n1_dist_path_rccl.json 2> $O/rccl.err || exit 1
python bench.py --gpus 2 --share-gpu --no-sdcfr --steps 500 > $O/bench_2ranks_shared_gpu_spawn.json 2> $O/2r.err || exit 1
for B in 4096 32768; do SCOPA_HIP_LIBRARY=$PWD/build/libscopa_stamps.so python tests/tools/sdcfr_stamps.py $B 10 2>&1 | grep -v amdgpu.ids; done > $O/sdcfr_stamps.txt
for B in 64 4096 32768; do SCOPA_HIP_LIBRARY=$PWD/build/libscopa_stamps.so python tests/tools/sdwalk_stamps.py $B 10 2>&1 | grep -v "amdgpu.ids\|Estimated"; done > $O/sdwalk_stamps.txt
python tests/tools/exact_cfr_timing.py > $O/exact_cfr_timing.json 2> $O/exact.err
ls -la $O | head -30
