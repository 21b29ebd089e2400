#!/usr/bin/env bash
# The round's measurement set (GPU box): the driver's invocation, the default invocation, large batches, SDCFR at both batches, the N > 1 code path on one rank and
# two ranks sharing the GPU, the stage stamps of the three traversal kernels (variant library built by tests/tools/build_variant.py -- rebuilt HERE when it is older
# than any kernel source, never a stale one), the exact-CFR timing.  Every step's status is checked; a Python traceback in any output fails the script.
#     gpurun --timeout 1200 -- 'bash tests/tools/measure.sh'          outputs: gpurun_out/measure/
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"; O=gpurun_out/measure; rm -rf $O; mkdir -p $O
STAMPS=build/libscopa_stamps.so
if [ ! -f $STAMPS ] || [ -n "$(find scopa_amd/csrc include -newer $STAMPS -print -quit)" ]; then
  echo "building $STAMPS (absent or older than the kernel sources)"
  python tests/tools/build_variant.py $STAMPS -DSCOPA_WALK_STAMPS > $O/build_stamps.log 2>&1
fi
run() { local out=$1; shift; "$@" > "$O/$out" 2> "$O/$out.err" || { echo "FAILED: $*" >&2; tail -5 "$O/$out.err" >&2; exit 1; }; }
run bench_n1_driver_shape_steps20.json python bench.py --gpus 1 --steps 20 --warmup 5
run bench_n1.json python bench.py
run bench_n1_b32768.json python bench.py --batch 32768 --steps 500 --no-sdcfr --no-subrecords
run bench_n1_b65536.json python bench.py --batch 65536 --steps 300 --no-sdcfr --no-subrecords
run bench_sdcfr_b4096.json python bench.py --workload sdcfr --steps 30 --warmup 3
run bench_sdcfr_b32768.json python bench.py --workload sdcfr --steps 30 --warmup 3 --batch 32768
run bench_sdcfr_b4096_trainbatch4096.json python bench.py --workload sdcfr --steps 30 --warmup 3 --sdcfr-train-batch 4096
run bench_n1_dist_path_p2p.json python bench.py --gpus 1 --force-dist --exchange p2p --no-sdcfr --no-subrecords
run bench_n1_dist_path_rccl.json python bench.py --gpus 1 --force-dist --exchange rccl --no-sdcfr --no-subrecords
run bench_2ranks_shared_gpu_spawn.json python bench.py --gpus 2 --share-gpu --no-sdcfr --no-subrecords --steps 500
export SCOPA_HIP_LIBRARY=$PWD/$STAMPS SCOPA_AB_OLD_LIBRARY=0
for B in 4096 65536; do run walk_stamps_b$B.txt python tests/tools/walk_stamps.py $B $([ $B = 4096 ] && echo 2000 || echo 300); done
for B in 4096 32768; do run sdcfr_stamps_b$B.txt python tests/tools/sdcfr_stamps.py $B 10; done
for B in 64 4096 32768; do run sdwalk_stamps_b$B.txt python tests/tools/sdwalk_stamps.py $B 10; done
unset SCOPA_HIP_LIBRARY
run exact_cfr_timing.json python tests/tools/exact_cfr_timing.py
run wg_starts.txt python tests/tools/wg_starts.py 4096
cat $O/walk_stamps_b4096.txt $O/walk_stamps_b65536.txt > $O/walk_stamps.txt
cat $O/sdcfr_stamps_b4096.txt $O/sdcfr_stamps_b32768.txt | grep -v "Estimated input" > $O/sdcfr_stamps.txt
cat $O/sdwalk_stamps_b64.txt $O/sdwalk_stamps_b4096.txt $O/sdwalk_stamps_b32768.txt | grep -v "Estimated input" > $O/sdwalk_stamps.txt
if grep -l "Traceback" $O/*.json $O/*.txt 2>/dev/null; then echo "a measurement output holds a Python traceback" >&2; exit 1; fi
ls -la $O | head -50
