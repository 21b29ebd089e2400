#!/usr/bin/env bash
# stamps (cheap form) at two batches + SQ passes a/b/c at B=4096 and 32768
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
SCOPA_HIP_LIBRARY=$PWD/build/libscopa_stamps.so timeout -k 10 120 python tests/tools/sdcfr_stamps.py 4096 10 > gpurun_out/r3b/stamps_b4096.txt 2>&1
SCOPA_HIP_LIBRARY=$PWD/build/libscopa_stamps.so timeout -k 10 120 python tests/tools/sdcfr_stamps.py 32768 5 > gpurun_out/r3b/stamps_b32768.txt 2>&1
cat gpurun_out/r3b/stamps_b4096.txt gpurun_out/r3b/stamps_b32768.txt | grep -v amdgpu.ids
bash tests/tools/profile_sdcfr.sh > gpurun_out/r3b/prof4096.log 2>&1; rm -rf gpurun_out/r3b/prof_b4096; mv gpurun_out/prof_sdcfr gpurun_out/r3b/prof_b4096
BATCH=32768 bash tests/tools/profile_sdcfr.sh > gpurun_out/r3b/prof32768.log 2>&1; rm -rf gpurun_out/r3b/prof_b32768; mv gpurun_out/prof_sdcfr gpurun_out/r3b/prof_b32768
