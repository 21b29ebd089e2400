#!/usr/bin/env bash
# the SDCFR profile passes at both batches (outputs: gpurun_out/prof_sdcfr_b4096, _b32768; fold with fold_profiles.py gpurun_out/prof r03)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"
for B in 4096 32768; do
  BATCH=$B bash tests/tools/profile_sdcfr.sh > gpurun_out/prof_sdcfr_b$B.log 2>&1 || { tail -20 gpurun_out/prof_sdcfr_b$B.log; exit 1; }
  rm -rf gpurun_out/prof_sdcfr_b$B; mv gpurun_out/prof_sdcfr gpurun_out/prof_sdcfr_b$B
done
du -sh gpurun_out/prof_sdcfr_b*
