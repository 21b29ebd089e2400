#!/usr/bin/env bash
# Round-3 profile set in one call on the GPU box: the MCCFR passes (profile_round.sh) and the SDCFR passes at both batches.
#     gpurun --timeout 1200 -- 'bash tests/tools/profile_all_r03.sh'    then (build container)  python tests/tools/fold_profiles.py gpurun_out/prof r03
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"
bash tests/tools/profile_round.sh > gpurun_out/prof_round.log 2>&1 || { tail -20 gpurun_out/prof_round.log; exit 1; }
for B in 4096 32768; do
  BATCH=$B bash tests/tools/profile_sdcfr.sh > gpurun_out/prof_sdcfr_b$B.log 2>&1 || { tail -20 gpurun_out/prof_sdcfr_b$B.log; exit 1; }
  rm -rf gpurun_out/prof_sdcfr_b$B; mv gpurun_out/prof_sdcfr gpurun_out/prof_sdcfr_b$B
done
du -sh gpurun_out/prof gpurun_out/prof_sdcfr_b*
