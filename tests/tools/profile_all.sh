#!/usr/bin/env bash
# The round's profile set in one call on the GPU box: the MCCFR passes (profile_round.sh: kernel stats of the bench command with its sub-records, FETCH_SIZE /
# WRITE_SIZE, SQ counters at two batches), the SDCFR passes at both batches (profile_sdcfr.sh), the lanes kernel's passes with the counter calibration
# (profile_lanes.sh) the kernel stats of the other kernels (stats_extra.sh) and the counter passes of the three step kernels (pmc_state_engines.sh -> gpurun_out/pmc_engines/summary.json, kept as profiles/<tag>_pmc_state_engines.json) and of the evaluator (pmc_eval.sh -> profiles/pmc_eval.json).  Any failing step stops the script with a non-zero status.
#     gpurun --timeout 1200 -- 'bash tests/tools/profile_all.sh'
# then, in the build container:  python tests/tools/fold_profiles.py gpurun_out/prof r04 && python tests/tools/fold_lanes.py gpurun_out/lanes r04
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/prof_sdcfr gpurun_out/prof_sdcfr_b4096 gpurun_out/prof_sdcfr_b32768
bash tests/tools/profile_round.sh > gpurun_out/prof_round.log 2>&1 || { tail -20 gpurun_out/prof_round.log; cat gpurun_out/prof/*.err | tail -20; exit 1; }
echo "profile_round done"
for B in 4096 32768; do
  BATCH=$B bash tests/tools/profile_sdcfr.sh > gpurun_out/prof_sdcfr_b$B.log 2>&1 || { tail -20 gpurun_out/prof_sdcfr_b$B.log; exit 1; }
  mv gpurun_out/prof_sdcfr gpurun_out/prof_sdcfr_b$B
  echo "profile_sdcfr $B done"
done
bash tests/tools/profile_lanes.sh > gpurun_out/prof_lanes.log 2>&1 || { tail -20 gpurun_out/prof_lanes.log; exit 1; }
echo "profile_lanes done"
bash tests/tools/stats_extra.sh > gpurun_out/stats_extra.log 2>&1 || { tail -20 gpurun_out/stats_extra.log; exit 1; }
cat gpurun_out/stats_extra.log
bash tests/tools/pmc_state_engines.sh > gpurun_out/pmc_engines.log 2>&1 || { tail -20 gpurun_out/pmc_engines.log; exit 1; }
cat gpurun_out/pmc_engines.log
bash tests/tools/pmc_eval.sh > gpurun_out/pmc_eval.log 2>&1 || { tail -20 gpurun_out/pmc_eval.log; exit 1; }
cat gpurun_out/pmc_eval.log
du -sh gpurun_out/prof gpurun_out/prof_sdcfr_b* gpurun_out/lanes gpurun_out/stats_extra
