#!/usr/bin/env bash
# The soak set (GPU box): 2e6 batched-MCCFR iterations with exact counters, the SDCFR loop for 5000 iterations, the exploitability curve.
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"; O=gpurun_out/soak; mkdir -p $O
[ -n "${SKIP_MCCFR:-}" ] || timeout -k 10 300 python tests/tools/soak.py 2000000 > $O/soak_2e6_iterations.json 2> $O/soak.err || { tail -5 $O/soak.err; exit 1; }
timeout -k 10 400 python tests/tools/sdcfr_soak.py ${SD_ITERS:-5000} > $O/sdcfr_soak.json 2> $O/sdsoak.err || { tail -5 $O/sdsoak.err; exit 1; }
timeout -k 10 300 python tests/tools/exploitability_curve.py > $O/exploitability_curve.json 2> $O/curve.err || { tail -5 $O/curve.err; exit 1; }
tail -3 $O/sdsoak.err; ls -la $O
