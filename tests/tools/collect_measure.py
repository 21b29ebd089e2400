#!/usr/bin/env python3
"""Copy the outputs of tests/tools/measure.sh (merged into gpurun_out/measure/ by gpurun) into the tracked evidence set under profiles/ (build container):
    python tests/tools/collect_measure.py [gpurun_out/measure] [round tag, default r04]
A bench output becomes profiles/<tag>_<name>.json (its last line: the one JSON line), the stamp listings <tag>_walk_stamps.txt etc.; refuses outputs that hold
a traceback."""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out", "measure")
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
P = os.path.join(R, "profiles")
for name in sorted(os.listdir(src)):
    path = os.path.join(src, name)
    if name.endswith(".err") or name.endswith(".log") or not os.path.isfile(path):
        continue
    text = open(path, errors="replace").read()
    if "Traceback (most recent call last)" in text:
        sys.exit(f"{path} holds a Python traceback: re-run tests/tools/measure.sh")
    if name.endswith(".json"):
        line = text.strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(P, f"{tag}_{name}"), "w").write(line + "\n")
    elif name in ("walk_stamps.txt", "sdcfr_stamps.txt", "sdwalk_stamps.txt", "wg_starts.txt"):
        shutil.copyfile(path, os.path.join(P, f"{tag}_{name}"))
print("collected into", P)
