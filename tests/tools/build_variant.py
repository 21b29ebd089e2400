#!/usr/bin/env python3
"""Build a variant of libscopa_hip.so for A/B measurements (tests/tools/ab_time.sh, walk_stamps.py):
    python tests/tools/build_variant.py OUT.so [--rev GIT_REV] [extra hipcc flags ...]
--rev takes scopa_amd/csrc and include/ from that commit instead of the working tree.  Outputs belong under build/ (git-ignored,
shipped to the GPU box)."""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scopa_amd import build as b

args = sys.argv[1:]
out = os.path.abspath(args.pop(0))
rev = None
if args and args[0] == "--rev":
    rev = args[1]; args = args[2:]
with tempfile.TemporaryDirectory() as tmp:
    if rev:
        subprocess.check_call(f"git -C {ROOT} archive {rev} scopa_amd/csrc include | tar -x -C {tmp}", shell=True)
    else:
        shutil.copytree(os.path.join(ROOT, "scopa_amd", "csrc"), os.path.join(tmp, "scopa_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + args + ["-o", out] + [os.path.join(tmp, "scopa_amd", "csrc", s) for s in b.SOURCES])
print(out)
