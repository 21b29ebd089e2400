"""Build libscopa_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to
the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libscopa_hip.so")
SOURCES = ["scopa_host.hip", "scopa_tree.hip", "scopa_mccfr.hip", "scopa_cfr.hip", "scopa_eval.hip", "scopa_sdcfr.hip", "scopa_train.hip", "scopa_multi.hip", "scopa_full.hip", "scopa_team.hip", "scopa_p2p.hip"]
# -ffp-contract=off: float64 arithmetic must round once per operation, like numpy (the one fused
#   product, np.dot, is written as an explicit fma chain); -munsafe-fp-atomics: float64 atomic adds
#   become ds_add_f64 / global_atomic_add_f64 instead of compare-and-swap loops.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "scopa.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    extra = os.environ.get("SCOPA_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DSCOPA_P2P_LIGHT=0 (scopa_p2p.h) for experiments
    cmd = [hipcc] + FLAGS + extra + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))


def source_fingerprint(name):
    """sha256 of a kernel source under csrc/ with comments and whitespace removed: what the profile summaries under profiles/ record
    (tests/tools/fold_profiles.py) and bench.py recomputes for `roofline.profile_stale` -- an edited comment is not a new kernel."""
    import hashlib
    import re
    with open(os.path.join(CSRC, name)) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return hashlib.sha256("".join(text.split()).encode()).hexdigest()
