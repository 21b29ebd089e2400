#!/usr/bin/env python3
"""Fold the outputs of tests/tools/profile_lanes.sh (merged into gpurun_out/lanes/ by gpurun) into the tracked summaries under profiles/
(run in the build container):
    python tests/tools/fold_lanes.py [gpurun_out/lanes] [round tag, default r04]
Writes  profiles/<tag>_lanes_kernel_stats.csv      rocprofv3 --kernel-trace --stats of benchmarks/multi_deal_lanes_bench.py (131 072 deals x 5 iterations)
        profiles/<tag>_lanes_bench.json            the line that program printed unprofiled (HIP-event timing)
        profiles/<tag>_lanes_pmc.json              every counter pass, per dispatch: the kernel and the calibration patterns (benchmarks/micro/row_gather.hip)
        profiles/lanes_hbm_traffic.json            HBM bytes per launch of k_cfr_exact_lanes with the correction the calibration gives -- what
                                                   benchmarks/subrecords.py reads for the `many_deals` sub-record's roofline.traffic"""
import collections, csv, glob, json, os, subprocess, sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out", "lanes")
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
P = os.path.join(R, "profiles")
sys.path.insert(0, R)
from scopa_amd.build import source_fingerprint
commit = subprocess.run(["git", "-C", R, "log", "-1", "--format=%h", "--", "scopa_amd/csrc/scopa_multi.hip"], capture_output=True, text=True).stdout.strip()


def counters(d):
    """{kernel short name: {counter: [value per dispatch]}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    f = glob.glob(f"{d}/*counter_collection.csv")
    if not f:
        return acc
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


for f in glob.glob(f"{src}/*.err"):                       # a pass that failed says so in its .err; never fold a traceback into a profile
    t = open(f).read()
    if "Traceback" in t:
        sys.exit(f"{f} holds a Python traceback: re-run tests/tools/profile_lanes.sh")
open(f"{P}/{tag}_lanes_kernel_stats.csv", "w").write(open(glob.glob(f"{src}/stats/*kernel_stats.csv")[0]).read())
bench = json.loads(open(f"{src}/lanes_bench.json").read().strip().splitlines()[-1])
json.dump(bench, open(f"{P}/{tag}_lanes_bench.json", "w"), indent=1)
gather = [json.loads(l) for l in open(f"{src}/row_gather.jsonl") if l.strip().startswith("{")]

allc = {"lanes": collections.defaultdict(dict), "calib": collections.defaultdict(dict)}
for d in sorted(glob.glob(f"{src}/pmc_*")):
    if not os.path.isdir(d):
        continue
    which = "lanes" if os.path.basename(d).startswith("pmc_lanes_") else "calib"
    for k, v in counters(d).items():
        for c, vals in v.items():
            allc[which][k][c] = vals
rows = gather[0]["rows"]
calib = {}
for k, v in allc["calib"].items():
    last = {c: vals[-1] for c, vals in v.items()}       # the last of a kernel's four dispatches
    calib[k] = {"rows_touched": rows, "counters_last_dispatch": last,
                "per_row": {c: x / rows for c, x in last.items()},
                "unprofiled": next((g for g in gather if g["kernel"] == k), None)}
L = allc["lanes"]["k_cfr_exact_lanes"]
big = {c: vals[-1] for c, vals in L.items()}            # dispatch 0 = the 1-iteration warm-up launch, the last = a 5-iteration launch
deals, iters = bench["deals"], bench["iterations_per_launch"]
rd128, rd64, rd32 = big.get("TCC_EA0_RDREQ_128B", 0.0), big.get("TCC_EA0_RDREQ_64B", 0.0), big.get("TCC_EA0_RDREQ_32B", 0.0)
wr, wr64 = big["TCC_EA0_WRREQ"], big.get("TCC_EA0_WRREQ_64B", big["TCC_EA0_WRREQ"])
read_bytes = 128.0 * rd128 + 64.0 * rd64 + 32.0 * rd32
write_bytes = 64.0 * wr64 + 32.0 * (wr - wr64)
out = {"kernel": "k_cfr_exact_lanes", "deals": deals, "iterations": iters, "commit": commit, "source_sha256": source_fingerprint("scopa_multi.hip"),
       "source": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python3 benchmarks/multi_deal_lanes_bench.py --deals %d --iters %d (tests/tools/profile_lanes.sh; "
                 "profiles/%s_lanes_pmc.json holds every pass); the %d-iteration launch" % (deals, iters, tag, iters),
       "FETCH_SIZE_KB_raw": big["FETCH_SIZE"], "WRITE_SIZE_KB": big["WRITE_SIZE"], "TCC_EA0_RDREQ": big["TCC_EA0_RDREQ"], "TCC_EA0_WRREQ": wr,
       "TCC_EA0_RDREQ_128B": rd128, "TCC_EA0_RDREQ_64B": rd64, "TCC_EA0_RDREQ_32B": rd32, "TCC_EA0_WRREQ_64B": wr64,
       "TCC_HIT": big.get("TCC_HIT"), "TCC_MISS": big.get("TCC_MISS"),
       "read_bytes_per_launch": read_bytes, "write_bytes_per_launch": write_bytes, "bytes_per_launch": read_bytes + write_bytes,
       "bytes_per_launch_uncorrected": (big["FETCH_SIZE"] + big["WRITE_SIZE"]) * 1024.0,
       "read_requests_per_deal_iteration": big["TCC_EA0_RDREQ"] / (deals * iters), "write_requests_per_deal_iteration": wr / (deals * iters),
       "hbm_bytes_per_deal_iteration": (read_bytes + write_bytes) / (deals * iters),
       "kernel_seconds_unprofiled": bench["kernel_seconds"],
       "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE = TCC_EA0_RDREQ x 64 B whatever the request size.  The request-size counters say every read request of this "
                     "kernel is a 128-byte one (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ; _64B and _32B ~ 0), so read bytes = 128 x TCC_EA0_RDREQ_128B + 64 x _64B + 32 x _32B = 2 x FETCH_SIZE; "
                     "writes are 64-byte requests (TCC_EA0_WRREQ_64B = 98 % of TCC_EA0_WRREQ, the rest 32-byte) and WRITE_SIZE reads them exactly",
       "calibration": "benchmarks/micro/row_gather.hip under the same passes, 2^27 distinct 64-byte rows of an 8 GiB table per launch: a coalesced 16 B/lane stream, a 32 B/lane gather from "
                      "distinct rows and a 64 B/lane gather ALL leave L2 as 128-byte read requests (one per row for the gathers: the whole 128-byte line of the row is fetched, "
                      "FETCH_SIZE tallies 64 B of it); 64 B/lane stores leave as one 64-byte write request per row, WRITE_SIZE exact (profiles/%s_lanes_pmc.json `calibration`)" % tag}
json.dump(out, open(f"{P}/lanes_hbm_traffic.json", "w"), indent=1)
json.dump({"what": "counter passes of tests/tools/profile_lanes.sh, values per dispatch (k_cfr_exact_lanes: dispatch 0 = the 1-iteration warm-up launch, then the 5-iteration launches; "
                   "calibration kernels: four launches each)", "lanes": allc["lanes"], "calibration": calib}, open(f"{P}/{tag}_lanes_pmc.json", "w"), indent=1)
t = bench["kernel_seconds"]
med = sorted(t)[len(t) // 2]
print(json.dumps({k: out[k] for k in ("read_bytes_per_launch", "write_bytes_per_launch", "bytes_per_launch", "hbm_bytes_per_deal_iteration")}))
print("HBM GB/s at the unprofiled median launch:", out["bytes_per_launch"] / med / 1e9, "of peak", out["bytes_per_launch"] / med / 8e12)
