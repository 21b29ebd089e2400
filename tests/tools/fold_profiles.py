#!/usr/bin/env python3
"""Fold rocprofv3 outputs merged into gpurun_out/ into the tracked summaries under profiles/ (run in the build container).
    python tests/tools/fold_profiles.py gpurun_out/fin_stats2 gpurun_out/fin_pmc_fetch gpurun_out/fin_pmc_write [stats_bench.json]"""
import collections, csv, glob, json, os, statistics as st, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stats, fetch, write = sys.argv[1:4]
open(f"{R}/profiles/r01_kernel_stats.csv", "w").write(open(glob.glob(f"{stats}/*kernel_stats.csv")[0]).read())
if len(sys.argv) > 4:
    open(f"{R}/profiles/r01_bench_n1_under_rocprof_stats.json", "w").write(open(sys.argv[4]).read().strip().splitlines()[-1] + "\n")
res = {}
for d, ctr in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{d}/*counter_collection.csv")[0])):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    rows = [dict(kernel=k, counter=ctr, dispatches=len(v), mean_KB=st.mean(v), min_KB=min(v), max_KB=max(v)) for k, v in acc.items()]
    with open(f"{R}/profiles/r01_pmc_{ctr}.csv", "w") as fh:
        w = csv.DictWriter(fh, fieldnames=["kernel", "counter", "dispatches", "mean_KB", "min_KB", "max_KB"])
        w.writeheader(); w.writerows(rows)
    res[ctr] = {r["kernel"]: r for r in rows}
f, w = res["FETCH_SIZE"]["k_mccfr_traverse"]["mean_KB"], res["WRITE_SIZE"]["k_mccfr_traverse"]["mean_KB"]
json.dump({"kernel": "k_mccfr_traverse", "workload": "bench.py default (B=4096 per traverser, 738 infosets)",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, %d dispatches each (profiles/r01_pmc_*.csv)" % res["FETCH_SIZE"]["k_mccfr_traverse"]["dispatches"],
           "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w,
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> read bytes doubled (an upper bound here: not all reads are 16 B/lane); WRITE_SIZE taken as is",
           "bytes_per_launch": (2 * f + w) * 1e3, "bytes_per_launch_uncorrected": (f + w) * 1e3,
           "reduce_apply_FETCH_KB_raw": res["FETCH_SIZE"].get("void k_mccfr_reduce_apply<false>", {}).get("mean_KB"),
           "reduce_apply_WRITE_KB": res["WRITE_SIZE"].get("void k_mccfr_reduce_apply<false>", {}).get("mean_KB")},
          open(f"{R}/profiles/hbm_traffic.json", "w"), indent=1)
print(open(f"{R}/profiles/hbm_traffic.json").read())
