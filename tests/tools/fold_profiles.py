#!/usr/bin/env python3
"""Fold the rocprofv3 outputs of tests/tools/profile_round.sh (merged into gpurun_out/prof/ by gpurun) into the tracked summaries
under profiles/ (run in the build container):
    python tests/tools/fold_profiles.py [gpurun_out/prof] [round-tag, default r02]
Writes  profiles/<tag>_kernel_stats.csv, <tag>_bench_n1_under_rocprof_stats.json   (kernel durations of the bench command)
        profiles/<tag>_pmc_FETCH_SIZE.csv, <tag>_pmc_WRITE_SIZE.csv, hbm_traffic.json (HBM bytes per launch, gfx950 correction)
        profiles/<tag>_pmc_sq_traverse_b<batch>.json, traverse_sq.json               (SQ counters of k_mccfr_traverse per pair)
hbm_traffic.json and traverse_sq.json are what bench.py reads for roofline.traffic and the VALU / LDS ceilings; both record the
commit of the kernel they were measured on."""
import collections, csv, glob, json, os, statistics as st, subprocess, sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out", "prof")
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
P = os.path.join(R, "profiles")
commit = subprocess.run(["git", "-C", R, "log", "-1", "--format=%h", "--", "scopa_amd/csrc/scopa_mccfr.hip"], capture_output=True, text=True).stdout.strip()
import hashlib
sys.path.insert(0, R)
from scopa_amd.build import source_fingerprint
source_sha256 = source_fingerprint("scopa_mccfr.hip")   # comments and whitespace stripped; bench.py recomputes it: roofline.profile_stale


def counters(d):
    """{kernel short name: {counter: [value per dispatch]}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(f"{d}/*counter_collection.csv")[0])):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


# 1. kernel durations
open(f"{P}/{tag}_kernel_stats.csv", "w").write(open(glob.glob(f"{src}/stats/*kernel_stats.csv")[0]).read())
open(f"{P}/{tag}_bench_n1_under_rocprof_stats.json", "w").write(open(f"{src}/stats_bench.json").read().strip().splitlines()[-1] + "\n")

# 2. HBM traffic
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = counters(f"{src}/pmc_{ctr}")
    rows = [dict(kernel=k, counter=ctr, dispatches=len(v[ctr]), mean_KB=st.mean(v[ctr]), min_KB=min(v[ctr]), max_KB=max(v[ctr])) for k, v in acc.items() if ctr in v]
    with open(f"{P}/{tag}_pmc_{ctr}.csv", "w") as fh:
        w = csv.DictWriter(fh, fieldnames=["kernel", "counter", "dispatches", "mean_KB", "min_KB", "max_KB"])
        w.writeheader(); w.writerows(rows)
    res[ctr] = {r["kernel"]: r for r in rows}
f, w = res["FETCH_SIZE"]["k_mccfr_traverse"]["mean_KB"], res["WRITE_SIZE"]["k_mccfr_traverse"]["mean_KB"]
af, aw = res["FETCH_SIZE"].get("k_mccfr_apply_groups", {}).get("mean_KB"), res["WRITE_SIZE"].get("k_mccfr_apply_groups", {}).get("mean_KB")
json.dump({"kernel": "k_mccfr_traverse", "batch": 4096, "commit": commit, "source_sha256": source_sha256,
           "workload": "bench.py default (B=4096 per traverser, 738 infosets)",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, %d dispatches each (profiles/%s_pmc_*.csv; tests/tools/profile_round.sh)"
                     % (res["FETCH_SIZE"]["k_mccfr_traverse"]["dispatches"], tag),
           "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w,
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while every read request is a 128-byte one (calibrated this round for gathers too: profiles/r04_lanes_pmc.json) -> read bytes doubled; WRITE_SIZE taken as is (exact for float atomics: one dword per lane); the counters' unit is 1024 bytes",
           "bytes_per_launch": (2 * f + w) * 1024.0, "bytes_per_launch_uncorrected": (f + w) * 1024.0,
           "apply_groups_FETCH_KB_raw": af, "apply_groups_WRITE_KB": aw,
           "iteration_bytes": (2 * f + w + 2 * (af or 0) + (aw or 0)) * 1024.0},
          open(f"{P}/hbm_traffic.json", "w"), indent=1)
print(open(f"{P}/hbm_traffic.json").read())

# 3. SQ counters of the traversal kernel
sq_all = {}
for batch in (4096, 65536):
    m = {}
    for part in ("a", "b"):
        for c, v in counters(f"{src}/sq_{part}_{batch}")["k_mccfr_traverse"].items():
            m[c] = st.mean(v)
    pairs = float(batch)
    d = {"valu_instr_per_pair": m["SQ_INSTS_VALU"] / pairs, "lds_instr_per_pair": m["SQ_INSTS_LDS"] / pairs, "salu_instr_per_pair": m["SQ_INSTS_SALU"] / pairs,
         "valu_busy_cycles_per_pair": 4.0 * m["SQ_ACTIVE_INST_VALU"] / pairs, "lds_array_cycles_per_pair": m["SQ_LDS_IDX_ACTIVE"] / pairs,
         "lds_bank_conflict_share_of_lds_active": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"],
         "wave_time_active": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], "wave_time_waiting_any": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
         "wave_time_waiting_on_lds_issue": m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"], "wave_quad_cycles_per_wave": m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]}
    sq_all[batch] = d
    json.dump({"kernel": "k_mccfr_traverse", "commit": commit, "workload": f"bench.py --batch {batch} ({batch} traversal pairs per launch)",
               "source": "rocprofv3 --kernel-trace --pmc SQ_*, two passes of 8 counters (tests/tools/profile_round.sh); SQ_* cycle counters are in quad-cycles",
               "per_launch_mean": m, "derived": d}, open(f"{P}/{tag}_pmc_sq_traverse_b{batch}.json", "w"), indent=1)
# what bench.py prices the ceilings with: the per-pair figures of the batch that keeps every wavefront in its loop (launch-time work amortised)
big = sq_all[65536]
json.dump({"kernel": "k_mccfr_traverse", "commit": commit, "source_sha256": source_sha256,
           "source": f"SQ counter passes at B=65536 (profiles/{tag}_pmc_sq_traverse_b65536.json); B=4096 figures beside them",
           "valu_instr_per_pair": big["valu_instr_per_pair"], "valu_busy_cycles_per_pair": big["valu_busy_cycles_per_pair"],
           "lds_array_cycles_per_pair": big["lds_array_cycles_per_pair"],
           "lds_instr_per_pair": big["lds_instr_per_pair"],
           "b4096": {k: sq_all[4096][k] for k in ("valu_instr_per_pair", "valu_busy_cycles_per_pair", "lds_array_cycles_per_pair", "lds_instr_per_pair")}},
          open(f"{P}/traverse_sq.json", "w"), indent=1)
print(open(f"{P}/traverse_sq.json").read())


# 4. SDCFR traversal kernels (tests/tools/profile_sdcfr.sh, run per batch; its outputs moved to gpurun_out/prof_sdcfr_b<batch>)
sd_sha = source_fingerprint("scopa_sdcfr.hip")
sd_commit = subprocess.run(["git", "-C", R, "log", "-1", "--format=%h", "--", "scopa_amd/csrc/scopa_sdcfr.hip"], capture_output=True, text=True).stdout.strip()


def sd_counters(path, kernel):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {c: st.mean(v) for c, v in acc.items()}, (min(len(v) for v in acc.values()) if acc else 0)


def sd_stats(path, kernel):
    # every instantiation whose name contains `kernel` (k_sdcfr_walk<2, 0> and <2, 1>: one per traverser), averaged over their calls
    ns, calls = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel in r["Name"]:
            ns += float(r["AverageNs"]) * int(r["Calls"]); calls += int(r["Calls"])
    return (ns / calls / 1e3, calls) if calls else (None, 0)


for batch in (4096, 32768):
    d = os.path.join(os.path.dirname(src), f"prof_sdcfr_b{batch}")
    if not os.path.isdir(d):
        continue
    # ---- 4a. the forward-per-visit kernel ----
    m, n_disp = {}, 0
    for part in ("a", "b", "c"):
        mm, n_disp = sd_counters(f"{d}/visit/{part}/sdcfr_counters.csv", "k_sdcfr_traverse")
        m.update(mm)
    waves_working = min(batch // 4, 12 * 256)                            # tasks of 4 traversals; 12 wavefronts x 256 compute units at most
    t_us, calls = sd_stats(glob.glob(f"{d}/visit/stats/*kernel_stats.csv")[0], "k_sdcfr_traverse")
    cyc = 4.0 * m["SQ_WAVE_CYCLES"] / waves_working
    derived = {
        "kernel_avg_us_under_kernel_trace": t_us, "calls_under_kernel_trace": calls,
        "mfma_busy_cycles_per_simd": m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0, "mfma_flop_per_launch": 512.0 * m["SQ_INSTS_VALU_MFMA_MOPS_F32"],
        "cycles_per_working_wave": cyc, "mfma_pipe_busy_share_of_a_working_wave_s_lifetime": (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / cyc,   # a SIMD's matrix pipe against the lifetime of the wavefront(s) it hosts
        "wave_time_issuing": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], "wave_time_waiting_to_issue": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        "wave_time_waiting_on_counters_or_barriers": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        "valu_instr_per_traversal": m["SQ_INSTS_VALU"] / batch, "mfma_instr_per_traversal": m["SQ_INSTS_MFMA"] / batch,
        "lds_instr_per_traversal": m["SQ_INSTS_LDS"] / batch, "lds_array_cycles_per_cu": m["SQ_LDS_IDX_ACTIVE"] / 256.0,
        "lds_bank_conflict_share_of_lds_active": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"],
        "vmem_reads_per_working_wave": m["SQ_INSTS_VMEM_RD"] / waves_working, "vmem_writes_per_working_wave": m["SQ_INSTS_VMEM_WR"] / waves_working,
        "working_waves": waves_working,
    }
    json.dump({"kernel": "k_sdcfr_traverse (a forward pass per visit, scopa_sdcfr_mode 1)", "commit": sd_commit, "source_sha256": sd_sha,
               "workload": f"SCOPA_SDCFR_MODE=1 bench.py --workload sdcfr --batch {batch} ({batch} traversals per launch, launches of both traversers averaged)",
               "dispatches": n_disp,
               "source": "rocprofv3 --kernel-trace --pmc SQ_*, three passes of 8 counters (tests/tools/profile_sdcfr.sh); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in "
                         "quad-cycles, SQ_LDS_IDX_ACTIVE and SQ_VALU_MFMA_BUSY_CYCLES in cycles (MI355X_MICROARCH.md)",
               "per_launch_mean": m, "derived": derived}, open(f"{P}/{tag}_pmc_sq_sdcfr_traverse_b{batch}.json", "w"), indent=1)
    open(f"{P}/{tag}_sdcfr_kernel_stats_forward_per_visit_b{batch}.csv", "w").write(open(glob.glob(f"{d}/visit/stats/*kernel_stats.csv")[0]).read())
    print(batch, "per visit", json.dumps(derived, indent=1))
    # ---- 4b. the default form: k_sdcfr_policy + k_sdcfr_walk ----
    w, n_w = {}, 0
    for part in ("a", "b"):
        mm, n_w = sd_counters(f"{d}/table/{part}/sdcfr_counters.csv", "k_sdcfr_walk")
        w.update(mm)
    pol = {}
    for part in ("a", "b"):
        pol.update(sd_counters(f"{d}/table/{part}/sdcfr_counters.csv", "k_sdcfr_policy")[0])
    stats_csv = glob.glob(f"{d}/table/stats/*kernel_stats.csv")[0]
    walk_us, walk_calls = sd_stats(stats_csv, "k_sdcfr_walk")
    pol_us, _ = sd_stats(stats_csv, "k_sdcfr_policy")
    hb = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for k in ("k_sdcfr_walk", "k_sdcfr_policy"):
            hb[(ctr, k)] = sd_counters(f"{d}/table/pmc_{ctr}/sdcfr_counters.csv", k)[0].get(ctr, 0.0)
    rows_bytes = 41 * 200 * batch
    line = json.loads(open(f"{d}/table/stats.json").read().strip().splitlines()[-1])
    wd = {"walk_avg_us_under_kernel_trace": walk_us, "policy_avg_us_under_kernel_trace": pol_us, "calls": walk_calls,
          "walk_valu_instr_per_traversal": w["SQ_INSTS_VALU"] / batch, "walk_lds_instr_per_traversal": w["SQ_INSTS_LDS"] / batch,
          "walk_wave_time_issuing": w["SQ_ACTIVE_INST_ANY"] / w["SQ_WAVE_CYCLES"], "walk_wave_time_waiting_to_issue": w["SQ_WAIT_INST_ANY"] / w["SQ_WAVE_CYCLES"],
          "walk_wave_time_waiting_on_counters_or_barriers": w["SQ_WAIT_ANY"] / w["SQ_WAVE_CYCLES"],
          "walk_valu_busy_share_of_simd_time": 4.0 * w["SQ_ACTIVE_INST_VALU"] / 1024.0 / (walk_us * 1e-6 * 2.4e9) if walk_us else None,
          "walk_lds_array_busy_share": w["SQ_LDS_IDX_ACTIVE"] / 256.0 / (walk_us * 1e-6 * 2.4e9) if walk_us else None,
          "policy_mfma_flop_per_launch": 512.0 * pol.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)}
    json.dump({"kernels": "k_sdcfr_policy + k_sdcfr_walk (the default form of scopa_sdcfr_traverse_fused)", "commit": sd_commit, "source_sha256": sd_sha,
               "workload": f"bench.py --workload sdcfr --batch {batch}", "dispatches": n_w,
               "per_launch_mean_walk": w, "per_launch_mean_policy": pol, "derived": wd}, open(f"{P}/{tag}_pmc_sq_sdcfr_walk_b{batch}.json", "w"), indent=1)
    fw, ww = hb[("FETCH_SIZE", "k_sdcfr_walk")], hb[("WRITE_SIZE", "k_sdcfr_walk")]
    fp, wp = hb[("FETCH_SIZE", "k_sdcfr_policy")], hb[("WRITE_SIZE", "k_sdcfr_policy")]
    json.dump({"kernels": "k_sdcfr_policy + k_sdcfr_walk", "batch": batch, "commit": sd_commit, "source_sha256": sd_sha,
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tests/tools/profile_sdcfr.sh)",
               "walk_FETCH_SIZE_KB_raw": fw, "walk_WRITE_SIZE_KB": ww, "policy_FETCH_SIZE_KB_raw": fp, "policy_WRITE_SIZE_KB": wp,
               "bytes_per_launch": (2 * (fw + fp) + ww + wp) * 1024.0, "bytes_per_launch_uncorrected": (fw + fp + ww + wp) * 1024.0,
               "memory_rows_bytes_per_launch": rows_bytes,
               "note": "per player's traversal call (both launches).  The necessary HBM traffic is the memory rows (41 rows x 200 B per traversal: features and regrets; the mask is a view of the features); policy table, node table and "
                       "frontier are LDS-resident; FETCH_SIZE doubled per the guide's gfx950 correction for wide coalesced reads"},
              open(f"{P}/sdcfr_hbm_traffic_b{batch}.json", "w"), indent=1)
    open(f"{P}/{tag}_sdcfr_kernel_stats_b{batch}.csv", "w").write(open(stats_csv).read())
    open(f"{P}/{tag}_bench_sdcfr_b{batch}_under_rocprof_stats.json", "w").write(json.dumps(line) + "\n")
    print(batch, "table", json.dumps(wd, indent=1), json.dumps({"bytes_per_launch": (2 * (fw + fp) + ww + wp) * 1024.0, "rows": rows_bytes}))

# the names round 2's review asked for (the B = 4096 files: BASELINE configs[3]'s batch)
import shutil
for a, b in ((f"{P}/{tag}_pmc_sq_sdcfr_traverse_b4096.json", f"{P}/{tag}_pmc_sq_sdcfr_traverse.json"), (f"{P}/{tag}_sdcfr_kernel_stats_b4096.csv", f"{P}/{tag}_sdcfr_kernel_stats.csv")):
    if os.path.exists(a):
        shutil.copyfile(a, b)
