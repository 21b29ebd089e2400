"""Sub-records of the default `python bench.py --gpus 1` line: the kernels of the hot path that the headline workload (BASELINE configs[1])
does not exercise, measured in the same process, each a few launches long, each with its own `roofline {achieved, peak, unit, frac, traffic}`.

    many_deals         k_cfr_exact_lanes: the reference's vanilla CFR (vanilla_cfr.py:56-110), one deal per lane, 131 072 deals x 5 iterations --
                       the kernel SURVEY 8(d)'s 192 B/visit HBM model binds (north_star's "thousands of independent games in HBM")
    sdcfr_large_batch  k_sdcfr_policy + k_sdcfr_walk at 32 768 traversals per player (the per-GPU shard of a 256k batch), traversal only
    state_engines      k_step_batch / k_team_step_batch / k_full_step_batch: packed games advanced one ply per launch
    evaluator          k_eval_tabular_step: episodes of "average policy vs uniform random" in lockstep

Timing: HIP events recorded on the stream the kernels are launched on (every context here is created on a torch stream, and the events are
that stream's).  `traffic` = HBM bytes per launch from the committed PMC passes named in `traffic_source` (separate FETCH_SIZE / WRITE_SIZE
passes, corrected as MI355X_MICROARCH.md's HBM section prescribes), or null where no pass of that shape was taken."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0
CFR_VISITS_PER_ITERATION = 3306        # 2 traversals x 1 653 decision nodes (SURVEY 8d check value)
CFR_ALG_BYTES_PER_VISIT = 192.0        # state 32 + regret row 32 + local_strategy read/write 64 + traverser half x 128


def _profile(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def _stale(prof, source):
    """True when `source` (a file under csrc/) has changed, comments and whitespace aside, since the PMC pass `prof` was taken."""
    if not prof:
        return None
    try:
        from scopa_amd.build import source_fingerprint
        return prof.get("source_sha256") != source_fingerprint(source)
    except OSError:
        return None


def _events(torch, stream, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    fn()
    e1.record(stream)
    e1.synchronize()
    return 1e-3 * e0.elapsed_time(e1)


def many_deals(device=0, deals=131072, iters=5, reps=3):
    import torch
    from scopa_amd import _lib
    stream = torch.cuda.Stream(device=device)
    ctx = _lib.Context(device, stream=stream.cuda_stream)
    m = _lib.MultiDeal(ctx, deals)
    try:
        m.deal_py_seeds(np.arange(deals))
        m.build()
        m.cfr_exact_iterate_lanes(1)                                   # packs the row image, loads the kernel
        d0, _ = m.counters()
        secs = [_events(torch, stream, lambda: m.cfr_exact_iterate_lanes(iters)) for _ in range(reps)]
        d1, _ = m.counters()
        assert d1 - d0 == deals * iters * reps * CFR_VISITS_PER_ITERATION, "k_cfr_exact_lanes visit counter"
        e = m.exploitability()
    finally:
        m.close()
        ctx.close()
    best, med = min(secs), sorted(secs)[len(secs) // 2]
    visits = deals * iters * CFR_VISITS_PER_ITERATION
    alg = visits * CFR_ALG_BYTES_PER_VISIT
    tr = _profile("lanes_hbm_traffic.json")
    same_shape = bool(tr) and tr.get("deals") == deals and tr.get("iterations") == iters
    traffic = tr.get("bytes_per_launch") if same_shape else None
    return {"kernel": "k_cfr_exact_lanes",
            "workload": f"the reference's vanilla CFR (vanilla_cfr.py:56-110), {deals} deals (seeds 0..{deals - 1}) x {iters} iterations in one launch, one deal per lane, "
                        "64-byte regret|strategy rows gathered from HBM; per deal bit-identical to the reference (tests/test_gpu_multi.py)",
            "deals": deals, "iterations_per_launch": iters, "launches_timed": reps, "kernel_avg_us": 1e6 * sum(secs) / len(secs), "kernel_seconds": secs,
            "visits_per_s": visits / med, "deal_iterations_per_s": deals * iters / med, "row_image_resident_GB": deals * 1653 * 64 / 1e9,
            "mean_exploitability_after": float(e[:, 0].mean()), "dtype": "f64",
            "roofline": {"bound": "hbm", "achieved": alg / med / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / med / 1e9 / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_GBps": (traffic / med / 1e9) if traffic else None,
                         "traffic_source": {k: tr.get(k) for k in ("source", "commit", "correction", "calibration")} if same_shape else None,
                         "profile_stale": _stale(tr, "scopa_multi.hip") if same_shape else None,
                         "algorithmic_bytes_per_visit": CFR_ALG_BYTES_PER_VISIT, "algorithmic_bytes_per_launch": alg,
                         "best_launch": {"seconds": best, "achieved": alg / best / 1e9, "frac": alg / best / 1e9 / HBM_PEAK_GBPS},
                         "note": "achieved = SURVEY 8(d)'s 192 B per decision visit x 3 306 visits x deals x iterations / the median launch.  The kernel moves fewer bytes "
                                 "than that price: local_strategy is implied by regret_sum (one 64-byte row fetch per visit, one 64-byte store per traverser visit = 96 B), "
                                 "so `traffic` is the figure to hold against the 8 TB/s"}}


def sdcfr_large_batch(device=0, batch=32768, launches=8):
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    torch.manual_seed(0)
    so = os.dup(1)
    os.dup2(2, 1)                                                      # the constructor prints the reference's "Estimated input dimension" line
    try:
        d = DeepCFR(load_game("mini_scopa"), device=f"cuda:{device}", batch=batch)
    finally:
        sys.stdout.flush()
        os.dup2(so, 1)
        os.close(so)
    ctx = d._engine.ctx
    for p in range(2):
        for _ in range(2):
            d._traverse_batch(p, batch, sync=False)
    d._stream.synchronize()
    v0 = ctx.sdcfr_visits()
    d.kernel_events = []
    for _ in range(launches):
        for p in range(2):
            d._traverse_batch(p, batch, sync=False)
    d._stream.synchronize()
    ms = [a.elapsed_time(b) for a, b in d.kernel_events]
    d.kernel_events = None
    assert ctx.sdcfr_visits() - v0 == (105 + 82) * batch * launches
    by_player = [sum(ms[p::2]) / len(ms[p::2]) * 1e-3 for p in range(2)]
    kern_s = sum(by_player) / 2.0
    row_bytes = d.advantage_nets[0].buffer.row_bytes
    rows_b = 41 * row_bytes * batch
    v_launch = (105 + 82) / 2.0 * batch
    tr = _profile(f"sdcfr_hbm_traffic_b{batch}.json")
    traffic = tr.get("bytes_per_launch") if tr else None
    del d
    torch.cuda.empty_cache()
    return {"kernel": "k_sdcfr_policy + k_sdcfr_walk", "workload": f"SDCFR traversal only, {batch} external-sampling traversals per player per call (two launches), "
            "memory rows into the device ring; nets as initialised (torch.manual_seed(0))", "batch": batch, "calls_timed": len(ms),
            "kernel_avg_us": 1e6 * kern_s, "kernel_avg_us_by_traverser": [1e6 * x for x in by_player], "visits_per_s": v_launch / kern_s, "dtype": "f32",
            "roofline": {"bound": "hbm", "achieved": rows_b / kern_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": rows_b / kern_s / 1e9 / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_GBps": (traffic / kern_s / 1e9) if traffic else None,
                         "traffic_source": {k: tr.get(k) for k in ("source", "commit")} if tr else None, "profile_stale": _stale(tr, "scopa_sdcfr.hip"),
                         "algorithmic_bytes_per_launch": rows_b,
                         "note": f"achieved = the memory rows a call must write (41 rows x {row_bytes} B per traversal) / the call's two launches, HIP events on the solver's stream; "
                                 "SURVEY 8(d)'s 412 B/visit is not a bound here (features, masks and advantages never reach HBM: the policy table is LDS-resident)"}}


def state_engines(device=0, n_mini=1 << 24, n_team=1 << 24, n_full=1 << 23):   # the sizes tests/tools/stats_extra.sh and pmc_state_engines.sh profile
    import torch
    from scopa_amd import _lib
    from benchmarks import state_engines_bench as seb
    stream = torch.cuda.Stream(device=device)
    ctx = _lib.Context(device, stream=stream.cuda_stream)
    try:
        out = seb.measure(ctx, n_mini, n_team, n_full, 1024, stream, device)
    finally:
        ctx.close()
    pmc = _profile("pmc_state_engines.json") or {}
    for key, kern in (("mini", "k_step_batch"), ("team", "k_team_step_batch"), ("full", "k_full_step_batch")):
        r = out.get(key)
        if not r:
            continue
        c = pmc.get(kern) or {}
        per_step = (c.get("fetch_bytes_per_game_step_raw_x2", 0) + c.get("write_bytes_per_game_step", 0)) or None
        wg = r["whole_game"]
        r["roofline"] = {"bound": "hbm", "achieved": wg["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": wg["frac"],
                         "traffic": per_step * r["games"] if per_step else None,
                         "traffic_source": "profiles/pmc_state_engines.json (FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE, per game-step; "
                                           "tests/tools/pmc_state_engines.sh) x games per launch" if per_step else None,
                         "best_ply_frac": r["best_ply"]["frac"],
                         "note": "achieved = algorithmic bytes per game-step (state in + action + state out) x games x plies / the whole game's launches"}
    return out


def evaluator(device=0, episodes=1 << 22, cfr_iterations=200):
    import torch
    from scopa_amd import _lib
    from benchmarks import eval_bench
    stream = torch.cuda.Stream(device=device)
    ctx = _lib.Context(device, stream=stream.cuda_stream)
    try:
        ctx.set_deal(_lib.deal_py_seed(42))
        ctx.cfr_exact_iterate(cfr_iterations)
        pol = torch.as_tensor(np.ascontiguousarray(ctx.exploitability(return_policy=True)["policy"], np.float64), device=f"cuda:{device}")
        eval_bench.measure_kernels(ctx, pol, 4096, stream, device)     # kernel load
        by_form, states = eval_bench.measure_kernels(ctx, pol, episodes, stream, device)
        b = states.view(torch.uint8).view(episodes, 16)               # scopa_state: ncap at bytes 12, 13; scopas at 14, 15
        r = b[:, 12:14].to(torch.float64) + 2.0 * b[:, 14:16].to(torch.float64)
        half = episodes // 2
        reward = float(((r[:half, 0] - r[:half, 1]).sum() + (r[half:, 1] - r[half:, 0]).sum()) / 2.0 / episodes)   # evaluate_game: own points - mean of both
    finally:
        ctx.close()
    f = by_form["integer thresholds per infoset (scopa_eval_tabular_prepare)"]
    c = (_profile("pmc_eval.json") or {}).get("thresholds") or {}
    per_ply = (c.get("fetch_bytes_per_episode_ply_raw_x2", 0) + c.get("write_bytes_per_episode_ply", 0)) or None
    return {"kernel": "k_eval_tabular_step", "workload": f"{episodes} episodes of 'average policy after {cfr_iterations} vanilla-CFR iterations vs uniform random', seats swapped at "
            "half time, eight launches (one per ply), one lane per episode", "episodes": episodes, "episodes_per_s": f["episodes_per_s"],
            "seconds_8_launches": f["seconds_8_launches"], "reward_vs_random": reward,
            "episodes_per_s_one_launch_match": by_form["one launch: walks over the deal's tree nodes, statistics summed in the kernel (scopa_eval_tabular_match)"]["episodes_per_s"],
            "by_sampling_form": by_form,
            "roofline": {"bound": "hbm", "achieved": f["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": f["frac_of_hbm_peak"],
                         "traffic": per_ply * episodes if per_ply else None,
                         "traffic_source": "profiles/pmc_eval.json (FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE, per episode-ply; tests/tools/pmc_eval.sh) x episodes per launch" if per_ply else None,
                         "valu_busy_share": c.get("valu_busy_share"), "algorithmic_bytes_per_episode_ply": 44,
                         "note": "achieved = 44 B per episode-ply (16-byte state in and out, 4-byte tree index in and out, 4-byte seat) x episodes x 8 / the eight launches; "
                                 "the 32-byte policy rows come from cache (measured traffic = 44 B).  The launch is VALU-bound, not HBM-bound: the vector unit is busy 0.97 of the time (245 instructions per wavefront: "
                                 "the move itself + Philox4x32-10 + the 64-bit threshold compares)"}}


def all_records(device=0):
    """Every sub-record, each guarded: a failing one reports its error instead of costing the headline line."""
    out = {}
    for name, fn in (("many_deals", many_deals), ("sdcfr_large_batch", sdcfr_large_batch), ("state_engines", state_engines), ("evaluator", evaluator)):
        try:
            out[name] = fn(device)
        except Exception as e:   # a side measurement, never a reason to lose the GPU line
            out[name] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["many_deals", "sdcfr_large_batch", "state_engines", "evaluator"]
    print(json.dumps({w: globals()[w]() for w in which}))
