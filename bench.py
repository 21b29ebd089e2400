#!/usr/bin/env python3
"""bench.py -- MiniScopa infoset-traversals/sec (BASELINE.json metric) on N MI355X of one node.

A "step" is one batched external-sampling MCCFR iteration: `batch` traversals per traverser per GPU against the
iteration's frozen tables (k_mccfr_traverse), [N>1: one sum-all-reduce of the [738][5] float64 delta], apply.
N=1 workload = BASELINE configs[1] ("External-sampling MCCFR, 4096 parallel traversals, 1 MI355X").
An infoset-traversal = one decision-node visit: 463 per traversal pair, counted exactly by the kernel.

    python bench.py                                   # N = 1, defaults finish within ~1.5 min (most of it the CPU baseline)
    python bench.py --gpus 4 --steps 20 --warmup 5    # spawns its 4 ranks itself (fresh child processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # or under torchrun: RANK / LOCAL_RANK / WORLD_SIZE from the env

Timing protocol (what the one JSON line reports).  The timed region is EXACTLY `--steps` iterations bracketed by
barrier + device synchronisation on both sides, max over ranks.  A region of 20 iterations is ~0.5 ms, and on a GPU that
has just idled through process start-up the first such region runs at half speed (clock and queue ramp), so:
(1) a fixed pre-phase of >= 0.4 s of iterations runs first, whatever --warmup says; (2) then --warmup iterations; (3) the
region is then timed R times back to back (R = 31, fewer only if a region is so long that 31 would exceed ~20 s) and the
MEDIAN region is reported (`ms_per_step`, `value`), with min / max / first beside it in `timing`.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VISITS_PER_PAIR = 463          # 291 + 172 decision visits per (traverser 0, traverser 1) traversal pair
ALG_BYTES_PER_VISIT = 111.6    # 32 B state + 32 B regret row + 0.3715 * 128 B table RMW
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
CLOCK_HZ = 2.4e9               # MI355X_MICROARCH.md: max clock
N_CUS, SIMDS_PER_CU = 256, 4
# The traversal kernel's LDS work counted from its DESIGN, not from its binary ("floor"): per traversal pair
#   6 ply rounds over 2,6,10,25,40,80 unique nodes (7 wave-instructions: ply 5 needs two) x {parent record by cross-lane read, infoset id,
#   threshold row (16 B), draw word, record store}                                                                    = 35
#   the pair's 58 Philox blocks stored as 4 draw arrays                                                                =  4
#   leaf stage: 240 leaves (4 wave-instructions) x {ply-5 record, payoff byte, payoff store}                           = 12
#   update step: 52 traverser nodes (1 wave-instruction) x {5 ancestor records, 5 ancestor sigmas, own sigma row (2 x 16 B), leaf
#   payoffs, 4 regret adds (ds_add_f64), visit count add}                                                               = 18
# and the LDS-array cycles those cost when conflict-free (MI355X_MICROARCH.md LDS table: 2 per 4/8-byte access, 4 per 16-byte one).
LDS_FLOOR_INSTR_PER_PAIR = 69
LDS_FLOOR_CYCLES_PER_PAIR = 164
REF_PY_VISITS_PER_S = 9300.0   # reference Python MCCFR, 1 Xeon core, survey container
PRE_PHASE_S = 0.4
REGIONS = 31


def cpu_baseline(batch, target_s=12.0):
    """The oracle's batched MCCFR (same workload, same RNG keying) on ONE host core, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    t = O.Tree(seed=42)
    R, S, _ = t.tables()
    sample_batch = min(batch, 1024)
    visits, iters, t0 = 0, 0, time.perf_counter()
    while True:
        visits += t.mccfr_batched(R, S, 0x5C09A, iters, 1, sample_batch)
        iters += 1
        dt = time.perf_counter() - t0
        if dt >= target_s or iters >= 10000:
            break
    return {"value": visits / dt, "unit": "infoset-traversals/s", "cores": 1, "kind": "port",
            "sample": f"{iters} iterations x {sample_batch} traversals/traverser of the same MCCFR workload "
                      f"(oracle/scopa_oracle.c og_mccfr_batched, {visits} visits in {dt:.1f} s)",
            "reference_python_visits_per_s": REF_PY_VISITS_PER_S,
            "reference_python_note": "rug-marl-group2/scopa MCCFRTrainer on 1 Xeon core, measured in the survey container (BASELINE.md); the reference cannot run on the GPU box"}


def _cpu_worker(args):
    """One oracle replica on one core (spawned process: no GPU state is inherited)."""
    batch, target_s, salt = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    t = O.Tree(seed=42)
    R, S, _ = t.tables()
    visits, iters, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < target_s:
        visits += t.mccfr_batched(R, S, 0x5C09A + salt, iters, 1, batch)
        iters += 1
    return visits, time.perf_counter() - t0


def cpu_baseline_all_cores(batch, target_s=6.0):
    """The same oracle workload as independent replicas on every host core: the reference is
    single-threaded, so 'all cores' can only mean replicas."""
    import multiprocessing as mp
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n = max(1, min(avail, 16))  # a 1-GPU box's CPU share is 16 cores
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(_cpu_worker, [(min(batch, 1024), target_s, i) for i in range(n)])
    rate = sum(v / dt for v, dt in res)
    return {"value": rate, "unit": "infoset-traversals/s", "cores": n, "kind": "port",
            "sample": f"{n} independent oracle replicas x {target_s:.0f} s of the same MCCFR workload"}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this process has made no GPU
    call and makes none), wire them like torchrun would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), pass rank 0's stdout --
    the one JSON line -- through, and exit non-zero if any rank failed (the others are then stopped by PID)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, deadline = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and rc == 0:
                rc = p.returncode
                deadline = time.time() + 20.0          # the others normally follow (a failed collective); then they are stopped
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    for p in procs:
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    return rc


def rank_roster(world, rank, local, per_rank_ms=None):
    """What lets a reader of the JSON line see N real ranks: every rank's (rank, pid, device index, device uuid / PCI bus id, device
    name[, its own ms_per_step]), gathered through the process group; world_size is the group's own answer."""
    import torch
    import torch.distributed as dist
    p = torch.cuda.get_device_properties(local)
    uuid = getattr(p, "uuid", None)
    me = {"rank": rank, "pid": os.getpid(), "device_index": local, "device_uuid": str(uuid) if uuid is not None else None,
          "pci_bus_id": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0)),
          "device_name": p.name, "host": socket.gethostname()}
    if per_rank_ms is not None:
        me["ms_per_step"] = per_rank_ms
    if world <= 1 or not dist.is_initialized():
        return {"world_size": 1, "ranks": [me]}
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, me)
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": got,
            "distinct_devices": len({(g["host"], g["pci_bus_id"], g["device_uuid"]) for g in got})}


def cpu_baseline_sdcfr(nets, target_s=10.0):
    """The oracle's SDCFR traversal (og_sdcfr_traverse: one traversal after the other, a batch-1 MLP forward per node, as the
    reference does) with the SAME weights on one host core, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    t = O.Tree(seed=42)
    visits, n, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < target_s:
        for trav in (0, 1):
            visits += t.sdcfr_traverse(nets, trav, seed=0x5C09A, iteration=n, b0=0, nb=64)[4]
        n += 1
    dt = time.perf_counter() - t0
    return {"value": visits / dt, "unit": "infoset-traversals/s", "cores": 1, "kind": "port",
            "sample": f"{n} x 64 traversals per player of the same SDCFR traversal with the same weights (oracle/scopa_oracle.c og_sdcfr_traverse, "
                      f"{visits} visits in {dt:.1f} s); traversal only, no training",
            "reference_python_visits_per_s": "930-2900 (BASELINE.md section 2: reference DeepCFR._external_sampling_cfr, 1 Xeon core)"}


def run_sdcfr(args, emit=True):
    """--workload sdcfr: BASELINE configs[3] (N = 1) / configs[4] (N > 1).  A step is one iteration of DeepCFR.train
    (deep_cfr.py:431-495) without its evaluation: per player, `batch` external-sampling traversals in one launch of
    k_sdcfr_traverse filling the device-resident memory ring, then the advantage net's Adam epochs on PyTorch-ROCm
    (N > 1: traversal ids sharded by rank, one flat gradient all-reduce per optimiser step).
    emit=False (the default `python bench.py` run, N = 1): returns the record instead of printing it -- it becomes the "sdcfr"
    sub-record of the one JSON line, measured in the same process after the MCCFR measurement."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the solver path has no CPU fallback")
    if local >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants GPU {local}, the node shows {torch.cuda.device_count()} (use --share-gpu to rehearse on fewer GPUs)")
    torch.cuda.set_device(local)
    saved_stdout = None
    if world > 1:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if args.share_gpu:   # rehearsal: all ranks on device 0, gloo carries the gradient all-reduce (RCCL refuses two ranks on one device)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    batch = args.batch
    torch.manual_seed(0)
    _so = os.dup(1)
    os.dup2(2, 1)                                              # the constructor prints the reference's "Estimated input dimension" line
    d = DeepCFR(load_game("mini_scopa"), device=f"cuda:{local}", batch=batch, rank=rank, world=world, graph_training=(world == 1),
                train_backend=(args.sdcfr_train_backend if world == 1 else "torch"))
    sys.stdout.flush()
    os.dup2(_so, 1)
    ctx = d._engine.ctx
    if os.environ.get("SCOPA_SDCFR_MODE"):                                        # kernel experiments: 1 = a forward pass per visit in the timed region too
        ctx.sdcfr_mode(int(os.environ["SCOPA_SDCFR_MODE"]))
    if os.environ.get("SCOPA_SDCFR_T") or os.environ.get("SCOPA_SDCFR_W"):       # kernel experiments: task shape of the traversal kernels
        ctx.sdcfr_tuning(int(os.environ.get("SCOPA_SDCFR_T", "0")), int(os.environ.get("SCOPA_SDCFR_W", "0")))
    epochs = args.sdcfr_epochs

    ahead = [None]

    def step():                                                  # as DeepCFR.train: the iteration is queued on the solver's stream, the host stays one iteration ahead
        q = d._queue_iteration(epochs, args.sdcfr_train_batch)
        if ahead[0] is not None:
            d._resolve(ahead[0])
        ahead[0] = q

    def fence():
        if ahead[0] is not None:
            d._resolve(ahead[0])
            ahead[0] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # pre-phase: at least 20 iterations and (single GPU) at least PRE_PHASE_S -- a fixed count where ranks must stay in step
    n_pre, t0 = 0, time.perf_counter()
    while n_pre < 20 or (world == 1 and time.perf_counter() - t0 < PRE_PHASE_S):
        step()
        n_pre += 1
    for _ in range(args.warmup):
        step()
    fence()
    v0 = ctx.sdcfr_visits()
    d.kernel_events = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    own_ms = 1e3 * elapsed / args.steps                          # this rank's own clock; `elapsed` becomes the maximum over ranks below
    if world > 1:
        tm = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.share_gpu else f"cuda:{local}")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm.item())
        # replicas must have stayed identical: the same averaged gradients, the same Adam steps
        flat = torch.cat([p.detach().reshape(-1) for a in d.advantage_nets for p in a.net.parameters()]).double()
        h = torch.tensor([float(flat.sum().item()), float((flat * flat).sum().item())], dtype=torch.float64, device="cpu" if args.share_gpu else f"cuda:{local}")
        hs = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(hs, h)
        replicas_identical = all(bool(torch.equal(x, hs[0])) for x in hs)
        assert replicas_identical, "SDCFR replicas' advantage nets differ after the run"
    else:
        replicas_identical = None
    roster = rank_roster(world, rank, local, own_ms)
    visits = (ctx.sdcfr_visits() - v0) * world
    assert visits == (105 + 82) * batch * args.steps * world
    kern_ms = [a.elapsed_time(b) for a, b in d.kernel_events]
    d.kernel_events = None
    # beside the timed region: the same traversal with a forward pass per VISIT (k_sdcfr_traverse, scopa_sdcfr_mode 1), for its MFMA
    # roofline -- the default path evaluates every decision node once per launch and is bound by the memory rows instead
    pv_ms = []
    if os.environ.get("SCOPA_SDCFR_MODE") != "1":
        ctx.sdcfr_mode(1)
        for a in d.advantage_nets:
            a.buffer.total = 0
        d.kernel_events = []
        for _ in range(6):
            for p in range(2):
                d._traverse_batch(p, batch, sync=False)
        d._stream.synchronize()
        pv_ms = [a.elapsed_time(b) for a, b in d.kernel_events][4:]
        d.kernel_events = None
        ctx.sdcfr_mode(0)
    per_visit_default = os.environ.get("SCOPA_SDCFR_MODE") == "1"
    # beside the timed region too: the same iteration with the OPT-IN hand-written optimiser step (train_backend="hip": two launches per Adam step
    # instead of PyTorch's thirty) -- the default trains on PyTorch-ROCm, as north_star asks
    hip_train = None
    if world == 1 and args.sdcfr_train_backend == "torch" and args.sdcfr_train_batch % 16 == 0:
        _so2 = os.dup(1); os.dup2(2, 1)
        d2 = DeepCFR(load_game("mini_scopa"), device=f"cuda:{local}", batch=batch, train_backend="hip")
        sys.stdout.flush(); os.dup2(_so2, 1)

        ahead2 = [None]

        def step2():
            q = d2._queue_iteration(epochs, args.sdcfr_train_batch)
            if ahead2[0] is not None:
                d2._resolve(ahead2[0])
            ahead2[0] = q
        for _ in range(10):
            step2()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(20):
            step2()
        d2._resolve(ahead2[0])
        torch.cuda.synchronize()
        hip_train = {"ms_per_step": 1e3 * (time.perf_counter() - t2) / 20, "iterations_timed": 20,
                     "what": "the same iteration with train_backend='hip' (scopa_sdcfr_train_steps: k_sdcfr_train_grad + k_sdcfr_train_adam per Adam step; tests hold it to the PyTorch step at 2e-5)"}
        del d2
    out = None
    if rank == 0:
        kern_s = 1e-3 * sum(kern_ms) / max(len(kern_ms), 1)              # one launch = one player's batch
        v_launch = (105 + 82) / 2.0 * batch                              # visits per launch, averaged over the two traversers
        flop_visit = 2.0 * (34 * 128 + 128 * 64 + 64 * 16)               # one MLP forward (27 136 FLOP) ...
        fwd_launch = (81 + 58) / 2.0 * batch                             # ... per visit that needs one: the 24 single-action opponent nodes of plies 6/7 are forced, the kernel skips them
        alg_b = 412.0
        row_bytes = d.advantage_nets[0].buffer.row_bytes                 # 200: 34 feature + 16 regret floats (the mask is a view of the features, DeviceMemory)
        rows_b = 41.0 * row_bytes * batch                                # the memory rows a launch must write
        pv_s = 1e-3 * sum(pv_ms) / max(len(pv_ms), 1) if pv_ms else (kern_s if per_visit_default else None)
        bounds = {"hbm-memory-rows": {"achieved": rows_b / kern_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                      "how": f"41 x {row_bytes} B of memory rows per traversal (34 feature + 16 regret floats; the mask is features[:16], not stored) x traversals per launch / traversal time (policy launch + walk launch): what the "
                                             "launch must write whatever the algorithm; nets, policy table, node table and frontier are LDS-resident"},
                  }
        if per_visit_default:
            bounds.pop("hbm-memory-rows")
        if pv_s:
            bounds["mfma-f32"] = {"achieved": flop_visit * fwd_launch / pv_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                  "how": "the forward-per-visit form (k_sdcfr_traverse, scopa_sdcfr_mode 1; measured beside the timed region unless it IS the timed path): 27 136 FLOP "
                                         "(one 34-128-64-16 forward) x forward passes per launch (81 / 58 of the 105 / 82 visits of a traversal) / kernel time against the f32 matrix "
                                         "peak at 2.4 GHz (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD; tiles of 16 nodes: 4.5 % more FLOP issued than counted).  The shader clock holds "
                                         "~2.15 GHz while it runs (profiles/r03_sdcfr_stamps.txt): 0.89 of this peak is the most the clock allows.  The default path does not pay "
                                         "this: it evaluates the deal's 1 653 decision nodes once per launch (k_sdcfr_policy) instead of once per visit"}
        for b in bounds.values():
            b["frac"] = b["achieved"] / b["peak"]
        top = "mfma-f32" if per_visit_default else "hbm-memory-rows"
        tr = load_profile_json(f"sdcfr_hbm_traffic_b{batch}.json") or {}     # FETCH_SIZE / WRITE_SIZE passes of the traversal launches at this batch, if taken
        traffic = tr.get("bytes_per_launch")
        try:
            from scopa_amd.build import source_fingerprint
            traffic_stale = (tr.get("source_sha256") != source_fingerprint("scopa_sdcfr.hip")) if tr else None
        except OSError:
            traffic_stale = None
        nets = np.stack([np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in a.net.state_dict().values()]) for a in d.advantage_nets])
        out = {"metric": "MiniScopa infoset-traversals/sec", "value": visits / elapsed, "unit": "infoset-traversals/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"BASELINE configs[{3 if world == 1 else 4}]: SDCFR on MiniScopa, {batch} external-sampling traversals per player per iteration per GPU "
                                      f"({'k_sdcfr_traverse fills' if per_visit_default else 'k_sdcfr_policy + k_sdcfr_walk fill'} the device memory ring), advantage MLP 34-128-64-16 f32, {epochs} Adam steps x batch {args.sdcfr_train_batch} per player on {'PyTorch-ROCm' if args.sdcfr_train_backend == 'torch' or world > 1 else 'the hand-written step'}",
                          "batch_per_gpu": batch, "global_batch": batch * world, "iterations": args.steps,
                          "parallelism": f"dp{world}" + (" + 1 gradient all-reduce of 55104 B per optimiser step (RCCL)" if world > 1 else ""),
                          "training": ("hand-written optimiser step (train_backend='hip')" if args.sdcfr_train_backend == "hip" else "HIP-graph-replayed PyTorch optimiser step") if world == 1 else "eager (gradient all-reduce between backward and step)",
                          "replicas_bit_identical": replicas_identical, "shared_gpu_rehearsal": bool(args.share_gpu)},
               "traversal_only": {"visits_per_s_per_gpu": v_launch / kern_s, "kernel_avg_us": 1e6 * kern_s, "launches_timed": len(kern_ms),
                                  "form": "forward pass per visit (k_sdcfr_traverse)" if per_visit_default else "policy table per launch + walks (k_sdcfr_policy + k_sdcfr_walk)",
                                  "forward_per_visit_kernel_avg_us": 1e6 * pv_s if pv_s else None,
                                  "forward_per_visit_visits_per_s_per_gpu": (v_launch / pv_s) if pv_s else None},
               "roofline": {"bound": top, "achieved": bounds[top]["achieved"], "peak": bounds[top]["peak"], "unit": bounds[top]["unit"], "frac": bounds[top]["frac"],
                            "traffic": traffic, "traffic_GBps": (traffic / kern_s / 1e9) if traffic else None, "profile_stale": traffic_stale,
                            "kernel": "k_sdcfr_traverse" if per_visit_default else "k_sdcfr_walk (+ k_sdcfr_policy)", "kernel_avg_us": 1e6 * kern_s, "launches_timed": len(kern_ms), "bounds": bounds,
                            "hbm_algorithmic": {"bytes_per_visit": alg_b, "GBps": alg_b * v_launch / kern_s / 1e9, "ratio_to_hbm_peak": alg_b * v_launch / kern_s / 1e9 / HBM_PEAK_GBPS,
                                                "note": "SURVEY 8(d)'s algorithmic price, 412 B per visit (state, features, mask, advantages in and out of HBM, memory rows) x visits per launch / "
                                                        "traversal time.  NOT a bound here: features, masks and advantages never exist in HBM (the policy table is LDS-resident), only the memory "
                                                        "rows do, so the ratio can exceed 1; the bound that applies is bounds.hbm-memory-rows, the traffic measured is `traffic`"},
                            "note": "time from events recorded on the kernels' stream around each player's traversal (both launches of the default form).  Default form: the "
                                    "advantage nets are frozen during a launch and a node's features depend on the tree node alone, so the deal's 1 653 decision nodes are "
                                    "evaluated once (MFMA tiles, k_sdcfr_policy) and the traversals walk the 26 KB policy table in LDS; HBM sees the 41 x 200 B memory rows per "
                                    "traversal, which is what bounds it.  Forward-per-visit form (bounds.mfma-f32): both nets as MFMA operand images in LDS, activations in "
                                    "registers; SQ counter passes: profiles/r03_pmc_sq_sdcfr_walk_b*.json (default form), profiles/r03_pmc_sq_sdcfr_traverse_b*.json"},
               "decision_visits": visits, "world": roster}
        if hip_train is not None:
            out["with_hip_training_step"] = hip_train
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sdcfr(nets)
            out["gpu_over_cpu_1core_traversal_only"] = out["traversal_only"]["visits_per_s_per_gpu"] / out["cpu_baseline"]["value"]
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        if emit:
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.sdcfr_visits()            # also asks the library whether a team barrier of the fused kernel ever gave up (raises if so)
    return out


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["mccfr", "sdcfr"], default="mccfr",
                    help="mccfr = BASELINE configs[1]/[2] (the headline); sdcfr = configs[3]/[4] (HIP traversal kernel + PyTorch-ROCm advantage MLP)")
    ap.add_argument("--steps", type=int, default=None, help="timed iterations (default 2000 for mccfr, 30 for sdcfr)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed iterations (default 100 for mccfr, 3 for sdcfr)")
    ap.add_argument("--batch", type=int, default=4096, help="traversals per traverser per GPU per iteration")
    ap.add_argument("--sdcfr-epochs", type=int, default=5, help="Adam steps per player per iteration (sdcfr workload)")
    ap.add_argument("--sdcfr-train-backend", choices=("torch", "hip"), default="torch", help="sdcfr workload: the optimiser step on PyTorch-ROCm (default) or the opt-in hand-written step")
    ap.add_argument("--sdcfr-train-batch", type=int, default=128, help="rows per Adam step (sdcfr workload; the reference trains on 128, SURVEY 8d also asks for a scaled setting of 4096)")
    ap.add_argument("--regions", type=int, default=REGIONS, help="how many times the --steps region is timed (median reported)")
    ap.add_argument("--pre-phase-s", type=float, default=PRE_PHASE_S, help="seconds of untimed iterations before anything is timed (0 for profiler passes that count every dispatch)")
    ap.add_argument("--prof-stride", type=int, default=0, help="bracket every n-th traversal launch with HIP events (0 = so that >= 16 launches are timed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sdcfr", action="store_true", help="N = 1 mccfr run: leave out the SDCFR sub-record (BASELINE configs[3], measured after the MCCFR workload in the same process)")
    ap.add_argument("--sdcfr-steps", type=int, default=20, help="timed SDCFR iterations of that sub-record")
    ap.add_argument("--exchange", choices=["auto", "p2p", "rccl"], default="auto",
                    help="N>1 delta all-reduce: library one-shot peer-memory exchange (validated against RCCL first), or torch.distributed/RCCL")
    ap.add_argument("--exchange-form", choices=["auto", "light", "fenced"], default="auto",
                    help="peer-memory exchange protocol form: light (sc0 sc1 accesses + s_waitcnt) only if it validates on this topology, else fenced")
    ap.add_argument("--inject-proof-failure", action="store_true", help="testing: treat the first sharded proof as failed, to exercise the fallback from the peer exchange to torch.distributed")
    ap.add_argument("--graph", action="store_true", help="N = 1 mccfr: replay the iteration loop as captured HIP graphs of up to 64 iterations (scopa_mccfr_graph_mode; A/B against the eager loop)")
    ap.add_argument("--force-dist", action="store_true", help="use the N>1 code path (process group, all-reduce) even with one rank")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: all ranks use device 0, the process group is gloo (RCCL refuses two ranks on one device)")
    ap.add_argument("--no-subrecords", action="store_true", help="N = 1 mccfr run: leave out the many_deals / sdcfr_large_batch / state_engines / evaluator sub-records (benchmarks/subrecords.py)")
    args = ap.parse_args()
    # the host driver only supports dmabuf IPC: ranks started by an external launcher (torchrun, the driver) must get this too, before any HIP call
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if args.steps is None:
        args.steps = 2000 if args.workload == "mccfr" else 30
    if args.warmup is None:
        args.warmup = 100 if args.workload == "mccfr" else 3
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.workload == "sdcfr":
        run_sdcfr(args)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    from scopa_amd import _lib
    from scopa_amd.distributed import ShardedMCCFR, make_gpu_engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the solver path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants GPU {local_rank}, the node shows {torch.cuda.device_count()} (use --share-gpu to rehearse on fewer GPUs)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    use_dist = world > 1 or args.force_dist
    saved_stdout = None
    if use_dist:
        # librccl prints a version banner on stdout when its first communicator comes up; stdout is reserved for the ONE JSON
        # line, so native-library output goes to stderr until that line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    coll_dev = torch.device("cpu") if args.share_gpu else dev     # where small collectives' tensors live

    perm = _lib.deal_py_seed(42)
    exchange_choice, forced_note = args.exchange, ""
    if args.share_gpu and world > 3 and args.exchange == "auto":
        # rehearsal with more than three ranks on ONE device: a rank spinning in the peer exchange's poll loop holds the device while
        # the peers' traversal launches wait behind it (the exchange assumes co-resident peers, as on one GPU per rank) -- round 2's
        # 4-rank rehearsal burnt its 5 s wait budgets before falling back.  Go to the collective directly.
        exchange_choice, forced_note = "rccl", "shared-GPU rehearsal with more than 3 ranks: the collective was chosen outright (the peer exchange needs co-resident peers)"
    ctx, delta, stream, all_reduce = make_gpu_engine(local_rank, perm, seed=0x5C09A, distributed=use_dist, rank=rank,
                                                      exchange=exchange_choice, exchange_form=args.exchange_form)
    if forced_note:
        ctx.exchange_note = forced_note
    if args.graph:
        ctx.mccfr_graph_mode(True)
    batch_total = args.batch * world
    # --force-dist with one rank still takes the exchange step (always_exchange), so the N>1 code path can be timed on one GPU
    drv = ShardedMCCFR(ctx, rank, world, all_reduce, fused_exchange=(use_dist and ctx.exchange == "p2p"), always_exchange=use_dist)

    def run(k):
        if not use_dist:
            ctx.mccfr_iterate(args.batch, k)  # in-library launch loop: traverse + apply per iteration
        else:
            drv.run(batch_total, k)           # raises (SCOPA_ETIMEOUT) if a peer did not answer: the process then exits non-zero

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def all_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # N > 1: before anything is timed, prove the sharded pipeline on THIS topology -- 10 iterations over `world` ranks must give
    # the tables of the same 10 iterations (same global traversal ids) on one GPU, up to the summation order of the rank deltas
    sharded_check = None

    def prove_sharded():
        """True on every rank iff 10 sharded iterations ran without a wait timing out and rank 0 found the one-GPU tables."""
        ok = 1
        try:
            run(10)
        except Exception as e:                                   # SCOPA_ETIMEOUT: a peer's rows did not arrive within the budget
            print(f"[bench rank {rank}] sharded proof run failed: {e}", file=sys.stderr, flush=True)
            ok = 0
        fence()
        if ok and rank == 0:
            Rn, Sn, _ = ctx.tables_get()
            ref = _lib.Context(local_rank)
            ref.set_deal(perm)
            ref.mccfr_seed(0x5C09A)
            ref.mccfr_iterate(batch_total, 10)
            R1, S1, _ = ref.tables_get()
            ref.close()
            ok = int(np.allclose(Rn, R1, rtol=1e-10, atol=1e-10) and np.allclose(Sn, S1, rtol=1e-10, atol=1e-10))
        t = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)                 # every rank learns the verdict: they switch paths together
        ctx.tables_reset()                                       # zero tables, iteration counter back to 0; the bound delta buffer stays bound (set_deal would drop it)
        fence()
        return bool(t.item())

    if use_dist:
        sharded_check = prove_sharded() and not args.inject_proof_failure
        if not sharded_check and ctx.exchange == "p2p" and exchange_choice == "auto":
            # the peer exchange passed its connect-time validation but not the solver-level proof on this topology: drop it, take the
            # torch.distributed all-reduce (split path) and prove that instead -- a slower valid number beats none
            ctx.exchange, ctx.exchange_note = "rccl", (ctx.exchange_note + "; dropped after the 10-iteration proof failed, fell back to torch.distributed").lstrip("; ")
            drv = ShardedMCCFR(ctx, rank, world, ctx.collective_all_reduce, fused_exchange=False, always_exchange=True)
            sharded_check = prove_sharded()

    # ---- (1) pre-phase: >= PRE_PHASE_S of iterations, the same count on every rank (calibrated, max over ranks) ----------------
    run(50)
    fence()
    t0 = time.perf_counter()
    run(100)
    fence()
    per_step = all_max((time.perf_counter() - t0) / 100)
    n_pre = int(min(max(args.pre_phase_s / per_step, 100), 200000))
    fence()
    t0 = time.perf_counter()
    run(n_pre)
    fence()
    pre_s = time.perf_counter() - t0
    # ---- (2) the caller's warm-up ----------------------------------------------------------------------------------------------
    run(args.warmup)
    fence()
    # ---- (3) the region: EXACTLY --steps iterations between fences, R times; the median region is the result -------------------
    per_step = all_max(pre_s / n_pre)
    regions = max(3, min(args.regions, int(20.0 / max(per_step * args.steps, 1e-9)))) if args.regions > 3 else max(1, args.regions)
    # A sampled launch carries a start and a stop event (hipExtLaunchKernelGGL) and costs the region it sits in ~18 us (measured on one box, 20-step regions:
    # median 13.65 us per step unsampled, 14.57 with 17 samples spread over the 31 regions -- more than half of them then hold one, and the median IS a sampled
    # region).  So the samples are taken in the LAST regions only -- an eighth of them, at least one, all of them timed regions like the others: the median,
    # `value`, is then an undisturbed region, and the kernel time still comes from HIP events around launches of the timed regions.
    n_sampled = min(max(1, regions // 8, -(-16 // max(args.steps, 1))), max(1, (regions - 1) // 2))   # >= 16 launches where the regions are short, always fewer than half the regions
    stride = args.prof_stride if args.prof_stride > 0 else max(1, (args.steps * n_sampled) // 16)
    d0, _ = ctx.counters()
    times, own_times = [], []
    for r_i in range(regions):
        if r_i == regions - n_sampled:
            ctx.prof_enable(stride)                             # (synchronises the stream: outside the region's clock)
        fence()
        t0 = time.perf_counter()
        run(args.steps)
        fence()
        own_times.append(time.perf_counter() - t0)             # this rank's own clock around the region ...
        times.append(all_max(own_times[-1]))                    # ... and the maximum over ranks, which is what counts
    if use_dist:
        drv.final_check()                                      # light exchange form: the replicas' tables are compared before anything is reported (no-op otherwise)
    launches, kernel_ms = ctx.prof_read()
    dev_n1, dev_ms1 = ctx.prof_device()
    phases = ctx.prof_phases()
    spread = ctx.prof_spread()
    ctx.prof_enable(0)
    d1, _ = ctx.counters()
    med = sorted(times)[len(times) // 2]
    roster = rank_roster(world if use_dist else 1, rank, local_rank, 1e3 * sorted(own_times)[len(own_times) // 2] / args.steps)

    if use_dist:
        cnt = torch.tensor([d1 - d0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        visits = int(cnt.item())
    else:
        visits = d1 - d0
    visits_per_region = VISITS_PER_PAIR * batch_total * args.steps
    assert visits == visits_per_region * regions, f"kernel visit counter {visits} != {visits_per_region * regions}"
    replicas_identical = None
    if use_dist:
        if ctx.exchange == "p2p":
            timeouts, exchanges = ctx.p2p_status()
            tt = torch.tensor([timeouts], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # every rank learns it, so every rank stops (none is left in a collective)
            assert int(tt.item()) == 0, f"peer exchange: wait(s) timed out on some rank (here: {timeouts}) -- the run is invalid"
        R, S, _ = ctx.tables_get()
        h = hashlib.sha256(R.tobytes() + S.tobytes()).digest()[:8]
        mine = torch.tensor(list(h), dtype=torch.uint8, device=coll_dev)
        allh = [torch.zeros(8, dtype=torch.uint8, device=coll_dev) for _ in range(world)]
        dist.all_gather(allh, mine)
        replicas_identical = all(bool((x == mine).all().item()) for x in allh)
        assert replicas_identical, "replicas' tables differ after the run: the exchange delivered different sums to different ranks"
        assert sharded_check is None or sharded_check, "10 sharded iterations do not reproduce the one-GPU tables"

    if rank == 0:
        kern_us = 1e3 * kernel_ms / max(launches, 1)
        kern_s = kern_us * 1e-6
        pairs_per_launch = args.batch                             # this rank's slice
        visits_per_launch = VISITS_PER_PAIR * pairs_per_launch
        alg_bytes = visits_per_launch * ALG_BYTES_PER_VISIT
        # per-pair work of k_mccfr_traverse from the SQ counters (rocprofv3 --pmc passes folded by tests/tools/fold_profiles.py)
        sq = load_profile_json("traverse_sq.json") or {}
        valu_cyc = sq.get("valu_busy_cycles_per_pair", 2411.0)    # SQ_ACTIVE_INST_VALU x 4 (quad-cycles) / pairs
        lds_cyc = sq.get("lds_array_cycles_per_pair", 510.3)      # SQ_LDS_IDX_ACTIVE / pairs
        tr = load_profile_json("hbm_traffic.json") or {}
        # the counter files were taken on ONE version of the kernel: say so when the source has moved on since
        try:
            from scopa_amd.build import source_fingerprint
            src_sha = source_fingerprint("scopa_mccfr.hip")       # comments and whitespace do not count
        except OSError:
            src_sha = None
        profile_stale = {name: (src_sha is None or prof.get("source_sha256") != src_sha) for name, prof in (("traverse_sq.json", sq), ("hbm_traffic.json", tr))}
        # the counters measured AT this run's batch, where the file has them (B = 4096): the units' busy fractions of this very launch shape
        at_batch = sq.get("b%d" % args.batch) or {}
        traffic = tr.get("bytes_per_launch") if tr.get("batch", 4096) == args.batch else None
        bounds = {}
        if launches:
            bounds = {
                "valu-issue": {"achieved": valu_cyc * pairs_per_launch / kern_s, "peak": N_CUS * SIMDS_PER_CU * CLOCK_HZ, "unit": "SIMD busy-cycles/s",
                               "how": f"{valu_cyc:.0f} VALU-busy cycles per traversal pair (SQ_ACTIVE_INST_VALU, quad-cycles x 4) x {pairs_per_launch} pairs per launch "
                                      "/ kernel time, against 1024 SIMDs x 2.4 GHz"},
                "lds": {"achieved": lds_cyc * pairs_per_launch / kern_s, "peak": N_CUS * CLOCK_HZ, "unit": "LDS-array cycles/s",
                        "how": f"{lds_cyc:.0f} LDS-array cycles per traversal pair (SQ_LDS_IDX_ACTIVE, bank-conflict cycles included) x {pairs_per_launch} pairs "
                               "/ kernel time, against 256 LDS arrays x 2.4 GHz"},
                "hbm-measured": {"achieved": (traffic / kern_s / 1e9) if traffic else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "how": "HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE PMC passes (profiles/hbm_traffic.json) / kernel time"},
            }
            for b in bounds.values():
                b["frac"] = (b["achieved"] / b["peak"]) if b["achieved"] else None
            # two readings of the same ceilings.  frac (above): the per-pair cost of the kernel's steady state (counters at B = 65536,
            # launch-time work amortised) x this launch's pairs / its time = what share of the launch the pair loop's work explains.
            # frac_busy: the counters of a launch of THIS batch (prologue, epilogue, idle wavefronts included) = how busy the unit was.
            if at_batch:
                bounds["valu-issue"]["frac_busy"] = at_batch["valu_busy_cycles_per_pair"] * pairs_per_launch / kern_s / (N_CUS * SIMDS_PER_CU * CLOCK_HZ)
                bounds["lds"]["frac_busy"] = at_batch["lds_array_cycles_per_pair"] * pairs_per_launch / kern_s / (N_CUS * CLOCK_HZ)
            # distance from an algorithm-level floor (not from the binary's own instruction stream)
            bounds["lds"]["floor"] = {"wave_instructions_per_pair": LDS_FLOOR_INSTR_PER_PAIR, "array_cycles_per_pair": LDS_FLOOR_CYCLES_PER_PAIR,
                                      "measured_wave_instructions_per_pair": sq.get("lds_instr_per_pair"), "measured_array_cycles_per_pair": lds_cyc,
                                      "frac_at_floor": LDS_FLOOR_CYCLES_PER_PAIR * pairs_per_launch / kern_s / (N_CUS * CLOCK_HZ),
                                      "how": "LDS accesses the design needs per traversal pair, conflict-free (bench.py LDS_FLOOR_*; DESIGN.md section 5): "
                                             "frac_at_floor = the LDS ceiling fraction this launch time would mean if the kernel issued only those"}
        top = max((k for k in bounds if bounds[k]["frac"] is not None), key=lambda k: bounds[k]["frac"], default=None)
        roofline = {
            "bound": top, "achieved": bounds[top]["achieved"] if top else None, "peak": bounds[top]["peak"] if top else None,
            "unit": bounds[top]["unit"] if top else None, "frac": bounds[top]["frac"] if top else None, "traffic": traffic,
            "kernel": "k_mccfr_traverse", "kernel_avg_us": kern_us, "launches_timed": launches,
            "kernel_avg_us_device_clock": 1e3 * dev_ms1 / max(dev_n1, 1), "launches_device_clock": dev_n1,
            "workgroup_phase_us": {"prologue": phases[0], "walks": phases[1], "epilogue": phases[2]},
            "workgroup_spread_us": {"mean_start_behind_first": spread[0], "last_start_behind_first": spread[1], "longest_workgroup": spread[2]},
            "bounds": bounds,
            "bound_note": "the kernel's working set (frozen strategy rows, delta table, tree maps) is LDS-resident by design, so the resources that can "
                          "bound it are VALU issue and the LDS array; `bound` is whichever of the candidate ceilings the kernel sits closest to.  Per-pair "
                          "cycle counts come from SQ counter passes of this kernel (profiles/traverse_sq.json: " + str(sq.get("source", "round-1 v7 kernel, B=65536")) + ")",
            "hbm_algorithmic": {"bytes_per_launch": alg_bytes, "GBps": alg_bytes / kern_s / 1e9 if launches else None,
                                "ratio_to_hbm_peak": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBPS if launches else None,
                                "note": "SURVEY 8(d)'s algorithmic price, 111.6 B/visit x 463 x batch visits per launch, divided by kernel time.  NOT a bound "
                                        "for this kernel: those bytes are served from LDS, so the ratio exceeds 1; the HBM traffic actually measured is `traffic`"},
            "traffic_source": {k: tr.get(k) for k in ("source", "commit", "batch")} if tr else None,
            "profile_stale": profile_stale,
            "profile_stale_note": "true = scopa_mccfr.hip has changed (comments and whitespace aside) since the named counter file was taken (sha256 recorded by "
                                  "tests/tools/fold_profiles.py): the per-pair figures then describe an older kernel",
            "timing_note": "kernel_avg_us: HIP start/stop events attached to the dispatch itself (hipExtLaunchKernelGGL) of every "
                           "prof-stride-th launch on the kernel's stream -- the kernel's own begin/end timestamps, what rocprofv3 "
                           "reports as its duration; kernel_avg_us_device_clock: first workgroup start -> last workgroup end on the "
                           "100 MHz device clock over the SAMPLED launches (the same prof-stride; the library keeps the last 2048 samples and "
                           "skips launches of more than 512 workgroups): launch ramp and end-of-kernel write-back excluded; sampled: every " + str(stride)
                           + "-th launch of the last " + str(n_sampled) + " of the " + str(regions) + " timed regions (a sampled launch costs its region ~18 us: the median region holds none)",
        }
        out = {
            "metric": "MiniScopa infoset-traversals/sec", "value": visits_per_region / med, "unit": "infoset-traversals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * med / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: external-sampling MCCFR on MiniScopa (seed-42 deal, 738 infosets), "
                                   f"{args.batch} parallel traversals per traverser per GPU per iteration, tables frozen per iteration",
                       "batch_per_gpu": args.batch, "global_batch": batch_total, "iterations": args.steps,
                       "parallelism": f"dp{world}" + ((" + 1 all-reduce of 29520 B per iteration (" + {"p2p": "one-shot peer-memory exchange over xGMI, rank-ordered sum", "rccl": "RCCL via torch.distributed"}.get(ctx.exchange, ctx.exchange) + ")") if use_dist else ""),
                       "exchange": ctx.exchange if use_dist else None, "exchange_form": ctx.exchange_form if use_dist else None,
                       "exchange_note": ctx.exchange_note if use_dist else None,
                       "iteration_loop": "captured HIP graphs of <= 64 iterations" if args.graph else "eager launches (in-library loop)",
                       "replicas_bit_identical": replicas_identical, "sharded_10_iterations_match_one_gpu": sharded_check,
                       "shared_gpu_rehearsal": bool(args.share_gpu),
                       "rng": "Philox4x32-10, key = seed, counter = (block of the node's (level, branch index), global traversal id, iteration, traverser); 31-bit draws"},
            "timing": {"protocol": f"pre-phase {n_pre} iterations ({pre_s:.2f} s), then --warmup, then the --steps region timed {regions} times "
                                   "(barrier + device sync both sides, max over ranks); value and ms_per_step are the MEDIAN region",
                       "regions": regions, "region_ms_median": 1e3 * med, "region_ms_min": 1e3 * min(times), "region_ms_max": 1e3 * max(times),
                       "region_ms_first": 1e3 * times[0], "region_ms_all": [round(1e3 * t, 4) for t in times], "ms_per_step_min": 1e3 * min(times) / args.steps, "ms_per_step_max": 1e3 * max(times) / args.steps,
                       "pre_phase_iterations": n_pre, "pre_phase_s": pre_s},
            "roofline": roofline,
            "decision_visits": visits,
            "world": roster,
            # the other half of BASELINE's metric ("exploitability vs iters"): where the average strategy stands after this run
            "exploitability": {"iterations": int(ctx.mccfr_iteration()), "traversals_per_iteration": 2 * batch_total,
                               "value": float(ctx.exploitability()["exploitability"]),
                               "note": "exact best-response exploitability of the average strategy on the full tree (k_exploitability); "
                                       "uniform play = 2.2604; the reference's MCCFR update rule (mc_cfr.py:79-84) plateaus -- 0.487 after 5000 of its own "
                                       "sequential iterations, reproduced bit for bit -- while its vanilla CFR reaches 0.0031 after 1000 "
                                       "(profiles/r02_exploitability_curve.json)"},
        }
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch)
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.batch)
            except Exception as e:  # a baseline, never a reason to lose the GPU line
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
            out["gpu_over_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
            # the same engine at a large batch (not the headline: BASELINE configs[1] is B=4096), for the throughput ceiling
            try:
                big = 65536
                ctx.mccfr_iterate(big, 200)
                torch.cuda.synchronize()
                c0, _ = ctx.counters()
                tb = time.perf_counter()
                ctx.mccfr_iterate(big, 1000)
                torch.cuda.synchronize()
                dtb = time.perf_counter() - tb
                out["large_batch"] = {"batch_per_gpu": big, "value": (ctx.counters()[0] - c0) / dtb, "ms_per_step": 1e3 * dtb / 1000}
            except Exception as e:
                out["large_batch"] = {"error": repr(e)}
        if world == 1 and not use_dist and not args.no_sdcfr:
            # BASELINE configs[3] under the same clock: SDCFR at 4096 traversals per player, its own step definition and roofline block
            try:
                sub = argparse.Namespace(**vars(args))
                sub.workload, sub.batch, sub.steps, sub.warmup, sub.gpus = "sdcfr", 4096, max(20, args.sdcfr_steps), 3, 1
                out["sdcfr"] = run_sdcfr(sub, emit=False)
            except Exception as e:
                out["sdcfr"] = {"error": repr(e)}
        if world == 1 and not use_dist and not args.no_subrecords:
            # the hot path's other kernels under the same clock: a few launches each, each with its own roofline block
            try:
                from benchmarks import subrecords
                out.update(subrecords.all_records(local_rank))
            except Exception as e:
                out["subrecords_error"] = repr(e)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
