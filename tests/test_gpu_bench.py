"""bench.py's contract, run for real on the GPU box (small step counts): the one JSON line, its required keys, a roofline fraction
that is a fraction, the CPU baseline leg, and `--gpus 2` started PLAINLY -- the script must spawn its ranks itself (here both on the
one GPU of the test box, --share-gpu) and come back with bit-identical replicas."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def test_bench_line_contract_n1(ctx):
    d = _bench("--gpus", "1", "--steps", "10", "--warmup", "2", "--regions", "5", "--pre-phase-s", "0.05", "--no-cpu-baseline", "--sdcfr-steps", "20")
    # the default invocation carries BASELINE configs[3] as a sub-record: SDCFR at 4096 traversals per player, >= 20 timed iterations
    sd = d["sdcfr"]
    assert "error" not in sd, sd
    assert sd["steps"] >= 20 and sd["config"]["batch_per_gpu"] == 4096 and sd["dtype"] == "f32"
    assert sd["decision_visits"] == (105 + 82) * 4096 * sd["steps"] and sd["ms_per_step"] > 0
    assert sd["roofline"]["kernel"].startswith("k_sdcfr_walk") and 0.0 < sd["roofline"]["frac"] <= 1.0 and sd["traversal_only"]["launches_timed"] == 2 * sd["steps"]
    assert set(sd["roofline"]["bounds"]) == {"hbm-memory-rows", "mfma-f32"} and sd["roofline"]["bound"] == "hbm-memory-rows" and sd["roofline"]["hbm_algorithmic"]["GBps"] > 0
    assert all(0.0 < b["frac"] <= 1.0 for b in sd["roofline"]["bounds"].values())
    assert sd["traversal_only"]["forward_per_visit_kernel_avg_us"] > sd["traversal_only"]["kernel_avg_us"]   # a forward pass per visit costs more than one per node
    assert 0.0 < sd["with_hip_training_step"]["ms_per_step"] < sd["ms_per_step"]                             # the opt-in hand-written optimiser step, measured beside the default
    w = d["world"]
    assert w["world_size"] == 1 and len(w["ranks"]) == 1 and w["ranks"][0]["rank"] == 0 and w["ranks"][0]["ms_per_step"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["higher_is_better"] is True and "workload" in d["config"]
    assert d["decision_visits"] == 463 * 4096 * 10 * d["timing"]["regions"]          # exact kernel counters over the timed regions
    assert abs(d["value"] - 463 * 4096 * 10 / (d["timing"]["region_ms_median"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["kernel"] == "k_mccfr_traverse" and r["launches_timed"] >= 16 and r["bound"] in r["bounds"]
    assert 0.0 < r["frac"] <= 1.0 and all(b["frac"] is None or 0.0 < b["frac"] <= 1.0 for b in r["bounds"].values())
    assert r["frac"] == max(b["frac"] for b in r["bounds"].values() if b["frac"] is not None)
    assert r["traffic"] is None or r["traffic"] > 0
    assert d["exploitability"]["value"] >= 0.0


def test_bench_spawns_its_own_ranks(ctx):
    d = _bench("--gpus", "2", "--share-gpu", "--steps", "5", "--warmup", "2", "--regions", "3", "--pre-phase-s", "0.05", "--batch", "512", "--no-cpu-baseline")
    c = d["config"]
    assert d["n_gpus"] == 2 and c["global_batch"] == 1024 and c["batch_per_gpu"] == 512
    assert c["exchange"] in ("p2p", "rccl") and c["replicas_bit_identical"] is True and c["sharded_10_iterations_match_one_gpu"] is True
    assert d["decision_visits"] == 463 * 1024 * 5 * d["timing"]["regions"]
    w = d["world"]                                   # the roster: what the process group itself says about its ranks
    assert w["world_size"] == 2 and [r["rank"] for r in w["ranks"]] == [0, 1] and len({r["pid"] for r in w["ranks"]}) == 2
    assert all(r["ms_per_step"] > 0 and r["device_index"] == 0 for r in w["ranks"]) and w["distinct_devices"] == 1   # --share-gpu: one device, and the line says so


def test_bench_five_ranks_rehearsal_goes_through_the_collective(ctx):
    """The 8-GPU control flow at the rank count the one-GPU box admits (its process guard allows six GPU processes: five ranks + this
    test): `--gpus 5 --share-gpu` started plainly.  With more than three ranks on one device the peer exchange is not even tried
    (its poll loops assume co-resident peers): the run goes through the torch.distributed all-reduce, proves the sharded
    pipeline against the one-GPU tables, and ends with identical replicas and the exact visit count."""
    d = _bench("--gpus", "5", "--share-gpu", "--steps", "4", "--warmup", "1", "--regions", "3", "--pre-phase-s", "0.05", "--batch", "256", "--no-cpu-baseline", timeout=600)
    c = d["config"]
    assert d["n_gpus"] == 5 and c["global_batch"] == 5 * 256 and c["exchange"] == "rccl" and "chosen outright" in c["exchange_note"]
    assert c["replicas_bit_identical"] is True and c["sharded_10_iterations_match_one_gpu"] is True
    assert d["decision_visits"] == 463 * 5 * 256 * 4 * d["timing"]["regions"]
    assert d["world"]["world_size"] == 5 and sorted(r["rank"] for r in d["world"]["ranks"]) == list(range(5))


@pytest.mark.parametrize("extra", [["--exchange", "rccl"], ["--inject-proof-failure"]])
def test_bench_two_ranks_through_the_collective(ctx, extra):
    """The split path traverse | torch.distributed all-reduce | apply on two ranks -- chosen outright, and as the fallback bench.py
    takes when a connected peer exchange does not pass the 10-iteration proof (injected here): valid lines with identical replicas."""
    d = _bench("--gpus", "2", "--share-gpu", "--steps", "5", "--warmup", "2", "--regions", "3", "--pre-phase-s", "0.05", "--batch", "512", "--no-cpu-baseline", *extra)
    c = d["config"]
    assert c["exchange"] == "rccl" and c["replicas_bit_identical"] is True and c["sharded_10_iterations_match_one_gpu"] is True
    assert ("fell back" in c["exchange_note"]) == (extra == ["--inject-proof-failure"])
    assert d["decision_visits"] == 463 * 1024 * 5 * d["timing"]["regions"]


def test_bench_sdcfr_workload(ctx):
    d = _bench("--workload", "sdcfr", "--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline")
    assert d["dtype"] == "f32" and d["decision_visits"] == (105 + 82) * 256 * 3
    assert d["roofline"]["kernel"].startswith("k_sdcfr_walk") and 0.0 < d["roofline"]["frac"] <= 1.0


def test_bench_sdcfr_with_the_hand_written_optimiser_step(ctx):
    d = _bench("--workload", "sdcfr", "--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline", "--sdcfr-train-backend", "hip")
    assert "hand-written" in d["config"]["training"] and "with_hip_training_step" not in d and d["decision_visits"] == (105 + 82) * 256 * 3


def test_bench_sdcfr_two_ranks_spawned(ctx):
    """BASELINE configs[4] at rehearsal scale: `--workload sdcfr --gpus 2` started plainly, both ranks on the one GPU."""
    d = _bench("--workload", "sdcfr", "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--batch", "128", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 256 and d["config"]["replicas_bit_identical"] is True
    assert d["decision_visits"] == (105 + 82) * 128 * 2 * 2
