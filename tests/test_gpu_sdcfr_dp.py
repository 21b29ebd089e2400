"""SDCFR data parallelism on the GPU (BASELINE configs[4]: sharded traversal batches + one gradient all-reduce per optimiser
step; DeepCFR.train, deep_cfr.py:431-495).  The test box has ONE GPU, so the ranks are separate processes sharing it over a gloo
group -- the product code path (DeepCFR(rank=, world=): broadcast of rank 0's nets, per-rank traversal ids, flat gradient
all-reduce on the device tensors) is the one the 8-GPU run takes with RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_sdcfr_keeps_replicas_identical(ctx, tmp_path):
    batch, iters, world = 64, 2, 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "sdcfr_dp_worker.py"), str(r), str(world), "29621",
                               str(tmp_path), str(batch), str(iters)], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    net_keys = [k for k in r0.files if k.startswith("net")]
    assert len(net_keys) == 2 * 6
    for k in [k for k in r0.files if k.startswith("init")]:
        assert np.array_equal(r0[k], r1[k]), k                      # rank 0's initial weights reached rank 1
    for k in net_keys:
        assert np.array_equal(r0[k], r1[k]), k                      # averaged gradients + same Adam step: bit-identical replicas
        assert not np.array_equal(r0[k], r0["init" + k[3:]]), k     # ... that did train
    for p in range(2):
        # rows: every rank appended batch x 41 rows per iteration to ITS ring -> the job holds world x the single-rank count
        assert int(r0[f"rows{p}"]) == int(r1[f"rows{p}"]) == batch * 41 * iters
        assert list(r0[f"buffer_sizes{p}"]) == [batch * 41 * (i + 1) for i in range(iters)]
        # the shards are different traversals (global ids rank*batch ...): the memories differ although the nets do not
        assert not np.array_equal(r0[f"feat{p}"], r1[f"feat{p}"])
        assert np.all(np.isfinite(r0[f"losses{p}"])) and np.all(np.isfinite(r1[f"losses{p}"]))
    assert int(r0["visits"]) == int(r1["visits"]) == (105 + 82) * batch * iters
