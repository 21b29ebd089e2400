#!/usr/bin/env python3
"""One rank of the SDCFR data-parallel test (tests/test_gpu_sdcfr_dp.py; BASELINE configs[4] at the scale of the one-GPU test box):
all ranks share GPU 0, the process group is gloo (RCCL refuses two ranks on one device), each rank runs DeepCFR(rank, world)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out, batch, iters = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    import torch
    import torch.distributed as dist
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)                       # different initial nets per rank: the constructor must broadcast rank 0's
    d = DeepCFR(load_game("mini_scopa"), num_players=2, device="cuda:0", batch=batch, rank=rank, world=world)
    init = {f"init{p}__{k}": v.cpu().numpy() for p in range(2) for k, v in d.advantage_nets[p].net.state_dict().items()}
    d.train(iterations=iters, advantage_epochs=3, eval_freq=100)
    res = dict(init)
    for p in range(2):
        for k, v in d.advantage_nets[p].net.state_dict().items():
            res[f"net{p}__{k}"] = v.cpu().numpy()
        mem = d.advantage_nets[p].buffer
        res[f"rows{p}"] = np.array(len(mem))
        res[f"feat{p}"] = mem.feat[:len(mem)].cpu().numpy()
        res[f"losses{p}"] = np.array(d.training_history["losses"][p])
        res[f"buffer_sizes{p}"] = np.array(d.training_history["buffer_sizes"][p])
    res["visits"] = np.array(d._engine.ctx.sdcfr_visits())
    np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
