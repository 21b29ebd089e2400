"""StrategyBuffer (deep_cfr.py:119-160) on the CPU: FIFO of <= max_size snapshots with weight iteration + 1, average policy = sum w_k / W * regret-match(net_k(x)).
The product keeps snapshots as slots of preallocated stacked tensors and evaluates small batches with batched matrix products over all of them; this file holds
that form against the reference's definition written out as a loop over the nets themselves."""
import numpy as np
import pytest


def test_fifo_weights_and_average_policy_against_the_loop_definition():
    import torch
    from scopa_amd.algorithms.deep_cfr.deep_cfr import StrategyBuffer
    from scopa_amd.algorithms.deep_cfr.nets import FlexibleNet, positive_regret_policy
    torch.manual_seed(0)
    nets = [FlexibleNet(mode="mlp", input_shape=(34,), output_dim=16, mlp_hidden=[128, 64]) for _ in range(7)]
    buf = StrategyBuffer(max_size=3)
    x = (torch.rand(9, 34) > 0.6).float()
    m = (torch.rand(9, 16) > 0.5).float()
    m[:, 0] = 1
    assert torch.allclose(buf.average_policy_batch(x, m), m / m.sum(1, keepdim=True))          # no snapshot yet: uniform over the mask (:146-148)
    for i, n in enumerate(nets):
        buf.add_strategy(n, i + 1)
        with torch.no_grad():
            n.head.bias.add_(0.5)                                                                  # the buffer holds a COPY: editing the net afterwards changes nothing
        keep = list(range(max(0, i - 2), i + 1))
        assert buf.weights == [k + 2 for k in keep] and len(buf.strategies) == len(keep)
        with torch.no_grad():
            n.head.bias.sub_(0.5)
        tot = float(sum(buf.weights))
        ref = torch.zeros_like(m)
        with torch.no_grad():
            for k in keep:
                ref += positive_regret_policy(nets[k](x), m) * ((k + 2) / tot)
        for _ in range(2):                                                                         # the second call reads the cached gather
            assert torch.allclose(buf.average_policy_batch(x, m), ref, atol=1e-6)
        big_x, big_m = x.repeat(600, 1), m.repeat(600, 1)                                          # > 4096 rows: the per-snapshot path
        assert torch.allclose(buf.average_policy_batch(big_x, big_m)[:9], ref, atol=1e-6)
        for view, k in zip(buf.strategies, keep):
            assert torch.allclose(view(x), nets[k](x), atol=1e-6)
    one = buf.get_average_policy(x[0].numpy(), m[0].numpy())
    assert one.shape == (16,) and np.allclose(one, ref[0].numpy(), atol=1e-6)


def test_lean_optimiser_step_equals_autograd_on_the_cpu():
    """AdvantageNetwork._step_lean (graph mode's step: the backward pass of the 34-128-64-16 MLP written out in PyTorch ops) against autograd's step
    (deep_cfr.py:99-112) on the same rows: same loss, and the same weights after three Adam steps -- bit for bit on the CPU, where both run the same kernels."""
    import copy
    import torch
    from scopa_amd.algorithms.deep_cfr.deep_cfr import AdvantageNetwork
    torch.manual_seed(3)
    a = AdvantageNetwork(34, 16, device="cpu", memory_size=256)
    b = AdvantageNetwork(34, 16, device="cpu", memory_size=256)
    b.net.load_state_dict(copy.deepcopy(a.net.state_dict()))
    for net in (a, b):
        net.buffer.feat[:200] = (torch.rand(200, 34, generator=torch.Generator().manual_seed(1)) > 0.7).float()
        net.buffer.regret[:200] = torch.randn(200, 16, generator=torch.Generator().manual_seed(2)) * 0.5
        net.buffer.mask[:200] = (torch.rand(200, 16, generator=torch.Generator().manual_seed(4)) > 0.6).float()
        net.buffer.total = 200
    rows = torch.arange(128)
    for _ in range(3):
        la, lb = a._step(rows), b._step_lean(rows)
        assert float(la.detach()) == float(lb)
    for (k, x), y in zip(a.net.state_dict().items(), b.net.state_dict().values()):
        assert torch.equal(x, y), k
