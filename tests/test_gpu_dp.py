"""Data-parallel exchange on the real device primitives (HipOps) with a single-rank RCCL group: with one rank the exchange must be
an identity on the gradients (packed over the union = the rank's own visible list), so a hooked trainer has to track an un-hooked one.  (The N>1 logic is covered on CPU by
tests/test_dp_gloo.py; multi-GPU runs are the driver's.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_single_rank_exchange_is_identity():
    import torch.distributed as dist
    from litegs_amd import dp, synthetic as S
    from litegs_amd.trainer import SyntheticTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        scene = S.make_scene(6000, seed=4)
        ta = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)
        tb = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)
        hook = dp.GradientExchange(tb.params, 1).hook
        for i in range(4):
            la = ta.step(i)
            lb = tb.step(i, hook, i % 2)
            assert abs(la.item() - lb.item()) < 1e-5
        moved = 0.0
        for pa, pb, p0 in zip(ta.params, tb.params, scene):
            assert (pa - pb).abs().max().item() < 5e-4
            moved = max(moved, (pa.detach().cpu() - torch.from_numpy(p0)).abs().max().item())
        assert moved > 1e-4, "parameters must actually have been updated"
        # union list == own visible set, ascending
        fr = tb.frames[0]
        img, vis_id, vis_num, _ = tb.forward(fr)
        img.sum().backward()
        n = int(vis_num.item())
        local = [p.grad.compacted_values.clone() for p in tb.params]
        uid, ucnt = hook(tb.params, vis_id, vis_num, 0)
        assert int(ucnt.item()) == n
        assert torch.equal(uid[:n], vis_id[:n])
        # one rank: the packed union buffer must hold exactly this rank's compact gradients
        for p, g0 in zip(tb.params, local):
            assert p.grad.shape == p.shape and hasattr(p.grad, "compacted_values")
            rows = g0.numel() // (g0.shape[-2] * g0.shape[-1])
            a = p.grad.compacted_values.reshape(rows, -1, g0.shape[-1])[:, :n]
            assert torch.equal(a, g0.reshape(rows, -1, g0.shape[-1])[:, :n])
    finally:
        dist.destroy_process_group()
