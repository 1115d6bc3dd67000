"""Sparse Adam over visible chunks -- host-side mirror of ``litegs/training/optimizer.py``.

``SparseGaussianAdam.step(visible_chunk, visible_chunks_num, primitive_visible)`` (reference :15-44) updates,
per parameter group, only the rows of the visible chunks using the compact gradient carried by
``CompactedTensor``; b1 = 0.9, b2 = 0.999, no bias correction (GR/compact.cu:333-342).  ``Scheduler`` is the
exponential position-lr decay of reference :46-71, ``get_optimizer`` the group layout of :74-95.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .binding import ops as fused
from .wrapper import CompactedTensor, sparse_adam_update


@dataclass
class OptimizationParams:
    """Defaults of litegs/arguments.py:80-92."""
    iterations: int = 30000
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_max_steps: int = 30000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.025
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    lambda_dssim: float = 0.2            # unused by the reference as well (the fused loss fixes 0.8 L1 + 0.2 D-SSIM)
    reg_weight: float = 0.0
    learnable_viewproj: bool = False


class SparseGaussianAdam(torch.optim.Adam):
    def __init__(self, params, lr, eps, bCluster=True):
        self.bCluster = bCluster
        super().__init__(params=params, lr=lr, eps=eps)

    @torch.no_grad()
    def step(self, visible_chunk=None, visible_chunks_num=None, primitive_visible=None):
        for group in self.param_groups:
            assert len(group["params"]) == 1, "one tensor per group"
            param = group["params"][0]
            if param.grad is None:
                continue
            state = self.state[param]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0)
                state["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
            lr, eps = group["lr"], group["eps"]
            if self.bCluster:
                chunks, S = param.shape[-2], param.shape[-1]
                grad = param.grad
                if isinstance(grad, CompactedTensor):
                    ids = visible_chunk if visible_chunk is not None else grad.chunk_ids
                    sparse_adam_update(param.data.view(-1, chunks, S), grad.compacted_values, state["exp_avg"].view(-1, chunks, S),
                                       state["exp_avg_sq"].view(-1, chunks, S), ids, visible_chunks_num, lr, 0.9, 0.999, eps)
                else:
                    # dense gradient [*, chunks, S] (data-parallel exchange): update exactly the chunks of `visible_chunk`
                    # (the union of all ranks' visibility); without a list, every chunk.
                    ids, cnt = visible_chunk, visible_chunks_num
                    if ids is None:
                        ids, cnt = torch.arange(chunks, device=param.device), None
                    fused.adamUpdate(param.data.view(-1, chunks, S), grad.reshape(-1, chunks, S), state["exp_avg"].view(-1, chunks, S),
                                     state["exp_avg_sq"].view(-1, chunks, S), ids, cnt, lr, 0.9, 0.999, eps, grad_dense=True)
            else:
                N = param.shape[-1]
                sparse_adam_update(param.data.view(-1, N), param.grad.reshape(-1, N), state["exp_avg"].view(-1, N),
                                   state["exp_avg_sq"].view(-1, N), primitive_visible, None, lr, 0.9, 0.999, eps)


class Scheduler:
    """Exponential interpolation of the position learning rate (reference :46-71); other groups keep their lr."""

    def __init__(self, optimizer, lr_init, lr_final, max_epochs=10000):
        self.optimizer, self.lr_init, self.lr_final, self.max_epochs = optimizer, lr_init, lr_final, max_epochs
        self.last_epoch = 0
        self._apply()

    def _xyz_lr(self):
        if self.lr_init == 0.0 and self.lr_final == 0.0:
            return 0.0
        t = min(max(self.last_epoch / self.max_epochs, 0.0), 1.0)
        return math.exp(math.log(self.lr_init) * (1 - t) + math.log(self.lr_final) * t)

    def _apply(self):
        for g in self.optimizer.param_groups:
            if g.get("name") == "xyz":
                g["lr"] = self._xyz_lr()

    def step(self):
        self.last_epoch += 1
        self._apply()


def get_optimizer(xyz, scale, rot, sh_0, sh_rest, opacity, spatial_lr_scale: float, opt: OptimizationParams, cluster: bool = True):
    groups = [
        {"params": [xyz], "lr": opt.position_lr_init * spatial_lr_scale, "name": "xyz"},
        {"params": [sh_0], "lr": opt.feature_lr, "name": "sh_0"},
        {"params": [sh_rest], "lr": opt.feature_lr / 10.0, "name": "sh_rest"},
        {"params": [opacity], "lr": opt.opacity_lr, "name": "opacity"},
        {"params": [scale], "lr": opt.scaling_lr, "name": "scale"},
        {"params": [rot], "lr": opt.rotation_lr, "name": "rot"},
    ]
    optimizer = SparseGaussianAdam(groups, lr=0, eps=1e-15, bCluster=cluster)
    scheduler = Scheduler(optimizer, opt.position_lr_init * spatial_lr_scale, opt.position_lr_final * spatial_lr_scale,
                          max_epochs=opt.position_lr_max_steps)
    return optimizer, scheduler
