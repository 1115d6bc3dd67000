"""Scene-side callers of the hot path -- host mirror of ``litegs/scene/{point,cluster}.py``.

* ``cluster_points`` / ``uncluster`` (cluster.py:7-29): the [.., N] <-> [.., chunks, S] views with the reference's padding rule.
* ``create_gaussians`` (point.py:7-20): initial parameters from a point cloud; the 3-NN distance is csrc/knn.hip.
* ``morton_order`` / ``spatial_refine`` (point.py:29-154): the periodic Morton re-sort of every parameter, gradient and Adam
  moment.  Codes, the stable 63-bit argsort and the column gathers are csrc/refine.hip (GPU only; CPU tensors are rejected).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ._lib import check, lib


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def cluster_points(chunksize: int, *tensors: torch.Tensor):
    """[..., N] -> [..., chunks, chunksize]; a ragged tail is padded by REPEATING the last elements (cluster.py:15-18)."""
    out = []
    for t in tensors:
        rem = t.shape[-1] % chunksize
        if rem != 0:
            t = torch.cat([t, t[..., -(chunksize - rem):]], dim=-1).contiguous()
        out.append(t.view(*t.shape[:-1], t.shape[-1] // chunksize, chunksize))
    return tuple(out)


def uncluster(*tensors: torch.Tensor):
    return tuple(t.view(*t.shape[:-2], t.shape[-2] * t.shape[-1]) for t in tensors)


@torch.no_grad()
def create_gaussians(xyz: torch.Tensor, color: torch.Tensor, sh_degree: int):
    """xyz [P,3], color [P,3] in [0,1] -> (xyz[3,P], scale[3,P], rot[4,P], sh_0[1,3,P], sh_rest[(d+1)^2-1,3,P], opacity[1,P])."""
    from simple_knn._C import distCUDA2
    dist2 = torch.clamp_min(distCUDA2(xyz), 0.0000001)
    P = xyz.shape[0]
    xyz_t = xyz.transpose(0, 1).contiguous()
    sh_0 = ((color.transpose(0, 1) - 0.5) / 0.28209479177387814).unsqueeze(0).contiguous()        # rgb_to_sh0 (utils/__init__.py)
    sh_rest = torch.zeros(((sh_degree + 1) ** 2 - 1, 3, P), dtype=torch.float32, device=xyz.device)
    scale = torch.log(torch.sqrt(dist2)).unsqueeze(0).repeat(3, 1)
    rot = torch.zeros((4, P), dtype=torch.float32, device=xyz.device)
    rot[0] = 1
    opacity = torch.full((1, P), 0.1, dtype=torch.float32, device=xyz.device)
    opacity = torch.log(opacity / (1 - opacity))
    return xyz_t, scale, rot, sh_0, sh_rest, opacity


@torch.no_grad()
def morton_order(xyz: torch.Tensor, want_codes: bool = False) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """xyz [3,N] float32 (GPU) -> (codes int64[N] or None, order int32[N]) with order = codes.sort(stable=True).indices."""
    if not xyz.is_cuda or xyz.dtype != torch.float32 or xyz.dim() != 2 or xyz.shape[0] != 3:
        raise RuntimeError("morton_order: float32 GPU tensor [3,N] required (no CPU path)")
    xyz = xyz.contiguous()
    N = xyz.shape[1]
    L = lib()
    codes = torch.empty((N,), dtype=torch.int64, device=xyz.device) if want_codes else None
    order = torch.empty((N,), dtype=torch.int32, device=xyz.device)
    tb = L.lg_morton_order_temp_bytes(N)
    temp = torch.empty((max(tb, 1),), dtype=torch.uint8, device=xyz.device)
    check(L.lg_morton_order(xyz.data_ptr(), N, codes.data_ptr() if want_codes else None, order.data_ptr(), temp.data_ptr(), tb, _s()),
          "morton_order")
    return codes, order


@torch.no_grad()
def permute_columns(t: torch.Tensor, order: torch.Tensor) -> torch.Tensor:
    """t [..., N] float32, order int32[M] -> t[..., order] as a new contiguous tensor (one launch)."""
    if not (t.is_cuda and order.is_cuda) or t.dtype != torch.float32 or order.dtype != torch.int32:
        raise RuntimeError("permute_columns: float32 tensor and int32 order on the GPU required (no CPU path)")
    t = t.contiguous()
    n_src, n_dst = t.shape[-1], order.shape[0]
    rows = t.numel() // max(n_src, 1)
    out = torch.empty((*t.shape[:-1], n_dst), dtype=torch.float32, device=t.device)
    check(lib().lg_permute_columns(t.data_ptr(), order.data_ptr(), rows, n_src, n_dst, out.data_ptr(), _s()), "permute_columns")
    return out


@torch.no_grad()
def spatial_refine(bClustered: bool, optimizer, xyz: torch.Tensor, *args: torch.Tensor):
    """Morton re-sort (point.py:86-154).  optimizer None: returns the re-sorted (xyz, *args).  Otherwise every parameter of every
    group, its gradient if present and every same-shaped state tensor (exp_avg, exp_avg_sq) is re-sorted in place and the six
    parameters are returned in the reference's order (xyz, scale, rot, sh_0, sh_rest, opacity)."""
    chunk = xyz.shape[-1] if bClustered else 0
    flat = uncluster(xyz)[0] if bClustered else xyz
    _, order = morton_order(flat.detach())

    def resort(t: torch.Tensor) -> torch.Tensor:
        f = uncluster(t)[0] if bClustered else t
        r = permute_columns(f.detach(), order)
        return cluster_points(chunk, r)[0] if bClustered else r

    if optimizer is None:
        return tuple(resort(t) for t in (xyz, *args))
    by_name = {}
    for group in optimizer.param_groups:
        for p in group["params"]:
            p.data.copy_(resort(p.data))
            if p.grad is not None and isinstance(p.grad, torch.Tensor) and type(p.grad) is torch.Tensor and p.grad.shape == p.shape:
                p.grad.data = resort(p.grad.data)
            for key, value in optimizer.state.get(p, {}).items():
                if isinstance(value, torch.Tensor) and value.shape == p.shape:
                    value.data = resort(value.data)
        by_name[group.get("name")] = group["params"][0]
    return tuple(by_name[n] for n in ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"))
