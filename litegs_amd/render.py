"""The ``litegs.render`` operator surface on MI355X -- host-side mirror of ``litegs/render/__init__.py``.

``render_preprocess`` (reference :11-48) and ``render`` (reference :50-94) keep the reference's names,
positional arguments, return tuples and error behaviour, so ``litegs.training`` / ``example_metrics.py``
style callers read unchanged.  ``pp`` is any object with the reference's PipelineParams attributes
(``cluster_size``, ``tile_size``, ``sparse_grad``, ``enable_transmitance``, ``enable_depth``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import wrapper
from .binding import ops as fused
from .statistics import STATS


@dataclass
class PipelineParams:
    """Defaults of litegs/arguments.py:68-76."""
    cluster_size: int = 128
    tile_size: Tuple[int, int] = (8, 16)
    sparse_grad: bool = True
    device_preload: bool = True
    enable_transmitance: bool = False
    enable_depth: bool = False
    input_color_type: str = "sh"


def uncluster(*tensors):
    """[..., chunks, S] -> [..., chunks*S] views (litegs/scene/cluster.py:23-27)."""
    return tuple(t.view(*t.shape[:-2], t.shape[-2] * t.shape[-1]) for t in tensors)


@torch.no_grad()
def get_cluster_AABB(clustered_xyz, clustered_scale, clustered_rot):
    """Chunk AABBs from activated scale / normalised rotation (litegs/scene/cluster.py:29-46)."""
    chunk_size = clustered_xyz.shape[-1]
    xyz, scale, rot = uncluster(clustered_xyz, clustered_scale, clustered_rot)
    T = fused.createTransformMatrix_forward(rot.contiguous(), scale.contiguous(), None)
    ext = (T * math.sqrt(2 * math.log(255))).abs().sum(dim=0)
    ext = ext.view(*ext.shape[:-1], ext.shape[-1] // chunk_size, chunk_size)
    mx = (clustered_xyz + ext).max(dim=-1).values
    mn = (clustered_xyz - ext).min(dim=-1).values
    return ((mx + mn) / 2).contiguous(), ((mx - mn) / 2).contiguous()


def render_preprocess(cluster_origin: Optional[torch.Tensor], cluster_extend: Optional[torch.Tensor], frustumplane: torch.Tensor,
                      view_matrix: torch.Tensor, xyz: torch.Tensor, scale: torch.Tensor, rot: torch.Tensor, sh_0: torch.Tensor,
                      sh_rest: torch.Tensor, opacity: torch.Tensor, feedback_buffer: Optional[torch.Tensor],
                      idx_tensor: Optional[torch.Tensor], pp, actived_sh_degree: int):
    """-> (visible_chunkid, visible_chunks_num, culled_xyz, culled_scale, culled_rot, color, culled_opacity)."""
    visible_chunkid = None
    visible_chunks_num = None
    if pp.cluster_size:
        if cluster_origin is None or cluster_extend is None:
            cluster_origin, cluster_extend = get_cluster_AABB(xyz, scale.exp(), torch.nn.functional.normalize(rot, dim=0))
        _, visible_chunks_num, visible_chunkid = fused.frustum_culling_aabb(cluster_origin, cluster_extend, frustumplane, feedback_buffer, idx_tensor)
        if STATS.active:
            STATS.set_compaction(visible_chunkid, visible_chunks_num)
        culled = wrapper.CullCompactActivateWithSparseGrad.apply(pp.sparse_grad, actived_sh_degree, visible_chunkid, visible_chunks_num,
                                                                 view_matrix, xyz, scale, rot, sh_0, sh_rest, opacity)
        culled_xyz, culled_scale, culled_rot, color, culled_opacity = uncluster(*culled)
    else:
        pad_one = torch.ones((1, xyz.shape[-1]), dtype=xyz.dtype, device=xyz.device)
        culled_xyz = torch.concat((xyz, pad_one), dim=0)
        culled_scale = scale.exp()
        culled_rot = torch.nn.functional.normalize(rot, dim=0)
        culled_opacity = opacity.sigmoid()
        with torch.no_grad():
            camera_center = (-view_matrix[..., 3:4, :3] @ (view_matrix[..., :3, :3].transpose(-1, -2))).squeeze(1)
            dirs = torch.nn.functional.normalize(culled_xyz[:3] - camera_center.unsqueeze(-1), dim=-2)
        color = wrapper.SphericalHarmonicToRGB.call_fused(actived_sh_degree, sh_0, sh_rest, dirs.contiguous())
    return visible_chunkid, visible_chunks_num, culled_xyz, culled_scale, culled_rot, color, culled_opacity


def render(view_matrix: torch.Tensor, proj_matrix: torch.Tensor, xyz: torch.Tensor, scale: torch.Tensor, rot: torch.Tensor,
           color: torch.Tensor, opacity: torch.Tensor, valid_length: Optional[torch.Tensor],
           feedback_binning_allocate_size: Optional[torch.Tensor], idx_tensor: Optional[torch.Tensor],
           actived_sh_degree: int, output_shape: Tuple[int, int], pp):
    """-> (img[V,3,H,W] clamped to [0,1], transmitance|None, depth|None, normal=None, primitive_visible)."""
    view_pos, ndc_pos = wrapper.MVPTransform.apply(xyz, view_matrix, proj_matrix, valid_length)
    transform_matrix = wrapper.CreateTransformMatrix.call_fused(scale, rot, valid_length)
    J = wrapper.CreateRaySpaceTransformMatrix.call_fused(view_pos, proj_matrix, output_shape, valid_length)
    cov2d = wrapper.CreateCov2dDirectly.call_fused(J, view_matrix, transform_matrix, valid_length)
    _, _, inv_cov2d = wrapper.EighAndInverse2x2Matrix.call_fused(cov2d, valid_length)
    view_depth = view_pos[:, 2, :]

    tile_start_index, sorted_pointId, primitive_visible = wrapper.Binning.call_fused(
        ndc_pos, view_depth, inv_cov2d, opacity, valid_length, feedback_binning_allocate_size, idx_tensor, output_shape, pp.tile_size)

    tiles = STATS.schedule_for_current_frame()
    img, transmitance, depth, normal, last = wrapper.GaussiansRasterFunc.apply(
        sorted_pointId, tile_start_index, ndc_pos, inv_cov2d, color, opacity, tiles,
        output_shape[0], output_shape[1], pp.tile_size[0], pp.tile_size[1], pp.enable_transmitance, pp.enable_depth)
    if STATS.active:
        STATS.update_tile_schedule(last, pp.tile_size[0], pp.tile_size[1])

    H, W = output_shape[0], output_shape[1]
    img = img[..., :H, :W].clamp(0, 1).contiguous()
    if transmitance is not None:
        transmitance = transmitance[..., :H, :W].contiguous()
    if depth is not None:
        depth = depth[..., :H, :W].contiguous()
    return img, transmitance, depth, normal, primitive_visible
