"""Host-side glue of the visibility table, entirely on litegs_amd's own HIP kernels.

Mirror of ``Binning.__binning_fused`` (litegs/utils/wrapper.py:717-763) with the two torch ops of the
reference (``view_depth.sort`` and ``cumsum``) replaced by the hand-written radix sort / scan, so that the
depth order is STABLE (ties keep Gaussian-index order; the reference's torch.sort is not stable, which
makes its output order for equal depths implementation defined) and the indices stay int32.
"""
from __future__ import annotations

import torch

from .binding import ops as fused
from ._lib import check, lib


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def depth_order_and_prefix(view_depth: torch.Tensor, allocate_size: torch.Tensor):
    """view_depth f32[V,N], allocate_size i32[V,N] -> (depth_sorted_index i32[V,N], prefix_sum i32[V,N] inclusive).

    wrapper.py:739-745: ascending depth order, allocate_size gathered into that order, int32 cumsum."""
    L = lib()
    view_depth = view_depth.contiguous()
    allocate_size = allocate_size.contiguous()
    V, N = view_depth.shape
    dev = view_depth.device
    ka = torch.empty((V, N), dtype=torch.int32, device=dev)
    va = torch.empty((V, N), dtype=torch.int32, device=dev)
    kb = torch.empty((V, N), dtype=torch.int32, device=dev)
    vb = torch.empty((V, N), dtype=torch.int32, device=dev)
    tb = max(L.lg_radix_sort_temp_bytes(N), L.lg_scan_temp_bytes(N))
    temp = torch.empty((tb,), dtype=torch.uint8, device=dev)
    prefix = torch.empty((V, N), dtype=torch.int32, device=dev)
    s = _s()
    odd = L.lg_radix_sort_num_passes(0, 32) % 2 == 1
    for v in range(V):
        check(L.lg_depth_sort_keys(view_depth[v].data_ptr(), N, ka[v].data_ptr(), va[v].data_ptr(), s), "depth_sort_keys")
        check(L.lg_radix_sort_pairs(ka[v].data_ptr(), va[v].data_ptr(), kb[v].data_ptr(), vb[v].data_ptr(), N, 0, 32, temp.data_ptr(), tb, s),
              "depth radix sort")
        idx = vb[v] if odd else va[v]
        check(L.lg_gather_inclusive_scan(allocate_size[v].data_ptr(), idx.data_ptr(), 0, N, prefix[v].data_ptr(), temp.data_ptr(), tb, s),
              "gather_inclusive_scan")
    return (vb if odd else va), prefix


@torch.no_grad()
def binning(ndc, view_depth, inv_cov2d, opacity, valid_length, feedback_binning_allocate_size, idx_tensor,
            img_pixel_shape, tile_size, on_visible=None):
    """-> (tile_start_index i32[V,T+2], sorted_pointId i32[V,L], primitive_visible i64[N]).

    Same contract as Binning.call_fused (wrapper.py:717-763).  `on_visible(b_visible)` is the statistics hook
    (wrapper.py:735-736)."""
    H, W = int(img_pixel_shape[0]), int(img_pixel_shape[1])
    th, tw = int(tile_size[0]), int(tile_size[1])
    tiles_num = ((H + th - 1) // th) * ((W + tw - 1) // tw)
    _, _, allocate_size = fused.get_allocate_size(ndc, view_depth, inv_cov2d, opacity, H, W, th, tw, valid_length)
    b_visible = allocate_size != 0
    if on_visible is not None:
        on_visible(b_visible)
    depth_sorted_index, prefix_sum = depth_order_and_prefix(view_depth, allocate_size)
    sorted_tile, sorted_point = fused.create_table(ndc, inv_cov2d, opacity, prefix_sum, depth_sorted_index,
                                                   feedback_binning_allocate_size, idx_tensor, H, W, th, tw)
    tile_start_index = fused.tileRange(sorted_tile, tiles_num)
    return tile_start_index, sorted_point, b_visible.sum(0)
