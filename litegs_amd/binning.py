"""Host-side glue of the visibility table, entirely on litegs_amd's own HIP kernels.

Mirror of ``Binning.__binning_fused`` (litegs/utils/wrapper.py:717-763) with the two torch ops of the
reference (``view_depth.sort`` and ``cumsum``) replaced by the hand-written radix sort / scan, so that the
depth order is STABLE (ties keep Gaussian-index order; the reference's torch.sort is not stable, which
makes its output order for equal depths implementation defined) and the indices stay int32.
"""
from __future__ import annotations

import os

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


def grouped_table(ndc, view_depth, inv_cov2d, opacity, allocate_size, feedback_binning_allocate_size, idx_tensor, H, W, th, tw, tiles_num):
    """The same table WITHOUT the two full-length sorts (single view): instances are emitted in ascending splat-id order (no depth sort of
    the splats: prefix sums in id order), grouped by tile with per-tile counts and cursors (lg_tile_group: no sort over the instances
    either) and every tile's list is then ordered by (view depth, id) in LDS (lg_tile_depth_sort_unordered) -- which is the order a STABLE
    depth sort followed by a STABLE tile sort leaves (csrc/tilesort.hip), so tile_start_index and sorted_pointId are bit for bit what
    depth_order_and_prefix + create_table + tileRange return.  Table sizing: the reference's protocol (GR/binning.cu:139-163: 1.5 x the
    frame's previous total from the pinned feedback buffer, blocking on the first visit).  One difference, in the failure mode only: a
    table that turns out too short drops the entries of the highest splat ids, not those of the deepest splats (the reference drops by
    emission position, binning.cu:63, silently in both cases).  -> (tile_start_index i32[1,T+2], sorted_pointId i32[1,L])"""
    L = lib()
    s = _s()
    dev = ndc.device
    N = view_depth.shape[1]
    if N <= 0:                               # nothing to size a table from (and prefix[N - 1] below would lie in front of the buffer)
        raise RuntimeError("error pred_allocate_size")
    ndc, inv_cov2d, opacity, view_depth = ndc.contiguous(), inv_cov2d.contiguous(), opacity.contiguous(), view_depth.contiguous()
    allocate_size = allocate_size.contiguous()
    tb = L.lg_scan_temp_bytes(N)
    temp = torch.empty((tb,), dtype=torch.uint8, device=dev)
    prefix = torch.empty((1, N), dtype=torch.int32, device=dev)
    check(L.lg_gather_inclusive_scan(allocate_size.data_ptr(), None, 0, N, prefix.data_ptr(), temp.data_ptr(), tb, s), "inclusive_scan")
    pred = 0
    if feedback_binning_allocate_size is not None and idx_tensor is not None:
        k = int(idx_tensor[0])
        pred = int(feedback_binning_allocate_size[k])
        check(L.lg_feedback_d2h(feedback_binning_allocate_size.data_ptr() + 4 * k, prefix.data_ptr() + 4 * (N - 1), s), "feedback copy")
    pred = int(1.5 * pred)
    if pred <= 0 and N > 0:                  # blocking path (binning.cu:152-163)
        pred = int(prefix[0, -1].item())
    if pred <= 0:
        raise RuntimeError("error pred_allocate_size")
    keys = torch.zeros((1, pred), dtype=torch.int32, device=dev)
    vals = torch.empty((1, pred), dtype=torch.int32, device=dev)
    db = L.lg_duplicate_with_keys_temp_bytes(1, N, pred)
    dtemp = torch.empty((db,), dtype=torch.uint8, device=dev)
    check(L.lg_duplicate_with_keys(ndc.data_ptr(), inv_cov2d.data_ptr(), opacity.data_ptr(), prefix.data_ptr(), None, 0, 1, N, H, W, th, tw, pred,
                                   keys.data_ptr(), vals.data_ptr(), dtemp.data_ptr(), db, s), "duplicate_with_keys")
    tile_start = torch.empty((1, tiles_num + 2), dtype=torch.int32, device=dev)
    grouped = torch.empty((1, pred), dtype=torch.int32, device=dev)
    gtemp = torch.empty((2 * (tiles_num + 2),), dtype=torch.int32, device=dev)
    check(L.lg_tile_group(keys.data_ptr(), vals.data_ptr(), pred, tiles_num, tile_start.data_ptr(), grouped.data_ptr(), gtemp.data_ptr(), s), "tile_group")
    check(L.lg_tile_depth_sort_unordered(grouped.data_ptr(), tile_start.data_ptr(), view_depth.data_ptr(), 1, pred, N, tiles_num,
                                         vals.data_ptr(), s), "tile_depth_sort")      # (vals is free again: scratch of the long lists)
    return tile_start, grouped


def reference_pattern_table(ndc, view_depth, inv_cov2d, opacity, allocate_size, feedback_binning_allocate_size, idx_tensor, H, W, th, tw, tiles_num):
    """Binning.__binning_fused's own sequence, operation for operation (litegs/utils/wrapper.py:738-761): torch.sort of the view depths
    (int64 indices, not stable), the per-view gather of the tile counts into depth order, an int32 cumsum, then create_table and tileRange
    through the boundary with exactly those tensors.  What tests/golden/reference_call_trace.json records the unmodified reference doing;
    tests/test_gpu_reference_call_pattern.py holds this function to that trace and bench.py times it (`reference_call_pattern_ms`)."""
    _, depth_sorted_index = view_depth.sort(dim=-1, descending=False)
    for i in range(ndc.shape[0]):
        allocate_size[i] = allocate_size[i, depth_sorted_index[i]]
    prefix_sum = allocate_size.cumsum(1, dtype=torch.int32)
    sorted_tile, sorted_point = fused.create_table(ndc, inv_cov2d, opacity, prefix_sum, depth_sorted_index, feedback_binning_allocate_size, idx_tensor,
                                                   H, W, th, tw)
    return fused.tileRange(sorted_tile, int(tiles_num)), sorted_point


# LITEGS_OPERATOR_BINNING: "grouped" (default: no full-length sorts, grouped_table), "sorted" (the reference's structure on this
# repository's stable radix sort / scan: depth_order_and_prefix + create_table + tileRange) or "reference" (the reference's own sequence
# of torch operations and boundary calls, reference_pattern_table)
LONG_LIST_PER_TILE = 640
_MODE = os.environ.get("LITEGS_OPERATOR_BINNING", "grouped")
_GROUPED = _MODE not in ("sorted", "reference")
last_route = None             # which structure built the last table: "reference" | "grouped" | "sorted" (bench.py reports it)


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
    global last_route
    if _MODE == "reference":
        last_route = "reference"
        tile_start_index, sorted_point = reference_pattern_table(ndc, view_depth, inv_cov2d, opacity, allocate_size, feedback_binning_allocate_size,
                                                                 idx_tensor, H, W, th, tw, tiles_num)
        return tile_start_index, sorted_point, b_visible.sum(0)
    # Long lists (the executor's rule, fast.py long_list_global: more than 640 instances per tile in the frame's previous visit): the per-tile
    # depth sort of grouped_table loses to the reference's structure there (training state, 3 M Gaussians: 5.66 against 4.48 ms per
    # iteration, gpurun r6o), so such frames take depth_order_and_prefix + create_table + tileRange.
    long_lists = False
    if _GROUPED and feedback_binning_allocate_size is not None and idx_tensor is not None:
        long_lists = int(feedback_binning_allocate_size[int(idx_tensor[0])]) > LONG_LIST_PER_TILE * tiles_num
    if _GROUPED and not long_lists and view_depth.shape[0] == 1 and view_depth.shape[1] < (1 << 24):
        last_route = "grouped"
        tile_start_index, sorted_point = grouped_table(ndc, view_depth, inv_cov2d, opacity, allocate_size, feedback_binning_allocate_size,
                                                       idx_tensor, H, W, th, tw, tiles_num)
        return tile_start_index, sorted_point, b_visible.sum(0)
    last_route = "sorted"
    depth_sorted_index, prefix_sum = depth_order_and_prefix(view_depth, allocate_size)
    sorted_tile, sorted_point = fused.create_table(ndc, inv_cov2d, opacity, prefix_sum, depth_sorted_index,
                                                   feedback_binning_allocate_size, idx_tensor, H, W, th, tw)
    tile_start_index = fused.tileRange(sorted_tile, tiles_num)
    return tile_start_index, sorted_point, b_visible.sum(0)
