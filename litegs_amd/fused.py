"""The ``litegs_fused`` operator surface of the reference, bound to the MI355X HIP library.

Same names, positional argument order and returned tensors (shape / dtype / layout) as the reference's
pybind11 module (GR/ext_cuda.cpp:9-35; signatures GR/{binning,compact,raster,transform}.h), so
``litegs/utils/wrapper.py`` can ``import litegs_fused`` (top-level shim ``litegs_fused.py``) unchanged.
Differences, all internal: ``packed_params`` is a [V,N,16] fp32 record (the reference's is [V,N,8] with
fp16 colour) and is only ever handed back to ``rasterize_backward``; ``visible_chunk_id`` is in ascending
order (the reference's order is non-deterministic); ``rasterize_forward(enable_depth=True)`` returns zeros
where the reference returns uninitialised memory.

torch is used for device memory and streams only; every computation is a HIP kernel behind the C ABI in
``include/litegs_hip.h``.  No CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional

import torch

from ._lib import check, lib

_DT = {torch.float32: 0, torch.int32: 1, torch.int64: 2, torch.float64: 3, torch.int16: 4, torch.int8: 5, torch.uint8: 5, torch.bool: 5}


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"litegs_fused: '{name}' must live on the GPU (litegs_amd has no CPU path)")
    return t if t.is_contiguous() else t.contiguous()


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    t = _dev(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"litegs_fused: '{name}' must be float32, got {t.dtype}")
    return t


def _vl(valid_length: Optional[torch.Tensor]):
    if valid_length is None:
        return None
    if valid_length.dtype != torch.int32 or not valid_length.is_cuda:
        raise RuntimeError("litegs_fused: valid_length must be a device int32 tensor")
    return valid_length.data_ptr()


def _tiles_shape(h: int, w: int, th: int, tw: int):
    gx, gy = (w + tw - 1) // tw, (h + th - 1) // th
    return gx, gy, gx * gy, gy * th, gx * tw


# ----------------------------------------------------------------------------------------------- compact.h
def frustum_culling_aabb(aabb_origin, aabb_ext, frustumplane, feedback_buffer_arg=None, data_idx_arg=None):
    """GR/compact.cu:503-551 -> [visibility bool[M], visible_chunks_num int32[1], visible_chunk_id int64[pred]]."""
    aabb_origin, aabb_ext, frustumplane = _f32(aabb_origin, "aabb_origin"), _f32(aabb_ext, "aabb_ext"), _f32(frustumplane, "frustumplane")
    V, M = frustumplane.shape[0], aabb_origin.shape[1]
    dev = frustumplane.device
    visibility = torch.empty((M,), dtype=torch.bool, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    ids = torch.empty((M,), dtype=torch.int64, device=dev)
    L = lib()
    check(L.lg_frustum_culling_aabb(_p(aabb_origin), _p(aabb_ext), _p(frustumplane), V, M, _p(visibility), _p(num), _p(ids), _s()),
          "frustum_culling_aabb")
    pred = 0
    if feedback_buffer_arg is not None and data_idx_arg is not None:
        base = feedback_buffer_arg.data_ptr()
        for i in range(data_idx_arg.shape[0]):
            idx = int(data_idx_arg[i])
            pred = max(pred, int(feedback_buffer_arg[idx]))
            check(L.lg_feedback_d2h(base + 4 * idx, _p(num), _s()), "feedback copy")
    pred = int(1.2 * pred)
    if pred <= 0:
        pred = int(num.item())          # blocking path, first time a frame is seen (compact.cu:543-546)
    return [visibility, num, ids[:pred]]


def cull_compact_activate(sh_degree, visible_chunk_id, visible_chunks_num, view_matrix, position, scale, rotation, sh_base, sh_rest, opacity):
    """GR/compact.cu:983-1085."""
    position, scale, rotation = _f32(position, "position"), _f32(scale, "scale"), _f32(rotation, "rotation")
    sh_base, sh_rest, opacity, view_matrix = _f32(sh_base, "sh_base"), _f32(sh_rest, "sh_rest"), _f32(opacity, "opacity"), _f32(view_matrix, "view_matrix")
    visible_chunk_id = _dev(visible_chunk_id, "visible_chunk_id")
    chunks, S = position.shape[-2], position.shape[-1]
    A, V = visible_chunk_id.shape[0], view_matrix.shape[0]
    dev = position.device
    o_pos = torch.empty((4, A, S), dtype=torch.float32, device=dev)
    o_scale = torch.empty((3, A, S), dtype=torch.float32, device=dev)
    o_rot = torch.empty((4, A, S), dtype=torch.float32, device=dev)
    o_color = torch.empty((V, 3, A, S), dtype=torch.float32, device=dev)
    o_opa = torch.empty((1, A, S), dtype=torch.float32, device=dev)
    check(lib().lg_cull_compact_activate(int(sh_degree), _p(visible_chunk_id), _p(visible_chunks_num), A, _p(view_matrix), V,
                                         _p(position), _p(scale), _p(rotation), _p(sh_base), _p(sh_rest), _p(opacity), chunks, S,
                                         _p(o_pos), _p(o_scale), _p(o_rot), _p(o_color), _p(o_opa), _s()), "cull_compact_activate")
    return [o_pos, o_scale, o_rot, o_color, o_opa]


def activate_backward(sh_degree, visible_chunk_id, visible_chunks_num, view_matrix, position, scale, rotation, sh_base, sh_rest, opacity,
                      activated_position_grad, activated_scale_grad, activated_rotation_grad, color_grad, activated_opacity_grad):
    """GR/compact.cu:1087-1212."""
    position, scale, rotation, opacity = _f32(position, "position"), _f32(scale, "scale"), _f32(rotation, "rotation"), _f32(opacity, "opacity")
    view_matrix = _f32(view_matrix, "view_matrix")
    visible_chunk_id = _dev(visible_chunk_id, "visible_chunk_id")
    g_pos, g_scale = _f32(activated_position_grad, "g_pos"), _f32(activated_scale_grad, "g_scale")
    g_rot, g_col, g_opa = _f32(activated_rotation_grad, "g_rot"), _f32(color_grad, "g_color"), _f32(activated_opacity_grad, "g_opa")
    chunks, S = position.shape[-2], position.shape[-1]
    A, V, R = visible_chunk_id.shape[0], view_matrix.shape[0], sh_rest.shape[0]
    dev = position.device
    d_pos = torch.empty((position.shape[0], A, S), dtype=torch.float32, device=dev)
    d_scale = torch.empty((3, A, S), dtype=torch.float32, device=dev)
    d_rot = torch.empty((4, A, S), dtype=torch.float32, device=dev)
    d_sh0 = torch.empty((sh_base.shape[0], sh_base.shape[1], A, S), dtype=torch.float32, device=dev)
    d_shr = torch.empty((R, sh_rest.shape[1], A, S), dtype=torch.float32, device=dev)
    d_opa = torch.empty((1, A, S), dtype=torch.float32, device=dev)
    check(lib().lg_activate_backward(int(sh_degree), _p(visible_chunk_id), _p(visible_chunks_num), A, _p(view_matrix), V,
                                     _p(position), _p(scale), _p(rotation), _p(opacity), chunks, S, R,
                                     _p(g_pos), _p(g_scale), _p(g_rot), _p(g_col), _p(g_opa),
                                     _p(d_pos), _p(d_scale), _p(d_rot), _p(d_sh0), _p(d_shr), _p(d_opa), _s()), "activate_backward")
    return [d_pos, d_scale, d_rot, d_sh0, d_shr, d_opa]


def adamUpdate(param, param_grad, exp_avg, exp_avg_sq, visible_index, valid_length, lr, b1, b2, eps, grad_dense=False):
    """GR/compact.cu:377-417 (in place).  grad_dense (extension for the DP path): param_grad is [E,chunks,S], indexed by chunk id."""
    for t, n in ((param, "param"), (param_grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise RuntimeError(f"adamUpdate: '{n}' must be a contiguous float32 device tensor")
    if param.dim() == 3:
        E, chunks, S = param.shape
        A = visible_index.shape[0]
        check(lib().lg_adam_update_chunk(_p(param), _p(param_grad), _p(exp_avg), _p(exp_avg_sq), _p(_dev(visible_index, "visible_index")),
                                         _vl(valid_length), E, chunks, A, S, 1 if grad_dense else 0, float(lr), float(b1), float(b2), float(eps), _s()), "adamUpdate")
    elif param.dim() == 2:
        E, N = param.shape
        check(lib().lg_adam_update_primitive(_p(param), _p(param_grad), _p(exp_avg), _p(exp_avg_sq), _p(_dev(visible_index, "visible_index")),
                                             E, N, float(lr), float(b1), float(b2), float(eps), _s()), "adamUpdate")
    else:
        raise RuntimeError("adamUpdate: param must be [E,chunks,S] or [E,N]")


def gpu_driven_pipeline_sparse_op(A, B, visible_chunk_ids, visible_count, op_name):
    """GR/compact.cu:1257-1336 (in place on A)."""
    for t, n in ((A, "A"), (B, "B"), (visible_chunk_ids, "visible_chunk_ids"), (visible_count, "visible_count")):
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
    if op_name in ("add", "sum"):
        op = 0
    elif op_name == "min":
        op = 1
    elif op_name == "max":
        op = 2
    else:
        raise RuntimeError(f"Unsupported op: {op_name}. Expected: add, min, max")
    if A.dtype not in _DT or A.dtype != B.dtype:
        raise RuntimeError("gpu_driven_pipeline_sparse_op: unsupported dtype")
    if not A.is_contiguous():
        raise RuntimeError("gpu_driven_pipeline_sparse_op: A must be contiguous (updated in place)")
    B = B.contiguous()
    E, chunks, S = A.shape
    alloc = B.shape[1]
    if S > 1024:
        raise RuntimeError("chunk_size exceeds max threads per block")
    check(lib().lg_sparse_scatter(_p(A), _p(B), _p(visible_chunk_ids.contiguous()), _p(visible_count), E, chunks, alloc, S, _DT[A.dtype], op, _s()),
          "gpu_driven_pipeline_sparse_op")


def create_viewproj_forward(view_params, recp_tan_half_fov_x, img_h, img_w, z_near, z_far):
    """GR/compact.cu:119-135: [V,7] quaternion+translation and [1] 1/tan(fov_x/2) -> (view, proj, viewproj [V,4,4], frustumplane [V,6,4])."""
    view_params, fov = _f32(view_params, "view_params").contiguous(), _f32(recp_tan_half_fov_x, "recp_tan_half_fov_x").contiguous()
    if view_params.dim() != 2 or view_params.shape[1] != 7:
        raise RuntimeError("create_viewproj_forward: view_params must be [views,7]")
    V = view_params.shape[0]
    view, proj, vp = (torch.empty((V, 4, 4), dtype=torch.float32, device=view_params.device) for _ in range(3))
    planes = torch.empty((V, 6, 4), dtype=torch.float32, device=view_params.device)
    check(lib().lg_create_viewproj_forward(_p(view_params), _p(fov), V, int(img_h), int(img_w), float(z_near), float(z_far),
                                           _p(view), _p(proj), _p(vp), _p(planes), _s()), "create_viewproj_forward")
    return [view, proj, vp, planes]


def create_viewproj_backward(view_matrix_grad, proj_matrix_grad, viewproj_matrix_grad, view_params, recp_tan_half_fov_x, img_h, img_w, z_near, z_far):
    """GR/compact.cu:287-316 -> (grad_view_params [V,7], grad_recp_tan_half_fov_x [1])."""
    view_params, fov = _f32(view_params, "view_params").contiguous(), _f32(recp_tan_half_fov_x, "recp_tan_half_fov_x").contiguous()
    V = view_params.shape[0]
    gs = [_f32(g, n).contiguous() for g, n in ((view_matrix_grad, "view_matrix_grad"), (proj_matrix_grad, "proj_matrix_grad"),
                                               (viewproj_matrix_grad, "viewproj_matrix_grad"))]
    for g in gs:
        if tuple(g.shape) != (V, 4, 4):
            raise RuntimeError("create_viewproj_backward: matrix gradients must be [views,4,4]")
    g_params, g_fov = torch.zeros_like(view_params), torch.zeros_like(fov)
    check(lib().lg_create_viewproj_backward(_p(gs[0]), _p(gs[1]), _p(gs[2]), _p(view_params), _p(fov), V, int(img_h), int(img_w),
                                            float(z_near), float(z_far), _p(g_params), _p(g_fov), _s()), "create_viewproj_backward")
    return [g_params, g_fov]


# --------------------------------------------------------------------------------------------- transform.h
def mvp_transform_forward(world_position, view_matrix, proj_matrix, valid_length=None):
    world_position, view_matrix, proj_matrix = _f32(world_position, "world_position"), _f32(view_matrix, "view_matrix"), _f32(proj_matrix, "proj_matrix")
    V, N = view_matrix.shape[0], world_position.shape[1]
    view_pos = torch.empty((V, 4, N), dtype=torch.float32, device=world_position.device)
    ndc_pos = torch.empty((V, 4, N), dtype=torch.float32, device=world_position.device)
    check(lib().lg_mvp_transform_forward(_p(world_position), _p(view_matrix), _p(proj_matrix), _vl(valid_length), V, N, _p(view_pos), _p(ndc_pos), _s()),
          "mvp_transform_forward")
    return [view_pos, ndc_pos]


def mvp_transform_backward(grad_ndc_pos, grad_view_pos, view_matrix, proj_matrix, view_pos, valid_length=None):
    grad_ndc_pos, grad_view_pos, view_pos = _f32(grad_ndc_pos, "grad_ndc_pos"), _f32(grad_view_pos, "grad_view_pos"), _f32(view_pos, "view_pos")
    V, N = grad_ndc_pos.shape[0], grad_ndc_pos.shape[2]
    g_world = torch.empty((4, N), dtype=torch.float32, device=grad_ndc_pos.device)
    check(lib().lg_mvp_transform_backward(_p(grad_ndc_pos), _p(grad_view_pos), _p(_f32(view_matrix, "view_matrix")), _p(_f32(proj_matrix, "proj_matrix")),
                                          _p(view_pos), _vl(valid_length), V, N, _p(g_world), _s()), "mvp_transform_backward")
    return g_world


def createTransformMatrix_forward(quaternion, scale, valid_length=None):
    quaternion, scale = _f32(quaternion, "quaternion"), _f32(scale, "scale")
    N = quaternion.shape[1]
    T = torch.empty((3, 3, N), dtype=torch.float32, device=scale.device)
    check(lib().lg_create_transform_matrix_forward(_p(quaternion), _p(scale), _vl(valid_length), N, _p(T), _s()), "createTransformMatrix_forward")
    return T


def createTransformMatrix_backward(transform_matrix_grad, quaternion, scale, valid_length=None):
    g, quaternion, scale = _f32(transform_matrix_grad, "transform_matrix_grad"), _f32(quaternion, "quaternion"), _f32(scale, "scale")
    N = quaternion.shape[1]
    g_quat = torch.empty((4, N), dtype=torch.float32, device=g.device)
    g_scale = torch.empty((3, N), dtype=torch.float32, device=g.device)
    check(lib().lg_create_transform_matrix_backward(_p(g), _p(quaternion), _p(scale), _vl(valid_length), N, _p(g_quat), _p(g_scale), _s()),
          "createTransformMatrix_backward")
    return [g_quat, g_scale]


def jacobianRayspace(translate_position, proj_matrix, output_h, output_w, valid_length=None):
    tp, proj_matrix = _f32(translate_position, "translate_position"), _f32(proj_matrix, "proj_matrix")
    V, N = tp.shape[0], tp.shape[2]
    J = torch.empty((V, 3, 3, N), dtype=torch.float32, device=tp.device)
    check(lib().lg_jacobian_rayspace(_p(tp), _p(proj_matrix), _vl(valid_length), V, N, int(output_h), int(output_w), _p(J), _s()), "jacobianRayspace")
    return J


def createCov2dDirectly_forward(J, view_matrix, transform_matrix, valid_length=None):
    J, view_matrix, T = _f32(J, "J"), _f32(view_matrix, "view_matrix"), _f32(transform_matrix, "transform_matrix")
    V, N = view_matrix.shape[0], T.shape[2]
    cov = torch.empty((V, 2, 2, N), dtype=torch.float32, device=T.device)
    check(lib().lg_create_cov2d_forward(_p(J), _p(view_matrix), _p(T), _vl(valid_length), V, N, _p(cov), _s()), "createCov2dDirectly_forward")
    return cov


def createCov2dDirectly_backward(cov2d_grad, J, view_matrix, transform_matrix, valid_length=None):
    g, J, view_matrix, T = _f32(cov2d_grad, "cov2d_grad"), _f32(J, "J"), _f32(view_matrix, "view_matrix"), _f32(transform_matrix, "transform_matrix")
    V, N = view_matrix.shape[0], T.shape[2]
    gT = torch.empty((3, 3, N), dtype=torch.float32, device=g.device)
    check(lib().lg_create_cov2d_backward(_p(g), _p(J), _p(view_matrix), _p(T), _vl(valid_length), V, N, _p(gT), _s()), "createCov2dDirectly_backward")
    return gT


def eigh_and_inv_2x2matrix_forward(input, valid_length=None):
    x = _f32(input, "input")
    V, N = x.shape[0], x.shape[3]
    val = torch.empty((V, 2, N), dtype=torch.float32, device=x.device)
    vec = torch.empty((V, 2, 2, N), dtype=torch.float32, device=x.device)
    inv = torch.empty((V, 2, 2, N), dtype=torch.float32, device=x.device)
    check(lib().lg_eigh_inv_2x2_forward(_p(x), _vl(valid_length), V, N, _p(val), _p(vec), _p(inv), _s()), "eigh_and_inv_2x2matrix_forward")
    return [val, vec, inv]


def inv_2x2matrix_backward(inv_matrix, dL_dInvMatrix, valid_length=None):
    inv, g = _f32(inv_matrix, "inv_matrix"), _f32(dL_dInvMatrix, "dL_dInvMatrix")
    V, N = inv.shape[0], inv.shape[3]
    out = torch.empty_like(g)
    check(lib().lg_inv_2x2_backward(_p(inv), _p(g), _vl(valid_length), V, N, 0, _p(out), _s()), "inv_2x2matrix_backward")
    return out


def sh2rgb_forward(degree, sh_base, sh_rest, dir):
    sh_base, sh_rest, dir = _f32(sh_base, "sh_base"), _f32(sh_rest, "sh_rest"), _f32(dir, "dir")
    V, N = dir.shape[0], dir.shape[2]
    rgb = torch.empty((V, 3, N), dtype=torch.float32, device=dir.device)
    check(lib().lg_sh2rgb_forward(int(degree), _p(sh_base), _p(sh_rest), _p(dir), V, N, _p(rgb), _s()), "sh2rgb_forward")
    return rgb


def sh2rgb_backward(degree, rgb_grad, sh_rest_dim, dir, SH_base, SH_rest):
    g, dir = _f32(rgb_grad, "rgb_grad"), _f32(dir, "dir")
    V, N = dir.shape[0], dir.shape[2]
    d0 = torch.empty((1, 3, N), dtype=torch.float32, device=g.device)
    dr = torch.empty((int(sh_rest_dim), 3, N), dtype=torch.float32, device=g.device)
    dd = torch.empty((V, 3, N), dtype=torch.float32, device=g.device)
    check(lib().lg_sh2rgb_backward(int(degree), _p(g), _p(dir), V, N, int(sh_rest_dim), _p(d0), _p(dr), _p(dd), _s()), "sh2rgb_backward")
    return [d0, dr, dd]


def world2ndc_forward(world_position, view_project_matrix):
    w, m = _f32(world_position, "world_position"), _f32(view_project_matrix, "view_project_matrix")
    V, N = m.shape[0], w.shape[1]
    ndc = torch.empty((V, 4, N), dtype=torch.float32, device=w.device)
    rw = torch.empty((V, 1, N), dtype=torch.float32, device=w.device)
    check(lib().lg_world2ndc_forward(_p(w), _p(m), V, N, _p(ndc), _p(rw), _s()), "world2ndc_forward")
    return [ndc, rw]


def world2ndc_backword(view_project_matrix, position, repc_hom_w, grad_ndcpos):
    m, ndc, rw, g = _f32(view_project_matrix, "vp"), _f32(position, "ndc_position"), _f32(repc_hom_w, "repc_hom_w"), _f32(grad_ndcpos, "grad_ndcpos")
    V, N = g.shape[0], g.shape[2]
    out = torch.empty((4, N), dtype=torch.float32, device=g.device)
    check(lib().lg_world2ndc_backward(_p(m), _p(ndc), _p(rw), _p(g), V, N, _p(out), _s()), "world2ndc_backword")
    return out


# ----------------------------------------------------------------------------------------------- binning.h
def get_allocate_size(ndc, view_space_z, inv_cov2d, opacity, height, width, tilesize_h, tilesize_w, valid_length=None):
    ndc, vz, ic, opacity = _f32(ndc, "ndc"), _f32(view_space_z, "view_space_z"), _f32(inv_cov2d, "inv_cov2d"), _f32(opacity, "opacity")
    V, N = ndc.shape[0], ndc.shape[2]
    dev = ndc.device
    left_up = torch.empty((V, 2, N), dtype=torch.int32, device=dev)
    right_down = torch.empty((V, 2, N), dtype=torch.int32, device=dev)
    alloc = torch.empty((V, N), dtype=torch.int32, device=dev)
    check(lib().lg_get_allocate_size(_p(ndc), _p(vz), _p(ic), _p(opacity), _vl(valid_length), V, N, int(height), int(width),
                                     int(tilesize_h), int(tilesize_w), _p(left_up), _p(right_down), _p(alloc), _s()), "get_allocate_size")
    return [left_up, right_down, alloc]


def sort_bits(height, width, th, tw) -> int:
    """GR/binning.cu:199-202."""
    max_tiles = ((height + th - 1) // th) * ((width + tw - 1) // tw)
    bit = 0
    while max_tiles >> 1:
        max_tiles >>= 1
        bit += 1
    return bit + 1


def radix_sort_pairs(keys: torch.Tensor, values: torch.Tensor, begin_bit: int, end_bit: int):
    """Stable LSD radix sort of int32/uint32 (key, value) pairs along the last dim of [V, n] (or [n]) tensors.
    The inputs are used as ping-pong scratch (destroyed).  Returns (sorted_keys, sorted_values)."""
    L = lib()
    k2 = keys.reshape(-1, keys.shape[-1])
    v2 = values.reshape(-1, values.shape[-1])
    n = k2.shape[1]
    kb, vb = torch.empty_like(k2), torch.empty_like(v2)
    tb = L.lg_radix_sort_temp_bytes(n)
    temp = torch.empty((tb,), dtype=torch.uint8, device=keys.device)
    for v in range(k2.shape[0]):       # per view (the reference sorts view 0 V times: GR/binning.cu:213-221, a bug)
        check(L.lg_radix_sort_pairs(_p(k2[v]), _p(v2[v]), _p(kb[v]), _p(vb[v]), n, int(begin_bit), int(end_bit), _p(temp), tb, _s()), "radix_sort_pairs")
    if L.lg_radix_sort_num_passes(int(begin_bit), int(end_bit)) % 2 == 1:
        return kb.reshape(keys.shape), vb.reshape(values.shape)
    return k2.reshape(keys.shape), v2.reshape(values.shape)


def create_table(ndc, inv_cov2d, opacity, offset, depth_sorted_pointid, feedback_buffer_cpu, idx_tensor_cpu, height, width, tile_size_h, tile_size_w):
    """GR/binning.cu:123-226 -> [tileId_sorted int32[V,L], pointId_sorted int32[V,L]]."""
    ndc, ic, opacity = _f32(ndc, "ndc"), _f32(inv_cov2d, "inv_cov2d"), _f32(opacity, "opacity")
    offset = _dev(offset, "offset")
    if offset.dtype != torch.int32:
        raise RuntimeError("create_table: offset must be int32 (cumsum dtype=torch.int32)")
    ids = _dev(depth_sorted_pointid, "depth_sorted_pointid")
    if ids.dtype not in (torch.int64, torch.int32):
        raise RuntimeError("create_table: depth_sorted_pointid must be int64 or int32")
    V, N = ndc.shape[0], ndc.shape[2]
    L = lib()
    pred = 0
    if feedback_buffer_cpu is not None and idx_tensor_cpu is not None:
        base = feedback_buffer_cpu.data_ptr()
        for i in range(V):
            idx = int(idx_tensor_cpu[i])
            pred = max(pred, int(feedback_buffer_cpu[idx]))
            check(L.lg_feedback_d2h(base + 4 * idx, offset.data_ptr() + 4 * (i * N + N - 1), _s()), "feedback copy")
    pred = int(1.5 * pred)
    if pred <= 0 and N > 0:                # blocking path (binning.cu:152-163)
        pred = int(offset[:, -1].max().item())
    if pred <= 0:
        raise RuntimeError("error pred_allocate_size")
    dev = ndc.device
    bits = sort_bits(int(height), int(width), int(tile_size_h), int(tile_size_w))
    if V == 1:                             # one native call: emission counts the sort's digits, no counting pass, no pre-cleared table
        ka, va = torch.empty((1, pred), dtype=torch.int32, device=dev), torch.empty((1, pred), dtype=torch.int32, device=dev)
        kb, vb = torch.empty_like(ka), torch.empty_like(va)
        tb = L.lg_create_table_temp_bytes(N, pred, bits)
        temp = torch.empty((tb,), dtype=torch.uint8, device=dev)
        check(L.lg_create_table(_p(ndc), _p(ic), _p(opacity), _p(offset), _p(ids), 1 if ids.dtype == torch.int64 else 0, N, int(height), int(width),
                                int(tile_size_h), int(tile_size_w), pred, bits, _p(ka), _p(va), _p(kb), _p(vb), _p(temp), tb, _s()), "create_table")
        return [kb, vb] if L.lg_radix_sort_num_passes(0, bits) % 2 == 1 else [ka, va]
    keys = torch.zeros((V, pred), dtype=torch.int32, device=dev)
    vals = torch.empty((V, pred), dtype=torch.int32, device=dev)
    tb = L.lg_duplicate_with_keys_temp_bytes(V, N, pred)
    temp = torch.empty((tb,), dtype=torch.uint8, device=dev)
    check(L.lg_duplicate_with_keys(_p(ndc), _p(ic), _p(opacity), _p(offset), _p(ids), 1 if ids.dtype == torch.int64 else 0, V, N,
                                   int(height), int(width), int(tile_size_h), int(tile_size_w), pred, _p(keys), _p(vals), _p(temp), tb, _s()),
          "duplicate_with_keys")
    ks, vs = radix_sort_pairs(keys, vals, 0, bits)
    return [ks, vs]


def tileRange(table_tileId, max_tileId):
    t = _dev(table_tileId, "table_tileId")
    V, Lh = t.shape
    out = torch.empty((V, int(max_tileId) + 2), dtype=torch.int32, device=t.device)
    check(lib().lg_tile_range(_p(t), V, Lh, int(max_tileId), _p(out), _s()), "tileRange")
    return out


# ------------------------------------------------------------------------------------------------ raster.h
def _raster_forward_impl(sorted_points, start_index, packed, specific_tiles, img_h, img_w, tile_h, tile_w, enable_statistic, enable_trans, enable_depth):
    sorted_points, start_index = _dev(sorted_points, "sorted_points"), _dev(start_index, "start_index")
    V, Lh = sorted_points.shape
    N = packed.shape[1]
    gx, gy, ntiles, Hp, Wp = _tiles_shape(int(img_h), int(img_w), int(tile_h), int(tile_w))
    dev = packed.device
    img = torch.empty((V, 3, Hp, Wp), dtype=torch.float32, device=dev)
    trans = torch.empty((V, 1, Hp, Wp), dtype=torch.float32, device=dev)
    depth = torch.zeros((V, 1, Hp, Wp), dtype=torch.float32, device=dev) if enable_depth else torch.empty((0, 0, 0, 0), dtype=torch.float32, device=dev)
    last = torch.empty((V, 1, Hp, Wp), dtype=torch.int16, device=dev)
    fc = torch.zeros((V, 1, N), dtype=torch.int32, device=dev)
    fw = torch.zeros((V, 1, N), dtype=torch.float32, device=dev)
    K, tp = 0, None
    if specific_tiles is not None:
        specific_tiles = _dev(specific_tiles, "specific_tiles")
        K, tp = specific_tiles.shape[1], specific_tiles.data_ptr()
        # tiles not listed are not rendered; give them a defined value
        img.zero_(); trans.fill_(1.0); last.zero_()
    check(lib().lg_raster_forward(_p(sorted_points), _p(start_index), _p(packed), tp, K, V, Lh, N, int(img_h), int(img_w), int(tile_h), int(tile_w),
                                  1 if enable_statistic else 0, _p(img), _p(trans), _p(last), _p(fc), _p(fw), None, None, _s()), "rasterize_forward")
    return img, trans, depth, last, fc, fw


def rasterize_forward(sorted_points, start_index, ndc, cov2d_inv, color, opacity, specific_tiles, img_h, img_w, tilesize_h, tilesize_w,
                      enable_statistic, enable_trans, enable_depth):
    """GR/raster.cu:386-492 -> [img, transmitance, depth, last_contributor, packed_params, fragment_count, fragment_weight_sum]."""
    ndc, ic, color, opacity = _f32(ndc, "ndc"), _f32(cov2d_inv, "cov2d_inv"), _f32(color, "color"), _f32(opacity, "opacity")
    V, N = ndc.shape[0], ndc.shape[2]
    L = lib()
    packed = torch.empty((V, N, L.lg_packed_record_floats()), dtype=torch.float32, device=ndc.device)
    check(L.lg_pack_forward_params(_p(ndc), _p(ic), _p(color), _p(opacity), None, V, N, int(img_h), int(img_w), _p(packed), _s()), "pack_forward_params")
    img, trans, depth, last, fc, fw = _raster_forward_impl(sorted_points, start_index, packed, specific_tiles, img_h, img_w, tilesize_h, tilesize_w,
                                                           enable_statistic, enable_trans, enable_depth)
    return [img, trans, depth, last, packed, fc, fw]


def rasterize_forward_packed(sorted_points, start_index, packed_params, specific_tiles_arg, img_h, img_w, tile_h, tile_w,
                             enable_statistic, enable_trans, enable_depth):
    """GR/raster.cu:495-586."""
    img, trans, depth, last, fc, fw = _raster_forward_impl(sorted_points, start_index, _f32(packed_params, "packed_params"), specific_tiles_arg,
                                                           img_h, img_w, tile_h, tile_w, enable_statistic, enable_trans, enable_depth)
    return [img, trans, depth, last, fc, fw]


def rasterize_backward(sorted_points, start_index, packed_params, specific_tiles, final_transmitance, last_contributor, d_img,
                       d_trans_img_arg, d_depth_img_arg, grad_inv_sacler_arg, img_h, img_w, tilesize_h, tilesize_w, enable_statistic):
    """GR/raster.cu:917-1037 -> [d_ndc, d_cov2d_inv, d_color, d_opacity, err_sum, err_square_sum]."""
    sorted_points, start_index = _dev(sorted_points, "sorted_points"), _dev(start_index, "start_index")
    packed = _f32(packed_params, "packed_params")
    final_T, d_img = _f32(final_transmitance, "final_transmitance"), _f32(d_img, "d_img")
    last = _dev(last_contributor, "last_contributor")
    V, Lh = sorted_points.shape
    N = packed.shape[1]
    dev = packed.device
    L = lib()
    pg = torch.zeros((V, N, L.lg_packed_grad_floats()), dtype=torch.float32, device=dev)
    err_sum = torch.zeros((V, 1, N), dtype=torch.float32, device=dev)
    err_sq = torch.zeros((V, 1, N), dtype=torch.float32, device=dev)
    K, tp = 0, None
    if specific_tiles is not None:
        specific_tiles = _dev(specific_tiles, "specific_tiles")
        K, tp = specific_tiles.shape[1], specific_tiles.data_ptr()
    d_trans = _f32(d_trans_img_arg, "d_trans") if d_trans_img_arg is not None else None
    order = None
    if specific_tiles is None:
        # heaviest tiles first (raster.hip: tile schedule): work per tile from last_contributor, then a counting sort
        gx, gy, ntiles, Hp, Wp = _tiles_shape(int(img_h), int(img_w), int(tilesize_h), int(tilesize_w))
        work = torch.empty((V, ntiles + 1), dtype=torch.int32, device=dev)
        order = torch.empty((V, ntiles), dtype=torch.int32, device=dev)
        check(L.lg_tile_work_from_last(_p(last), V, int(img_h), int(img_w), int(tilesize_h), int(tilesize_w), _p(work), _s()), "tile_work")
        check(L.lg_tile_order(_p(work), V, ntiles, _p(order), _s()), "tile_order")
    check(L.lg_raster_backward(_p(sorted_points), _p(start_index), _p(packed), tp, K, _p(final_T), _p(last), _p(d_img), _p(d_trans),
                               V, Lh, N, int(img_h), int(img_w), int(tilesize_h), int(tilesize_w), 1 if enable_statistic else 0,
                               _p(pg), _p(err_sq), None, _p(order), _s()), "rasterize_backward")
    d_ndc = torch.empty((V, 4, N), dtype=torch.float32, device=dev)
    d_ic = torch.empty((V, 2, 2, N), dtype=torch.float32, device=dev)
    d_color = torch.empty((V, 3, N), dtype=torch.float32, device=dev)
    d_opa = torch.empty((1, N), dtype=torch.float32, device=dev)
    sc = None
    if grad_inv_sacler_arg is not None:
        sc = _f32(grad_inv_sacler_arg.reshape(1), "grad_inv_scaler")
    check(L.lg_unpack_gradient(_p(pg), _p(packed), _p(sc), None, V, N, int(img_h), int(img_w), _p(d_ndc), _p(d_ic), _p(d_color), _p(d_opa), _s()), "unpack_gradient")
    return [d_ndc, d_ic, d_color, d_opa, err_sum, err_sq]


EXPORTS = [
    "create_viewproj_forward", "create_viewproj_backward", "create_table", "tileRange", "get_allocate_size", "rasterize_forward",
    "rasterize_forward_packed", "rasterize_backward", "jacobianRayspace", "createTransformMatrix_forward", "createTransformMatrix_backward",
    "world2ndc_forward", "world2ndc_backword", "mvp_transform_forward", "mvp_transform_backward", "createCov2dDirectly_forward",
    "createCov2dDirectly_backward", "sh2rgb_forward", "sh2rgb_backward", "eigh_and_inv_2x2matrix_forward", "inv_2x2matrix_backward",
    "cull_compact_activate", "activate_backward", "adamUpdate", "frustum_culling_aabb", "gpu_driven_pipeline_sparse_op",
]
