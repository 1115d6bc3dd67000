"""CPU oracle for the LiteGS render hot path -- numpy front-end over ``litegs_oracle.c``.

TEST INFRASTRUCTURE ONLY (see the header of ``litegs_oracle.c``): imported by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; never by ``litegs_amd``.

Pinning status: the per-Gaussian chain (SH, frustum planes/culling, transform matrix, ray-space
Jacobian, cov2d, 2x2 eigen/inverse) is pinned against the reference's own ``_script`` twins and
helpers imported from /root/reference (fixtures in ``tests/golden/``, generator
``tests/golden/make_golden.py``); so are the learnable-camera matrices (litegs/data.py).  The reference ships
NO executable twin for binning, the tile sort, the blend forward/backward, cull/activate or Adam,
so for those the oracle is pinned only against independent formulations (``tests/test_oracle_autograd.py``: dense
float64 torch-autograd blend, float64 ellipse/rectangle intersection for the tile walk, numpy/torch for the rest):
"parity unpinned vs the reference binary" for those rows, stated in DESIGN.md.

Array conventions follow the reference: SoA with the Gaussian index innermost, float32, C-contiguous.
"""
from __future__ import annotations

import contextlib
import ctypes
import dataclasses
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblitegs_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, a second or two)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("litegs_oracle.c", "litegs_oracle_fp16.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblitegs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_logf_export.restype = ctypes.c_float
        _lib.orc_logf_export.argtypes = [ctypes.c_float]
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


_i = ctypes.c_int
_l = ctypes.c_int64
_f = ctypes.c_float


# --------------------------------------------------------------------------- per-op wrappers
def frustum_culling_aabb(origin, ext, planes):
    """-> (visibility bool[M], visible_chunk_id int64[nvis] ascending)."""
    origin, ext, planes = _f32(origin), _f32(ext), _f32(planes)
    M = origin.shape[1]
    vis = np.zeros(M, dtype=np.uint8)
    lib().orc_frustum_culling_aabb(_p(origin), _p(ext), _p(planes), _i(planes.shape[0]), _i(M), _p(vis))
    return vis.astype(bool), np.nonzero(vis)[0].astype(np.int64)


def activate_forward(degree, chunk_id, nvis, view, pos, scale, rot, sh0, shr, opa, alloc=None):
    chunk_id = np.ascontiguousarray(chunk_id, dtype=np.int64)
    A = int(alloc if alloc is not None else len(chunk_id))
    C, S = pos.shape[-2:]
    V = view.shape[0]
    o_pos = np.zeros((4, A, S), np.float32)
    o_scale = np.zeros((3, A, S), np.float32)
    o_rot = np.zeros((4, A, S), np.float32)
    o_color = np.zeros((V, 3, A, S), np.float32)
    o_opa = np.zeros((1, A, S), np.float32)
    lib().orc_activate_forward(_i(degree), _p(chunk_id), _i(nvis), _i(A), _p(_f32(view)), _i(V),
                               _p(_f32(pos)), _p(_f32(scale)), _p(_f32(rot)), _p(_f32(sh0)), _p(_f32(shr)),
                               _p(_f32(opa)), _i(C), _i(S), _p(o_pos), _p(o_scale), _p(o_rot), _p(o_color), _p(o_opa))
    return o_pos, o_scale, o_rot, o_color, o_opa


def activate_backward(degree, chunk_id, nvis, view, pos, scale, rot, sh0, shr, opa,
                      g_pos, g_scale, g_rot, g_color, g_opa):
    chunk_id = np.ascontiguousarray(chunk_id, dtype=np.int64)
    A = g_pos.shape[-2]
    C, S = pos.shape[-2:]
    R = shr.shape[0]
    V = view.shape[0]
    d_pos = np.zeros((3, A, S), np.float32)
    d_scale = np.zeros((3, A, S), np.float32)
    d_rot = np.zeros((4, A, S), np.float32)
    d_sh0 = np.zeros((1, 3, A, S), np.float32)
    d_shr = np.zeros((R, 3, A, S), np.float32)
    d_opa = np.zeros((1, A, S), np.float32)
    lib().orc_activate_backward(_i(degree), _p(chunk_id), _i(nvis), _i(A), _p(_f32(view)), _i(V),
                                _p(_f32(pos)), _p(_f32(scale)), _p(_f32(rot)), _p(_f32(sh0)), _p(_f32(shr)),
                                _p(_f32(opa)), _i(C), _i(S), _i(R),
                                _p(_f32(g_pos)), _p(_f32(g_scale)), _p(_f32(g_rot)), _p(_f32(g_color)), _p(_f32(g_opa)),
                                _p(d_pos), _p(d_scale), _p(d_rot), _p(d_sh0), _p(d_shr), _p(d_opa))
    return d_pos, d_scale, d_rot, d_sh0, d_shr, d_opa


def sh2rgb_forward(degree, sh0, shr, dirs):
    V, _, N = dirs.shape
    rgb = np.zeros((V, 3, N), np.float32)
    lib().orc_sh2rgb_forward(_i(degree), _p(_f32(sh0)), _p(_f32(shr)), _p(_f32(dirs)), _i(V), _i(N), _p(rgb))
    return rgb


def sh2rgb_backward(degree, g_rgb, rest_dim, dirs):
    V, _, N = dirs.shape
    d_sh0 = np.zeros((1, 3, N), np.float32)
    d_shr = np.zeros((rest_dim, 3, N), np.float32)
    d_dirs = np.zeros((V, 3, N), np.float32)
    lib().orc_sh2rgb_backward(_i(degree), _p(_f32(g_rgb)), _p(_f32(dirs)), _i(V), _i(N), _i(rest_dim),
                              _p(d_sh0), _p(d_shr), _p(d_dirs))
    return d_sh0, d_shr, d_dirs


def mvp_forward(world, view, proj, valid=None):
    V, N = view.shape[0], world.shape[1]
    valid = N if valid is None else int(valid)
    vp = np.zeros((V, 4, N), np.float32)
    ndc = np.zeros((V, 4, N), np.float32)
    lib().orc_mvp_forward(_p(_f32(world)), _p(_f32(view)), _p(_f32(proj)), _i(V), _i(N), _i(valid), _p(vp), _p(ndc))
    return vp, ndc


def mvp_backward(g_ndc, g_view, view, proj, view_pos, valid=None):
    V, _, N = g_ndc.shape
    valid = N if valid is None else int(valid)
    gw = np.zeros((4, N), np.float32)
    lib().orc_mvp_backward(_p(_f32(g_ndc)), _p(_f32(g_view)), _p(_f32(view)), _p(_f32(proj)), _p(_f32(view_pos)),
                           _i(V), _i(N), _i(valid), _p(gw))
    return gw


def transform_matrix_forward(quat, scale, valid=None):
    N = quat.shape[1]
    valid = N if valid is None else int(valid)
    T = np.zeros((3, 3, N), np.float32)
    lib().orc_transform_matrix_forward(_p(_f32(quat)), _p(_f32(scale)), _i(N), _i(valid), _p(T))
    return T


def transform_matrix_backward(gT, quat, scale, valid=None):
    N = quat.shape[1]
    valid = N if valid is None else int(valid)
    gq = np.zeros((4, N), np.float32)
    gs = np.zeros((3, N), np.float32)
    lib().orc_transform_matrix_backward(_p(_f32(gT)), _p(_f32(quat)), _p(_f32(scale)), _i(N), _i(valid), _p(gq), _p(gs))
    return gq, gs


def jacobian_rayspace(view_pos, proj, H, W, valid=None):
    V, _, N = view_pos.shape
    valid = N if valid is None else int(valid)
    J = np.zeros((V, 3, 3, N), np.float32)
    lib().orc_jacobian_rayspace(_p(_f32(view_pos)), _p(_f32(proj)), _i(V), _i(N), _i(valid), _i(H), _i(W), _p(J))
    return J


def cov2d_forward(J, view, T, valid=None):
    V, N = view.shape[0], T.shape[2]
    valid = N if valid is None else int(valid)
    cov = np.zeros((V, 2, 2, N), np.float32)
    lib().orc_cov2d_forward(_p(_f32(J)), _p(_f32(view)), _p(_f32(T)), _i(V), _i(N), _i(valid), _p(cov))
    return cov


def cov2d_backward(g_cov, J, view, T, valid=None):
    V, N = view.shape[0], T.shape[2]
    valid = N if valid is None else int(valid)
    gT = np.zeros((3, 3, N), np.float32)
    lib().orc_cov2d_backward(_p(_f32(g_cov)), _p(_f32(J)), _p(_f32(view)), _p(_f32(T)), _i(V), _i(N), _i(valid), _p(gT))
    return gT


def eigh_inv_forward(cov, valid=None):
    V, _, _, N = cov.shape
    valid = N if valid is None else int(valid)
    val = np.zeros((V, 2, N), np.float32)
    vec = np.zeros((V, 2, 2, N), np.float32)
    inv = np.zeros((V, 2, 2, N), np.float32)
    lib().orc_eigh_inv_forward(_p(_f32(cov)), _i(V), _i(N), _i(valid), _p(val), _p(vec), _p(inv))
    return val, vec, inv


def inv2x2_backward(inv, g_inv, valid=None):
    V, _, _, N = inv.shape
    valid = N if valid is None else int(valid)
    g = np.zeros((V, 2, 2, N), np.float32)
    lib().orc_inv2x2_backward(_p(_f32(inv)), _p(_f32(g_inv)), _i(V), _i(N), _i(valid), _p(g))
    return g


def get_allocate_size(ndc, view_z, inv_cov, opacity, H, W, TH, TW, valid=None):
    V, _, N = ndc.shape
    valid = N if valid is None else int(valid)
    lu = np.zeros((V, 2, N), np.int32)
    rd = np.zeros((V, 2, N), np.int32)
    al = np.zeros((V, N), np.int32)
    lib().orc_get_allocate_size(_p(_f32(ndc)), _p(_f32(view_z)), _p(_f32(inv_cov)), _p(_f32(opacity)),
                                _i(V), _i(N), _i(valid), _i(H), _i(W), _i(TH), _i(TW), _p(lu), _p(rd), _p(al))
    return lu, rd, al


def sort_bits(H, W, TH, TW):
    """GR/binning.cu:199-202."""
    max_tiles = ((H + TH - 1) // TH) * ((W + TW - 1) // TW)
    bit = 0
    while max_tiles >> 1:
        max_tiles >>= 1
        bit += 1
    return bit + 1


def stable_sort_pairs(keys, values, bits):
    keys = np.ascontiguousarray(keys, np.int32)
    values = np.ascontiguousarray(values, np.int32)
    ko, vo = np.empty_like(keys), np.empty_like(values)
    if keys.ndim == 1:
        lib().orc_stable_sort_pairs(_p(keys), _p(values), _l(keys.shape[0]), _i(bits), _p(ko), _p(vo))
    else:
        for v in range(keys.shape[0]):
            kk, vv = np.ascontiguousarray(keys[v]), np.ascontiguousarray(values[v])
            k1, v1 = np.empty_like(kk), np.empty_like(vv)
            lib().orc_stable_sort_pairs(_p(kk), _p(vv), _l(kk.shape[0]), _i(bits), _p(k1), _p(v1))
            ko[v], vo[v] = k1, v1
    return ko, vo


def create_table(ndc, inv_cov, opacity, prefix, sorted_id, H, W, TH, TW, table_len=None):
    """duplicate_with_keys + stable sort -> (tileId_sorted, pointId_sorted) int32[V, table_len]."""
    V, _, N = ndc.shape
    prefix = np.ascontiguousarray(prefix, np.int32)
    sorted_id = np.ascontiguousarray(sorted_id, np.int64)
    if table_len is None:
        table_len = int(prefix[:, -1].max())
    table_len = max(int(table_len), 1)
    keys = np.zeros((V, table_len), np.int32)
    values = np.zeros((V, table_len), np.int32)
    lib().orc_duplicate_with_keys(_p(_f32(ndc)), _p(_f32(inv_cov)), _p(_f32(opacity)), _p(prefix), _p(sorted_id),
                                  _i(V), _i(N), _i(H), _i(W), _i(TH), _i(TW), _l(table_len), _p(keys), _p(values))
    ks, vs = stable_sort_pairs(keys, values, sort_bits(H, W, TH, TW))
    return ks, vs, keys, values


def tile_range(sorted_keys, max_tile):
    sorted_keys = np.ascontiguousarray(sorted_keys, np.int32)
    V, L = sorted_keys.shape
    out = np.empty((V, max_tile + 2), np.int32)
    lib().orc_tile_range(_p(sorted_keys), _i(V), _l(L), _i(max_tile), _p(out))
    return out


def pack_params(ndc, inv_cov, color, opacity, H, W):
    V, _, N = ndc.shape
    packed = np.zeros((V, N, 16), np.float32)
    lib().orc_pack_params(_p(_f32(ndc)), _p(_f32(inv_cov)), _p(_f32(color)), _p(_f32(opacity)), _i(V), _i(N), _i(H), _i(W), _p(packed))
    return packed


def padded_hw(H, W, TH, TW):
    return ((H + TH - 1) // TH) * TH, ((W + TW - 1) // TW) * TW


def raster_forward(sorted_points, start_index, packed, H, W, TH, TW, tiles=None, enable_stat=False):
    sorted_points = np.ascontiguousarray(sorted_points, np.int32)
    start_index = np.ascontiguousarray(start_index, np.int32)
    V, L = sorted_points.shape
    N = packed.shape[1]
    Hp, Wp = padded_hw(H, W, TH, TW)
    img = np.zeros((V, 3, Hp, Wp), np.float32)
    trans = np.ones((V, 1, Hp, Wp), np.float32)
    last = np.zeros((V, 1, Hp, Wp), np.int16)
    fc = np.zeros((V, 1, N), np.int32)
    fw = np.zeros((V, 1, N), np.float32)
    K = 0
    if tiles is not None:
        tiles = np.ascontiguousarray(tiles, np.int32)
        K = tiles.shape[1]
    lib().orc_raster_forward(_p(sorted_points), _p(start_index), _p(_f32(packed)), _p(tiles), _i(K),
                             _i(V), _l(L), _i(N), _i(H), _i(W), _i(TH), _i(TW), _i(int(enable_stat)),
                             _p(img), _p(trans), _p(last), _p(fc), _p(fw))
    return img, trans, last, fc, fw


def raster_backward(sorted_points, start_index, packed, final_T, last, d_img, H, W, TH, TW,
                    d_trans=None, inv_scaler=1.0, tiles=None, enable_stat=False):
    sorted_points = np.ascontiguousarray(sorted_points, np.int32)
    start_index = np.ascontiguousarray(start_index, np.int32)
    V, L = sorted_points.shape
    N = packed.shape[1]
    d_ndc = np.zeros((V, 4, N), np.float32)
    d_ic = np.zeros((V, 2, 2, N), np.float32)
    d_color = np.zeros((V, 3, N), np.float32)
    d_opa = np.zeros((1, N), np.float32)
    esq = np.zeros((V, 1, N), np.float32)
    K = 0
    if tiles is not None:
        tiles = np.ascontiguousarray(tiles, np.int32)
        K = tiles.shape[1]
    lib().orc_raster_backward(_p(sorted_points), _p(start_index), _p(_f32(packed)), _p(tiles), _i(K),
                              _p(_f32(final_T)), _p(np.ascontiguousarray(last, np.int16)), _p(_f32(d_img)),
                              _p(_f32(d_trans)) if d_trans is not None else None, _f(inv_scaler),
                              _i(V), _l(L), _i(N), _i(H), _i(W), _i(TH), _i(TW), _i(int(enable_stat)),
                              _p(d_ndc), _p(d_ic), _p(d_color), _p(d_opa), _p(esq))
    return d_ndc, d_ic, d_color, d_opa, esq


@contextlib.contextmanager
def blend_thresholds(alpha_scale=1.0, t_scale=1.0):
    """the blend's two decision thresholds (alpha >= 1/256, T > 1/8192) scaled for the duration of the block -- tests/util.py's bracket
    rule; every oracle blend outside such a block uses the reference's constants"""
    lib().orc_set_blend_thresholds(_f(np.float32(1.0 / 256) * np.float32(alpha_scale)), _f(np.float32(1.0 / 8192) * np.float32(t_scale)))
    try:
        yield
    finally:
        lib().orc_set_blend_thresholds(_f(1.0 / 256), _f(1.0 / 8192))


def reblend(res, H, W, tile=(8, 16)):
    """a copy of a PipelineResult whose blend outputs (img, trans, last) are recomputed from its table and records -- under whatever
    blend_thresholds block is active"""
    img, trans, last, fc, fw = raster_forward(res.sorted_point, res.tile_start, res.packed, H, W, tile[0], tile[1])
    return dataclasses.replace(res, img=img, trans=trans, last=last, frag_count=fc, frag_weight=fw)


def raster_decisions(sorted_points, start_index, packed, H, W, TH, TW, delta=1e-3):
    """-> (near_px bool[V,1,Hp,Wp], near_splat bool[V,N]): where a blend threshold decision (alpha >= 1/256, T > 1/8192) sits within a
    relative `delta` of its threshold, and which splats' gradient sums contain terms behind such a decision (orc_raster_decisions)."""
    sorted_points = np.ascontiguousarray(sorted_points, np.int32)
    start_index = np.ascontiguousarray(start_index, np.int32)
    V, L = sorted_points.shape
    N = packed.shape[1]
    Hp, Wp = padded_hw(H, W, TH, TW)
    near_px = np.zeros((V, 1, Hp, Wp), np.uint8)
    near_splat = np.zeros((V, N), np.uint8)
    lib().orc_raster_decisions(_p(sorted_points), _p(start_index), _p(_f32(packed)), _i(V), _l(L), _i(N), _i(H), _i(W), _i(TH), _i(TW),
                               _f(delta), _p(near_px), _p(near_splat))
    return near_px.astype(bool), near_splat.astype(bool)


# ---- the reference BINARY's blend arithmetic (half2, x128 transmittance scale), emulated: litegs_oracle_fp16.c -------------------
def pack_params_fp16(packed):
    """colour and opacity rounded to binary16 as GR/raster.cu:353-354 packs them (oracle record slots 5..8)"""
    out = np.array(packed, dtype=np.float32, copy=True)
    out[..., 5:9] = out[..., 5:9].astype(np.float16).astype(np.float32)
    return out


def raster_forward_fp16(sorted_points, start_index, packed, H, W, TH, TW):
    """-> img, trans, last as the reference's half2 forward kernel computes them (packed: pack_params_fp16 output)"""
    sorted_points = np.ascontiguousarray(sorted_points, np.int32)
    start_index = np.ascontiguousarray(start_index, np.int32)
    V, L = sorted_points.shape
    N = packed.shape[1]
    Hp, Wp = padded_hw(H, W, TH, TW)
    img = np.zeros((V, 3, Hp, Wp), np.float32)
    trans = np.ones((V, 1, Hp, Wp), np.float32)
    last = np.zeros((V, 1, Hp, Wp), np.int16)
    lib().orc_raster_forward_fp16(_p(sorted_points), _p(start_index), _p(_f32(packed)), _i(V), _l(L), _i(N), _i(H), _i(W), _i(TH), _i(TW),
                                  _p(img), _p(trans), _p(last))
    return img, trans, last


def raster_backward_fp16(sorted_points, start_index, packed, final_T, last, d_img, H, W, TH, TW, inv_scaler=1.0):
    """-> d_ndc, d_inv_cov, d_color, d_opacity as the reference's half2 backward kernel + unpack compute them"""
    sorted_points = np.ascontiguousarray(sorted_points, np.int32)
    start_index = np.ascontiguousarray(start_index, np.int32)
    V, L = sorted_points.shape
    N = packed.shape[1]
    d_ndc = np.zeros((V, 4, N), np.float32)
    d_ic = np.zeros((V, 2, 2, N), np.float32)
    d_color = np.zeros((V, 3, N), np.float32)
    d_opa = np.zeros((1, N), np.float32)
    lib().orc_raster_backward_fp16(_p(sorted_points), _p(start_index), _p(_f32(packed)), _p(_f32(final_T)), _p(np.ascontiguousarray(last, np.int16)),
                                   _p(_f32(d_img)), _f(inv_scaler), _i(V), _l(L), _i(N), _i(H), _i(W), _i(TH), _i(TW),
                                   _p(d_ndc), _p(d_ic), _p(d_color), _p(d_opa))
    return d_ndc, d_ic, d_color, d_opa


def adam_chunk(param, grad, m, v, chunk_id, nvis, lr, b1=0.9, b2=0.999, eps=1e-15):
    """in place on param/m/v ([C,chunks,S] float32 contiguous)."""
    C, chunks, S = param.shape
    A = grad.shape[1]
    lib().orc_adam_chunk(_p(param), _p(_f32(grad)), _p(m), _p(v), _p(np.ascontiguousarray(chunk_id, np.int64)),
                         _i(nvis), _i(C), _i(chunks), _i(A), _i(S), _f(lr), _f(b1), _f(b2), _f(eps))


def adam_primitive(param, grad, m, v, mask, lr, b1=0.9, b2=0.999, eps=1e-15):
    C, N = param.shape
    lib().orc_adam_primitive(_p(param), _p(_f32(grad)), _p(m), _p(v), _p(np.ascontiguousarray(mask, np.int64)),
                             _i(C), _i(N), _f(lr), _f(b1), _f(b2), _f(eps))


def sparse_scatter(A, B, chunk_id, nvis, op):
    opc = {"add": 0, "sum": 0, "min": 1, "max": 2}[op]
    E, chunks, S = A.shape
    alloc = B.shape[1]
    cid = np.ascontiguousarray(chunk_id, np.int64)
    if A.dtype == np.float32:
        lib().orc_sparse_scatter_f32(_p(A), _p(np.ascontiguousarray(B, np.float32)), _p(cid), _i(nvis), _i(E), _i(chunks), _i(alloc), _i(S), _i(opc))
    elif A.dtype == np.int32:
        lib().orc_sparse_scatter_i32(_p(A), _p(np.ascontiguousarray(B, np.int32)), _p(cid), _i(nvis), _i(E), _i(chunks), _i(alloc), _i(S), _i(opc))
    else:
        raise TypeError(A.dtype)


def logf(x: float) -> float:
    return float(lib().orc_logf_export(_f(x)))


# --------------------------------------------------------------------------- whole pipeline
@dataclass
class PipelineResult:
    visible_chunkid: np.ndarray
    nvis: int
    act: tuple            # activated pos, scale, rot, color, opacity (flattened [C, N])
    view_pos: np.ndarray
    ndc: np.ndarray
    T: np.ndarray
    J: np.ndarray
    cov2d: np.ndarray
    inv_cov: np.ndarray
    alloc: np.ndarray
    depth_sorted_index: np.ndarray
    prefix: np.ndarray
    sorted_tile: np.ndarray
    sorted_point: np.ndarray
    tile_start: np.ndarray
    packed: np.ndarray
    img: np.ndarray        # padded [V,3,Hp,Wp], min(C,1)
    trans: np.ndarray
    last: np.ndarray
    frag_count: np.ndarray
    frag_weight: np.ndarray
    n_instances: int


def render_forward(params, view, proj, planes, H, W, degree=3, tile=(8, 16), cluster_aabb=None,
                   enable_stat=False) -> PipelineResult:
    """render_preprocess + render (litegs/render/__init__.py:11-94) for V=1 in fp32 on the CPU.

    params = (xyz[3,C,S], scale[3,C,S], rot[4,C,S], sh0[1,3,C,S], shr[R,3,C,S], opacity[1,C,S]) raw
    (pre-activation) values, chunked as litegs/scene/cluster.py:7-21 does.
    """
    xyz, scale, rot, sh0, shr, opa = [_f32(p) for p in params]
    TH, TW = tile
    C, S = xyz.shape[-2:]
    view, proj, planes = _f32(view), _f32(proj), _f32(planes)
    if cluster_aabb is None:
        cluster_aabb = cluster_AABB(xyz, scale, rot)
    origin, extend = cluster_aabb
    _, chunk_id = frustum_culling_aabb(origin, extend, planes)
    nvis = len(chunk_id)
    if nvis == 0:                                   # nothing in the frustum: empty tables, background image (the kernels are never launched)
        V = view.shape[0]
        Hp, Wp = (H + TH - 1) // TH * TH, (W + TW - 1) // TW * TW
        ntiles = (Hp // TH) * (Wp // TW)
        e = lambda *shape: np.zeros(shape, np.float32)
        ei = lambda *shape: np.zeros(shape, np.int32)
        return PipelineResult(chunk_id, 0, (e(4, 0), e(3, 0), e(4, 0), e(V, 3, 0), e(1, 0)), e(V, 4, 0), e(V, 4, 0), e(3, 3, 0), e(V, 3, 3, 0),
                              e(V, 2, 2, 0), e(V, 2, 2, 0), ei(V, 0), np.zeros((V, 0), np.int64), ei(V, 0), ei(V, 0), ei(V, 0),
                              np.full((V, ntiles + 2), -1, np.int32), e(V, 0, 16), e(V, 3, Hp, Wp), np.ones((V, 1, Hp, Wp), np.float32),
                              np.zeros((V, 1, Hp, Wp), np.int16), ei(V, 1, 0), e(V, 1, 0), 0)
    a_pos, a_scale, a_rot, a_color, a_opa = activate_forward(degree, chunk_id, nvis, view, xyz, scale, rot, sh0, shr, opa)
    N = nvis * S
    pos = a_pos.reshape(4, N); sc = a_scale.reshape(3, N); rt = a_rot.reshape(4, N)
    col = a_color.reshape(view.shape[0], 3, N); op = a_opa.reshape(1, N)
    view_pos, ndc = mvp_forward(pos, view, proj)
    T = transform_matrix_forward(rt, sc)
    J = jacobian_rayspace(view_pos, proj, H, W)
    cov = cov2d_forward(J, view, T)
    _, _, inv_cov = eigh_inv_forward(cov)
    view_depth = np.ascontiguousarray(view_pos[:, 2, :])
    _, _, alloc = get_allocate_size(ndc, view_depth, inv_cov, op, H, W, TH, TW)
    # wrapper.py:739-745 -- stable sort so ties are well defined
    dsi = np.argsort(view_depth, axis=-1, kind="stable").astype(np.int64)
    alloc_sorted = np.take_along_axis(alloc, dsi, axis=-1)
    prefix = np.cumsum(alloc_sorted, axis=-1, dtype=np.int64).astype(np.int32)
    st, spt, _, _ = create_table(ndc, inv_cov, op, prefix, dsi, H, W, TH, TW)
    ntiles = ((H + TH - 1) // TH) * ((W + TW - 1) // TW)
    ts = tile_range(st, ntiles)
    packed = pack_params(ndc, inv_cov, col, op, H, W)
    img, trans, last, fc, fw = raster_forward(spt, ts, packed, H, W, TH, TW, enable_stat=enable_stat)
    return PipelineResult(chunk_id, nvis, (pos, sc, rt, col, op), view_pos, ndc, T, J, cov, inv_cov, alloc, dsi, prefix,
                          st, spt, ts, packed, img, trans, last, fc, fw, int(prefix[:, -1].max()))


def render_backward(res: PipelineResult, params, view, proj, d_img_padded, H, W, degree=3, tile=(8, 16)):
    """autograd chain of wrapper.py:481-524, :588-592, :404-407, :190-193, :278-285, :820-845 on the CPU.

    Returns the six compacted parameter gradients [C, nvis, S] (what CompactedTensor carries).
    """
    xyz, scale, rot, sh0, shr, opa = [_f32(p) for p in params]
    TH, TW = tile
    S = xyz.shape[-1]
    pos, sc, rt, col, op = res.act
    d_ndc, d_ic, d_color, d_opa, _ = raster_backward(res.sorted_point, res.tile_start, res.packed, res.trans, res.last,
                                                     d_img_padded, H, W, TH, TW)
    g_cov = np.nan_to_num(inv2x2_backward(res.inv_cov, d_ic), nan=0.0, posinf=0.0, neginf=0.0)
    gT = cov2d_backward(g_cov, res.J, view, res.T)
    g_rot, g_scale = transform_matrix_backward(gT, rt, sc)
    g_pos = mvp_backward(d_ndc, np.zeros_like(res.view_pos), view, proj, res.view_pos)
    A = res.nvis
    return activate_backward(degree, res.visible_chunkid, res.nvis, view, xyz, scale, rot, sh0, shr, opa,
                             g_pos.reshape(4, A, S), g_scale.reshape(3, A, S), g_rot.reshape(4, A, S),
                             d_color.reshape(view.shape[0], 3, A, S), d_opa.reshape(1, A, S)), (d_ndc, d_ic, d_color, d_opa)


def cluster_AABB(xyz, scale_raw, rot_raw):
    """litegs/scene/cluster.py:29-46 on activated scale/rot."""
    C, S = xyz.shape[-2:]
    N = C * S
    sc = np.exp(_f32(scale_raw)).reshape(3, N)
    rt = _f32(rot_raw).reshape(4, N)
    rt = rt / np.maximum(np.linalg.norm(rt, axis=0, keepdims=True), 1e-12)
    T = transform_matrix_forward(rt.astype(np.float32), sc.astype(np.float32))
    coefficient = 2 * np.log(255.0)
    ext_axis = T * np.float32(np.sqrt(coefficient))
    point_ext = np.abs(ext_axis).sum(axis=0).reshape(3, C, S)
    mx = (xyz + point_ext).max(axis=-1)
    mn = (xyz - point_ext).min(axis=-1)
    return ((mx + mn) / 2).astype(np.float32), ((mx - mn) / 2).astype(np.float32)


# ------------------------------------------------------------------------------------------- learnable cameras
def create_viewproj_forward(view_params, fov, H, W, zn, zf):
    """GR/compact.cu:17-135"""
    vp7 = _f32(view_params); V = vp7.shape[0]
    view, proj, vp = (np.zeros((V, 4, 4), np.float32) for _ in range(3))
    planes = np.zeros((V, 6, 4), np.float32)
    lib().orc_create_viewproj_forward(_p(vp7), _p(_f32(fov)), _i(V), _i(H), _i(W), ctypes.c_float(zn), ctypes.c_float(zf),
                                      _p(view), _p(proj), _p(vp), _p(planes))
    return view, proj, vp, planes


def create_viewproj_backward(g_view, g_proj, g_vp, view_params, fov, H, W, zn, zf):
    """GR/compact.cu:137-316"""
    vp7 = _f32(view_params); V = vp7.shape[0]
    gp, gf = np.zeros((V, 7), np.float32), np.zeros((1,), np.float32)
    lib().orc_create_viewproj_backward(_p(_f32(g_view)), _p(_f32(g_proj)), _p(_f32(g_vp)), _p(vp7), _p(_f32(fov)), _i(V), _i(H), _i(W),
                                       ctypes.c_float(zn), ctypes.c_float(zf), _p(gp), _p(gf))
    return gp, gf
