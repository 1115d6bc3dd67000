// `litegs_fused` as a compiled torch extension (the reference builds the same module name from GR/ext_cuda.cpp:9-35 with
// GR/setup.py:20-35).  Host-only C++: every function checks its tensors, allocates the outputs with ATen and calls the C ABI of
// liblitegs_hip.so (include/litegs_hip.h) on torch's current HIP stream.  Names, positional order and returned tensors follow
// GR/{binning,compact,raster,transform}.h; litegs_amd/fused.py is the same binding through ctypes (kept for environments without a
// C++ toolchain) and the two are tested against each other.  No computation happens here and there is no CPU path.
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <optional>
#include <string>
#include <vector>
#include "litegs_hip.h"

namespace {

using at::Tensor;
typedef std::optional<Tensor> OptTensor;

inline void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

inline void check(int rc, const char* what) { TORCH_CHECK(rc == 0, "litegs_fused: ", what, " failed with hipError ", rc); }

inline Tensor dev(const Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), "litegs_fused: '", name, "' must live on the GPU (litegs_amd has no CPU path)");
    return t.is_contiguous() ? t : t.contiguous();
}
inline Tensor f32(const Tensor& t, const char* name)
{
    Tensor c = dev(t, name);
    TORCH_CHECK(c.scalar_type() == at::kFloat, "litegs_fused: '", name, "' must be float32");
    return c;
}
inline const int* vl(const OptTensor& v)
{
    if (!v.has_value()) return nullptr;
    TORCH_CHECK(v->is_cuda() && v->scalar_type() == at::kInt, "litegs_fused: valid_length must be a device int32 tensor");
    return v->data_ptr<int>();
}
inline const float* fp(const Tensor& t) { return t.data_ptr<float>(); }
inline at::TensorOptions like(const Tensor& t, at::ScalarType dt) { return t.options().dtype(dt); }

struct TileShape { int gx, gy, ntiles, Hp, Wp; };
inline TileShape tiles_shape(int64_t h, int64_t w, int64_t th, int64_t tw)
{
    TileShape s;
    s.gx = (int)((w + tw - 1) / tw); s.gy = (int)((h + th - 1) / th);
    s.ntiles = s.gx * s.gy; s.Hp = s.gy * (int)th; s.Wp = s.gx * (int)tw;
    return s;
}

// ------------------------------------------------------------------------------------------------ compact.h
std::vector<Tensor> frustum_culling_aabb(Tensor aabb_origin, Tensor aabb_ext, Tensor frustumplane, OptTensor feedback_buffer_arg, OptTensor data_idx_arg)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(aabb_origin));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor o = f32(aabb_origin, "aabb_origin"), e = f32(aabb_ext, "aabb_ext"), p = f32(frustumplane, "frustumplane");
    const int V = (int)p.size(0), M = (int)o.size(1);
    Tensor visibility = at::empty({M}, like(p, at::kBool)), num = at::empty({1}, like(p, at::kInt)), ids = at::empty({M}, like(p, at::kLong));
    void* s = cur_stream();
    check(lg_frustum_culling_aabb(fp(o), fp(e), fp(p), V, M, (uint8_t*)visibility.data_ptr(), num.data_ptr<int>(), ids.data_ptr<int64_t>(), s),
          "frustum_culling_aabb");
    int64_t pred = 0;
    if (feedback_buffer_arg.has_value() && data_idx_arg.has_value()) {
        // the reference reads `(*data_idx_arg)[i].item()` (GR/compact.cu:527): any device, any integer dtype
        TORCH_CHECK(!feedback_buffer_arg->is_cuda() && feedback_buffer_arg->scalar_type() == at::kInt && feedback_buffer_arg->is_contiguous(),
                    "frustum_culling_aabb: feedback_buffer must be a contiguous (pinned) CPU int32 tensor");
        const Tensor idx_host = data_idx_arg->to(at::kCPU, at::kLong).contiguous();
        int* base = feedback_buffer_arg->data_ptr<int>();
        const int64_t* idx = idx_host.data_ptr<int64_t>();
        for (int64_t i = 0; i < idx_host.size(0); i++) {
            TORCH_CHECK(idx[i] >= 0 && idx[i] < feedback_buffer_arg->numel(), "frustum_culling_aabb: data index outside the feedback buffer");
            pred = std::max<int64_t>(pred, base[idx[i]]);
            check(lg_feedback_d2h(base + idx[i], num.data_ptr<int>(), s), "feedback copy");
        }
    }
    pred = (int64_t)(1.2 * (double)pred);
    if (pred <= 0) pred = num.item<int>();          // blocking path, first time a frame is seen (compact.cu:543-546)
    return { visibility, num, ids.slice(0, 0, pred) };
}

std::vector<Tensor> cull_compact_activate(int sh_degree, Tensor visible_chunk_id, Tensor visible_chunks_num, Tensor view_matrix,
                                          Tensor position, Tensor scale, Tensor rotation, Tensor sh_base, Tensor sh_rest, Tensor opacity)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(visible_chunk_id));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor pos = f32(position, "position"), sc = f32(scale, "scale"), rot = f32(rotation, "rotation");
    Tensor s0 = f32(sh_base, "sh_base"), sr = f32(sh_rest, "sh_rest"), op = f32(opacity, "opacity"), vm = f32(view_matrix, "view_matrix");
    Tensor ids = dev(visible_chunk_id, "visible_chunk_id");
    const int chunks = (int)pos.size(-2), S = (int)pos.size(-1), A = (int)ids.size(0), V = (int)vm.size(0);
    auto o = pos.options();
    Tensor o_pos = at::empty({4, A, S}, o), o_scale = at::empty({3, A, S}, o), o_rot = at::empty({4, A, S}, o);
    Tensor o_color = at::empty({V, 3, A, S}, o), o_opa = at::empty({1, A, S}, o);
    check(lg_cull_compact_activate(sh_degree, ids.data_ptr<int64_t>(), visible_chunks_num.data_ptr<int>(), A, fp(vm), V, fp(pos), fp(sc), fp(rot),
                                   fp(s0), fp(sr), fp(op), chunks, S, o_pos.data_ptr<float>(), o_scale.data_ptr<float>(), o_rot.data_ptr<float>(),
                                   o_color.data_ptr<float>(), o_opa.data_ptr<float>(), cur_stream()), "cull_compact_activate");
    return { o_pos, o_scale, o_rot, o_color, o_opa };
}

std::vector<Tensor> activate_backward(int sh_degree, Tensor visible_chunk_id, Tensor visible_chunks_num, Tensor view_matrix,
                                      Tensor position, Tensor scale, Tensor rotation, Tensor sh_base, Tensor sh_rest, Tensor opacity,
                                      Tensor activated_position_grad, Tensor activated_scale_grad, Tensor activated_rotation_grad,
                                      Tensor color_grad, Tensor activated_opacity_grad)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(visible_chunk_id));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor pos = f32(position, "position"), sc = f32(scale, "scale"), rot = f32(rotation, "rotation"), op = f32(opacity, "opacity");
    Tensor vm = f32(view_matrix, "view_matrix"), ids = dev(visible_chunk_id, "visible_chunk_id");
    Tensor gp = f32(activated_position_grad, "g_pos"), gs = f32(activated_scale_grad, "g_scale"), gr = f32(activated_rotation_grad, "g_rot");
    Tensor gc = f32(color_grad, "g_color"), go = f32(activated_opacity_grad, "g_opa");
    const int chunks = (int)pos.size(-2), S = (int)pos.size(-1), A = (int)ids.size(0), V = (int)vm.size(0), R = (int)sh_rest.size(0);
    auto o = pos.options();
    Tensor d_pos = at::empty({pos.size(0), A, S}, o), d_scale = at::empty({3, A, S}, o), d_rot = at::empty({4, A, S}, o);
    Tensor d_sh0 = at::empty({sh_base.size(0), sh_base.size(1), A, S}, o), d_shr = at::empty({R, sh_rest.size(1), A, S}, o);
    Tensor d_opa = at::empty({1, A, S}, o);
    check(lg_activate_backward(sh_degree, ids.data_ptr<int64_t>(), visible_chunks_num.data_ptr<int>(), A, fp(vm), V, fp(pos), fp(sc), fp(rot), fp(op),
                               chunks, S, R, fp(gp), fp(gs), fp(gr), fp(gc), fp(go), d_pos.data_ptr<float>(), d_scale.data_ptr<float>(),
                               d_rot.data_ptr<float>(), d_sh0.data_ptr<float>(), d_shr.data_ptr<float>(), d_opa.data_ptr<float>(), cur_stream()),
          "activate_backward");
    return { d_pos, d_scale, d_rot, d_sh0, d_shr, d_opa };
}

void adamUpdate(Tensor param, Tensor param_grad, Tensor exp_avg, Tensor exp_avg_sq, Tensor visible_index, OptTensor valid_length,
                double lr, double b1, double b2, double eps)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(param));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    for (const Tensor* t : { &param, &param_grad, &exp_avg, &exp_avg_sq })
        TORCH_CHECK(t->is_cuda() && t->is_contiguous() && t->scalar_type() == at::kFloat, "adamUpdate: tensors must be contiguous float32 device tensors");
    Tensor vi = dev(visible_index, "visible_index");
    if (param.dim() == 3) {
        check(lg_adam_update_chunk(param.data_ptr<float>(), fp(param_grad), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(), vi.data_ptr<int64_t>(),
                                   vl(valid_length), (int)param.size(0), (int)param.size(1), (int)vi.size(0), (int)param.size(2), 0,
                                   (float)lr, (float)b1, (float)b2, (float)eps, cur_stream()), "adamUpdate");
    } else {
        TORCH_CHECK(param.dim() == 2, "adamUpdate: param must be [E,chunks,S] or [E,N]");
        check(lg_adam_update_primitive(param.data_ptr<float>(), fp(param_grad), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(),
                                       vi.data_ptr<int64_t>(), (int)param.size(0), (int)param.size(1), (float)lr, (float)b1, (float)b2, (float)eps,
                                       cur_stream()), "adamUpdate");
    }
}

int dtype_code(at::ScalarType t)
{
    switch (t) {
    case at::kFloat: return 0; case at::kInt: return 1; case at::kLong: return 2; case at::kDouble: return 3; case at::kShort: return 4;
    case at::kChar: case at::kByte: case at::kBool: return 5;
    default: TORCH_CHECK(false, "gpu_driven_pipeline_sparse_op: unsupported dtype"); return -1;
    }
}

void gpu_driven_pipeline_sparse_op(Tensor A, Tensor B, Tensor visible_chunk_ids, Tensor visible_count, std::string op_name)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(A));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    TORCH_CHECK(A.is_cuda() && B.is_cuda() && visible_chunk_ids.is_cuda() && visible_count.is_cuda(), "inputs must be CUDA tensors");
    int op;
    if (op_name == "add" || op_name == "sum") op = 0;
    else if (op_name == "min") op = 1;
    else if (op_name == "max") op = 2;
    else { TORCH_CHECK(false, "Unsupported op: ", op_name, ". Expected: add, min, max"); op = -1; }
    TORCH_CHECK(A.scalar_type() == B.scalar_type(), "gpu_driven_pipeline_sparse_op: dtype mismatch");
    TORCH_CHECK(A.is_contiguous(), "gpu_driven_pipeline_sparse_op: A must be contiguous (updated in place)");
    Tensor Bc = B.contiguous(), ids = visible_chunk_ids.contiguous();
    TORCH_CHECK(A.size(2) <= 1024, "chunk_size exceeds max threads per block");
    check(lg_sparse_scatter(A.data_ptr(), Bc.data_ptr(), ids.data_ptr<int64_t>(), visible_count.data_ptr<int>(), (int)A.size(0), (int)A.size(1),
                            (int)Bc.size(1), (int)A.size(2), dtype_code(A.scalar_type()), op, cur_stream()), "gpu_driven_pipeline_sparse_op");
}

std::vector<Tensor> create_viewproj_forward(Tensor view_params, Tensor recp_tan_half_fov_x, int img_h, int img_w, float z_near, float z_far)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(view_params));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor vp = f32(view_params, "view_params"), fov = f32(recp_tan_half_fov_x, "recp_tan_half_fov_x");
    TORCH_CHECK(vp.dim() == 2 && vp.size(1) == 7, "create_viewproj_forward: view_params must be [views,7]");
    const int V = (int)vp.size(0);
    auto o = vp.options();
    Tensor view = at::empty({V, 4, 4}, o), proj = at::empty({V, 4, 4}, o), vpm = at::empty({V, 4, 4}, o), planes = at::empty({V, 6, 4}, o);
    check(lg_create_viewproj_forward(fp(vp), fp(fov), V, img_h, img_w, z_near, z_far, view.data_ptr<float>(), proj.data_ptr<float>(),
                                     vpm.data_ptr<float>(), planes.data_ptr<float>(), cur_stream()), "create_viewproj_forward");
    return { view, proj, vpm, planes };
}

std::vector<Tensor> create_viewproj_backward(Tensor view_matrix_grad, Tensor proj_matrix_grad, Tensor viewproj_matrix_grad, Tensor view_params,
                                             Tensor recp_tan_half_fov_x, int img_h, int img_w, float z_near, float z_far)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(view_matrix_grad));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor vp = f32(view_params, "view_params"), fov = f32(recp_tan_half_fov_x, "recp_tan_half_fov_x");
    const int V = (int)vp.size(0);
    Tensor g0 = f32(view_matrix_grad, "view_matrix_grad"), g1 = f32(proj_matrix_grad, "proj_matrix_grad"), g2 = f32(viewproj_matrix_grad, "viewproj_matrix_grad");
    for (const Tensor* g : { &g0, &g1, &g2 })
        TORCH_CHECK(g->dim() == 3 && g->size(0) == V && g->size(1) == 4 && g->size(2) == 4, "create_viewproj_backward: matrix gradients must be [views,4,4]");
    Tensor gp = at::zeros_like(vp), gf = at::zeros_like(fov);
    check(lg_create_viewproj_backward(fp(g0), fp(g1), fp(g2), fp(vp), fp(fov), V, img_h, img_w, z_near, z_far, gp.data_ptr<float>(), gf.data_ptr<float>(),
                                      cur_stream()), "create_viewproj_backward");
    return { gp, gf };
}

// ---------------------------------------------------------------------------------------------- transform.h
std::vector<Tensor> mvp_transform_forward(Tensor world_position, Tensor view_matrix, Tensor proj_matrix, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(world_position));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor w = f32(world_position, "world_position"), vm = f32(view_matrix, "view_matrix"), pm = f32(proj_matrix, "proj_matrix");
    const int V = (int)vm.size(0), N = (int)w.size(1);
    Tensor view_pos = at::empty({V, 4, N}, w.options()), ndc_pos = at::empty({V, 4, N}, w.options());
    check(lg_mvp_transform_forward(fp(w), fp(vm), fp(pm), vl(valid_length), V, N, view_pos.data_ptr<float>(), ndc_pos.data_ptr<float>(), cur_stream()),
          "mvp_transform_forward");
    return { view_pos, ndc_pos };
}

Tensor mvp_transform_backward(Tensor grad_ndc_pos, Tensor grad_view_pos, Tensor view_matrix, Tensor proj_matrix, Tensor view_pos, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(grad_ndc_pos));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor gn = f32(grad_ndc_pos, "grad_ndc_pos"), gv = f32(grad_view_pos, "grad_view_pos"), vp = f32(view_pos, "view_pos");
    Tensor vm = f32(view_matrix, "view_matrix"), pm = f32(proj_matrix, "proj_matrix");
    const int V = (int)gn.size(0), N = (int)gn.size(2);
    Tensor g_world = at::empty({4, N}, gn.options());
    check(lg_mvp_transform_backward(fp(gn), fp(gv), fp(vm), fp(pm), fp(vp), vl(valid_length), V, N, g_world.data_ptr<float>(), cur_stream()),
          "mvp_transform_backward");
    return g_world;
}

Tensor createTransformMatrix_forward(Tensor quaternion, Tensor scale, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(quaternion));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor q = f32(quaternion, "quaternion"), sc = f32(scale, "scale");
    const int N = (int)q.size(1);
    Tensor T = at::empty({3, 3, N}, sc.options());
    check(lg_create_transform_matrix_forward(fp(q), fp(sc), vl(valid_length), N, T.data_ptr<float>(), cur_stream()), "createTransformMatrix_forward");
    return T;
}

std::vector<Tensor> createTransformMatrix_backward(Tensor transform_matrix_grad, Tensor quaternion, Tensor scale, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(transform_matrix_grad));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor g = f32(transform_matrix_grad, "transform_matrix_grad"), q = f32(quaternion, "quaternion"), sc = f32(scale, "scale");
    const int N = (int)q.size(1);
    Tensor gq = at::empty({4, N}, g.options()), gs = at::empty({3, N}, g.options());
    check(lg_create_transform_matrix_backward(fp(g), fp(q), fp(sc), vl(valid_length), N, gq.data_ptr<float>(), gs.data_ptr<float>(), cur_stream()),
          "createTransformMatrix_backward");
    return { gq, gs };
}

Tensor jacobianRayspace(Tensor translate_position, Tensor proj_matrix, int64_t output_h, int64_t output_w, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(translate_position));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor tp = f32(translate_position, "translate_position"), pm = f32(proj_matrix, "proj_matrix");
    const int V = (int)tp.size(0), N = (int)tp.size(2);
    Tensor J = at::empty({V, 3, 3, N}, tp.options());
    check(lg_jacobian_rayspace(fp(tp), fp(pm), vl(valid_length), V, N, (int)output_h, (int)output_w, J.data_ptr<float>(), cur_stream()), "jacobianRayspace");
    return J;
}

Tensor createCov2dDirectly_forward(Tensor J, Tensor view_matrix, Tensor transform_matrix, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(J));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor j = f32(J, "J"), vm = f32(view_matrix, "view_matrix"), T = f32(transform_matrix, "transform_matrix");
    const int V = (int)vm.size(0), N = (int)T.size(2);
    Tensor cov = at::empty({V, 2, 2, N}, T.options());
    check(lg_create_cov2d_forward(fp(j), fp(vm), fp(T), vl(valid_length), V, N, cov.data_ptr<float>(), cur_stream()), "createCov2dDirectly_forward");
    return cov;
}

Tensor createCov2dDirectly_backward(Tensor cov2d_grad, Tensor J, Tensor view_matrix, Tensor transform_matrix, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(cov2d_grad));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor g = f32(cov2d_grad, "cov2d_grad"), j = f32(J, "J"), vm = f32(view_matrix, "view_matrix"), T = f32(transform_matrix, "transform_matrix");
    const int V = (int)vm.size(0), N = (int)T.size(2);
    Tensor gT = at::empty({3, 3, N}, g.options());
    check(lg_create_cov2d_backward(fp(g), fp(j), fp(vm), fp(T), vl(valid_length), V, N, gT.data_ptr<float>(), cur_stream()), "createCov2dDirectly_backward");
    return gT;
}

std::vector<Tensor> eigh_and_inv_2x2matrix_forward(Tensor input, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(input));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor x = f32(input, "input");
    const int V = (int)x.size(0), N = (int)x.size(3);
    Tensor val = at::empty({V, 2, N}, x.options()), vec = at::empty({V, 2, 2, N}, x.options()), inv = at::empty({V, 2, 2, N}, x.options());
    check(lg_eigh_inv_2x2_forward(fp(x), vl(valid_length), V, N, val.data_ptr<float>(), vec.data_ptr<float>(), inv.data_ptr<float>(), cur_stream()),
          "eigh_and_inv_2x2matrix_forward");
    return { val, vec, inv };
}

Tensor inv_2x2matrix_backward(Tensor inv_matrix, Tensor dL_dInvMatrix, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(inv_matrix));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor inv = f32(inv_matrix, "inv_matrix"), g = f32(dL_dInvMatrix, "dL_dInvMatrix");
    const int V = (int)inv.size(0), N = (int)inv.size(3);
    Tensor out = at::empty_like(g);
    check(lg_inv_2x2_backward(fp(inv), fp(g), vl(valid_length), V, N, 0, out.data_ptr<float>(), cur_stream()), "inv_2x2matrix_backward");
    return out;
}

Tensor sh2rgb_forward(int64_t degree, Tensor sh_base, Tensor sh_rest, Tensor dir)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(sh_base));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor s0 = f32(sh_base, "sh_base"), sr = f32(sh_rest, "sh_rest"), d = f32(dir, "dir");
    const int V = (int)d.size(0), N = (int)d.size(2);
    Tensor rgb = at::empty({V, 3, N}, d.options());
    check(lg_sh2rgb_forward((int)degree, fp(s0), fp(sr), fp(d), V, N, rgb.data_ptr<float>(), cur_stream()), "sh2rgb_forward");
    return rgb;
}

std::vector<Tensor> sh2rgb_backward(int64_t degree, Tensor rgb_grad, int64_t sh_rest_dim, Tensor dir, Tensor SH_base, Tensor SH_rest)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(rgb_grad));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor g = f32(rgb_grad, "rgb_grad"), d = f32(dir, "dir");
    const int V = (int)d.size(0), N = (int)d.size(2);
    Tensor d0 = at::empty({1, 3, N}, g.options()), dr = at::empty({sh_rest_dim, 3, N}, g.options()), dd = at::empty({V, 3, N}, g.options());
    check(lg_sh2rgb_backward((int)degree, fp(g), fp(d), V, N, (int)sh_rest_dim, d0.data_ptr<float>(), dr.data_ptr<float>(), dd.data_ptr<float>(), cur_stream()),
          "sh2rgb_backward");
    return { d0, dr, dd };
}

std::vector<Tensor> world2ndc_forward(Tensor world_position, Tensor view_project_matrix)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(world_position));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor w = f32(world_position, "world_position"), m = f32(view_project_matrix, "view_project_matrix");
    const int V = (int)m.size(0), N = (int)w.size(1);
    Tensor ndc = at::empty({V, 4, N}, w.options()), rw = at::empty({V, 1, N}, w.options());
    check(lg_world2ndc_forward(fp(w), fp(m), V, N, ndc.data_ptr<float>(), rw.data_ptr<float>(), cur_stream()), "world2ndc_forward");
    return { ndc, rw };
}

Tensor world2ndc_backword(Tensor view_project_matrix, Tensor position, Tensor repc_hom_w, Tensor grad_ndcpos)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(view_project_matrix));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor m = f32(view_project_matrix, "vp"), ndc = f32(position, "ndc_position"), rw = f32(repc_hom_w, "repc_hom_w"), g = f32(grad_ndcpos, "grad_ndcpos");
    const int V = (int)g.size(0), N = (int)g.size(2);
    Tensor out = at::empty({4, N}, g.options());
    check(lg_world2ndc_backward(fp(m), fp(ndc), fp(rw), fp(g), V, N, out.data_ptr<float>(), cur_stream()), "world2ndc_backword");
    return out;
}

// ------------------------------------------------------------------------------------------------ binning.h
std::vector<Tensor> get_allocate_size(Tensor ndc, Tensor view_space_z, Tensor inv_cov2d, Tensor opacity, int64_t height, int64_t width,
                                      int64_t tilesize_h, int64_t tilesize_w, OptTensor valid_length)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(ndc));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor n = f32(ndc, "ndc"), vz = f32(view_space_z, "view_space_z"), ic = f32(inv_cov2d, "inv_cov2d"), op = f32(opacity, "opacity");
    const int V = (int)n.size(0), N = (int)n.size(2);
    Tensor left_up = at::empty({V, 2, N}, like(n, at::kInt)), right_down = at::empty({V, 2, N}, like(n, at::kInt)), alloc = at::empty({V, N}, like(n, at::kInt));
    check(lg_get_allocate_size(fp(n), fp(vz), fp(ic), fp(op), vl(valid_length), V, N, (int)height, (int)width, (int)tilesize_h, (int)tilesize_w,
                               left_up.data_ptr<int>(), right_down.data_ptr<int>(), alloc.data_ptr<int>(), cur_stream()), "get_allocate_size");
    return { left_up, right_down, alloc };
}

int sort_bits(int64_t height, int64_t width, int64_t th, int64_t tw)       // GR/binning.cu:199-202
{
    int64_t max_tiles = ((height + th - 1) / th) * ((width + tw - 1) / tw);
    int bit = 0;
    while (max_tiles >> 1) { max_tiles >>= 1; bit++; }
    return bit + 1;
}

std::vector<Tensor> create_table(Tensor ndc, Tensor inv_cov2d, Tensor opacity, Tensor offset, Tensor depth_sorted_pointid,
                                 OptTensor feedback_buffer_cpu, OptTensor idx_tensor_cpu, int64_t height, int64_t width, int64_t tile_size_h, int64_t tile_size_w)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(ndc));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor n = f32(ndc, "ndc"), ic = f32(inv_cov2d, "inv_cov2d"), op = f32(opacity, "opacity"), off = dev(offset, "offset");
    TORCH_CHECK(off.scalar_type() == at::kInt, "create_table: offset must be int32 (cumsum dtype=torch.int32)");
    Tensor ids = dev(depth_sorted_pointid, "depth_sorted_pointid");
    TORCH_CHECK(ids.scalar_type() == at::kLong || ids.scalar_type() == at::kInt, "create_table: depth_sorted_pointid must be int64 or int32");
    const int V = (int)n.size(0), N = (int)n.size(2);
    void* s = cur_stream();
    int64_t pred = 0;
    if (feedback_buffer_cpu.has_value() && idx_tensor_cpu.has_value()) {
        TORCH_CHECK(!feedback_buffer_cpu->is_cuda() && feedback_buffer_cpu->scalar_type() == at::kInt && feedback_buffer_cpu->is_contiguous(),
                    "create_table: feedback_buffer must be a contiguous (pinned) CPU int32 tensor");
        const Tensor idx_host = idx_tensor_cpu->to(at::kCPU, at::kLong).contiguous();       // any device / integer dtype, as GR/binning.cu:139-150 reads it
        int* base = feedback_buffer_cpu->data_ptr<int>();
        const int64_t* idx = idx_host.data_ptr<int64_t>();
        for (int i = 0; i < V; i++) {
            pred = std::max<int64_t>(pred, base[idx[i]]);
            check(lg_feedback_d2h(base + idx[i], off.data_ptr<int>() + ((int64_t)i * N + N - 1), s), "feedback copy");
        }
    }
    pred = (int64_t)(1.5 * (double)pred);
    if (pred <= 0 && N > 0) pred = off.select(1, N - 1).max().item<int>();          // blocking path (binning.cu:152-163)
    TORCH_CHECK(pred > 0, "error pred_allocate_size");
    const int bits = sort_bits(height, width, tile_size_h, tile_size_w);
    const int is64 = ids.scalar_type() == at::kLong ? 1 : 0;
    auto oi = like(n, at::kInt);
    if (V == 1) {           // one native call: emission counts the sort's digits, no counting pass, no pre-cleared table
        Tensor ka = at::empty({1, pred}, oi), va = at::empty({1, pred}, oi), kb = at::empty({1, pred}, oi), vb = at::empty({1, pred}, oi);
        const long long tb = lg_create_table_temp_bytes(N, pred, bits);
        Tensor temp = at::empty({tb}, like(n, at::kByte));
        check(lg_create_table(fp(n), fp(ic), fp(op), off.data_ptr<int>(), ids.data_ptr(), is64, N, (int)height, (int)width, (int)tile_size_h, (int)tile_size_w,
                              pred, bits, ka.data_ptr<int>(), va.data_ptr<int>(), kb.data_ptr<int>(), vb.data_ptr<int>(), temp.data_ptr(), tb, s), "create_table");
        if (lg_radix_sort_num_passes(0, bits) % 2 == 1) return { kb, vb };
        return { ka, va };
    }
    Tensor keys = at::zeros({V, pred}, oi), vals = at::empty({V, pred}, oi);
    const long long tb = lg_duplicate_with_keys_temp_bytes(V, N, pred);
    Tensor temp = at::empty({tb}, like(n, at::kByte));
    check(lg_duplicate_with_keys(fp(n), fp(ic), fp(op), off.data_ptr<int>(), ids.data_ptr(), is64, V, N, (int)height, (int)width, (int)tile_size_h,
                                 (int)tile_size_w, pred, keys.data_ptr<int>(), vals.data_ptr<int>(), temp.data_ptr(), tb, s), "duplicate_with_keys");
    Tensor kb = at::empty_like(keys), vb = at::empty_like(vals);
    const long long sb = lg_radix_sort_temp_bytes(pred);
    Tensor stemp = at::empty({sb}, like(n, at::kByte));
    for (int v = 0; v < V; v++)          // per view (the reference sorts view 0 V times: GR/binning.cu:213-221, a bug)
        check(lg_radix_sort_pairs((uint32_t*)keys[v].data_ptr<int>(), (uint32_t*)vals[v].data_ptr<int>(), (uint32_t*)kb[v].data_ptr<int>(),
                                  (uint32_t*)vb[v].data_ptr<int>(), pred, 0, bits, stemp.data_ptr(), sb, s), "radix_sort_pairs");
    if (lg_radix_sort_num_passes(0, bits) % 2 == 1) return { kb, vb };
    return { keys, vals };
}

Tensor tileRange(Tensor table_tileId, int64_t max_tileId)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(table_tileId));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor t = dev(table_tileId, "table_tileId");
    const int V = (int)t.size(0);
    Tensor out = at::empty({V, max_tileId + 2}, like(t, at::kInt));
    check(lg_tile_range(t.data_ptr<int>(), V, t.size(1), (int)max_tileId, out.data_ptr<int>(), cur_stream()), "tileRange");
    return out;
}

// ------------------------------------------------------------------------------------------------- raster.h
struct RasterOut { Tensor img, trans, depth, last, fc, fw; };

RasterOut raster_forward_impl(const Tensor& sorted_points_, const Tensor& start_index_, const Tensor& packed, const OptTensor& specific_tiles,
                              int64_t img_h, int64_t img_w, int64_t tile_h, int64_t tile_w, bool enable_statistic, bool enable_depth)
{
    Tensor sp = dev(sorted_points_, "sorted_points"), si = dev(start_index_, "start_index");
    const int V = (int)sp.size(0), N = (int)packed.size(1);
    const TileShape ts = tiles_shape(img_h, img_w, tile_h, tile_w);
    auto o = packed.options();
    RasterOut r;
    r.img = at::empty({V, 3, ts.Hp, ts.Wp}, o);
    r.trans = at::empty({V, 1, ts.Hp, ts.Wp}, o);
    r.depth = enable_depth ? at::zeros({V, 1, ts.Hp, ts.Wp}, o) : at::empty({0, 0, 0, 0}, o);
    r.last = at::empty({V, 1, ts.Hp, ts.Wp}, like(packed, at::kShort));
    r.fc = at::zeros({V, 1, N}, like(packed, at::kInt));
    r.fw = at::zeros({V, 1, N}, o);
    int K = 0;
    const int* tp = nullptr;
    Tensor tiles;
    if (specific_tiles.has_value()) {
        tiles = dev(*specific_tiles, "specific_tiles");
        K = (int)tiles.size(1); tp = tiles.data_ptr<int>();
        r.img.zero_(); r.trans.fill_(1.0f); r.last.zero_();          // tiles not listed are not rendered; give them a defined value
    }
    check(lg_raster_forward(sp.data_ptr<int>(), si.data_ptr<int>(), fp(packed), tp, K, V, sp.size(1), N, (int)img_h, (int)img_w, (int)tile_h, (int)tile_w,
                            enable_statistic ? 1 : 0, r.img.data_ptr<float>(), r.trans.data_ptr<float>(), r.last.data_ptr<short>(), r.fc.data_ptr<int>(),
                            r.fw.data_ptr<float>(), nullptr, nullptr, cur_stream()), "rasterize_forward");
    return r;
}

std::vector<Tensor> rasterize_forward(Tensor sorted_points, Tensor start_index, Tensor ndc, Tensor cov2d_inv, Tensor color, Tensor opacity,
                                      OptTensor specific_tiles, int64_t img_h, int64_t img_w, int64_t tilesize_h, int64_t tilesize_w,
                                      bool enable_statistic, bool enable_trans, bool enable_depth)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(sorted_points));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor n = f32(ndc, "ndc"), ic = f32(cov2d_inv, "cov2d_inv"), c = f32(color, "color"), op = f32(opacity, "opacity");
    const int V = (int)n.size(0), N = (int)n.size(2);
    Tensor packed = at::empty({V, N, lg_packed_record_floats()}, n.options());
    check(lg_pack_forward_params(fp(n), fp(ic), fp(c), fp(op), nullptr, V, N, (int)img_h, (int)img_w, packed.data_ptr<float>(), cur_stream()),
          "pack_forward_params");
    RasterOut r = raster_forward_impl(sorted_points, start_index, packed, specific_tiles, img_h, img_w, tilesize_h, tilesize_w, enable_statistic, enable_depth);
    return { r.img, r.trans, r.depth, r.last, packed, r.fc, r.fw };
}

std::vector<Tensor> rasterize_forward_packed(Tensor sorted_points, Tensor start_index, Tensor packed_params, OptTensor specific_tiles_arg,
                                             int64_t img_h, int64_t img_w, int64_t tile_h, int64_t tile_w, bool enable_statistic, bool enable_trans,
                                             bool enable_depth)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(sorted_points));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    RasterOut r = raster_forward_impl(sorted_points, start_index, f32(packed_params, "packed_params"), specific_tiles_arg, img_h, img_w, tile_h, tile_w,
                                      enable_statistic, enable_depth);
    return { r.img, r.trans, r.depth, r.last, r.fc, r.fw };
}

std::vector<Tensor> rasterize_backward(Tensor sorted_points, Tensor start_index, Tensor packed_params, OptTensor specific_tiles,
                                       Tensor final_transmitance, Tensor last_contributor, Tensor d_img, OptTensor d_trans_img_arg,
                                       OptTensor d_depth_img_arg, OptTensor grad_inv_sacler_arg, int64_t img_h, int64_t img_w, int64_t tilesize_h,
                                       int64_t tilesize_w, bool enable_statistic)
{
    const c10::OptionalDeviceGuard device_guard(at::device_of(sorted_points));      // launches go to the tensors' device (one process per GPU: LOCAL_RANK != 0)
    Tensor sp = dev(sorted_points, "sorted_points"), si = dev(start_index, "start_index"), packed = f32(packed_params, "packed_params");
    Tensor fT = f32(final_transmitance, "final_transmitance"), dimg = f32(d_img, "d_img"), last = dev(last_contributor, "last_contributor");
    const int V = (int)sp.size(0), N = (int)packed.size(1);
    auto o = packed.options();
    void* s = cur_stream();
    Tensor pg = at::zeros({V, N, lg_packed_grad_floats()}, o), err_sum = at::zeros({V, 1, N}, o), err_sq = at::zeros({V, 1, N}, o);
    int K = 0;
    const int* tp = nullptr;
    Tensor tiles, order;
    const int* order_p = nullptr;
    if (specific_tiles.has_value()) {
        tiles = dev(*specific_tiles, "specific_tiles");
        K = (int)tiles.size(1); tp = tiles.data_ptr<int>();
    } else {
        // heaviest tiles first (raster.hip: tile schedule): work per tile from last_contributor, then a counting sort
        const TileShape ts = tiles_shape(img_h, img_w, tilesize_h, tilesize_w);
        Tensor work = at::empty({V, ts.ntiles + 1}, like(packed, at::kInt));
        order = at::empty({V, ts.ntiles}, like(packed, at::kInt));
        check(lg_tile_work_from_last(last.data_ptr<short>(), V, (int)img_h, (int)img_w, (int)tilesize_h, (int)tilesize_w, work.data_ptr<int>(), s), "tile_work");
        check(lg_tile_order(work.data_ptr<int>(), V, ts.ntiles, order.data_ptr<int>(), s), "tile_order");
        order_p = order.data_ptr<int>();
    }
    Tensor dtr;
    const float* dtp = nullptr;
    if (d_trans_img_arg.has_value()) { dtr = f32(*d_trans_img_arg, "d_trans"); dtp = fp(dtr); }
    check(lg_raster_backward(sp.data_ptr<int>(), si.data_ptr<int>(), fp(packed), tp, K, fp(fT), last.data_ptr<short>(), fp(dimg), dtp, V, sp.size(1), N,
                             (int)img_h, (int)img_w, (int)tilesize_h, (int)tilesize_w, enable_statistic ? 1 : 0, pg.data_ptr<float>(),
                             err_sq.data_ptr<float>(), nullptr, order_p, s), "rasterize_backward");
    Tensor d_ndc = at::empty({V, 4, N}, o), d_ic = at::empty({V, 2, 2, N}, o), d_color = at::empty({V, 3, N}, o), d_opa = at::empty({1, N}, o);
    Tensor sc;
    const float* scp = nullptr;
    if (grad_inv_sacler_arg.has_value()) { sc = f32(grad_inv_sacler_arg->reshape({1}), "grad_inv_scaler"); scp = fp(sc); }
    check(lg_unpack_gradient(fp(pg), fp(packed), scp, nullptr, V, N, (int)img_h, (int)img_w, d_ndc.data_ptr<float>(), d_ic.data_ptr<float>(),
                             d_color.data_ptr<float>(), d_opa.data_ptr<float>(), s), "unpack_gradient");
    return { d_ndc, d_ic, d_color, d_opa, err_sum, err_sq };
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "litegs_fused on MI355X: the reference's 26 operators bound to liblitegs_hip.so (csrc/ext/litegs_fused_ext.cpp)";
    m.def("create_viewproj_forward", &create_viewproj_forward);
    m.def("create_viewproj_backward", &create_viewproj_backward);
    m.def("create_table", &create_table);
    m.def("tileRange", &tileRange);
    m.def("get_allocate_size", &get_allocate_size);
    m.def("rasterize_forward", &rasterize_forward);
    m.def("rasterize_forward_packed", &rasterize_forward_packed);
    m.def("rasterize_backward", &rasterize_backward);
    m.def("jacobianRayspace", &jacobianRayspace);
    m.def("createTransformMatrix_forward", &createTransformMatrix_forward);
    m.def("createTransformMatrix_backward", &createTransformMatrix_backward);
    m.def("world2ndc_forward", &world2ndc_forward);
    m.def("world2ndc_backword", &world2ndc_backword);
    m.def("mvp_transform_forward", &mvp_transform_forward);
    m.def("mvp_transform_backward", &mvp_transform_backward);
    m.def("createCov2dDirectly_forward", &createCov2dDirectly_forward);
    m.def("createCov2dDirectly_backward", &createCov2dDirectly_backward);
    m.def("sh2rgb_forward", &sh2rgb_forward);
    m.def("sh2rgb_backward", &sh2rgb_backward);
    m.def("eigh_and_inv_2x2matrix_forward", &eigh_and_inv_2x2matrix_forward);
    m.def("inv_2x2matrix_backward", &inv_2x2matrix_backward);
    m.def("cull_compact_activate", &cull_compact_activate);
    m.def("activate_backward", &activate_backward);
    m.def("adamUpdate", &adamUpdate);
    m.def("frustum_culling_aabb", &frustum_culling_aabb);
    m.def("gpu_driven_pipeline_sparse_op", &gpu_driven_pipeline_sparse_op);
}
