"""Top-level shim so that ``import litegs_fused`` (litegs/utils/wrapper.py:8-12 of the reference) resolves to the MI355X
implementation: the compiled torch extension (litegs_amd/csrc/ext/litegs_fused_ext.cpp, built by ``litegs_amd.build``) when present,
else the ctypes binding of the same C ABI (litegs_amd/fused.py).  Put the repository root on PYTHONPATH; see INTEGRATION.md."""
from litegs_amd.binding import compiled as _compiled

if _compiled is not None:
    from litegs_amd._litegs_fused_C import (  # noqa: F401
        create_viewproj_forward, create_viewproj_backward, create_table, tileRange, get_allocate_size, rasterize_forward,
        rasterize_forward_packed, rasterize_backward, jacobianRayspace, createTransformMatrix_forward, createTransformMatrix_backward,
        world2ndc_forward, world2ndc_backword, mvp_transform_forward, mvp_transform_backward, createCov2dDirectly_forward,
        createCov2dDirectly_backward, sh2rgb_forward, sh2rgb_backward, eigh_and_inv_2x2matrix_forward, inv_2x2matrix_backward,
        cull_compact_activate, activate_backward, adamUpdate, frustum_culling_aabb, gpu_driven_pipeline_sparse_op,
    )
    BINDING = "ext"
else:
    from litegs_amd.fused import (  # noqa: F401
        create_viewproj_forward, create_viewproj_backward, create_table, tileRange, get_allocate_size, rasterize_forward,
        rasterize_forward_packed, rasterize_backward, jacobianRayspace, createTransformMatrix_forward, createTransformMatrix_backward,
        world2ndc_forward, world2ndc_backword, mvp_transform_forward, mvp_transform_backward, createCov2dDirectly_forward,
        createCov2dDirectly_backward, sh2rgb_forward, sh2rgb_backward, eigh_and_inv_2x2matrix_forward, inv_2x2matrix_backward,
        cull_compact_activate, activate_backward, adamUpdate, frustum_culling_aabb, gpu_driven_pipeline_sparse_op,
    )
    BINDING = "ctypes"
