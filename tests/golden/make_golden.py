#!/usr/bin/env python
"""Generates tests/golden/reference_chain.npz by RUNNING THE REFERENCE's own Python twins on the CPU.

Runs only in the build container (needs /root/reference; never at test time on the GPU box -- the fixture is
committed).  The reference's rasteriser, binning, sort, cull/activate and Adam have no executable twin
(SURVEY.md 4), so the golden vectors pin exactly the pieces the reference can evaluate without CUDA:

  * litegs/utils/spherical_harmonics.py:38-93           sh_to_rgb             (degrees 0..3)
  * litegs/utils/__init__.py:63-136                     viewproj_to_frustumplane, frustum_culling_aabb
  * litegs/utils/wrapper.py:198-220  (_script)          CreateTransformMatrix
  * litegs/utils/wrapper.py:243-255  (_script)          CreateRaySpaceTransformMatrix (no +-1.3 clamp: inputs kept inside it)
  * litegs/utils/wrapper.py:419-442  (call_script)      CreateCov2dDirectly, forward and autograd backward
  * litegs/utils/wrapper.py:569-577  (_script)          EighAndInverse2x2Matrix (torch.linalg eigh / inv), fwd + inverse backward
  * litegs/data.py:35-57, 139-176                       PinHoleCameraInfo projection matrix, frustum planes
  * litegs/scene/cluster.py:7-46                        cluster_points padding rule, get_cluster_AABB (keys cl_*)
  * litegs/training/optimizer.py:46-108                 parameter groups, learning rates, position-lr schedule (keys opt_*)
  * litegs/scene/point.py:29-76, :94                    _gen_morton_code and the stable re-sort order (keys mo_*)
  * litegs/training/densify.py:228-363                  DensityControllerTamingGS.step on the CPU with injected statistics; the two
                                                        random draws are recorded so that the test can replay them (keys dn_*)

The reference package cannot be imported as a whole here (litegs/__init__.py pulls in CUDA-only extensions and
StatisticsHelper allocates on 'cuda' at import), so the needed modules are loaded with stub modules for
`litegs_fused`, `simple_knn`, `cv2` and the statistics singleton.  Nothing of the reference is copied: its
functions are CALLED and only their numeric outputs are stored.
"""
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True          # /root/reference is read-only by contract: importing it must not drop __pycache__ there

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_chain.npz")


def load_reference():
    pkg = types.ModuleType("litegs")
    pkg.__path__ = [os.path.join(REF, "litegs")]
    sys.modules["litegs"] = pkg
    sys.modules["litegs_fused"] = types.ModuleType("litegs_fused")
    stat = types.ModuleType("litegs.utils.statistic_helper")

    class _S:
        bStart = False
    stat.StatisticsHelperInst = _S()
    stat.StatisticsHelper = _S
    sys.modules["litegs.utils.statistic_helper"] = stat
    sys.modules["cv2"] = types.ModuleType("cv2")
    knn, knn_c = types.ModuleType("simple_knn"), types.ModuleType("simple_knn._C")
    knn_c.distCUDA2 = None                                    # litegs/scene/__init__.py imports it; never called here
    knn._C = knn_c
    sys.modules["simple_knn"], sys.modules["simple_knn._C"] = knn, knn_c
    for missing in ("plyfile",):                              # I/O packages absent from this image, imported at package import
        if missing not in sys.modules:
            try:
                importlib.import_module(missing)
            except Exception:
                stub = types.ModuleType(missing)
                stub.PlyData = stub.PlyElement = None
                sys.modules[missing] = stub
    import importlib
    utils = importlib.import_module("litegs.utils")
    wrapper = importlib.import_module("litegs.utils.wrapper")
    sh = importlib.import_module("litegs.utils.spherical_harmonics")
    data = importlib.import_module("litegs.data")
    return utils, wrapper, sh, data


def main():
    utils, wrapper, sh, data = load_reference()
    g = torch.Generator().manual_seed(1234)
    N = 257
    out = {}

    # --- camera conventions (litegs/data.py) -------------------------------------------------
    cam = data.PinHoleCameraInfo(0, 640, 360, np.array([500.0, 480.0, 320.0, 180.0]))
    proj = cam.get_project_matrix()                                             # [4,4] row-vector convention
    qvec = np.array([0.9, 0.1, -0.3, 0.2]); qvec /= np.linalg.norm(qvec)
    tvec = np.array([0.3, -0.2, 2.5])
    frame = data.ImageFrame(0, qvec, tvec, 0, "f", "", np.zeros((0, 2)))
    view = frame.get_viewmatrix()
    vp = torch.tensor(view @ proj)[None]
    planes = utils.viewproj_to_frustumplane(vp)
    out.update(cam_proj=proj.astype(np.float32), cam_view=view.astype(np.float32), cam_qvec=qvec, cam_tvec=tvec, cam_planes=planes.numpy())
    # square-pixel camera (fy == fx): the configuration create_viewproj (GR/compact.cu:54-55) can express from one fov parameter
    cam_sq = data.PinHoleCameraInfo(1, 640, 360, np.array([500.0, 500.0, 320.0, 180.0]))
    proj_sq = cam_sq.get_project_matrix()
    planes_sq = utils.viewproj_to_frustumplane(torch.tensor(view @ proj_sq)[None])
    out.update(camsq_proj=proj_sq.astype(np.float32), camsq_intr=np.asarray(cam_sq.intr_params, np.float32).reshape(1),
               camsq_extr=frame.extr_params.astype(np.float32), camsq_viewproj=(view @ proj_sq).astype(np.float32),
               camsq_planes=planes_sq.numpy())

    # --- frustum culling of AABBs (litegs/utils/__init__.py:109-136) --------------------------
    M = 300
    origin = (torch.rand((3, M), generator=g) * 2 - 1) * 6
    ext = torch.rand((3, M), generator=g) * 0.5
    vis = utils.frustum_culling_aabb(planes, origin, ext)                        # [1, M]
    out.update(cull_origin=origin.numpy(), cull_ext=ext.numpy(), cull_visible=vis.numpy())

    # --- SH (litegs/utils/spherical_harmonics.py) ---------------------------------------------
    sh_all = torch.randn((16, 3, N), generator=g)
    dirs = torch.nn.functional.normalize(torch.randn((2, 3, N), generator=g), dim=1)
    out.update(sh_coeffs=sh_all.numpy(), sh_dirs=dirs.numpy())
    for deg in range(4):
        out[f"sh_rgb_deg{deg}"] = sh.sh_to_rgb(deg, sh_all, dirs).numpy()

    # --- transform matrix (wrapper.py:198-220; the script hard-codes device='cuda' in torch.zeros) ----
    scale = torch.rand((3, N), generator=g) + 0.1
    quat = torch.nn.functional.normalize(torch.randn((4, N), generator=g), dim=0)
    real_zeros = torch.zeros
    torch.zeros = lambda *a, **k: real_zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        T = wrapper.CreateTransformMatrix.call_script(scale, quat)
    finally:
        torch.zeros = real_zeros
    out.update(tm_scale=scale.numpy(), tm_quat=quat.numpy(), tm_T=T.numpy())

    # --- ray-space Jacobian (wrapper.py:243-255) -----------------------------------------------
    view_pos = torch.randn((1, 4, N), generator=g)
    view_pos[:, 2] = view_pos[:, 2].abs() * 3 + 0.5
    view_pos[:, 0] *= 0.15 * view_pos[:, 2]          # inside the +-1.3 z/P clamp so both definitions agree
    view_pos[:, 1] *= 0.1 * view_pos[:, 2]
    view_pos[:, 3] = 1
    projm = torch.tensor(proj)[None]
    J = wrapper.CreateRaySpaceTransformMatrix.call_script(view_pos.clone(), projm, (360, 640))
    out.update(j_view_pos=view_pos.numpy(), j_proj=projm.numpy(), j_J=J.numpy())

    # --- cov2d forward + backward (wrapper.py:419-442 with :312-371 autograd functions) -----------
    Tm = T.clone().requires_grad_(True)
    viewm = torch.tensor(view)[None]
    cov = wrapper.CreateCov2dDirectly.call_script(J, viewm, Tm)
    gcov = torch.randn(cov.shape, generator=g)
    gcov[:, 0, 1] = gcov[:, 1, 0]
    (cov * gcov).sum().backward()
    out.update(cov_view=viewm.numpy(), cov_cov2d=cov.detach().numpy(), cov_gcov=gcov.numpy(), cov_gT=Tm.grad.numpy())

    # --- eigh + inverse (wrapper.py:569-577) --------------------------------------------------------
    c2 = cov.detach().clone().requires_grad_(True)
    val, vec, inv = wrapper.EighAndInverse2x2Matrix.call_script(c2)
    ginv = torch.randn(inv.shape, generator=g)
    ginv[:, 0, 1] = ginv[:, 1, 0]
    (inv * ginv).sum().backward()
    out.update(eig_val=val.numpy(), eig_vec=vec.numpy(), eig_inv=inv.detach().numpy(), eig_ginv=ginv.numpy(), eig_gcov=c2.grad.numpy())

    # --- chunking and chunk AABBs (litegs/scene/cluster.py:7-46; the AABB routine calls CreateTransformMatrix.call == the fused
    #     kernel, so it is pointed at the script twin for this run; its own RNG so the fixtures above keep their values) --------
    cluster = importlib.import_module("litegs.scene.cluster")
    g2 = torch.Generator().manual_seed(4321)
    M = 300                                                   # not a multiple of 128: exercises the padding rule
    cxyz = torch.randn((3, M), generator=g2) * 3
    cscale = torch.rand((3, M), generator=g2) * 0.3 + 0.02     # ACTIVATED scale / rotation, as trainer.py passes them
    crot = torch.nn.functional.normalize(torch.randn((4, M), generator=g2), dim=0)
    kxyz, kscale, krot = cluster.cluster_points(128, cxyz, cscale, crot)
    real_call, real_zeros = wrapper.CreateTransformMatrix.call, torch.zeros
    wrapper.CreateTransformMatrix.call = wrapper.CreateTransformMatrix.call_script
    torch.zeros = lambda *a, **k: real_zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        origin_c, extend_c = cluster.get_cluster_AABB(kxyz, kscale, krot)
    finally:
        wrapper.CreateTransformMatrix.call, torch.zeros = real_call, real_zeros
    # --- optimizer groups and the position learning-rate schedule (litegs/training/optimizer.py:46-108, arguments.py:80-92) ------
    training_pkg = types.ModuleType("litegs.training")          # do not run training/__init__.py (it imports the trainer and its I/O stack)
    training_pkg.__path__ = [os.path.join(REF, "litegs", "training")]
    sys.modules["litegs.training"] = training_pkg
    opt_ref = importlib.import_module("litegs.training.optimizer")
    args_ref = importlib.import_module("litegs.arguments")
    ps = [torch.nn.Parameter(torch.zeros(sh)) for sh in [(3, 2, 128), (3, 2, 128), (4, 2, 128), (1, 3, 2, 128), (15, 3, 2, 128), (1, 2, 128)]]
    pipe = args_ref.PipelineParams
    o_ref, s_ref = opt_ref.get_optimizer(*ps, 2.5, args_ref.OptimizationParams, pipe)
    names = [grp["name"] for grp in o_ref.param_groups]
    traj = []
    for it in range(0, 30001, 1500):
        s_ref.last_epoch = it - 1
        o_ref.step = lambda *a, **k: None                      # the scheduler only needs the counter
        s_ref.step()
        traj.append([grp["lr"] for grp in o_ref.param_groups])
    out.update(opt_group_names=np.array(names), opt_lr_traj=np.array(traj, np.float64), opt_eps=np.float64(o_ref.param_groups[0]["eps"]))

    out.update(cl_xyz=cxyz.numpy(), cl_scale=cscale.numpy(), cl_rot=crot.numpy(), cl_xyz_chunked=kxyz.numpy(),
               cl_origin=origin_c.numpy(), cl_extend=extend_c.numpy())

    # --- Morton codes and the stable re-sort order (litegs/scene/point.py:29-76, :94) ---------------------------------------------
    point = importlib.import_module("litegs.scene.point")
    g3 = torch.Generator().manual_seed(99)
    mxyz = torch.randn((3, 5000), generator=g3) * 2
    mxyz[:, 100:200] = mxyz[:, 300:400]                     # exact duplicates: ties must keep the input order
    mxyz[:, 4000:4100] = mxyz[:, 4000:4001]                 # a run of identical points
    codes = point._gen_morton_code(mxyz)
    out.update(mo_xyz=mxyz.numpy(), mo_codes=codes.numpy(), mo_order=codes.sort(stable=True)[1].numpy().astype(np.int32))
    flat = mxyz[:, :700].clone()
    flat[2] = 0.25                                          # degenerate axis: extent 0 -> the 1e-12 clamp
    codes_flat = point._gen_morton_code(flat)
    out.update(mo_flat_xyz=flat.numpy(), mo_flat_codes=codes_flat.numpy(),
               mo_flat_order=codes_flat.sort(stable=True)[1].numpy().astype(np.int32))

    # --- density control (litegs/training/densify.py:245-363 DensityControllerTamingGS; :228-243 step) on the CPU ------------------
    # statistics are injected (a stand-in for StatisticsHelperInst), the two random draws are recorded so that the test can replay them
    densify_ref = importlib.import_module("litegs.training.densify")
    g4 = torch.Generator().manual_seed(2024)
    S_, C_ = 128, 4
    N_ = S_ * C_
    shapes = dict(xyz=(3, C_, S_), scale=(3, C_, S_), rot=(4, C_, S_), sh_0=(1, 3, C_, S_), sh_rest=(3, 3, C_, S_), opacity=(1, C_, S_))
    init = {k: torch.randn(v, generator=g4) for k, v in shapes.items()}
    init["scale"] = init["scale"] * 0.5 - 3.0                # exp(scale) around 0.05: both sides of percent_dense * extent
    init["opacity"] = init["opacity"] * 2.0
    ps = {k: torch.nn.Parameter(v.clone()) for k, v in init.items()}
    o_ref, _ = opt_ref.get_optimizer(ps["xyz"], ps["scale"], ps["rot"], ps["sh_0"], ps["sh_rest"], ps["opacity"], 2.5,
                                     args_ref.OptimizationParams, pipe)
    mom = {}
    for grp in o_ref.param_groups:
        p0 = grp["params"][0]
        m0, v0 = torch.randn(p0.shape, generator=g4) * 0.01, torch.rand(p0.shape, generator=g4) * 0.001
        o_ref.state[p0] = {"step": torch.tensor(7.0), "exp_avg": m0.clone(), "exp_avg_sq": v0.clone()}
        mom[grp["name"]] = (m0, v0)

    class FakeStats:
        def __init__(self, n, gen):
            self.err_var = torch.rand((1, n), generator=gen)
            self.err_cnt = torch.randint(0, 40, (n,), generator=gen, dtype=torch.int32)
            self.w_mean = torch.rand((1, n), generator=gen)
            self.w_cnt = torch.randint(0, 3, (n,), generator=gen, dtype=torch.int32)      # ~1/3 never contributed -> pruned
            self.culled = torch.rand((n,), generator=gen) < 0.1
        def get_var(self, key): return self.err_var, self.err_cnt
        def get_mean(self, key): return self.w_mean, self.w_cnt
        def get_global_culling(self): return self.culled
        def reset(self, *a, **k): pass

    class NS:
        pass
    dparams = NS()
    for k_, v_ in vars(args_ref.DensifyParams).items():
        if not k_.startswith("_"):
            setattr(dparams, k_, v_)
    dparams.densify_until, dparams.target_primitives = 41, 3000
    rec = {}
    real_multinomial, real_normal, real_zeros2 = torch.multinomial, torch.normal, torch.zeros
    real_fused = wrapper.CreateTransformMatrix.call_fused

    def rec_multinomial(*a, **k):
        r = real_multinomial(*a, **k); rec["picked"] = r.clone(); return r

    def rec_normal(*a, **k):
        r = real_normal(*a, **k); rec["samples"] = r.clone(); return r
    torch.multinomial, torch.normal = rec_multinomial, rec_normal
    torch.zeros = lambda *a, **k: real_zeros2(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    wrapper.CreateTransformMatrix.call_fused = wrapper.CreateTransformMatrix.call_script
    real_empty_cache = torch.cuda.empty_cache
    torch.cuda.empty_cache = lambda: None
    try:
        torch.manual_seed(777)
        for tag, epoch, mode in (("a", 5, "weight"), ("b", 10, "threshold")):
            n_now = o_ref.param_groups[0]["params"][0].shape[-2] * S_
            stats = FakeStats(n_now, g4)
            densify_ref.StatisticsHelperInst = stats
            dparams.prune_mode = mode
            ctrl = densify_ref.DensityControllerTamingGS(2.0, dparams, True, N_)
            out.update({f"dn_{tag}_err_var": stats.err_var.numpy(), f"dn_{tag}_err_cnt": stats.err_cnt.numpy(),
                        f"dn_{tag}_w_mean": stats.w_mean.numpy(), f"dn_{tag}_w_cnt": stats.w_cnt.numpy(),
                        f"dn_{tag}_culled": stats.culled.numpy()})
            ctrl.step(o_ref, epoch)
            out.update({f"dn_{tag}_picked": rec["picked"].numpy(), f"dn_{tag}_samples": rec["samples"].numpy()})
            for grp in o_ref.param_groups:
                p0 = grp["params"][0]
                out[f"dn_{tag}_p_{grp['name']}"] = p0.detach().numpy().copy()
                st = o_ref.state.get(p0)
                out[f"dn_{tag}_has_state_{grp['name']}"] = np.array(bool(st))
                if st:
                    out[f"dn_{tag}_m_{grp['name']}"] = st["exp_avg"].numpy().copy()
                    out[f"dn_{tag}_v_{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
    finally:
        torch.multinomial, torch.normal, torch.zeros = real_multinomial, real_normal, real_zeros2
        wrapper.CreateTransformMatrix.call_fused = real_fused
        torch.cuda.empty_cache = real_empty_cache
    for k_, v_ in init.items():
        out[f"dn_init_{k_}"] = v_.numpy()
        out[f"dn_init_m_{k_}"], out[f"dn_init_v_{k_}"] = mom[k_][0].numpy(), mom[k_][1].numpy()
    out.update(dn_until=np.int64(41), dn_target=np.int64(3000), dn_extent=np.float64(2.0))

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
