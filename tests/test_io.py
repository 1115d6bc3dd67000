"""Data formats either side of the hot path: COLMAP models, image frames, 3DGS .ply, checkpoints, the
command line.  Pinned against the REFERENCE's own loaders run on the committed tiny scene (tests/golden/make_golden_io.py ->
reference_io.npz): same frames in the same order, same matrices, same pixels, same .ply bytes."""
import hashlib
import os
import shutil

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "colmap_scene")
G = np.load(os.path.join(HERE, "golden", "reference_io.npz"))


@pytest.fixture()
def scene(tmp_path):
    dst = tmp_path / "scene"
    shutil.copytree(SCENE, dst)
    return str(dst)


@pytest.mark.parametrize("tag", ["bin", "txt"])
def test_colmap_frames_match_reference_loader(scene, tag):
    from litegs_amd.io import colmap
    sp = os.path.join(scene, "sparse", "0")
    if tag == "txt":
        os.remove(os.path.join(sp, "cameras.bin")); os.remove(os.path.join(sp, "images.bin"))
    cams, frames = colmap.load_frames(scene, "images")
    assert [f.name for f in frames] == list(G[f"{tag}_names"])                     # sorted by name, non-PINHOLE frames dropped
    assert [f.camera_id for f in frames] == list(G[f"{tag}_camera_ids"])
    assert sorted(cams.keys()) == list(G[f"{tag}_cam_ids"])
    np.testing.assert_array_equal(np.stack([f.view_matrix for f in frames]), G[f"{tag}_view"])
    np.testing.assert_allclose(np.stack([f.camera_center for f in frames]), G[f"{tag}_center"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(np.stack([f.extr_params for f in frames]), G[f"{tag}_extr"])
    np.testing.assert_array_equal(np.stack([cams[k].proj_matrix for k in sorted(cams)]), G[f"{tag}_proj"])
    assert frames[0].view_matrix.dtype == np.float32 and cams[1].proj_matrix.dtype == np.float32


def test_text_model_with_a_non_pinhole_camera_is_an_error(scene):
    from litegs_amd.io import colmap
    with open(os.path.join(scene, "sparse", "0", "cameras.txt"), "a") as f:
        f.write("9 SIMPLE_RADIAL 64 48 50.0 32.0 24.0 0.01\n")
    with pytest.raises(ValueError):
        colmap.read_cameras_text(os.path.join(scene, "sparse", "0", "cameras.txt"))


def test_points_and_cached_ply_match_reference(scene):
    from litegs_amd.io import colmap
    sp = os.path.join(scene, "sparse", "0")
    xyz, rgb, _ = colmap.read_points3d_binary(os.path.join(sp, "points3D.bin"))
    np.testing.assert_array_equal(xyz, G["pts_bin_xyz"]); np.testing.assert_array_equal(rgb, G["pts_bin_rgb"])
    xyz, rgb, _ = colmap.read_points3d_text(os.path.join(sp, "points3D.txt"))
    np.testing.assert_array_equal(xyz, G["pts_txt_xyz"]); np.testing.assert_array_equal(rgb, G["pts_txt_rgb"])
    assert not os.path.exists(os.path.join(sp, "points3D.ply"))
    pos, col = colmap.load_pointcloud(scene)
    assert os.path.exists(os.path.join(sp, "points3D.ply"))                          # cached on first load, like the reference
    np.testing.assert_array_equal(pos, G["cloud_xyz"]); np.testing.assert_array_equal(col, G["cloud_rgb"])
    pos2, col2 = colmap.load_pointcloud(scene)                                       # second load reads the cache
    np.testing.assert_array_equal(pos2, pos); np.testing.assert_array_equal(col2, col)


def test_image_loading_and_norm_match_reference(scene):
    from litegs_amd import data as D
    from litegs_amd.io import colmap
    cams, frames = colmap.load_frames(scene, "images")
    for f in frames:
        for ds in (-1, 1, 2, 30):
            got = f.load_image(ds)
            want = G[f"img_{f.name}_{ds}"]
            assert got.dtype == np.uint8 and got.shape == want.shape, (f.name, ds, got.shape, want.shape)
            np.testing.assert_array_equal(got, want)
    wide = [f for f in frames if f.name == "wide.png"][0]
    assert wide.load_image(-1).shape == (3, 112, 1600)                               # the >1600 px rule (data.py:95-103)
    ds = D.CameraFrameDataset(cams, [f for f in frames if f.camera_id == 1], 1, None)
    full = D.CameraFrameDataset(cams, frames, 1, None)
    trans, radius = full.get_norm()
    np.testing.assert_allclose(trans, G["norm_translate"], rtol=0, atol=1e-12)
    assert abs(radius - float(G["norm_radius"])) < 1e-12
    view, proj, planes, img, idx = ds[2]
    assert view.shape == (4, 4) and proj.shape == (4, 4) and planes.shape == (6, 4) and img.dtype == torch.uint8 and idx == 2
    assert ds.image_size() == (23, 37)


def test_colmap_writers_round_trip(tmp_path):
    from litegs_amd.io import colmap as C
    rng = np.random.default_rng(0)
    cams = {3: C.Camera(3, "PINHOLE", 640, 480, [500.0, 501.0, 320.0, 240.0]), 4: C.Camera(4, "OPENCV", 10, 20, np.arange(8.0))}
    imgs = {7: C.Image(7, [1, 0, 0, 0], [0.1, 0.2, 0.3], 3, "ünï.png"), 8: C.Image(8, rng.standard_normal(4), rng.standard_normal(3), 4, "b.png")}
    xyz, rgb = rng.standard_normal((9, 3)), rng.integers(0, 256, (9, 3))
    C.write_cameras_binary(tmp_path / "c.bin", cams); C.write_images_binary(tmp_path / "i.bin", imgs); C.write_points3d_binary(tmp_path / "p.bin", xyz, rgb)
    c2, i2 = C.read_cameras_binary(tmp_path / "c.bin"), C.read_images_binary(tmp_path / "i.bin")
    assert [(c.id, c.model, c.width, c.height) for c in c2.values()] == [(3, "PINHOLE", 640, 480), (4, "OPENCV", 10, 20)]
    np.testing.assert_array_equal(c2[4].params, np.arange(8.0))
    assert i2[7].name == "ünï.png" and i2[8].camera_id == 4
    np.testing.assert_array_equal(i2[8].qvec, imgs[8].qvec); np.testing.assert_array_equal(i2[8].tvec, imgs[8].tvec)
    x2, r2, _ = C.read_points3d_binary(tmp_path / "p.bin")
    np.testing.assert_array_equal(x2, xyz); np.testing.assert_array_equal(r2, rgb)
    C.write_images_text(tmp_path / "i.txt", imgs); C.write_points3d_text(tmp_path / "p.txt", xyz, rgb)
    i3 = C.read_images_text(tmp_path / "i.txt")
    np.testing.assert_array_equal(i3[8].qvec, imgs[8].qvec)                          # repr() round-trips doubles exactly
    x3, r3, _ = C.read_points3d_text(tmp_path / "p.txt")
    np.testing.assert_array_equal(x3, xyz); np.testing.assert_array_equal(r3, rgb)
    with pytest.raises(Exception):
        C.read_cameras_binary(tmp_path / "i.bin")                                    # garbage in: an error, not silence


def test_quaternion_helpers():
    from litegs_amd import data as D
    rng = np.random.default_rng(3)
    for _ in range(20):
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        if q[0] < 0:
            q = -q
        R = D.qvec2rotmat(q)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(D.rotmat2qvec(R), q, atol=1e-9)


@pytest.mark.parametrize("deg", [3, 1])
def test_gaussian_ply_bytes_match_reference_writer(tmp_path, deg):
    from litegs_amd.io import ply
    t = {k: G[f"ply{deg}_in_{k}"] for k in ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity")}
    path = str(tmp_path / "sub" / "dir" / "point_cloud.ply")                          # directories are created, like the reference
    ply.save_ply(path, t["xyz"], t["scale"], t["rot"], t["sh_0"], t["sh_rest"], t["opacity"])
    raw = open(path, "rb").read()
    assert raw[: raw.index(b"end_header\n") + 11].decode("ascii") == str(G[f"ply{deg}_header"])
    assert hashlib.sha256(raw).hexdigest() == str(G[f"ply{deg}_sha256"])             # byte-identical to the reference's save_ply
    back = ply.load_ply(path, deg)
    for k, v in zip(("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"), back):
        want = G[f"ply{deg}_load_{k}"]
        assert v.shape == want.shape, (k, v.shape, want.shape)
        np.testing.assert_array_equal(v, want.astype(np.float32))
        np.testing.assert_array_equal(v, t[k])                                        # and a round trip of the inputs
    # torch tensors take the same path (device tensors on the GPU box: transpose on the device, one copy to the host)
    path2 = str(tmp_path / "torch.ply")
    ply.save_ply(path2, *[torch.from_numpy(t[k]) for k in ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity")])
    assert open(path2, "rb").read() == raw
    with pytest.raises(ValueError):
        ply.load_ply(path, 2 if deg == 3 else 3)                                     # wrong SH degree for the file (reference: assert)


def test_ply_container_ascii_and_big_endian(tmp_path):
    from litegs_amd.io import plyformat as F
    t = np.zeros((5,), dtype=[("x", "f4"), ("n", "i4"), ("c", "u1"), ("d", "f8")])
    t["x"] = np.linspace(-1, 1, 5); t["n"] = np.arange(5) - 2; t["c"] = [0, 1, 128, 254, 255]; t["d"] = np.pi * np.arange(5)
    for kw in (dict(), dict(text=True), dict(big_endian=True)):
        p = str(tmp_path / "t.ply")
        F.write(p, [("vertex", t)], comments=["hello"], **kw)
        back, comments = F.read(p)
        assert comments == ["hello"]
        for n in t.dtype.names:
            np.testing.assert_array_equal(back["vertex"][n], t[n])
    open(tmp_path / "bad.ply", "wb").write(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nend_header\n\x00\x00")
    with pytest.raises(ValueError):
        F.read(str(tmp_path / "bad.ply"))                                            # truncated body
    open(tmp_path / "face.ply", "wb").write(b"ply\nformat ascii 1.0\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n3 0 1 2\n")
    with pytest.raises(ValueError):
        F.read(str(tmp_path / "face.ply"))


def test_compat_plyfile_surface(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "compat"))
    try:
        import plyfile
        t = np.zeros((4,), dtype=[("x", "f4"), ("red", "u1")])
        t["x"] = [1, 2, 3, 4]; t["red"] = [9, 8, 7, 6]
        plyfile.PlyData([plyfile.PlyElement.describe(t, "vertex")]).write(str(tmp_path / "a.ply"))
        d = plyfile.PlyData.read(str(tmp_path / "a.ply"))
        assert [p.name for p in d.elements[0].properties] == ["x", "red"]
        np.testing.assert_array_equal(np.asarray(d.elements[0]["x"]), t["x"]); np.testing.assert_array_equal(d["vertex"]["red"], t["red"])
    finally:
        sys.path.pop(0)


def test_checkpoint_round_trip(tmp_path):
    from litegs_amd.io import checkpoint
    names = ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity")
    params = {n: torch.nn.Parameter(torch.randn(2, 3, 4)) for n in names}
    opt = torch.optim.Adam([{"params": [params[n]], "lr": 0.1, "name": n} for n in ("xyz", "sh_0", "sh_rest", "opacity", "scale", "rot")])
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 1.0)
    path = checkpoint.save_checkpoint(str(tmp_path / "m"), 6, opt, None)
    assert os.path.basename(path) == "chkpnt6.pth"
    *ps, start_epoch, opt2, sched2 = checkpoint.load_checkpoint(path)
    assert start_epoch == 7 and sched2 is None
    for n, p in zip(names, ps):
        assert torch.equal(p, params[n])


def test_command_line_matches_reference_flags():
    from argparse import ArgumentParser
    from litegs_amd import arguments as A
    p = ArgumentParser(); A.add_cmdline_args(p)
    a = p.parse_args(["-s", "/data/garden", "-m", "out", "-i", "images_4", "-r", "2", "--eval", "--iterations", "7000", "--sh_degree", "2",
                      "--target_primitives", "500000", "--prune_mode", "threshold", "--position_lr_init", "0.0002"])
    lp, op, pp, dp = A.extract(a)
    assert (lp.source_path, lp.model_path, lp.images, lp.resolution, lp.eval, lp.sh_degree) == ("/data/garden", "out", "images_4", 2, True, 2)
    assert op.iterations == 7000 and op.position_lr_init == 0.0002 and op.learnable_viewproj is False
    assert pp.cluster_size == 128 and tuple(pp.tile_size) == (8, 16) and pp.sparse_grad
    assert dp.target_primitives == 500000 and dp.prune_mode == "threshold"
    d = A.get_default_arg()
    assert d[0].sh_degree == 3 and d[1].iterations == 30000 and d[3].densification_interval == 5


def test_epoch_schedule_properties():
    from litegs_amd.training import epoch_schedule
    for n, w in ((13, 1), (13, 4), (8, 2), (5, 8)):
        e0, e1 = epoch_schedule(n, w, 0), epoch_schedule(n, w, 1)
        assert len(e0) == (n + w - 1) // w
        assert sorted(s for s, _ in e0) == list(range(len(e0))) and all(len(p) == w for _, p in e0)
        assert dict(e0) == dict(e1)                                                   # the frame sets are fixed, only their order moves
        seen = [f for _, p in e0 for f in p]
        assert set(seen) == set(range(n)) if n >= w else set(seen) <= set(range(n))
    many = [tuple(s for s, _ in epoch_schedule(13, 1, e)) for e in range(6)]
    assert len(set(many)) > 1                                                         # the order is re-drawn per epoch


def test_compat_torchmetrics_stand_ins():
    """PSNR / SSIM stand-ins used when the reference's own scripts run on this image (compat/README.md)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "compat"))
    try:
        for m in [k for k in sys.modules if k == "torchmetrics" or k.startswith("torchmetrics.")]:
            del sys.modules[m]
        from torchmetrics.image import lpip, psnr, ssim
        g = torch.Generator().manual_seed(0)
        a = torch.rand((1, 3, 40, 48), generator=g)
        b = (a + 0.05 * torch.randn((1, 3, 40, 48), generator=g)).clamp(0, 1)
        p = psnr.PeakSignalNoiseRatio(data_range=(0.0, 1.0))(a, b)
        assert abs(float(p) - float(10 * torch.log10(1.0 / (a - b).square().mean()))) < 1e-5
        s_same = ssim.StructuralSimilarityIndexMeasure(data_range=(0.0, 1.0))(a, a)
        s_diff = ssim.StructuralSimilarityIndexMeasure(data_range=(0.0, 1.0))(a, b)
        assert abs(float(s_same) - 1.0) < 1e-6 and 0.0 < float(s_diff) < 1.0
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert torch.isnan(lpip.LearnedPerceptualImagePatchSimilarity(net_type="vgg")(a, b))
    finally:
        sys.path.pop(0)
        for m in [k for k in sys.modules if k == "torchmetrics" or k.startswith("torchmetrics.")]:
            del sys.modules[m]
