#!/usr/bin/env python
"""Writes a synthetic scene in the on-disk layout the reference trainer reads (COLMAP sparse model + images):

    <out>/sparse/0/{cameras,images,points3D}.bin      PINHOLE camera, posed frames, SfM-like points (teacher positions + colours)
    <out>/images/frame_XXXX.png                       the teacher cloud rendered from each pose by this repository's executor (GPU)

The teacher is a seeded Gaussian cloud (litegs_amd.synthetic.make_scene); cameras orbit OUTSIDE the cloud on two elevation rings and
look at its centre.  There are no datasets in this image (no network): this is the stand-in that lets `example_train.py` of the
reference and `litegs_amd.training.start` train on identical files."""
import argparse
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from litegs_amd import data as D                 # noqa: E402
from litegs_amd import fast                      # noqa: E402
from litegs_amd import render as R               # noqa: E402
from litegs_amd import synthetic as S            # noqa: E402
from litegs_amd.io import colmap as C            # noqa: E402


def make(out, points=20000, frames=32, width=480, height=320, focal=420.0, radius=4.0, cam_radius=2.6, sfm_fraction=0.5, seed=7, text=False):
    a = argparse.Namespace(out=out, points=points, frames=frames, width=width, height=height, focal=focal, radius=radius,
                           cam_radius=cam_radius, sfm_fraction=sfm_fraction, seed=seed, text=text)
    return _make(a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--focal", type=float, default=420.0)
    ap.add_argument("--radius", type=float, default=4.0)
    ap.add_argument("--cam_radius", type=float, default=2.6, help="camera distance in units of the scene radius")
    ap.add_argument("--sfm_fraction", type=float, default=0.5, help="fraction of the teacher's Gaussians that appear as SfM points")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--text", action="store_true", help="write the .txt model instead of .bin")
    print(_make(ap.parse_args()))


def _make(a):
    import PIL.Image
    dev = torch.device("cuda", 0)
    scene = S.make_scene(a.points, seed=a.seed, radius=a.radius)
    params = [torch.from_numpy(p).to(dev) for p in scene]
    xyz, scale, rot, sh0, shr, opa = params
    origin, extend = R.get_cluster_AABB(xyz, scale.exp(), torch.nn.functional.normalize(rot, dim=0))
    os.makedirs(os.path.join(a.out, "images"), exist_ok=True)
    os.makedirs(os.path.join(a.out, "sparse", "0"), exist_ok=True)
    cams = []
    half = a.frames // 2
    for k in range(a.frames):
        ring = 0 if k < half else 1
        n_ring = half if ring == 0 else a.frames - half
        az = 2 * math.pi * (k - ring * half) / n_ring + 0.3 * ring
        el = math.radians(12.0 if ring == 0 else 38.0)
        r = a.cam_radius * a.radius
        pos = (r * math.cos(el) * math.cos(az), -r * math.sin(el), r * math.cos(el) * math.sin(az))
        cams.append(S.make_camera(a.width, a.height, a.focal, a.focal, pos))
    renderer = fast.FusedRenderer(a.frames, a.height, a.width)
    images = {}
    with torch.no_grad():
        for k, (view, proj, planes) in enumerate(cams):
            fr = fast.CameraFrame(*[torch.from_numpy(x).to(dev) for x in (view, proj, planes)], k)
            img = renderer.render(fr, origin, extend, xyz, scale, rot, sh0, shr, opa, 3)[0]
            u8 = (img[0].clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
            name = f"frame_{k:04d}.png"
            PIL.Image.fromarray(u8).save(os.path.join(a.out, "images", name))
            V = view[0].T.astype(np.float64)                 # row-vector view matrix -> [R t; 0 1]
            images[k + 1] = C.Image(k + 1, D.rotmat2qvec(V[:3, :3]), V[:3, 3], 1, name)
    cameras = {1: C.Camera(1, "PINHOLE", a.width, a.height, [a.focal, a.focal, a.width / 2, a.height / 2])}
    # SfM-like points: a seeded subset of the teacher's Gaussians with their view-independent colour
    rng = np.random.default_rng(a.seed + 1)
    flat_xyz = scene[0].reshape(3, -1).T
    flat_col = np.clip(0.5 + 0.28209479177387814 * scene[3].reshape(3, -1).T, 0, 1)
    n = flat_xyz.shape[0]
    pick = np.sort(rng.choice(n, size=max(int(a.sfm_fraction * n), 16), replace=False))
    pts, rgb = flat_xyz[pick].astype(np.float64), np.round(flat_col[pick] * 255).astype(np.uint8)
    sp = os.path.join(a.out, "sparse", "0")
    if a.text:
        C.write_cameras_text(os.path.join(sp, "cameras.txt"), cameras)
        C.write_images_text(os.path.join(sp, "images.txt"), images)
        C.write_points3d_text(os.path.join(sp, "points3D.txt"), pts, rgb)
    else:
        C.write_cameras_binary(os.path.join(sp, "cameras.bin"), cameras)
        C.write_images_binary(os.path.join(sp, "images.bin"), images)
        C.write_points3d_binary(os.path.join(sp, "points3D.bin"), pts, rgb)
    return f"wrote {a.out}: {a.frames} frames {a.width}x{a.height}, {len(pick)} SfM points of {n} teacher Gaussians"


if __name__ == "__main__":
    main()
