"""COLMAP sparse-model I/O: what ``litegs.io_manager.load_colmap_result`` (litegs/io_manager/colmap.py:186-324) hands to the
trainer -- pinhole cameras, posed image frames sorted by name, and the SfM point cloud as (xyz, rgb in [0,1]).

Formats (COLMAP's documented binary and text model files under ``<scene>/sparse/0``):

* ``cameras.bin``: u64 count; per camera  i32 id, i32 model, u64 width, u64 height, f64 params[n(model)].
* ``images.bin``:  u64 count; per image   i32 id, f64 qvec[4] (w x y z), f64 tvec[3], i32 camera id, NUL-terminated name,
  u64 n2d, n2d x (f64 x, f64 y, i64 point3D id).
* ``points3D.bin``: u64 count; per point  u64 id, f64 xyz[3], u8 rgb[3], f64 error, u64 track length, track x (i32, i32).
* the ``.txt`` twins (one record per line; images.txt uses two lines per image).

The reader is written against that layout (memoryview + ``struct.unpack_from``; the 2-D observations and tracks, which the
trainer never looks at, are skipped rather than materialised as Python tuples).  Writers exist for tests and for the synthetic
scenes of ``tools/make_colmap_scene.py``.  Like the reference, the point cloud is cached as ``points3D.ply`` on first load.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np

from . import ply as ply_io

# model id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
MODEL_IDS = {name: mid for mid, (name, _) in CAMERA_MODELS.items()}


class Camera:
    __slots__ = ("id", "model", "width", "height", "params")

    def __init__(self, id, model, width, height, params):
        self.id, self.model, self.width, self.height, self.params = int(id), model, int(width), int(height), np.asarray(params, dtype=np.float64)


class Image:
    __slots__ = ("id", "qvec", "tvec", "camera_id", "name", "n_points2d")

    def __init__(self, id, qvec, tvec, camera_id, name, n_points2d=0):
        self.id, self.camera_id, self.name, self.n_points2d = int(id), int(camera_id), name, int(n_points2d)
        self.qvec, self.tvec = np.asarray(qvec, dtype=np.float64), np.asarray(tvec, dtype=np.float64)


# -- binary -------------------------------------------------------------------------------------------------------------------------
def read_cameras_binary(path: str) -> Dict[int, Camera]:
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    (count,), off = struct.unpack_from("<Q", buf, 0), 8
    cams = {}
    for _ in range(count):
        cid, mid, w, h = struct.unpack_from("<iiQQ", buf, off)
        off += 24
        if mid not in CAMERA_MODELS:
            raise ValueError(f"cameras.bin: unknown camera model id {mid}")
        name, n = CAMERA_MODELS[mid]
        cams[cid] = Camera(cid, name, w, h, struct.unpack_from(f"<{n}d", buf, off))
        off += 8 * n
    return cams


def read_images_binary(path: str) -> Dict[int, Image]:
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    raw = buf.tobytes()
    (count,), off = struct.unpack_from("<Q", buf, 0), 8
    images = {}
    for _ in range(count):
        iid, qw, qx, qy, qz, tx, ty, tz, cid = struct.unpack_from("<idddddddi", buf, off)
        off += 64
        end = raw.index(b"\x00", off)
        name = raw[off:end].decode("utf-8")
        off = end + 1
        (n2d,) = struct.unpack_from("<Q", buf, off)
        off += 8 + 24 * n2d                                  # (x, y, point3D id) observations: unused on this path
        images[iid] = Image(iid, (qw, qx, qy, qz), (tx, ty, tz), cid, name, n2d)
    return images


def read_points3d_binary(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> xyz [P,3] f64, rgb [P,3] (0..255 as f64, like the reference), error [P,1]"""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    (count,), off = struct.unpack_from("<Q", buf, 0), 8
    xyz, rgb, err = np.empty((count, 3)), np.empty((count, 3)), np.empty((count, 1))
    for p in range(count):
        _, x, y, z, r, g, b, e = struct.unpack_from("<QdddBBBd", buf, off)
        off += 43
        (track,) = struct.unpack_from("<Q", buf, off)
        off += 8 + 8 * track
        xyz[p], rgb[p], err[p] = (x, y, z), (r, g, b), e
    return xyz, rgb, err


def write_cameras_binary(path: str, cameras: Dict[int, Camera]) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for c in cameras.values():
            f.write(struct.pack("<iiQQ", c.id, MODEL_IDS[c.model], c.width, c.height))
            f.write(struct.pack(f"<{len(c.params)}d", *c.params))


def write_images_binary(path: str, images: Dict[int, Image]) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images.values():
            f.write(struct.pack("<idddddddi", im.id, *im.qvec, *im.tvec, im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", 0))


def write_points3d_binary(path: str, xyz: np.ndarray, rgb: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", xyz.shape[0]))
        for p in range(xyz.shape[0]):
            f.write(struct.pack("<QdddBBBd", p + 1, *[float(v) for v in xyz[p]], *[int(v) for v in rgb[p]], 0.0))
            f.write(struct.pack("<Q", 0))


# -- text ---------------------------------------------------------------------------------------------------------------------------
def _records(path: str):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line.split()


def read_cameras_text(path: str) -> Dict[int, Camera]:
    """like the reference's text reader (colmap.py:67), anything but PINHOLE is an error here (its binary reader drops them instead)"""
    cams = {int(t[0]): Camera(int(t[0]), t[1], int(t[2]), int(t[3]), [float(v) for v in t[4:]]) for t in _records(path)}
    for c in cams.values():
        if c.model != "PINHOLE":
            raise ValueError(f"cameras.txt: camera {c.id} is {c.model}; only PINHOLE is supported")
    return cams


def read_images_text(path: str) -> Dict[int, Image]:
    images = {}
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f]
    k = 0
    while k < len(lines):
        ln = lines[k]
        k += 1
        if not ln or ln.startswith("#"):
            continue
        t = ln.split()
        obs = lines[k].split() if k < len(lines) else []    # the second line of the record: observations (may be empty)
        k += 1
        images[int(t[0])] = Image(int(t[0]), [float(v) for v in t[1:5]], [float(v) for v in t[5:8]], int(t[8]), t[9], len(obs) // 3)
    return images


def read_points3d_text(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    rows = [(t[1:4], t[4:7], t[7]) for t in _records(path)]
    xyz = np.array([[float(v) for v in r[0]] for r in rows], dtype=np.float64).reshape(-1, 3)
    rgb = np.array([[int(v) for v in r[1]] for r in rows], dtype=np.float64).reshape(-1, 3)
    err = np.array([[float(r[2])] for r in rows], dtype=np.float64).reshape(-1, 1)
    return xyz, rgb, err


def write_cameras_text(path: str, cameras: Dict[int, Camera]) -> None:
    with open(path, "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        for c in cameras.values():
            f.write(f"{c.id} {c.model} {c.width} {c.height} " + " ".join(repr(float(v)) for v in c.params) + "\n")


def write_images_text(path: str, images: Dict[int, Image]) -> None:
    with open(path, "w") as f:
        f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
        for im in images.values():
            f.write(f"{im.id} " + " ".join(repr(float(v)) for v in (*im.qvec, *im.tvec)) + f" {im.camera_id} {im.name}\n\n")


def write_points3d_text(path: str, xyz: np.ndarray, rgb: np.ndarray) -> None:
    with open(path, "w") as f:
        f.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n")
        for p in range(xyz.shape[0]):
            f.write(f"{p + 1} " + " ".join(repr(float(v)) for v in xyz[p]) + " " + " ".join(str(int(v)) for v in rgb[p]) + " 0.0\n")


# -- the trainer's entry points -------------------------------------------------------------------------------------------------------
def load_frames(path: str, image_dir: str):
    """-> ({camera id: PinHoleCameraInfo}, [ImageFrame sorted by name]) (colmap.py:186-211): binary model preferred, text as the
    fallback; non-PINHOLE cameras and the frames that use them are dropped, as in the reference."""
    from ..data import ImageFrame, PinHoleCameraInfo
    sparse = os.path.join(path, "sparse", "0")
    if os.path.exists(os.path.join(sparse, "images.bin")) and os.path.exists(os.path.join(sparse, "cameras.bin")):
        images, cams = read_images_binary(os.path.join(sparse, "images.bin")), read_cameras_binary(os.path.join(sparse, "cameras.bin"))
    else:
        images, cams = read_images_text(os.path.join(sparse, "images.txt")), read_cameras_text(os.path.join(sparse, "cameras.txt"))
    infos = {c.id: PinHoleCameraInfo(c.id, c.width, c.height, c.params) for c in cams.values() if c.model == "PINHOLE"}
    frames = [ImageFrame(im.id, im.qvec, im.tvec, im.camera_id, im.name, os.path.join(path, image_dir, im.name))
              for im in images.values() if im.camera_id in infos]
    return infos, sorted(frames, key=lambda fr: fr.name)


def load_pointcloud(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> xyz [P,3], rgb [P,3] in [0,1]; converts points3D.bin/.txt to points3D.ply on first use (colmap.py:311-322)"""
    sparse = os.path.join(path, "sparse", "0")
    ply_path = os.path.join(sparse, "points3D.ply")
    if not os.path.exists(ply_path):
        if os.path.exists(os.path.join(sparse, "points3D.bin")):
            xyz, rgb, _ = read_points3d_binary(os.path.join(sparse, "points3D.bin"))
        else:
            xyz, rgb, _ = read_points3d_text(os.path.join(sparse, "points3D.txt"))
        # Several ranks may get here at once on a fresh scene: each writes its own temporary file and publishes it with an atomic
        # rename (identical bytes whichever rank wins), so no reader ever sees a truncated cache; and this call returns what the
        # cache holds without re-reading a file another rank may be replacing.
        tmp = f"{ply_path}.tmp.{os.getpid()}"
        try:
            ply_io.store_points_ply(tmp, xyz, rgb)
            out = ply_io.fetch_points_ply(tmp)
            os.replace(tmp, ply_path)
        except OSError:                                  # read-only dataset directory: serve the arrays without the cache
            out = None
            try:
                os.remove(tmp)
            except OSError:
                pass
        if out is not None:
            return out
        return np.asarray(xyz, dtype=np.float32), np.asarray(rgb, dtype=np.float32) / 255.0
    return ply_io.fetch_points_ply(ply_path)


def load_colmap_result(path: str, image_dir: str):
    """-> cameras, frames, xyz [P,3], rgb [P,3] in [0,1] -- the reference's tuple (colmap.py:324 ff.)"""
    cameras, frames = load_frames(path, image_dir)
    xyz, rgb = load_pointcloud(path)
    return cameras, frames, xyz, rgb
