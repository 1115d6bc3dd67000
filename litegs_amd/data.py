"""Cameras, posed frames and the training set the epoch loop iterates -- host mirror of ``litegs/data.py`` (same class names and
attributes, so ``litegs_amd.training.start`` reads like the reference's ``start``).

Conventions (litegs/data.py:13): row-major matrices applied to row vectors, ``view_pos = [x y z 1] @ V``; COLMAP's world-to-camera
rotation R (from qvec, w first) and translation t become ``V = [[R t],[0 1]]^T``; the projection depends on the focal length
over the half size only, so a down-scaled image keeps its camera's matrix.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import synthetic as S


def qvec2rotmat(q) -> np.ndarray:
    """COLMAP quaternion (w, x, y, z) -> 3x3 rotation (litegs/utils/__init__.py:7-18)"""
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def rotmat2qvec(R: np.ndarray) -> np.ndarray:
    """inverse of qvec2rotmat, w >= 0 (Shepperd's method on the symmetric 4x4 form, litegs/utils/__init__.py:21-33)"""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = np.asarray(R, dtype=np.float64).flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0], [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
                  [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


class CameraInfo:
    def __init__(self, id: int, model_name: str, width: int, height: int):
        self.id, self.model, self.width, self.height = id, model_name, width, height


class PinHoleCameraInfo(CameraInfo):
    """parameters = COLMAP PINHOLE (fx, fy, cx, cy); the principal point is ignored, as in the reference (data.py:33-51)"""

    def __init__(self, id: int, width: int, height: int, parameters, z_near: float = 0.01, z_far: float = 5000.0):
        super().__init__(id, "PINHOLE", width, height)
        fx, fy = float(parameters[0]), float(parameters[1])
        self.focal = (fx, fy)
        self.intr_params = np.float32(fx / (width * 0.5))
        self.proj_matrix = S.pinhole_proj(width, height, fx, fy, z_near, z_far)

    def get_project_matrix(self) -> np.ndarray:
        return self.proj_matrix


class ImageFrame:
    def __init__(self, id: int, qvec, tvec, camera_id: int, name: str, img_source: Optional[str], xys=None):
        self.id, self.camera_id, self.name, self.img_source = id, camera_id, name, img_source
        R, t = qvec2rotmat(qvec), np.asarray(tvec, dtype=np.float64)
        self.extr_params = np.concatenate([np.asarray(qvec, dtype=np.float64), t]).astype(np.float32)
        V = np.zeros((4, 4), dtype=np.float64)
        V[:3, :3], V[:3, 3], V[3, 3] = R, t, 1.0
        self.view_matrix = np.float32(V).T.copy()
        self.camera_center = -R.T @ t
        self.image: Dict[int, np.ndarray] = {}

    def load_image(self, downsample: int = -1) -> np.ndarray:
        """uint8 [3, H, W]; resolution rule of data.py:87-112: 1/2/4/8 divide, -1 caps the width at 1600, any other value is the
        target width"""
        if downsample not in self.image:
            import PIL.Image
            img = PIL.Image.open(self.img_source)
            w, h = img.size
            if downsample in (1, 2, 4, 8):
                res = (round(w / downsample), round(h / downsample))
            else:
                scale = (w / 1600 if w > 1600 else 1.0) if downsample == -1 else w / downsample
                res = (int(w / scale), int(h / scale))
            arr = np.array(img.convert("RGB").resize(res) if res != (w, h) else img.convert("RGB"), dtype=np.uint8)
            self.image[downsample] = np.ascontiguousarray(arr.transpose(2, 0, 1))
        return self.image[downsample]

    def get_viewmatrix(self) -> np.ndarray:
        return self.view_matrix

    def get_camera_center(self) -> np.ndarray:
        return self.camera_center


class CameraFrameDataset:
    """index -> (view[4,4], proj[4,4], frustum planes[6,4], image uint8[3,H,W], index); with ``device`` the matrices and images live
    on that device (the reference's ``device_preload``, data.py:180-190)."""

    def __init__(self, cameras: Dict[int, PinHoleCameraInfo], frames: List[ImageFrame], downsample: int = -1, device=None):
        self.cameras, self.frames, self.downsample, self.device = cameras, frames, downsample, device
        self.items = []
        for fr in frames:
            view, proj = fr.get_viewmatrix(), cameras[fr.camera_id].get_project_matrix()
            planes = S.frustum_planes(view, proj)
            img = fr.load_image(downsample)
            t = [torch.from_numpy(np.ascontiguousarray(a)) for a in (view, proj, planes, img)]
            if device is not None:
                t = [x.to(device) for x in t]
            self.items.append(tuple(t))

    def __len__(self) -> int:
        return len(self.frames)

    def __getitem__(self, idx: int):
        return (*self.items[idx], idx)

    def image_size(self, idx: int = 0) -> Tuple[int, int]:
        return tuple(self.items[idx][3].shape[1:])

    def get_norm(self) -> Tuple[np.ndarray, float]:
        """-> (translate, radius): minus the mean camera centre, 1.1 x the largest distance of a camera from it (data.py:214-232)"""
        centers = np.stack([fr.get_camera_center() for fr in self.frames], axis=1)
        center = centers.mean(axis=1, keepdims=True)
        radius = float(np.linalg.norm(centers - center, axis=0).max()) * 1.1
        return -center.reshape(-1), radius
