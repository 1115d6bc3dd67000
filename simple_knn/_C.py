"""``distCUDA2(points[P,3]) -> float[P]``: mean squared distance to the 3 nearest neighbours (reference:
litegs/submodules/simple-knn/simple_knn.cu:186-222, spatial.cu:14-25).  Used once at scene initialisation
(litegs/scene/point.py:8).  HIP implementation in litegs_amd/csrc/knn.hip (Morton order from the library's radix sort, boxes of
256 consecutive points, exact box-pruned search): 3 M points in well under a second, where an O(P^2) torch.cdist cannot run."""
import torch


@torch.no_grad()
def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    from litegs_amd._lib import check, lib
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: GPU tensor required (no CPU path)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("distCUDA2: points must be [P,3]")
    pts = points.float().contiguous()
    P = pts.shape[0]
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    L = lib()
    tb = L.lg_knn3_temp_bytes(P)
    temp = torch.empty((tb,), dtype=torch.uint8, device=pts.device)
    check(L.lg_knn3_mean_dist2(pts.data_ptr(), P, out.data_ptr(), temp.data_ptr(), tb, torch.cuda.current_stream().cuda_stream), "distCUDA2")
    return out
