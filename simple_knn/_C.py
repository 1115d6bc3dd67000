"""``distCUDA2(points[P,3]) -> float[P]``: mean squared distance to the 3 nearest neighbours (reference:
litegs/submodules/simple-knn/simple_knn.cu:186-222, spatial.cu:14-25).  Used once at scene initialisation
(litegs/scene/point.py:8), so this shim is plumbing, not hot path: exact brute force in chunks on the GPU with
torch.cdist/topk (the reference uses a Morton-ordered box search; the result is the same quantity)."""
import torch


@torch.no_grad()
def distCUDA2(points: torch.Tensor, chunk: int = 4096) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: GPU tensor required")
    P = points.shape[0]
    out = torch.empty((P,), dtype=torch.float32, device=points.device)
    pts = points.float().contiguous()
    k = min(4, P)
    for s in range(0, P, chunk):
        d = torch.cdist(pts[s:s + chunk], pts)            # [c, P]
        nn = d.topk(k, dim=1, largest=False).values[:, 1:]  # drop self (distance 0)
        out[s:s + chunk] = (nn * nn).mean(dim=1) if k > 1 else 0.0
    return out
