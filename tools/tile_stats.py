"""Developer probe: per-tile blend depth statistics + isolated blend kernel timings for a config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
n, W, H, f = S.CONFIGS[cfg]
tr = SyntheticTrainer(n, W, H, f, n_frames=8)
for fi in range(3):
    img, vis_id, vis_num, _ = tr.forward(tr.frames[fi])
    torch.cuda.synchronize()
import litegs_amd.fast as fast
# re-render frame 0 through the operator path to get `last`
from litegs_amd import fused, wrapper, render as R
fr = tr.frames[0]
with torch.no_grad():
    xyz, scale, rot, sh_0, sh_rest, opacity = tr.params
    vis_id, vis_num, cx, cs, cr, cc, co = R.render_preprocess(tr.cluster_origin, tr.cluster_extend, fr.planes, fr.view, xyz, scale, rot, sh_0, sh_rest, opacity, None, None, tr.pp, 3)
    vl = vis_num * 128
    view_pos, ndc = fused.mvp_transform_forward(cx, fr.view, fr.proj, vl)
    T = fused.createTransformMatrix_forward(cr, cs, vl)
    J = fused.jacobianRayspace(view_pos, fr.proj, H, W, vl)
    cov = fused.createCov2dDirectly_forward(J, fr.view, T, vl)
    _, _, inv = fused.eigh_and_inv_2x2matrix_forward(cov, vl)
    ts, sp, _ = wrapper.Binning.call_fused(ndc, view_pos[:, 2, :], inv, co, vl, None, None, (H, W), (8, 16))
    out = fused.rasterize_forward(sp, ts, ndc, inv, cc, co, None, H, W, 8, 16, False, False, False)
    last = out[3][0, 0].to(torch.int32)
    per_tile = last.reshape(H // 8, 8, W // 16, 16).amax(dim=(1, 3)).flatten().float()
    lens = (ts[0, 2:] - ts[0, 1:-1]).float()
    q = torch.tensor([0.5, 0.9, 0.99, 0.999])
    print("tiles", per_tile.numel(), "visited-per-tile mean %.1f max %d quantiles" % (per_tile.mean().item(), int(per_tile.max().item())), torch.quantile(per_tile, q.cuda()).tolist())
    import numpy as np
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed(f"gpurun_out/tile_stats_{cfg}.npz", visited=per_tile.cpu().numpy().astype(np.int32), length=lens.cpu().numpy().astype(np.int32),
                        last=last.cpu().numpy().astype(np.int16))
    print("list-length-per-tile mean %.1f max %d" % (lens[lens >= 0].mean().item(), int(lens.max().item())), "sum visited", int(per_tile.sum().item()))
print(bench.frame_units(tr, 0))
