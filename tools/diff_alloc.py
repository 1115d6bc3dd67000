"""Developer probe: find splats whose tile count differs between the HIP get_allocate_size and the oracle at full size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import case, oracle_forward
from oracle import oracle as O
from litegs_amd import fused as F
name = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
c = case(name); res = oracle_forward(name)
H, W = c["H"], c["W"]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
op = res.act[4]
vd = np.ascontiguousarray(res.view_pos[:, 2, :])
lu, rd, al = F.get_allocate_size(dev(res.ndc), dev(vd), dev(res.inv_cov), dev(op), H, W, 8, 16, None)
al = al.cpu().numpy()
bad = np.nonzero(al[0] != res.alloc[0])[0]
print("differing splats:", len(bad), "of", al.shape[1])
lu_r, rd_r, al_r = O.get_allocate_size(res.ndc, vd, res.inv_cov, op, H, W, 8, 16)
print("oracle self-consistent:", np.array_equal(al_r, res.alloc))
np.savez("gpurun_out/diff_alloc.npz", idx=bad, ndc=res.ndc[:, :, bad], vd=vd[:, bad], inv=res.inv_cov[:, :, :, bad], op=op[..., bad],
         gpu=al[0][bad], ref=res.alloc[0][bad], lu=lu.cpu().numpy()[:, :, bad], rd=rd.cpu().numpy()[:, :, bad], lu_r=lu_r[:, :, bad], rd_r=rd_r[:, :, bad])
for i in bad[:10]:
    print(i, al[0][i], res.alloc[0][i])
