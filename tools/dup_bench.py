"""Developer tool: time lg_duplicate_with_keys alone on the 3M @1080p case (inputs from the CPU oracle's pipeline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import case, oracle_forward
from litegs_amd._lib import lib, check

name = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
c = case(name); res = oracle_forward(name)
H, W = c["H"], c["W"]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ndc, ic, op, prefix = dev(res.ndc), dev(res.inv_cov), dev(res.act[4]), dev(res.prefix)
ids = dev(res.depth_sorted_index.astype(np.int32))
N = ndc.shape[-1]; Ltab = int(res.n_instances)
keys = torch.zeros((1, Ltab), dtype=torch.int32, device="cuda"); vals = torch.empty_like(keys)
L = lib(); tb = L.lg_duplicate_with_keys_temp_bytes(1, N, Ltab); temp = torch.empty((tb,), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run():
    check(L.lg_duplicate_with_keys(ndc.data_ptr(), ic.data_ptr(), op.data_ptr(), prefix.data_ptr(), ids.data_ptr(), 0, 1, N, H, W, 8, 16, Ltab,
                                   keys.data_ptr(), vals.data_ptr(), temp.data_ptr(), tb, s), "dup")
run(); torch.cuda.synchronize()
ok = np.array_equal(np.sort(keys.cpu().numpy()[0]), res.sorted_tile[0])
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): run()
b.record(); torch.cuda.synchronize()
print(f"dup {name}: N={N} instances={Ltab} keys_match={ok} {a.elapsed_time(b)/30*1e3:.1f} us per call (incl. 1 memset)", flush=True)
