"""Developer tool: cost of the data-parallel exchange glue on ONE GPU (single-rank RCCL group: the collectives degenerate to copies, what
remains is the compaction / accumulation work every rank does per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from litegs_amd import dp, synthetic as S
from litegs_amd.trainer import SyntheticTrainer
n, W, H, f = S.CONFIGS["3m_1080p"]
tr = SyntheticTrainer(n, W, H, f, n_frames=4)
out = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dp_glue.log"), "w")
for mode in (sys.argv[1:] or ["sparse"]):
    ex = dp.GradientExchange(tr.params, 1, mode=mode)
    for i in range(8):
        tr.step(i, ex.hook, i % 4)
    torch.cuda.synchronize(); t = time.perf_counter()
    K = 24
    for i in range(K):
        tr.step(i, ex.hook, i % 4)
    torch.cuda.synchronize()
    print(f"{mode}: {(time.perf_counter() - t) / K * 1e3:.3f} ms/step with the exchange hook (world 1), last K = {ex.last_k}", file=out, flush=True)
dist.destroy_process_group()
