"""Developer tool: cost of the data-parallel exchange glue on ONE GPU (single-rank RCCL group: the collectives degenerate to copies, what
remains is the compaction / map / accumulation work every rank does per step).  Frozen parameters (lr 0): the workload does not drift.
    python tools/dp_glue_bench.py [--outside] [--trained <steps>] [modes...]
--trained N: first train the cloud for N steps (noise targets, as bench.py's steady_state): most visible Gaussians then carry gradients and
Adam history, and the moment records per rank grow from ~10^4 to ~10^6 -- the state the exchange has to be sized for."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from litegs_amd import dp, synthetic as S
from litegs_amd.trainer import SyntheticTrainer
cfg = "3m_1080p"
n, W, H, f = S.CONFIGS[cfg]
outside = "--outside" in sys.argv                      # camera outside the cloud: a frame touches ~10x more Gaussians
tr = SyntheticTrainer(n, W, H, f, n_frames=4, cam_radius_frac=(1.6 if outside else 0.5))
trained = int(sys.argv[sys.argv.index("--trained") + 1]) if "--trained" in sys.argv else 0
if trained:
    tr.speculative = True
    for i in range(trained):
        tr.step(i % 4)
    tr.flush()
    tr.speculative = False
for g in tr.opt.param_groups:
    g["lr"] = 0.0
tr.sched.step = lambda: None
out = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dp_glue.log"), "a")
modes = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()] or ["none", "moments", "moments_spec", "sparse"]
label = ("outside " if outside else "") + (f"trained {trained} steps " if trained else "")
for mode in modes:
    moments = mode in ("moments", "moments_spec")
    ex = None if mode == "none" else (dp.MomentExchange(tr.params, 1) if moments else dp.GradientExchange(tr.params, 1, mode=mode))
    hook = None if ex is None else (ex if moments else ex.hook)
    # moments_spec: rank-consistent speculative culling (no gated repeat launches; record blocks at 1.125x instead of 1.5x the last count)
    tr.speculative = mode == "moments_spec"
    r0 = tr.spec_replays
    for i in range(8):
        tr.step(i, hook, i % 4, [i % 4])
    tr.flush()
    torch.cuda.synchronize(); t = time.perf_counter()
    K = 24
    for i in range(K):
        tr.step(i, hook, i % 4, [i % 4])
    tr.flush()
    torch.cuda.synchronize()
    extra = ""
    if moments:
        ex.check()
        extra = f", record capacity {ex.last_cap}, block bytes {(1 + ex.last_cap) * 40}"
        if mode == "moments_spec":
            extra += f", replayed steps {tr.spec_replays - r0} (overflows {ex.overflow_replays})"
        tr.speculative = False
        tr.flush()
    elif ex is not None:
        extra = f", last K = {ex.last_k}"
    line = f"{label}{mode}: {(time.perf_counter() - t) / K * 1e3:.3f} ms/step (world 1){extra}"
    print(line, file=out, flush=True)
    print(line, flush=True)
dist.destroy_process_group()
