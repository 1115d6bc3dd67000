"""``start(lp, op, pp, dp, ...)`` -- the reference's training entry point (litegs/training/trainer.py:26-226) on the native
executor, data-parallel when launched with one process per GPU.

Same signature, same phases, same outputs (``<model_path>/point_cloud/finish/point_cloud.ply``, ``chkpnt<epoch>.pth``):

    load the COLMAP scene -> train/test split (``--eval``: ``train_test_split.json`` or every 8th frame) -> Gaussians from the SfM
    points (3-NN scale, csrc/knn.hip) in 128-point chunks -> per epoch: Morton re-sort one epoch after a densification, active SH
    degree = epoch // 5, one iteration per frame (render_preprocess + render + L1/SSIM + backward + sparse Adam + position-lr decay,
    all inside ``FrameTrainer.step``), PSNR evaluation on request, density control, .ply / checkpoint on request.

Differences from the reference, all deliberate:

* the iteration runs on the native executor (``litegs_amd/fast.py``) instead of the operator-by-operator wrappers -- pass
  ``fused=False`` to drive the same loop through the drop-in ``litegs_fused`` surface;
* world size > 1 (``torch.distributed`` initialised, or RANK/WORLD_SIZE in the environment): every step trains ``world`` frames,
  one per rank, the blend backward's moment records are exchanged (``dp.MomentExchange``), statistics are summed before the density
  controller reads them and the controller's random draws are seeded by (seed, epoch) -- replicas stay bit-identical, nothing is
  ever broadcast.  Frames are grouped into fixed sets of ``world`` (seeded once); the ORDER of the sets is re-drawn every epoch,
  so the exchange's per-set size predictions stay valid.  Only rank 0 writes files and prints;
* frames are drawn by a seeded permutation per epoch instead of a ``DataLoader(shuffle=True)`` (same distribution, reproducible);
* ``op.learnable_viewproj`` (trainer.py:84-91, 117-123, 158-162, 183-191, 221-222): the cameras are built from their parameters -- the
  frames' COLMAP poses as a sparse ``nn.Embedding`` [frames, 7] (quaternion + translation) and the first camera's 1 / tan(fov_x / 2) --
  by the ``create_viewproj`` operator, ``SparseAdam(lr 1e-4)`` steps the embedding after every iteration, evaluation frames take the
  correction of their nearest training pose, and ``viewproj.pth`` is written next to every ``.ply``.  As in the reference's cluster path,
  NO gradient reaches the camera parameters (``MVPTransform.backward`` / the cov2d and activation backwards return None for the matrices,
  litegs/utils/wrapper.py:285): the embedding never moves, the optimizer step is a no-op, and the matrices are therefore built once per
  frame instead of once per iteration.  What the flag changes is where the matrices come from (the operator's float32 quaternion
  arithmetic instead of the loader's float64) and the extra file.
"""
from __future__ import annotations

import json
import os
import time
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import data as data_mod
from . import io as io_manager
from . import optimizer as opt_mod
from . import scene
from .trainer import Frame, FrameTrainer


def _dist_state():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        rank = int(os.environ.get("RANK", "0"))
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
        return dist, rank, world
    return None, 0, 1


def split_frames(lp, camera_frames):
    """--eval: ``train_test_split.json`` if the scene has one, else every 8th frame is a test frame (trainer.py:39-51)"""
    if not lp.eval:
        return list(camera_frames), []
    split = os.path.join(lp.source_path, "train_test_split.json")
    if os.path.exists(split):
        with open(split, "r") as f:
            names = json.load(f)
        return [c for c in camera_frames if c.name in names["train"]], [c for c in camera_frames if c.name in names["test"]]
    return [c for i, c in enumerate(camera_frames) if i % 8 != 0], [c for i, c in enumerate(camera_frames) if i % 8 == 0]


def frames_of(dataset: data_mod.CameraFrameDataset, device) -> List[Frame]:
    out = []
    for k in range(len(dataset)):
        view, proj, planes, img, _ = dataset[k]
        gt = (img.to(device).to(torch.float32) / 255.0).unsqueeze(0).contiguous()
        out.append(Frame(view.to(device)[None].contiguous(), proj.to(device)[None].contiguous(), planes.to(device)[None].contiguous(), gt, k))
    return out


def _learnable_cameras(cameras_info, training_frames, test_frames, frames: List[Frame], test_frames_dev: List[Frame], H: int, W: int, device):
    """op.learnable_viewproj (trainer.py:84-91): camera parameters as learnable tensors, matrices from the ``create_viewproj`` operator
    (litegs/utils/wrapper.py:772-791 CreateViewProj).  Rebuilds the frames' view / projection matrices and frustum planes in place."""
    from . import fast
    from .wrapper import CreateViewProj
    noise_extr = torch.stack([torch.from_numpy(np.asarray(f.extr_params, dtype=np.float32)) for f in training_frames]).to(device)
    extr = torch.nn.Embedding(noise_extr.shape[0], noise_extr.shape[1], _weight=noise_extr.clone(), sparse=True)
    intr0 = float(list(cameras_info.values())[0].intr_params)            # (the reference's "todo fix multi cameras": first camera only)
    # shape [1], as the reference builds it (torch.tensor(scalar).unsqueeze(0), trainer.py:87-88) and as its consumers of viewproj.pth
    # expect (CreateViewProj takes a 1-D accessor: GR/compact.cu:19, example_metrics.py:90)
    intr = torch.nn.Parameter(torch.tensor([intr0], dtype=torch.float32, device=device))
    view_opt = torch.optim.SparseAdam(extr.parameters(), lr=1e-4)

    def rebuild(frs, params7):
        with torch.no_grad():
            view, proj, _vp, planes = CreateViewProj.apply(params7, intr.detach(), H, W, 0.01, 5000.0)
        for k, fr in enumerate(frs):
            fr.view, fr.proj, fr.planes = view[k:k + 1].contiguous(), proj[k:k + 1].contiguous(), planes[k:k + 1].contiguous()
            fr.cam = fast.CameraFrame(fr.view, fr.proj, fr.planes, fr.cam.index)
    rebuild(frames, extr.weight.detach())
    if test_frames_dev:
        # evaluation frames (trainer.py:183-190): own pose + the learnt correction of the nearest training pose (zero: the embedding never moves)
        t_extr = torch.stack([torch.from_numpy(np.asarray(f.extr_params, dtype=np.float32)) for f in test_frames]).to(device)
        nearest = (t_extr[:, None, :] - extr.weight.detach()[None]).abs().sum(dim=2).argmin(dim=1)
        rebuild(test_frames_dev, t_extr + (extr.weight.detach()[nearest] - noise_extr[nearest]))
    return dict(extr=extr, intr=intr, view_opt=view_opt, noise_extr=noise_extr)


def psnr(img: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """peak signal-to-noise ratio for data range [0,1] (what torchmetrics' PeakSignalNoiseRatio(data_range=(0,1)) returns for one image)"""
    return 10.0 * torch.log10(1.0 / (img - gt).square().mean().clamp_min(1e-20))


@torch.no_grad()
def evaluate(trainer: FrameTrainer, frames: Sequence[Frame]) -> float:
    """mean PSNR over ``frames``, forward only (trainer.py:165-193)"""
    trainer.flush()                          # speculative mode: the parameters must reflect every step enqueued so far
    vals = []
    for fr in frames:
        img = trainer.forward(fr)[0]
        vals.append(psnr(img[..., : trainer.H, : trainer.W], fr.gt))
    return float(torch.stack(vals).mean().item()) if vals else float("nan")


def epoch_schedule(n_frames: int, world: int, epoch: int, seed: int = 0):
    """-> list of steps, each a list of ``world`` frame indices (rank order).  world == 1: a fresh permutation per epoch.  world > 1:
    the frames are dealt into fixed sets once (seed only) and the order of the sets is drawn per epoch; a ragged last set wraps
    around to the first frames of the permutation."""
    base = np.random.default_rng(seed).permutation(n_frames)
    n_steps = (n_frames + world - 1) // world
    sets = [[int(base[(s * world + r) % n_frames]) for r in range(world)] for s in range(n_steps)]
    order = np.random.default_rng((seed + 1) * 1_000_003 + epoch).permutation(n_steps)
    return [(int(s), sets[int(s)]) for s in order]


def start(lp, op, pp, dp, test_epochs: Sequence[int] = (), save_ply: Sequence[int] = (), save_checkpoint: Sequence[int] = (),
          start_checkpoint: Optional[str] = None, *, fused: bool = True, seed: int = 0, log=print, on_epoch=None):
    dist, rank, world = _dist_state()
    device = torch.device("cuda", torch.cuda.current_device())
    say = log if rank == 0 else (lambda *a, **k: None)
    # loss terms of the reference's trainer (litegs/training/trainer.py:141-150) that this loop does not compute: refuse, never ignore
    if float(getattr(op, "reg_weight", 0.0) or 0.0) > 0.0:
        raise ValueError("op.reg_weight > 0 (scale regularisation) is not implemented in litegs_amd.training.start")
    if getattr(pp, "enable_transmitance", False):
        raise ValueError("pp.enable_transmitance (transmittance loss term) is not implemented in litegs_amd.training.start")

    cameras_info, camera_frames, init_xyz, init_color = io_manager.load_colmap_result(lp.source_path, lp.images)
    training_frames, test_frames = split_frames(lp, camera_frames)
    trainingset = data_mod.CameraFrameDataset(cameras_info, training_frames, lp.resolution, device if pp.device_preload else None)
    testset = data_mod.CameraFrameDataset(cameras_info, test_frames, lp.resolution, device if pp.device_preload else None) if test_frames else None
    norm_trans, norm_radius = trainingset.get_norm()
    H, W = trainingset.image_size()
    for k in range(len(trainingset)):
        if trainingset.image_size(k) != (H, W):
            raise ValueError("litegs_amd.training.start: all training images must have one size (use --resolution to rescale)")

    init_points_num = init_xyz.shape[0]
    if start_checkpoint is None:
        xyz0 = torch.tensor(init_xyz, dtype=torch.float32, device=device)
        col0 = torch.tensor(init_color, dtype=torch.float32, device=device)
        tensors = scene.create_gaussians(xyz0, col0, lp.sh_degree)
        if not pp.cluster_size:
            raise ValueError("litegs_amd.training.start needs pp.cluster_size > 0 (the executor works on chunks)")
        tensors = scene.cluster_points(pp.cluster_size, *tensors)
        params = [torch.nn.Parameter(t.contiguous()) for t in tensors]
        opt, sched = opt_mod.get_optimizer(*params, norm_radius, op, cluster=True)
        start_epoch = 0
    else:
        *params, start_epoch, opt, sched = io_manager.load_checkpoint(start_checkpoint)
        params = list(params)

    frames = frames_of(trainingset, device)
    test_frames_dev = frames_of(testset, device) if testset is not None else []
    for k, fr in enumerate(test_frames_dev):                         # evaluation frames: feedback slots behind the training set's
        fr.cam.index = len(frames) + k
        fr.idx_tensor = torch.tensor([len(frames) + k], dtype=torch.int64)
    cam_state = None
    if getattr(op, "learnable_viewproj", False):
        cam_state = _learnable_cameras(cameras_info, training_frames, test_frames, frames, test_frames_dev, H, W, device)
    trainer = FrameTrainer(params, frames, H, W, opt, sched, pp, sh_degree=0, device=device, fused=fused, extra_slots=len(test_frames_dev))

    total_epoch = int(op.iterations / len(trainingset))
    dp.resolve_until(total_epoch)
    group = None
    trainer.enable_densify(dp, total_epoch, norm_radius, seed, group, init_points_num)
    exchange = None
    if world > 1:
        from . import dp as dp_mod
        # one feedback slot per frame set of an epoch: a slot must never be shared by two sets (its capacity prediction and the
        # rank-consistent sizing of the collectives are per set, litegs_amd/dp.py)
        exchange = dp_mod.MomentExchange(trainer.params, world, n_slots=(len(frames) + world - 1) // world)
    trainer.exchange = exchange
    trainer.sched_ticks = world                           # the lr schedule counts frames, not optimizer steps (FrameTrainer.sched_ticks)
    # speculative depth-bound culling (csrc/fused.hip), flushed at every epoch boundary.  Across ranks (litegs_amd/dp.py "rank-consistent
    # speculation": lock-step replay) it is OPT-IN -- LITEGS_DP_SPECULATIVE=1 -- until the protocol has run on more than one GPU: it has
    # only ever seen two gloo ranks sharing a device, and a rank-local host exception would leave the peers blocked in a collective
    # (VERDICT / ADVICE round 5).  The default across ranks is the gated repeat.
    trainer.speculative = bool(fused) and (world == 1 or os.environ.get("LITEGS_DP_SPECULATIVE", "0") == "1")
    say(f"[litegs_amd] {len(frames)} training frames {W}x{H}, {len(test_frames_dev)} test frames, {init_points_num} initial points, "
        f"{total_epoch} epochs, world {world}, scene radius {norm_radius:.3f}")

    t_start = time.time()
    history = []
    for epoch in range(start_epoch, total_epoch):
        trainer.degree = min(int(epoch / 5), lp.sh_degree)
        with trainer.begin_epoch(epoch):
            for slot, peers in epoch_schedule(len(frames), world, epoch, seed):
                trainer.step(peers[rank], exchange, slot, peers)
                if cam_state is not None:             # trainer.py:158-160: no gradient ever reaches the embedding (module docstring): a no-op
                    cam_state["view_opt"].step()
                    cam_state["view_opt"].zero_grad()
        if exchange is not None:
            exchange.check()
        trainer.flush()
        record = {"epoch": epoch, "points": trainer.n_chunks * trainer.S}
        if epoch in test_epochs:
            record["psnr_train"] = evaluate(trainer, frames)
            say("\n[EPOCH {}] {} Evaluating: PSNR {}".format(epoch, "Trainingset", record["psnr_train"]))
            if test_frames_dev:
                record["psnr_test"] = evaluate(trainer, test_frames_dev)
                say("\n[EPOCH {}] {} Evaluating: PSNR {}".format(epoch, "Testset", record["psnr_test"]))
        trainer.end_epoch(epoch)
        record["points_after"] = trainer.n_chunks * trainer.S
        history.append(record)
        if on_epoch is not None:
            on_epoch(epoch, trainer, record)

        last = epoch == total_epoch - 1
        if (epoch in save_ply or last) and rank == 0:
            if last:
                torch.cuda.synchronize()
                say("{} takes: {}".format(lp.model_path, time.time() - t_start))
            sub = "finish" if last else "iteration_{}".format(epoch)
            flat = scene.uncluster(*[p.detach() for p in trainer.params])
            io_manager.save_ply(os.path.join(lp.model_path, "point_cloud", sub, "point_cloud.ply"), *flat)
            if cam_state is not None:                 # trainer.py:221-222
                torch.save(list(cam_state["extr"].parameters()) + [cam_state["intr"]], os.path.join(lp.model_path, "point_cloud", sub, "viewproj.pth"))
        if epoch in save_checkpoint and rank == 0:
            io_manager.save_checkpoint(lp.model_path, epoch, trainer.opt, trainer.sched)
    if dist is not None:
        dist.barrier()
    return trainer, history
