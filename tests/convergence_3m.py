"""Long-horizon convergence at the headline size (north_star: "PSNR within 0.1 dB at 30k iters" on a 3 M-Gaussian scene).

Teacher -> student at BASELINE configs[2]'s size: a seeded 3 M-Gaussian teacher cloud is rendered from `frames` orbit cameras at
1920x1080 (targets); a perturbed student is trained for 30 000 iterations with the reference's schedule (litegs/training/trainer.py:
108-195: epochs over the frames, SH degree = min(epoch // 5, 3), density control every 5 epochs with opacity decay, Morton re-sort after
every densification, position-lr decay) by
  executor x3  -- the native executor, three runs from the SAME student: float atomics reorder the blend backward's sums, so the three
                  trajectories differ; their spread is the noise floor every other difference is to be read against;
  operator     -- the same loop through the litegs_fused operator surface (what the reference's unmodified Python drives).
PSNR (mean over 16 fixed frames against the teacher renders) every `eval_every` epochs.

    python tests/convergence_3m.py --out gpurun_out/convergence_3m.md           (GPU box, ~5 minutes)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch_reference as TR                      # noqa: E402
from convergence import perturb                   # noqa: E402
from litegs_amd import synthetic as S             # noqa: E402
from litegs_amd.trainer import SyntheticTrainer   # noqa: E402


def _snapshot(path, tag, tr):
    """debugging aid (LITEGS_CONV_SNAPSHOT=<file>): one JSON line per epoch with every device segment / block of the caching allocator and
    the addresses of the trainer's long-lived tensors -- enough to attribute the address of a GPU memory access fault afterwards"""
    segs = [[s["address"], s["total_size"], [[b.get("address", 0), b["size"], b["state"][0]] for b in s["blocks"]]] for s in torch.cuda.memory_snapshot()]
    named = {}
    if tr is not None:
        for i, prm in enumerate(tr.params):
            named[f"param{i}"] = [prm.data_ptr(), prm.numel() * prm.element_size()]
        rd = getattr(tr, "renderer", None)
        for name in ("sched", "tile_order", "hot_counter", "_cull_scratch"):
            t = getattr(rd, name, None) if rd is not None else None
            if t is not None:
                named[name] = [t.data_ptr(), t.numel() * t.element_size()]
    with open(path, "a") as f:
        f.write(json.dumps(dict(tag=tag, segments=segs, named=named)) + "\n")
        f.flush()
        os.fsync(f.fileno())


def save_state(path, tr, epoch):
    """the student cloud + its Adam moments at an epoch boundary (tools/late_phase.py replays steps on it under a profiler)"""
    by = {g["name"]: g["params"][0] for g in tr.opt.param_groups}
    order = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]
    st = dict(epoch=epoch, degree=tr.degree, params=[by[n].detach().cpu() for n in order],
              exp_avg=[tr.opt.state[by[n]].get("exp_avg", torch.zeros_like(by[n])).cpu() for n in order],
              exp_avg_sq=[tr.opt.state[by[n]].get("exp_avg_sq", torch.zeros_like(by[n])).cpu() for n in order],
              lr={g["name"]: float(g["lr"]) for g in tr.opt.param_groups})
    torch.save(st, path)


def train(student, targets, cfg, fused, epochs, eval_frames, eval_every, densify, log, settings=None):
    from litegs_amd import densify as D
    from litegs_amd.statistics import STATS
    tr = SyntheticTrainer(cfg["n"], cfg["W"], cfg["H"], cfg["focal"], n_frames=cfg["frames"], seed=cfg["seed"], scene=student, fused=fused, noise_targets=False)
    for k, t in enumerate(targets):
        tr.frames[k].gt = t
    tr.speculative = fused                                   # the executor's speculative culling, as bench.py and training.start run it
    for key, val in (settings or {}).items():                # A/B runs: attributes of the executor (litegs_amd/fast.py FusedRenderer)
        if not hasattr(tr.renderer, key):
            raise AttributeError(f"FusedRenderer has no option '{key}'")
        setattr(tr.renderer, key, val)
    save_at = os.environ.get("LITEGS_CONV_SAVE")             # "<file>:<epoch>"
    epoch_ms, epoch_inst = [], []
    ctl = tr.enable_densify(D.DensifyParams(**densify), total_epochs=epochs, seed=cfg["seed"]) if densify else None

    def evaluate():
        with torch.no_grad():
            return float(np.mean([TR.psnr(tr.forward_only(k)[0].clamp(0, 1), targets[k][0]) for k in eval_frames]))

    curve, sizes, at = [evaluate()], [tr.n_chunks * tr.S], [0]
    torch.cuda.synchronize()
    t0 = time.time()
    rng = np.random.default_rng(cfg["seed"] + 7)
    for epoch in range(epochs):
        te = time.time()
        tr.degree = min(epoch // 5, 3)                       # trainer.py:111
        order = rng.permutation(cfg["frames"])               # the reference's DataLoader shuffles
        if ctl is not None:
            with tr.begin_epoch(epoch):
                for k in order:
                    tr.step(int(k))
            tr.end_epoch(epoch)
        else:
            for k in order:
                tr.step(int(k))
        torch.cuda.synchronize()
        epoch_ms.append((time.time() - te) / cfg["frames"] * 1e3)
        if fused:
            epoch_inst.append(float(np.mean(tr.renderer.fb_total)))
        for spec in filter(None, (save_at or "").split(",")):          # "<file>:<epoch>[,<file>:<epoch>...]"
            path, at_epoch = spec.rsplit(":", 1)
            if fused and epoch + 1 == int(at_epoch) and not os.path.exists(path):
                save_state(path, tr, epoch + 1)
        if os.environ.get("LITEGS_CONV_SNAPSHOT"):
            _snapshot(os.environ["LITEGS_CONV_SNAPSHOT"], f"{'executor' if fused else 'operator'} epoch {epoch}", tr)
        if os.environ.get("LITEGS_CONV_VERBOSE"):
            torch.cuda.synchronize()
            log(f"   epoch {epoch} done: {tr.n_chunks * tr.S} points, degree {tr.degree}")
        if (epoch + 1) % eval_every == 0 or epoch == epochs - 1:
            curve.append(evaluate()); sizes.append(tr.n_chunks * tr.S); at.append((epoch + 1) * cfg["frames"])
    torch.cuda.synchronize()
    secs = time.time() - t0
    rd = tr.renderer
    tr.flush()                                               # (collects the always-on sanitised-word counters, csrc/lg_sanity.h)
    info = dict(sanitised=dict(tr.sanitised), psnr=curve, size=sizes, iterations=at, seconds=secs, ms_per_iteration=secs / (epochs * cfg["frames"]) * 1e3,
                unculled_reruns=int(rd.fallbacks), replayed_steps=int(tr.spec_replays), truncated=int(rd.truncated_visits), finite=all(bool(torch.isfinite(p).all()) for p in tr.params),
                epoch_ms=epoch_ms, epoch_instances=epoch_inst)
    if ctl is not None:
        STATS.reset(1, 1, enabled_for_epoch=lambda e: False, device="cuda")
        STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
    tr.close()
    del tr
    torch.cuda.empty_cache()
    log(f"{'executor' if fused else 'operator'}: {curve[0]:.3f} -> {curve[-1]:.3f} dB, {sizes[-1]} points, {secs:.1f} s ({info['ms_per_iteration']:.3f} ms / iteration)")
    return info


def run(iterations=30000, frames=150, n=3_000_000, W=1920, H=1080, focal=1200.0, seed=0, runs=3, eval_every=10, log=lambda *a: print(*a, flush=True),
        settings=None):
    cfg = dict(n=n, W=W, H=H, focal=focal, frames=frames, seed=seed)
    epochs = iterations // frames
    t0 = time.time()
    teacher = S.make_scene(n, seed=seed)
    student = perturb(teacher, seed + 1, amount=0.5)
    teach = SyntheticTrainer(n, W, H, focal, n_frames=frames, seed=seed, scene=teacher, noise_targets=False)
    targets = [teach.forward_only(k).clamp(0, 1).clone() for k in range(frames)]
    del teach
    torch.cuda.empty_cache()
    eval_frames = list(range(0, frames, max(1, frames // 16)))[:16]
    densify = dict(target_primitives=int(1.1 * n))                       # the reference's defaults otherwise (arguments.py:95-110)
    out = dict(config=cfg, iterations=iterations, epochs=epochs, densify=densify, eval_frames=eval_frames, settings=settings or {})
    log(f"targets rendered ({frames} frames {W}x{H}), {epochs} epochs; {time.time() - t0:.0f} s")
    out["executor"] = []
    for _ in range(runs):
        out["executor"].append(train(student, targets, cfg, True, epochs, eval_frames, eval_every, densify, log, settings))
        if os.environ.get("LITEGS_CONV_PARTIAL"):                       # completed curves survive a later failure
            with open(os.environ["LITEGS_CONV_PARTIAL"], "w") as f:
                json.dump(out, f)
    if os.environ.get("LITEGS_CONV_SKIP_OPERATOR"):                     # executor runs only: the operator curve is taken from an earlier run's JSON
        out["operator"] = json.load(open(os.environ["LITEGS_CONV_SKIP_OPERATOR"]))["operator"]
    else:
        out["operator"] = train(student, targets, cfg, False, epochs, eval_frames, eval_every, densify, log)
    out["seconds"] = time.time() - t0
    return out


def to_markdown(out):
    cfg = out["config"]
    ex, op = out["executor"], out["operator"]
    L = [f"# Convergence at the headline size: {cfg['n']} Gaussians, {cfg['frames']} cameras {cfg['W']}x{cfg['H']}, {out['iterations']} iterations", "",
         f"Teacher -> student (tests/convergence_3m.py): student = teacher + noise; {out['epochs']} epochs over {cfg['frames']} shuffled frames; SH degree "
         f"min(epoch // 5, 3); density control with the reference's defaults (every 5 epochs from epoch 3 to 80 % of the run, opacity decay every "
         f"10, prune by weight, budget {out['densify']['target_primitives']} primitives), Morton re-sort after each densification; reference learning "
         f"rates and position-lr decay; L1 + 0.2 D-SSIM.  PSNR = mean over {len(out['eval_frames'])} fixed frames against the teacher renders.  "
         f"Whole script: {out['seconds']:.0f} s on one MI355X.", "",
         "| iterations | " + " | ".join(f"executor run {i + 1}" for i in range(len(ex))) + " | operator path | executor points | operator points |",
         "|---:|" + "---:|" * (len(ex) + 3)]
    for j, it in enumerate(ex[0]["iterations"]):
        L.append(f"| {it} | " + " | ".join(f"{r['psnr'][j]:.3f}" for r in ex) + f" | {op['psnr'][j]:.3f} | {ex[0]['size'][j]} | {op['size'][j]} |")
    fin = np.array([r["psnr"][-1] for r in ex])
    tail = np.array([np.mean(r["psnr"][-3:]) for r in ex])
    L += ["", "| quantity | value |", "|---|---:|",
          f"| executor, final PSNR: mean of the runs | {fin.mean():.3f} dB |",
          f"| executor, run-to-run spread of the final PSNR (max - min: the atomics-order noise floor) | {fin.max() - fin.min():.3f} dB |",
          f"| operator path, final PSNR | {op['psnr'][-1]:.3f} dB |",
          f"| final dPSNR, mean executor vs operator | {abs(fin.mean() - op['psnr'][-1]):.3f} dB |",
          f"| the same on the mean of the last three evaluations | {abs(tail.mean() - np.mean(op['psnr'][-3:])):.3f} dB |",
          f"| largest |dPSNR| between executor run 1 and the operator path over the whole curve | {np.abs(np.array(ex[0]['psnr']) - np.array(op['psnr'])).max():.3f} dB |" if len(ex[0]['psnr']) == len(op['psnr']) else "",
          f"| largest |dPSNR| between executor runs 1 and 2 over the whole curve | {np.abs(np.array(ex[0]['psnr']) - np.array(ex[1]['psnr'])).max():.3f} dB |" if len(ex) > 1 else "",
          f"| ms per iteration (training + density control + evaluation), executor / operator | {np.mean([r['ms_per_iteration'] for r in ex]):.3f} / {op['ms_per_iteration']:.3f} |",
          f"| frames repeated unculled (a depth bound was violated), executor runs | {', '.join(str(r['unculled_reruns']) for r in ex)} |",
          f"| steps replayed by the speculative executor (the failed step and those enqueued behind it) | {', '.join(str(r.get('replayed_steps', 0)) for r in ex)} |",
          f"| truncated tables observed | {', '.join(str(r['truncated']) for r in ex)} |",
          f"| garbage table words neutralised by a kernel (lg_sanity.h sites; truncations excluded) | {', '.join(str(sum(v for k, v in r.get('sanitised', {}).items() if k != 'truncated_tables')) for r in ex)} |",
          f"| parameters finite at the end | {all(r['finite'] for r in ex) and op['finite']} |"]
    if ex and ex[0].get("epoch_ms"):
        L += ["", "Cost per iteration over the run (executor run 1; plain epochs and statistics / density-control epochs alike; wall clock per epoch / frames):", "",
              "| epoch | ms / iteration | mean emitted instances per frame (M) |", "|---:|---:|---:|"]
        em, ei = ex[0]["epoch_ms"], ex[0].get("epoch_instances", [])
        for e in list(range(0, len(em), 10)) + [len(em) - 1]:
            lo, hi = e, min(e + 10, len(em))
            inst = f"{np.mean(ei[lo:hi]) / 1e6:.1f}" if ei else ""
            L.append(f"| {lo}-{hi - 1} | {np.mean(em[lo:hi]):.3f} | {inst} |")
    if out.get("settings"):
        L += ["", f"Executor options of this run: {out['settings']}"]
    return "\n".join(x for x in L if x != "") + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/convergence_3m.md")
    ap.add_argument("--iterations", type=int, default=30000)
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--n", type=int, default=3_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--eval-every", type=int, default=10)
    ap.add_argument("--set", default="", help="executor options for an A/B run: attr=value[,attr=value...] (FusedRenderer attributes)")
    a = ap.parse_args()
    settings = {}
    for kv in filter(None, a.set.split(",")):
        key, val = kv.split("=")
        settings[key] = {"true": True, "false": False}.get(val.lower(), None)
        if settings[key] is None:
            settings[key] = int(val)
    res = run(iterations=a.iterations, frames=a.frames, n=a.n, W=a.width, H=a.height, runs=a.runs, eval_every=a.eval_every, settings=settings)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        f.write(to_markdown(res))
    with open(os.path.splitext(a.out)[0] + ".json", "w") as f:
        json.dump(res, f)
    print(open(a.out).read())
