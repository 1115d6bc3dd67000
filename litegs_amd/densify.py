"""Density control on top of the hot path's statistics -- host mirror of ``litegs/training/densify.py``.

The controller is control plane (pure torch in the reference too); it is here because the statistics it consumes are produced by
the blend kernels in statistic mode (fragment counts / weights / error moments, csrc/raster.hip) and because, in a data-parallel
job, every rank must take IDENTICAL clone / split / prune decisions or the replicas diverge:

* ``Statistics.all_reduce`` is called first, so the moments equal those of one process that rendered every rank's frames;
* the two random draws (the multinomial pick of densification candidates and the normal offsets of split children) come from a
  ``Sampler`` seeded by (seed, epoch) only -- parameters are bit-identical across ranks (synchronous updates), so are the draws.

Layout: every parameter is [..., chunks, S]; all decisions are taken on the flat [.., N] view and whole chunks are appended /
removed (the reference's truncation rules: the appended set and the pruned set are cut to a multiple of S, densify.py:153-160,
209-217).  Rules mirrored: DensityControllerTamingGS (the controller trainer.py:105 instantiates) with both prune modes and both
opacity-reset modes; budget schedule densify.py:286-287; score densify.py:276-281; children = parents' scale / 1.6 (densify.py:190).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from .statistics import STATS, Statistics

PARAM_NAMES = ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity")


@dataclass
class DensifyParams:
    """Defaults of litegs/arguments.py:95-110."""
    densification_interval: int = 5
    densify_from: int = 3
    densify_until: int = -1
    opacity_reset_interval: int = 10
    opacity_reset_mode: str = "decay"          # 'decay' | 'reset'
    prune_mode: str = "weight"                 # 'weight' | 'threshold'
    target_primitives: int = 1000000
    densify_grad_threshold: float = 0.00015
    opacity_threshold: float = 0.005
    screen_size_threshold: int = 128
    percent_dense: float = 0.01

    def resolve_until(self, total_epochs: int) -> None:
        """trainer.py:103-104: a negative densify_until means 80 % of the run, rounded down to a reset boundary, plus one."""
        if self.densify_until < 0:
            self.densify_until = int(total_epochs * 0.8 / self.opacity_reset_interval) * self.opacity_reset_interval + 1


class Sampler:
    """The controller's two random draws.  Default: torch generators on the parameters' device seeded from (seed, epoch)."""

    def __init__(self, seed: int = 0):
        self.seed = seed
        self._gen: Dict[str, torch.Generator] = {}

    def begin(self, epoch: int, device: torch.device) -> None:
        g = torch.Generator(device=device)
        g.manual_seed(self.seed * 1_000_003 + epoch)
        self._gen = {"g": g}

    def multinomial(self, score: torch.Tensor, budget: int) -> torch.Tensor:
        return torch.multinomial(score, budget, replacement=False, generator=self._gen["g"])

    def normal(self, std: torch.Tensor) -> torch.Tensor:
        return torch.normal(mean=torch.zeros_like(std), std=std, generator=self._gen["g"])


def rotation_rows(rot: torch.Tensor) -> torch.Tensor:
    """unit quaternions [4,n] (r,x,y,z) -> [3,3,n], the transform matrix of GR/transform.cu:120-160 with unit scale."""
    r, x, y, z = rot[0], rot[1], rot[2], rot[3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y),
            2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x),
            2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=0).reshape(3, 3, -1)


def _inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    return torch.log(x / (1 - x))


class DensityController:
    def __init__(self, screen_extent: float, params: DensifyParams, chunk_size: int, init_points_num: int,
                 stats: Statistics = STATS, sampler: Optional[Sampler] = None, group=None):
        assert params.target_primitives != 0
        self.p = params
        self.screen_extent = float(screen_extent)
        self.S = int(chunk_size)
        self.init_points_num = int(init_points_num)
        self.stats = stats
        self.sampler = sampler or Sampler(0)
        self.group = group
        self.on_change: Optional[Callable[[], None]] = None       # trainer hook: cached pointers / feedback buffers are stale
        self.last: Dict[str, int] = {}

    # -- schedule (densify.py:221-226) -------------------------------------------------------------------------------------------
    def is_densify_actived(self, epoch: int) -> bool:
        return self.p.densify_from <= epoch < self.p.densify_until and epoch % self.p.densification_interval == 0

    # -- optimizer plumbing ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _groups(optimizer) -> Dict[str, dict]:
        return {g["name"]: g for g in optimizer.param_groups}

    def _flat(self, optimizer) -> Dict[str, torch.Tensor]:
        return {n: g["params"][0].data.reshape(*g["params"][0].shape[:-2], -1) for n, g in self._groups(optimizer).items() if n in PARAM_NAMES}

    def _chunked(self, t: torch.Tensor) -> torch.Tensor:
        return t.reshape(*t.shape[:-1], t.shape[-1] // self.S, self.S)

    @staticmethod
    def _swap(optimizer, group: dict, new_param: torch.Tensor, new_state: Optional[dict]) -> None:
        old = group["params"][0]
        optimizer.state.pop(old, None)
        group["params"][0] = torch.nn.Parameter(new_param.contiguous())
        if new_state is not None:
            optimizer.state[group["params"][0]] = new_state

    def _append(self, optimizer, new: Dict[str, torch.Tensor]) -> None:
        """whole chunks appended; their Adam moments start at zero (densify.py:38-55)."""
        for name, group in self._groups(optimizer).items():
            p = group["params"][0]
            ext = new[name]
            st = optimizer.state.get(p)
            if st is not None and "exp_avg" in st:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=-2).contiguous()
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=-2).contiguous()
            self._swap(optimizer, group, torch.cat((p.data, ext), dim=-2), st)

    def _keep(self, optimizer, keep: torch.Tensor) -> None:
        """keep: int64 indices into the flat view (a multiple of S of them are dropped, so the result re-chunks exactly)."""
        for name, group in self._groups(optimizer).items():
            p = group["params"][0]
            st = optimizer.state.get(p)
            take = lambda t: self._chunked(t.reshape(*t.shape[:-2], -1)[..., keep])
            if st is not None and "exp_avg" in st:
                st["exp_avg"] = take(st["exp_avg"]).contiguous()
                st["exp_avg_sq"] = take(st["exp_avg_sq"]).contiguous()
            self._swap(optimizer, group, take(p.data), st)

    # -- decisions ---------------------------------------------------------------------------------------------------------------
    def prune_mask(self, opacity_act: torch.Tensor) -> torch.Tensor:
        """bool[N] (densify.py:111-118, 259-271).  Statistics cover the Gaussians that existed when they were reset: anything
        appended since is never flagged by them."""
        n = opacity_act.shape[-1]
        if self.p.prune_mode == "weight":
            mask = torch.zeros((n,), dtype=torch.bool, device=opacity_act.device)
            got = self.stats.mean("fragment_weight")
            if got is not None:
                w, cnt = got
                invisible = (w * cnt).nan_to_num(0).reshape(-1) == 0
                mask[: invisible.shape[0]] |= invisible
            return mask
        mask = (opacity_act < self.p.opacity_threshold).reshape(-1).clone()
        invisible = self.stats.never_visible()
        mask[: invisible.shape[0]] |= invisible
        return mask

    def score(self, opacity: torch.Tensor) -> torch.Tensor:
        got = self.stats.var("fragment_err")
        n = opacity.shape[-1]
        if got is None:
            return torch.zeros((n,), device=opacity.device)
        var, cnt = got
        s = (var * cnt).reshape(-1)
        if s.shape[0] < n:                                           # grown since the statistics were reset: no evidence yet
            s = torch.cat((s, torch.zeros((n - s.shape[0],), device=s.device)))
        sig = opacity.reshape(-1).sigmoid()
        return (s * (sig * sig)).nan_to_num(0).clamp_min(0)

    def budget(self, epoch: int, n_points: int, prune_num: int) -> int:
        p = self.p
        target = (p.target_primitives - self.init_points_num) / (p.densify_until - p.densify_from) * (epoch - p.densify_from) + self.init_points_num
        return int(min(max(int(target - n_points), 1) + prune_num, n_points))

    # -- operations --------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def split_and_clone(self, optimizer, epoch: int) -> None:
        f = self._flat(optimizer)
        xyz, scale, rot, opacity = f["xyz"], f["scale"], f["rot"], f["opacity"]
        n = xyz.shape[-1]
        prune_num = int(self.prune_mask(opacity.sigmoid()).sum())
        budget = self.budget(epoch, n, prune_num)
        picked = self.sampler.multinomial(self.score(opacity), budget)
        largest = scale[:, picked].exp().max(dim=0).values
        limit = self.p.percent_dense * self.screen_extent
        clone_idx, split_idx = picked[largest <= limit], picked[largest > limit]
        # split children: parent + R^T-rotated normal offset with the parent's std, scale / 1.6; the parent stays (densify.py:296-310)
        std = scale[:, split_idx].exp()
        samples = self.sampler.normal(std)                                                   # [3, n_split]
        T = rotation_rows(torch.nn.functional.normalize(rot[:, split_idx], dim=0))          # [3, 3, n_split]
        shift = torch.einsum("ip,ijp->jp", samples, T)
        new = {"xyz": torch.cat((xyz[:, split_idx] + shift, xyz[:, clone_idx]), dim=-1),
               "scale": torch.cat(((std / (0.8 * 2)).log(), scale[:, clone_idx]), dim=-1)}
        for name in ("rot", "sh_0", "sh_rest", "opacity"):
            new[name] = torch.cat((f[name][..., split_idx], f[name][..., clone_idx]), dim=-1)
        m = new["xyz"].shape[-1] // self.S * self.S                     # whole chunks only; the tail is dropped (densify.py:330-337)
        self.last.update(budget=budget, split=int(split_idx.shape[0]), clone=int(clone_idx.shape[0]), appended=m)
        self._append(optimizer, {k: self._chunked(v[..., :m].contiguous()) for k, v in new.items()})

    @torch.no_grad()
    def prune(self, optimizer, epoch: int) -> None:
        f = self._flat(optimizer)
        mask = self.prune_mask(f["opacity"].sigmoid())
        n = mask.shape[0]
        flagged = int(mask.sum())
        assert flagged <= 0.8 * n, "prune would remove more than 80 % of the Gaussians"          # densify.py:148-149
        limit = flagged // self.S * self.S                                                       # whole chunks' worth (densify.py:150-156)
        drop = mask.nonzero()[:limit, 0]
        keep_mask = torch.ones((n,), dtype=torch.bool, device=mask.device)
        keep_mask[drop] = False
        self.last.update(pruned=limit)
        self._keep(optimizer, keep_mask.nonzero()[:, 0])

    @torch.no_grad()
    def reset_opacity(self, optimizer, epoch: int) -> None:
        group = self._groups(optimizer)["opacity"]
        p = group["params"][0]
        act = p.data.sigmoid()
        if self.p.opacity_reset_mode == "decay":
            p.data = _inverse_sigmoid((act * 0.5).clamp_min(1.0 / 128))
            optimizer.state.clear()                                      # ALL moments restart (densify.py:208-209)
        elif self.p.opacity_reset_mode == "reset":
            st = optimizer.state.get(p)
            new = _inverse_sigmoid(act.clamp_max(0.005))
            if st is not None and "exp_avg" in st:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
            self._swap(optimizer, group, new, st)

    @torch.no_grad()
    def step(self, optimizer, epoch: int):
        """End-of-epoch hook (densify.py:228-243, trainer.py:195).  Returns (xyz, scale, rot, sh_0, sh_rest, opacity)."""
        p = self.p
        if p.densify_from <= epoch < p.densify_until:
            changed = False
            if epoch % p.densification_interval == 0 or epoch % p.opacity_reset_interval == 0:
                self.stats.all_reduce(self.group)                     # identical evidence on every rank (no-op for one process)
                g0 = self._groups(optimizer)["xyz"]["params"][0]
                self.sampler.begin(epoch, g0.device)
            if epoch % p.densification_interval == 0:
                self.split_and_clone(optimizer, epoch)
                self.prune(optimizer, epoch)
                changed = True
            if epoch % p.opacity_reset_interval == 0:
                self.reset_opacity(optimizer, epoch)
                changed = True
            if changed:
                xyz = self._groups(optimizer)["xyz"]["params"][0]
                self.stats.reset(xyz.shape[-2], xyz.shape[-1], self.is_densify_actived, device=xyz.device)
                if self.on_change is not None:
                    self.on_change()
        groups = self._groups(optimizer)
        return tuple(groups[n]["params"][0] for n in PARAM_NAMES)
