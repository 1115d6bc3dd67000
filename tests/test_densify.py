"""Density controller (litegs_amd/densify.py) against the reference's DensityControllerTamingGS run on the CPU
(tests/golden/make_golden.py, fixture keys dn_*): same statistics in, the two random draws replayed, parameters and Adam moments
out.  Copies must be bit-identical; the split children's positions / scales go through differently-ordered float sums (1e-6)."""
import os

import numpy as np
import torch

from litegs_amd import densify as D
from litegs_amd import optimizer as opt_mod

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_chain.npz"))
NAMES = ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity")


class FixtureStats:
    def __init__(self, tag):
        self.t = {k: torch.from_numpy(GOLD[f"dn_{tag}_{k}"]) for k in ("err_var", "err_cnt", "w_mean", "w_cnt", "culled")}
        self.resets = []
        self.reduced = 0

    def var(self, key):
        assert key == "fragment_err"
        return self.t["err_var"], self.t["err_cnt"]

    def mean(self, key):
        assert key == "fragment_weight"
        return self.t["w_mean"], self.t["w_cnt"]

    def never_visible(self):
        return self.t["culled"]

    def all_reduce(self, group=None):
        self.reduced += 1

    def reset(self, chunks, S, handle, device=None):
        self.resets.append((chunks, S))


class ReplaySampler(D.Sampler):
    def __init__(self, tag):
        super().__init__(0)
        self.picked = torch.from_numpy(GOLD[f"dn_{tag}_picked"])
        self.samples = torch.from_numpy(GOLD[f"dn_{tag}_samples"])

    def begin(self, epoch, device):
        pass

    def multinomial(self, score, budget):
        assert budget == self.picked.shape[0], (budget, self.picked.shape)
        assert (score[self.picked] > 0).all(), "the reference can only have drawn candidates with a positive score"
        return self.picked

    def normal(self, std):
        assert std.shape == self.samples.shape
        return self.samples


def _optimizer():
    ps = {n: torch.nn.Parameter(torch.from_numpy(GOLD[f"dn_init_{n}"]).clone()) for n in NAMES}
    opt, _ = opt_mod.get_optimizer(*[ps[n] for n in NAMES], 2.5, opt_mod.OptimizationParams())
    for g in opt.param_groups:
        p = g["params"][0]
        opt.state[p] = {"step": torch.tensor(7.0), "exp_avg": torch.from_numpy(GOLD[f"dn_init_m_{g['name']}"]).clone(),
                        "exp_avg_sq": torch.from_numpy(GOLD[f"dn_init_v_{g['name']}"]).clone()}
    return opt


def _check(opt, tag):
    for g in opt.param_groups:
        n = g["name"]
        p = g["params"][0]
        want = GOLD[f"dn_{tag}_p_{n}"]
        assert tuple(p.shape) == want.shape, (n, p.shape, want.shape)
        if n in ("xyz", "scale", "opacity"):
            np.testing.assert_allclose(p.detach().numpy(), want, rtol=2e-6, atol=2e-6, err_msg=n)
        else:
            assert np.array_equal(p.detach().numpy(), want), n
        st = opt.state.get(p)
        assert bool(st) == bool(GOLD[f"dn_{tag}_has_state_{n}"]), n
        if st:
            assert np.array_equal(st["exp_avg"].numpy(), GOLD[f"dn_{tag}_m_{n}"]), n
            assert np.array_equal(st["exp_avg_sq"].numpy(), GOLD[f"dn_{tag}_v_{n}"]), n
            assert st["exp_avg"].shape == p.shape


def test_densify_matches_reference_controller():
    opt = _optimizer()
    dp = D.DensifyParams(densify_until=int(GOLD["dn_until"]), target_primitives=int(GOLD["dn_target"]))
    changes = []
    # epoch 5: split + clone + prune by fragment weight; moments of the survivors are carried, new chunks start at zero
    stats = FixtureStats("a")
    ctl = D.DensityController(float(GOLD["dn_extent"]), dp, 128, 512, stats, ReplaySampler("a"))
    ctl.on_change = lambda: changes.append(1)
    out = ctl.step(opt, 5)
    assert [tuple(t.shape) for t in out] == [GOLD[f"dn_a_p_{n}"].shape for n in NAMES]
    _check(opt, "a")
    assert stats.reduced == 1 and stats.resets == [(5, 128)] and changes == [1]
    assert ctl.last["split"] + ctl.last["clone"] == ctl.last["budget"] and ctl.last["appended"] % 128 == 0 and ctl.last["pruned"] % 128 == 0
    # epoch 10: threshold pruning (transparent or never visible) and the 'decay' opacity reset, which drops every Adam moment
    dp.prune_mode = "threshold"
    stats = FixtureStats("b")
    ctl = D.DensityController(float(GOLD["dn_extent"]), dp, 128, 512, stats, ReplaySampler("b"))
    ctl.step(opt, 10)
    _check(opt, "b")
    assert len(opt.state) == 0
    # outside the window / off the interval: nothing happens
    before = [g["params"][0] for g in opt.param_groups]
    ctl.step(opt, 11)
    ctl.step(opt, 45)
    assert all(a is b for a, b in zip(before, [g["params"][0] for g in opt.param_groups]))


def test_reset_mode_and_schedule():
    opt = _optimizer()
    dp = D.DensifyParams(opacity_reset_mode="reset")
    dp.resolve_until(100)
    assert dp.densify_until == 81                                  # int(100 * 0.8 / 10) * 10 + 1 (trainer.py:103-104)
    ctl = D.DensityController(2.0, dp, 128, 512, FixtureStats("a"), ReplaySampler("a"))
    assert [e for e in range(0, 30) if ctl.is_densify_actived(e)] == [5, 10, 15, 20, 25]
    old = {g["name"]: g["params"][0] for g in opt.param_groups}
    ctl.reset_opacity(opt, 10)
    new = {g["name"]: g["params"][0] for g in opt.param_groups}
    assert new["opacity"] is not old["opacity"] and all(new[n] is old[n] for n in NAMES if n != "opacity")
    assert float(new["opacity"].detach().sigmoid().max()) <= 0.005 + 1e-6
    st = opt.state[new["opacity"]]
    assert float(st["exp_avg"].abs().max()) == 0 and float(st["exp_avg_sq"].abs().max()) == 0
    assert float(opt.state[new["xyz"]]["exp_avg"].abs().max()) > 0  # the other groups keep their moments
    # budget schedule (densify.py:286-287): linear growth towards target_primitives, at least 1, plus what pruning will free
    dp2 = D.DensifyParams(densify_until=41, target_primitives=3000)
    c2 = D.DensityController(2.0, dp2, 128, 512, FixtureStats("a"), ReplaySampler("a"))
    assert c2.budget(5, 512, 152) == int((3000 - 512) / 38 * 2 + 512 - 512) + 152
    assert c2.budget(5, 100000, 0) == 1 and c2.budget(40, 512, 10_000) == 512


def test_default_sampler_is_a_function_of_seed_and_epoch_only():
    std = torch.rand((3, 50)) + 0.1
    score = torch.rand((400,))
    draws = []
    for _ in range(2):
        s = D.Sampler(3)
        s.begin(7, torch.device("cpu"))
        draws.append((s.multinomial(score, 60), s.normal(std)))
    assert torch.equal(draws[0][0], draws[1][0]) and torch.equal(draws[0][1], draws[1][1])
    s.begin(8, torch.device("cpu"))
    assert not torch.equal(s.multinomial(score, 60), draws[0][0])


def test_rotation_rows_matches_transform_fixture():
    q, sc, T = GOLD["tm_quat"], GOLD["tm_scale"], GOLD["tm_T"]      # reference CreateTransformMatrix: T[i,j] = scale[i] * R[i,j]
    Rm = D.rotation_rows(torch.from_numpy(q)).numpy()
    np.testing.assert_allclose(Rm * sc[:, None, :], T, rtol=1e-5, atol=1e-6)
