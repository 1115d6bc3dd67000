"""Host-side logic that needs no GPU: synthetic workloads, the sparse-gradient container, sort sizing, oracle binning invariants."""
import numpy as np
import torch

from litegs_amd import synthetic as S


def test_scene_is_seeded_chunked_and_morton_sorted():
    a = S.make_scene(1000, seed=0)
    b = S.make_scene(1000, seed=0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    xyz, scale, rot, sh0, shr, opa = a
    assert xyz.shape == (3, 8, 128) and shr.shape == (15, 3, 8, 128) and opa.shape == (1, 8, 128)
    assert np.array_equal(xyz[:, -1, -24:], xyz[:, -1, 1000 - 7 * 128 - 24:1000 - 7 * 128])     # padded by repeating the tail
    flat = xyz.reshape(3, -1)[:, :1000]
    assert np.array_equal(S.morton_order(flat), np.arange(1000))


def test_camera_looks_at_target():
    view, proj, planes = S.make_camera(640, 480, 500.0, 500.0, (3.0, -1.0, 2.0), (0, 0, 0))
    p = np.array([0, 0, 0, 1.0], np.float32) @ view[0]
    assert abs(p[0]) < 1e-5 and abs(p[1]) < 1e-5 and p[2] > 0
    h = p @ proj[0]
    assert abs(h[3] - p[2]) < 1e-5
    assert (planes[0] @ np.array([0, 0, 0, 1.0]) > 0).all(), "target must be inside all six planes"


def test_sort_bits_rule():
    from litegs_amd import fused
    from oracle import oracle as O
    for (h, w) in [(400, 400), (1080, 1920), (1200, 1600), (141, 250)]:
        assert fused.sort_bits(h, w, 8, 16) == O.sort_bits(h, w, 8, 16)
    assert fused.sort_bits(1080, 1920, 8, 16) == 14 and fused.sort_bits(400, 400, 8, 16) == 11


def test_compacted_tensor_claims_full_shape():
    from litegs_amd.wrapper import CompactedTensor
    ids = torch.tensor([4, 1, 7])
    vals = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    g = CompactedTensor((2, 9, 5), ids, vals)
    assert g.shape == (2, 9, 5) and g.compacted_values is vals
    p = torch.nn.Parameter(torch.zeros(2, 9, 5))
    p.grad = g                                        # autograd accepts it as the gradient of a full-size parameter
    d = g.to_dense()
    assert torch.equal(d[:, 4], vals[:, 0]) and torch.equal(d[:, 7], vals[:, 2]) and d[:, 0].abs().sum() == 0
    assert isinstance(g.detach(), CompactedTensor)


def test_oracle_binning_invariants(oracle):
    """Every emitted instance lies inside its splat's tile rectangle, tiles are sorted, depth order holds inside tiles, and the
    table is a permutation of the unsorted emission (size-independent properties also used at full size on the GPU)."""
    from tests.util import oracle_forward
    res = oracle_forward("small")
    keys, vals = res.sorted_tile[0], res.sorted_point[0]
    assert (np.diff(keys) >= 0).all()
    assert keys.min() >= 1
    depth = res.view_pos[0, 2]
    same = keys[1:] == keys[:-1]
    assert (depth[vals[1:]][same] >= depth[vals[:-1]][same]).all(), "front-to-back inside every tile"
    assert len(keys) == res.alloc.sum()
    counts = np.bincount(vals, minlength=res.alloc.shape[1])
    assert np.array_equal(counts, res.alloc[0])
    ts = res.tile_start[0]
    for t in np.unique(keys):
        assert keys[ts[t]] == t and (ts[t] == 0 or keys[ts[t] - 1] != t)


def test_train_loop_schedule_with_a_fake_trainer():
    """litegs_amd.trainer.train(): the reference's epoch shape (trainer.py:108-195) -- begin_epoch guard, ceil(frames / world) steps of
    disjoint frames per rank, end_epoch hook, the exchange attached to the trainer (re-bound by the trainer only when the parameters
    are replaced, not per epoch), slots recurring with the same frame set, every rank knowing its peers' frames."""
    import contextlib
    from litegs_amd import trainer as T

    log = []

    class FakeTrainer:
        frames = list(range(6))
        params = ["p"]

        def begin_epoch(self, epoch):
            log.append(("begin", epoch))
            return contextlib.nullcontext()

        def step(self, frame, hook, slot, peers=None):
            log.append(("step", frame, hook is not None, slot, peers))

        def end_epoch(self, epoch):
            log.append(("end", epoch))

    class FakeExchange:
        def __init__(self):
            self.rebinds = 0

        def rebind(self, params):
            self.rebinds += 1

        def hook(self, *a):
            return None

    ex = FakeExchange()
    seen = []
    T.train(FakeTrainer(), 3, ex, rank=1, world=4, start_epoch=1, on_epoch=lambda e, t: seen.append(e))
    steps = [x for x in log if x[0] == "step"]
    assert [x for x in log if x[0] in ("begin", "end")] == [("begin", 1), ("end", 1), ("begin", 2), ("end", 2)]
    assert len(steps) == 2 * 2 and ex.rebinds == 0 and seen == [1, 2]            # ceil(6 / 4) = 2 steps per epoch, epochs 1 and 2
    # global step counter continues from start_epoch: steps 2,3 | 4,5 -> frames (step * 4 + 1) % 6
    assert [s[1] for s in steps] == [(k * 4 + 1) % 6 for k in (2, 3, 4, 5)]
    assert [s[3] for s in steps] == [0, 1, 0, 1] and all(s[2] for s in steps)
    assert [s[4] for s in steps] == [[(k * 4 + r) % 6 for r in range(4)] for k in (2, 3, 4, 5)]
    log.clear()
    T.train(FakeTrainer(), 1)                                                      # single process: no hook, every frame once
    assert [s[1:4] for s in log if s[0] == "step"] == [(k, False, k) for k in range(6)]


def test_guard_allocator_places_buffers_at_the_end_of_their_allocation(monkeypatch):
    """LITEGS_GUARD_ALLOC=1 (tools/fault_hunt.py): every per-frame buffer of the executor ends within `align` bytes of the end of a
    page-granular allocation of its own, so that an access past its end leaves the mapping; shapes, dtypes and contents are unchanged"""
    import torch
    from litegs_amd import fast
    monkeypatch.setattr(fast, "_GUARD_ALLOC", True)
    for shape, dtype, align in (((3, 5), torch.float32, 16), ((7,), torch.bool, 16), ((1, 1, 9, 3), torch.int16, 16), ((4097,), torch.uint8, 64)):
        t = fast._empty(shape, dtype, "cpu", zero=True, align=align)
        assert tuple(t.shape) == shape and t.dtype == dtype and t.is_contiguous() and not t.any()
        st = t.untyped_storage()
        end_of_tensor = t.data_ptr() + t.numel() * t.element_size()
        assert st.nbytes() % 4096 == 0 and 0 <= st.data_ptr() + st.nbytes() - end_of_tensor < align
        assert (t.data_ptr() - st.data_ptr()) % align == 0
    monkeypatch.setattr(fast, "_GUARD_ALLOC", False)
    t = fast._empty((2, 3), torch.float32, "cpu")
    assert t.untyped_storage().nbytes() == 24
