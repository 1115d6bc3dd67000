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


def test_oracle_bounds_a_degenerate_splats_entries_to_its_share(oracle):
    """The first tile slice of a needle-like splat can count NEGATIVE tiles (GR/speedy_splat.cuh:118-125 adds the difference to the splat's
    count all the same): the count is then smaller than the tiles its slices hold.  Pinned here on the CPU: the signed counts of the fourteen
    fixture splats (found by tests/host/walk_check.cpp's generator, the numbers in their comments), and the oracle's rule that a splat owns
    exactly [offset, offset + count) of the table -- every word of it a valid tile id, the surplus tiles of the last slices dropped
    (oracle/litegs_oracle.c process_tiles; csrc/binning.hip follows it bit for bit: tests/test_gpu_edge.py, profiles/r05_fault_root_cause.md)."""
    from test_gpu_edge import DEGENERATE_SPLATS, degenerate_table_inputs
    H, W = 1080, 1920
    rows = np.array(DEGENERATE_SPLATS, dtype=np.float32)
    N = len(rows)
    ndc = np.zeros((1, 4, N), np.float32); ndc[0, 0] = rows[:, 0]; ndc[0, 1] = rows[:, 1]; ndc[0, 2] = 0.5; ndc[0, 3] = 1.0
    inv = np.zeros((1, 2, 2, N), np.float32); inv[0, 0, 0] = rows[:, 2]; inv[0, 0, 1] = inv[0, 1, 0] = rows[:, 3]; inv[0, 1, 1] = rows[:, 4]
    op = np.ascontiguousarray(rows[:, 5][None])
    vz = np.linspace(1.0, 2.0, N, dtype=np.float32)[None]
    lu, rd, al = oracle.get_allocate_size(ndc, vz, inv, op, H, W, 8, 16)
    assert al[0].tolist() == [54, 44, 103, 38, 46, 40, 33, 38, 55, 45, 43, 67, 74, 66]
    area = (rd[0, 0] - lu[0, 0]) * (rd[0, 1] - lu[0, 1])
    assert (al[0] <= area).all()                                          # (a count never exceeds the tile rectangle)
    for copies in (1, 7):
        ndc, inv, op, vz = degenerate_table_inputs(copies)
        _, _, al = oracle.get_allocate_size(ndc, vz, inv, op, H, W, 8, 16)
        dsi = np.argsort(vz, axis=-1, kind="stable").astype(np.int64)
        prefix = np.cumsum(np.take_along_axis(al, dsi, axis=-1), axis=-1, dtype=np.int64).astype(np.int32)
        ks, vs, ks_u, vs_u = oracle.create_table(ndc, inv, op, prefix, dsi, H, W, 8, 16)
        assert ks.shape[1] == int(prefix[0, -1]) and (ks > 0).all() and ks.max() <= (H // 8) * (W // 16)
        assert (np.diff(ks[0]) >= 0).all()
        assert np.array_equal(np.bincount(vs[0], minlength=al.shape[1]), al[0])      # every splat: exactly its count, no more, no less
        if ks_u is not None:                                                         # unsorted emission: splat j's entries sit in its own share
            off = np.concatenate([[0], prefix[0]])
            for slot in range(al.shape[1]):
                assert (vs_u[0, off[slot]:off[slot + 1]] == dsi[0, slot]).all()


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
