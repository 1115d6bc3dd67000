"""The executor's tile sort as an operator (`lg_tile_sort_ranges`): packed two-pass radix sort whose last pass leaves the tile range table
(csrc/binning.hip radix_onesweep_kernel PACK, tile_range_close_kernel) against numpy's stable sort and the oracle's tileRange
(GR/binning.cu:228-287), on key distributions the rendered tables of the other files do not produce: empty tiles in every position, one tile
only, a first tile > 0, single elements, lengths around the sort's 4096-key tiles, a device-side count below the buffer length, values at the
top of their bit range -- in both routes (`lg_set_tuning(26, 2)`: ranges from the sort; `(26, 0)`: key / value passes + range scan)."""
import ctypes
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(keys, vals, max_tile, oracle):
    order = np.argsort(keys, kind="stable")
    ranges = oracle.tile_range(keys[order][None].astype(np.int32), max_tile)[0]
    return vals[order], ranges


def _run(keys, vals, max_tile, value_bits, n_dev=None):
    from litegs_amd._lib import check, lib
    L = lib()
    n = len(keys)
    dev = "cuda"
    ka = torch.from_numpy(keys.astype(np.int32)).to(dev); va = torch.from_numpy(vals.astype(np.int32)).to(dev)
    kb = torch.empty_like(ka); vb = torch.empty_like(va)
    temp = torch.empty((int(L.lg_radix_sort_temp_bytes(max(n, 1))),), dtype=torch.uint8, device=dev)
    rng = torch.empty((max_tile + 2,), dtype=torch.int32, device=dev)
    nd = torch.tensor([n_dev], dtype=torch.int32, device=dev) if n_dev is not None else None
    done = ctypes.c_int(-1)
    check(L.lg_tile_sort_ranges(ka.data_ptr(), va.data_ptr(), kb.data_ptr(), vb.data_ptr(), n, nd.data_ptr() if nd is not None else None, max_tile,
                                value_bits, temp.data_ptr(), temp.numel(), rng.data_ptr(), ctypes.byref(done), torch.cuda.current_stream().cuda_stream),
          "lg_tile_sort_ranges")
    torch.cuda.synchronize()
    bits = max(1, int(max_tile).bit_length())
    passes = (bits + 7) // 8
    out = (va if passes % 2 == 0 else vb).cpu().numpy()
    return out, rng.cpu().numpy(), done.value


CASES = [
    # (name, n, max_tile, key generator(rng, n, max_tile))
    ("uniform_1080p", 300_000, 16200, lambda r, n, m: r.integers(1, m + 1, n)),
    ("with_key_zero", 50_000, 16200, lambda r, n, m: r.integers(0, m + 1, n)),
    ("even_tiles_only", 100_000, 16200, lambda r, n, m: 2 * r.integers(1, m // 2, n)),
    ("sparse_tiles", 20_000, 16200, lambda r, n, m: r.choice(r.integers(1, m + 1, 37), n)),
    ("one_tile", 9_000, 16200, lambda r, n, m: np.full(n, 4711)),
    ("last_tile_only", 5_000, 16200, lambda r, n, m: np.full(n, m)),
    ("first_and_last", 8_200, 16200, lambda r, n, m: np.where(r.random(n) < 0.5, 1, m)),
    ("single_element", 1, 16200, lambda r, n, m: np.array([300])),
    ("tile_boundary_4096", 4096, 16200, lambda r, n, m: r.integers(1, m + 1, n)),
    ("tile_boundary_4097", 4097, 16200, lambda r, n, m: r.integers(1, m + 1, n)),
    ("two_tiles_8193", 8193, 16200, lambda r, n, m: r.integers(1, m + 1, n)),
    ("skewed", 400_000, 16200, lambda r, n, m: np.minimum((r.exponential(400.0, n)).astype(np.int64) + 1, m)),
    ("small_grid_two_passes", 60_000, 299, lambda r, n, m: r.integers(1, m + 1, n)),
    ("big_grid_15_bits", 120_000, 32767, lambda r, n, m: r.integers(1, m + 1, n)),
]


@pytest.mark.parametrize("route", [2, 0], ids=["ranges_from_sort", "key_value_passes"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_tile_sort_and_ranges_match_stable_sort_and_tile_range(oracle, case, route):
    from litegs_amd._lib import check, lib
    name, n, max_tile, gen = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    keys = np.asarray(gen(rng, n, max_tile), dtype=np.int64)
    value_bits = 22
    vals = rng.integers(0, 1 << value_bits, n)
    vals[: min(n, 3)] = (1 << value_bits) - 1                       # the top of the value range
    check(lib().lg_set_tuning(26, route), "tuning 26")
    try:
        got_v, got_r, done = _run(keys, vals, max_tile, value_bits)
    finally:
        check(lib().lg_set_tuning(26, 2), "tuning 26")
    ref_v, ref_r = _reference(keys, vals, max_tile, oracle)
    passes = (max(1, int(max_tile).bit_length()) + 7) // 8
    assert done == (1 if (route == 2 and passes == 2) else 0), (done, route, passes)
    assert np.array_equal(got_v, ref_v), f"{name}: values"
    assert np.array_equal(got_r, ref_r), f"{name}: range table (first differing word {np.nonzero(got_r != ref_r)[0][:4]})"


def test_tile_sort_respects_the_device_side_count(oracle):
    """only the first min(n, *n_dev) pairs are a table (the executor's buffers are 1.5 x over-allocated): the rest is neither read nor written"""
    rng = np.random.default_rng(9)
    n, used, max_tile = 50_000, 31_337, 16200
    keys = rng.integers(1, max_tile + 1, n)
    keys[used:] = 0x7fffffff                                        # garbage behind the count must not matter
    vals = rng.integers(0, 1 << 22, n)
    got_v, got_r, done = _run(keys, vals, max_tile, 22, n_dev=used)
    ref_v, ref_r = _reference(keys[:used], vals[:used], max_tile, oracle)
    assert done == 1
    assert np.array_equal(got_v[:used], ref_v)
    assert np.array_equal(got_r, ref_r)


def test_empty_table_leaves_an_empty_range_table(oracle):
    got_v, got_r, done = _run(np.zeros(0, np.int64), np.zeros(0, np.int64), 16200, 22)
    assert done == 0 and (got_r == -1).all()
