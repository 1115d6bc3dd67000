"""The per-tile depth sort's network and its three list-length regimes, emulated on the host: litegs_amd/csrc/lg_tilesort_body.h is
written so that the same text compiles as device code (tilesort.hip) and as a sequential C++ program (tests/host/bitonic_check.cpp),
which compares it with std::stable_sort for 745 lists (every length 2..700, chunk edges up to 20 000, duplicate-heavy depths)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_tile_sort_body_matches_stable_sort_on_the_host(tmp_path):
    exe = str(tmp_path / "bitonic_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "litegs_amd", "csrc"), os.path.join(ROOT, "tests", "host", "bitonic_check.cpp"),
                    "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
