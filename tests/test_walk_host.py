"""The key emission's two tile walks, compared on the host: litegs_amd/csrc/lg_tilewalk.h compiles as sequential C++
(tests/host/walk_check.cpp).  The projection counts a splat's tiles with the serial AccuTile walk (the reference's, GR/binning.cu:310-373),
dup_small_kernel emits with the same walk, dup_big_kernel evaluates it one slice per lane (slice_bounds): every entry of the table is
written only if the two always agree, also for the degenerate conics a trained cloud contains (nearly singular, ten decades of scale,
centres on the frustum limit, opacity at the 1/255 threshold).  4 million splats here; 150 million were run once for DESIGN.md section 9."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_sliced_walk_counts_the_tiles_the_serial_walk_counts(tmp_path):
    exe = str(tmp_path / "walk_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "litegs_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "walk_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe, "4000000"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.rstrip().endswith(" 0 mismatches"), out.stdout[-2000:] + out.stderr
