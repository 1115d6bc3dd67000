"""Guards on the compiled gfx950 code of the blend kernels (hipcc cross-compiles here: no GPU needed).

The blend kernels fetch every splat record with an asynchronous scalar load into 16 SGPRs (csrc/raster.hip rec_request) and tell the
compiler the block is defined at the request.  That holds only while the register allocator never moves such a block between request and
wait, which it does not as long as each of the two record variables of a loop keeps ONE register block.  Round 6 broke it once (two extra
masks alive across the requests -> four blocks, copies of in-flight records, a wrong image): the shape is asserted here so that the next
change that raises the scalar register pressure fails at build time, not on a GPU box."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "litegs_amd", "csrc")


@pytest.fixture(scope="module")
def raster_isa():
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "raster.s")
        cmd = [hipcc, "-S", "--cuda-device-only", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I", CSRC,
               "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, "raster.hip"), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return open(out).read()


def _kernels(isa):
    """{mangled name: body} of every kernel of the translation unit"""
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*;? ?@?\S*\n(.*?)\n\s*s_endpgm", isa, flags=re.S | re.M):
        out[m.group(1)] = m.group(2)
    return out


def test_record_loads_of_the_blend_kernels_keep_two_register_blocks(raster_isa):
    kernels = {k: v for k, v in _kernels(raster_isa).items() if re.match(r"_Z\d+raster_(forward|backward)", k)}
    assert len(kernels) >= 8, sorted(kernels)
    checked = 0
    for name, body in kernels.items():
        blocks = set(re.findall(r"s_load_dwordx16 (s\[\d+:\d+\])", body))
        if not blocks:
            continue                      # (the splat-parallel variant loads its records through the vector path)
        checked += 1
        # per LOOP (the compiler annotates every basic block with its loop header): two destination blocks, the ping-pong pair.  Blocks
        # outside loops -- the prologue's first request, a one-record tail loop -- are requested and waited for in place.
        loops = {}
        header = None
        lines = body.split("\n")
        for i, line in enumerate(lines):
            lab = re.match(r"^\.LBB(\d+_\d+):(.*)", line)
            if lab:
                note = lab.group(2)
                j = i + 1
                while j < len(lines) and re.match(r"^\s*;", lines[j]):         # the annotation may continue on comment lines
                    note += lines[j]; j += 1
                inloop = re.search(r"in Loop: Header=BB(\d+_\d+)", note)
                header = lab.group(1) if "Loop Header" in note else (inloop.group(1) if inloop else None)
            ld = re.search(r"s_load_dwordx16 (s\[\d+:\d+\])", line)
            if ld and header is not None:
                loops.setdefault(header, set()).add(ld.group(1))
        assert loops, f"{name}: no record load inside a loop?"
        for hdr, blk in loops.items():
            assert len(blk) <= 2, f"{name}: loop BB{hdr} loads records into {sorted(blk)} -- more than two blocks: the allocator will copy in-flight records"
        # no scalar move may read a register of a record block between a request and the next scalar wait
        for m in re.finditer(r"s_load_dwordx16 s\[(\d+):(\d+)\][^\n]*\n(.*?)s_waitcnt lgkmcnt\(0\)", body, flags=re.S):
            lo, hi = int(m.group(1)), int(m.group(2))
            window = re.split(r"^\.LBB", m.group(3), maxsplit=1, flags=re.M)[0]       # straight-line code behind the request only
            for mv in re.finditer(r"s_mov_b(?:32|64) s\[?(\d+)(?::(\d+))?\]?, s\[?(\d+)(?::(\d+))?\]?", window):
                src_lo = int(mv.group(3)); src_hi = int(mv.group(4) or mv.group(3))
                assert src_hi < lo or src_lo > hi, f"{name}: '{mv.group(0)}' copies a record register while its load is in flight"
    assert checked >= 6
