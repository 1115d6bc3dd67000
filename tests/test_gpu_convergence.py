"""Training-level parity: the native executor, the operator-by-operator path and an independent dense torch+autograd formulation
(tests/torch_reference.py, pinned against the oracle on CPU) train the same perturbed student towards the same teacher renders and
must produce the same PSNR curve (tolerance 0.1 dB at every epoch).  The long run with density control lives in
tests/convergence.py -> profiles/r02_convergence.md."""
import pytest

pytestmark = pytest.mark.gpu


def test_three_paths_converge_together():
    import convergence as C
    out = C.run(n=2048, epochs=12, densify_epochs=0, log=lambda *_: None)
    d = C.deltas(out)
    assert out["teacher_render_agreement_db"] > 60.0, out["teacher_render_agreement_db"]
    for path in ("executor", "operator", "torch"):
        assert out[path]["psnr"][-1] > out[path]["psnr"][0] + 2.0, (path, out[path]["psnr"])
    for pair, v in d.items():
        assert v["smoothed"] <= 0.1 and v["final5"] <= 0.1, (pair, v, {k: out[k]["psnr"] for k in ("executor", "operator", "torch")})


def test_density_control_keeps_paths_together():
    import convergence as C
    import numpy as np
    out = C.run(n=2048, epochs=1, densify_epochs=18, densify_until=9, with_torch=False, log=lambda *_: None)   # densifies after epochs 4 and 8
    a, b = out["executor_densify"], out["operator_densify"]
    assert a["size"][-1] > a["size"][0], a["size"]                       # a densification happened
    # After a densification the two trajectories are chaotic copies of each other (float atomics reorder the sums): the long run in
    # profiles/r02_convergence.md shows up to 0.73 dB per epoch / 0.49 dB in a 5-epoch average between the two paths while both keep
    # climbing; the executor differs from ITSELF by 0.18 dB run to run at fixed topology.  0.75 dB on a 5-epoch mean is the sanity bound.
    assert abs(np.mean(a["psnr"][-5:]) - np.mean(b["psnr"][-5:])) <= 0.75, (a["psnr"], b["psnr"])
    assert abs(a["size"][-1] - b["size"][-1]) <= 0.02 * a["size"][-1], (a["size"], b["size"])
