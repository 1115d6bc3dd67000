"""Training-level parity: the native executor, the operator-by-operator path and an independent dense torch+autograd formulation
(tests/torch_reference.py, pinned against the oracle on CPU) train the same perturbed student towards the same teacher renders and
must produce the same PSNR curve (tolerance: see the test).  The long run with density control lives in
tests/convergence.py -> profiles/r02_convergence.md."""
import pytest

from tests.util import noise_log

pytestmark = pytest.mark.gpu


@pytest.mark.stochastic
def test_three_paths_converge_together():
    import convergence as C
    out = C.run(n=2048, epochs=12, densify_epochs=0, log=lambda *_: None)
    d = C.deltas(out)
    assert out["teacher_render_agreement_db"] > 60.0, out["teacher_render_agreement_db"]
    for path in ("executor", "operator", "torch"):
        assert out[path]["psnr"][-1] > out[path]["psnr"][0] + 2.0, (path, out[path]["psnr"])
    # The pair executor ~ executor_again (the same path twice) measures the atomics-order noise of THIS scenario; the cross-path pairs are
    # held to a fixed bound that sits between that noise (observed over repeated runs: profiles/r05_noise_calibration.md) and what a
    # defect in one path produces (a wrong gradient row or a lost optimizer step separates the curves by more than 1 dB within 12
    # epochs).  Every measured value is logged (tests/util.py noise_log).
    for pair, v in d.items():
        noise_log(what=pair, smoothed=v["smoothed"], final5=v["final5"], bound=0.25)
    for pair, v in d.items():
        assert v["smoothed"] <= 0.25 and v["final5"] <= 0.25, (pair, v, {k: out[k]["psnr"] for k in ("executor", "operator", "torch")})


@pytest.mark.stochastic
def test_density_control_keeps_paths_together():
    import convergence as C
    import numpy as np
    out = C.run(n=2048, epochs=1, densify_epochs=18, densify_until=9, with_torch=False, log=lambda *_: None)   # densifies after epochs 4 and 8
    a, b = out["executor_densify"], out["operator_densify"]
    assert a["size"][-1] > a["size"][0], a["size"]                       # a densification happened
    # After a densification the two trajectories are chaotic copies of each other (float atomics reorder the sums): the long run in
    # profiles/r02_convergence.md shows up to 0.73 dB per epoch / 0.49 dB in a 5-epoch average between the two paths while both keep
    # climbing; the executor differs from ITSELF by 0.18 dB run to run at fixed topology.  Observed on the 5-epoch mean of this scenario:
    # 0.12 ... 0.35 dB (profiles/r05_noise_calibration.md, and 0.34 once in round 3); a path with a defect stops climbing and falls several dB
    # behind.  1.0 dB is the sanity bound.
    noise_log(what="executor_densify~operator_densify", final5=abs(np.mean(a["psnr"][-5:]) - np.mean(b["psnr"][-5:])), bound=1.0,
              size_rel=abs(a["size"][-1] - b["size"][-1]) / a["size"][-1])
    assert abs(np.mean(a["psnr"][-5:]) - np.mean(b["psnr"][-5:])) <= 1.0, (a["psnr"], b["psnr"])
    assert abs(a["size"][-1] - b["size"][-1]) <= 0.02 * a["size"][-1], (a["size"], b["size"])
