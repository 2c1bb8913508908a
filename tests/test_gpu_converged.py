"""Which update rule renders the better image?  Both rules at equal pass and photon counts against a long render.

tests/golden/converged_320x240_rule0.npz is 65 536 passes x 200k photons of the per-photon rule (the reference's live
src/sppm2.h:391-421, whose image is pinned against the compiled reference), made on an MI355X by
tools/render_converged.py with seeds of its own; converged_320x240_rule1.npz is the same length of the per-pass rule
(src/stochastic_ppm.h:721-741).  Each is within 0.003 rel-RMSE of its own image at half the length and the two differ
by 0.0016: the rules converge to the same image.  The per-pass rule carries bench.py's primary line, so it must be no
worse than the per-photon rule against the per-photon rule's own long render: its error may exceed the other's by at
most 10 % after 16 and after 256 passes."""
import os

import numpy as np
import pytest

from raytracer_amd import sppm

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def errors_vs_converged(pass_counts=(16, 256), verbose=False):
    """{rule: {passes: rel-RMSE against the long per-photon-rule render}} (also used by bench.py)"""
    ref = {r: np.load(os.path.join(GOLDEN, f"converged_320x240_rule{r}.npz")) for r in (0, 1)}
    w, h, photons = [int(x) for x in ref[0]["meta"][:3]]
    out = {}
    for rule in (0, 1):
        ctx = sppm.SppmContext(sppm.make_config(w, h, photons_per_pass=photons, update_rule=rule))
        out[rule] = {}
        for i in range(max(pass_counts)):
            ctx.eye_pass(i)
            ctx.photon_pass(i, photons)
            ctx.end_pass()
            if i + 1 in pass_counts:
                img = ctx.resolve(i, photons).cpu().numpy()
                out[rule][i + 1] = rel_rmse(img, ref[0]["image"].astype(np.float64))
                if verbose:
                    print(f"rule {rule} after {i + 1} passes: {out[rule][i + 1]:.5f} against the per-photon rule's long render, "
                          f"{rel_rmse(img, ref[1]['image'].astype(np.float64)):.5f} against the per-pass rule's")
        ctx.check()
        ctx.close()
    return out


def test_the_long_renders_of_the_two_rules_agree():
    a = np.load(os.path.join(GOLDEN, "converged_320x240_rule0.npz"))
    b = np.load(os.path.join(GOLDEN, "converged_320x240_rule1.npz"))
    assert rel_rmse(b["image"].astype(np.float64), a["image"].astype(np.float64)) < 2e-3
    assert float(a["half_vs_full"][0]) < 4e-3 and float(b["half_vs_full"][0]) < 4e-3


def test_per_pass_rule_is_no_worse_than_the_per_photon_rule():
    e = errors_vs_converged(verbose=True)
    for n in (16, 256):
        assert e[1][n] <= 1.1 * e[0][n], f"after {n} passes: per-pass rule {e[1][n]:.5f}, per-photon rule {e[0][n]:.5f}"
