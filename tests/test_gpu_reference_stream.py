"""GPU path in reference-stream mode (sppm_config.sampler = 1) against the golden vectors made by the
compiled, UNMODIFIED reference (tests/golden/make_golden.py): same seeds, same sequential mt19937
streams, so the GPU renders the reference's own image.

What can differ: the GPU takes sin/cos/x^5 from raytracer_amd/csrc/detmath.h, the reference from
glibc (last-bit differences), so values that pass through them agree to rounding, not bit for bit;
everything discrete (which pixels own a hitpoint, object ids, photon counts) and everything that depends
only on the discrete state (the radii) must be identical.  north_star's bound is rel-RMSE <= 1e-3 on
the image; the bounds asserted here are much tighter and written next to each check.
"""
import numpy as np
import pytest

from raytracer_amd import sppm

pytestmark = pytest.mark.gpu


def _rel_rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


@pytest.mark.parametrize("tag", ["64x48", "64x48_r5", "320x240", "64x48_spp"])
def test_gpu_renders_the_reference_image(golden, tag, tmp_path):
    g = golden(tag)
    w, h, photons, passes, traced, k, se, sp = [int(x) for x in g["meta"]]
    st = int(g["stride"][0])
    spp = dict(samples=int(g["spp"][0]), supersamples=int(g["spp"][1])) if "spp" in g else {}   # several eye paths per pixel
    cfg = sppm.make_config(w, h, gather_radius2=float(g["gather_r2"][0]), gather_max=k, seed_eye=se, seed_photon=sp,
                           photons_per_pass=photons, sampler=1, **spp)
    ctx = sppm.SppmContext(cfg)
    img = None
    for i in range(passes):
        ctx.eye_pass(i)
        ctx.check()
        s = ctx.hitpoints()
        v = s["valid"] > 0
        assert np.array_equal(v, g[f"hp_valid_pass{i}"] > 0), f"pass {i}: which pixels own a hitpoint"
        assert np.array_equal(s["obj"].astype(np.int8)[v], g[f"hp_obj_pass{i}"][v]), f"pass {i}: hitpoint object ids"
        vs = v[::st]
        ref_hp = g[f"hp_f_pass{i}"][:, :10]
        # hitpoint records: distance, position, normal, weight.  Paths through glass carry x^5 in the weight.
        np.testing.assert_allclose(s["hp_f"][::st][vs], ref_hp[vs], rtol=1e-12, atol=1e-12, err_msg=f"pass {i} hitpoint records")
        np.testing.assert_allclose(s["ld"][::st], g[f"ld_pass{i}"], rtol=1e-11, atol=1e-13, err_msg=f"pass {i} direct term")
        exact = (s["hp_f"][::st][vs].view(np.uint64) == ref_hp[vs].view(np.uint64)).mean()
        assert exact > 0.95, f"pass {i}: only {exact:.3f} of the hitpoint record words are bit-identical"
        ctx.photon_pass(i, photons)
        ctx.end_pass()
        ctx.check()
        s = ctx.hitpoints()
        assert np.array_equal(s["n"][v], g[f"state_n_pass{i}"][v]), f"pass {i}: photon counts per hitpoint"
        ref_state = g[f"state_f_pass{i}"]
        assert np.array_equal(s["r2"][::st][vs].view(np.uint64), ref_state[:, 0][vs].view(np.uint64)), f"pass {i}: radii (a function of the counts)"
        np.testing.assert_allclose(s["flux"][::st][vs], ref_state[:, 1:][vs], rtol=1e-10, atol=1e-12, err_msg=f"pass {i} flux")
        img = ctx.resolve(i, photons).cpu().numpy()
        assert _rel_rmse(img[::st], g[f"image_pass{i}"]) < 1e-12, f"pass {i} image"
    # the file the reference wrote, byte for byte (last-bit differences stay below the RGBE rounding steps on these fixtures)
    path = str(tmp_path / "out.hdr")
    sppm.save_hdr_file(path, img, w, h)
    mine = np.frombuffer(open(path, "rb").read(), np.uint8)
    ref = g["final_hdr"]
    assert mine.size == ref.size
    assert (mine != ref).sum() == 0, f"{(mine != ref).sum()} bytes of the .hdr differ"
    ctx.close()
