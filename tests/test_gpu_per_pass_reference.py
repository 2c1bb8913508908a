"""The per-pass rule on the GPU (update_rule = 1) against runs of the compiled, UNMODIFIED src/stochastic_ppm.h
(tests/golden/ref_pp_*.npz, made by oracle/ref_driver_pp.cpp: the file's create_point_loop and create_photon called as
they are, its update of :721-741 applied per pass).

The GPU runs in reference-stream mode (sampler = 1: the reference's sequential mt19937 streams with the fixture's
seeds) with that file's photon tracer (tracer_variant = 1), so it follows the same photons to the same hitpoints.
Identical, per hitpoint and pass: which pixels own a hitpoint, object ids, the deposit count M, hence N and r^2 (bit
for bit: they depend on the counts only).  To rounding: everything that passes through sin/cos/x^5 (detmath.h here,
glibc there) and tau, which the GPU sums in 2^-s fixed point where the reference adds doubles -- bounds written at each
check (the fixed-point sum alone is bounded in tests/test_oracle_vs_reference.py)."""
import numpy as np
import pytest

from raytracer_amd import sppm

pytestmark = pytest.mark.gpu


def _rel_rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


# _k8, _k100: the K cap of :425-427 binds (the accepted sets then depend on the tree's visiting order: sppm_cap.hpp)
@pytest.mark.parametrize("tag", ["64x48", "320x240", "64x48_k8", "320x240_k100"])
def test_gpu_reproduces_the_compiled_per_pass_file(golden, tag, tmp_path):
    g = golden("pp_" + tag)
    w, h, photons, passes, traced, k, se, sp, layout = [int(x) for x in g["meta"]]
    assert layout == 0
    st = int(g["stride"][0])
    cfg = sppm.make_config(w, h, gather_radius2=float(g["search_r2_pass0"][0]), gather_max=k, seed_eye=se, seed_photon=sp,
                           photons_per_pass=photons, sampler=1, update_rule=1, tracer_variant=1)
    ctx = sppm.SppmContext(cfg)
    nd = np.zeros(w * h)
    img = None
    for i in range(passes):
        ctx.eye_pass(i)
        ctx.check()
        s = ctx.hitpoints()
        v = s["valid"] > 0
        assert np.array_equal(v, g[f"hp_valid_pass{i}"] > 0), f"pass {i}: which pixels own a hitpoint"
        assert np.array_equal(s["obj"].astype(np.int8)[v], g[f"hp_obj_pass{i}"][v]), f"pass {i}: hitpoint object ids"
        vs = v[::st]
        np.testing.assert_allclose(s["hp_f"][::st][vs], g[f"hp_f_pass{i}"][vs], rtol=1e-12, atol=1e-12, err_msg=f"pass {i} hitpoint records")
        np.testing.assert_allclose(s["ld"][::st], g[f"ld_pass{i}"], rtol=1e-11, atol=1e-13, err_msg=f"pass {i} direct term")
        ctx.photon_pass(i, photons)
        dep = ctx.export_deposits().cpu().numpy()
        assert np.array_equal(dep[3][v], g[f"dep_m_pass{i}"][v].astype(np.int64)), f"pass {i}: deposit count M of every hitpoint"
        # the fixed-point sums against the reference's double sums of :452-454: M/2 units of the last place, plus the
        # last-bit differences of the photon flux itself (x^5 at glass)
        scale = ctx.fixed_point_scale()
        from raytracer_amd.scene import CORNELL_SPHERES
        f = (np.array([c[3] for c in CORNELL_SPHERES])[np.maximum(s["obj"], 0)] / np.pi)[::st]
        fixed = dep[:3].T[::st] / scale * f
        ref_sum = g[f"dep_f_pass{i}"]
        bound = (0.5 * dep[3][::st, None] / scale) * f + 1e-10 * np.abs(ref_sum)
        assert (np.abs(fixed - ref_sum) <= bound).all(), f"pass {i}: fixed-point sums"
        ctx.end_pass()
        ctx.check()
        s = ctx.hitpoints()
        assert np.array_equal(s["r2"][v].view(np.uint64), g[f"state_r2_pass{i}"][v].view(np.uint64)), f"pass {i}: r^2 of every hitpoint"
        nd = nd + 0.7 * dep[3] * v  # N' = N + alpha M (:728), the double the reference keeps
        assert np.array_equal(nd[v].view(np.uint64), g[f"state_n_pass{i}"][v].view(np.uint64)), f"pass {i}: N"
        assert np.array_equal(s["n"][v], g[f"state_n_pass{i}"][v].astype(np.int32)), f"pass {i}: N as the tap reports it"
        np.testing.assert_allclose(s["flux"][::st], g[f"tau_pass{i}"], rtol=2e-6, atol=1e-9, err_msg=f"pass {i} tau")
        img = ctx.resolve(i, photons).cpu().numpy()
        assert _rel_rmse(img[::st], g[f"image_pass{i}"]) < 1e-7, f"pass {i} image (north_star: <= 1e-3)"
    path = str(tmp_path / "out.hdr")
    sppm.save_hdr_file(path, img, w, h)
    mine = np.frombuffer(open(path, "rb").read(), np.uint8)
    ref = g["final_hdr"]
    assert mine.size == ref.size
    # RGBE has 8-bit mantissas: a relative difference of 1e-7 moves a byte only where a channel sits on a rounding step
    assert (mine != ref).sum() <= 4, f"{(mine != ref).sum()} bytes of the .hdr differ"
    ctx.close()
