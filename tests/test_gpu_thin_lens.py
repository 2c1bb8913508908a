"""Thin-lens camera (sppm_config.camera_model = 1; DoFCamera::get_ray, src/cameras.h:66-139, the member
src/sppm2.h:73-79 keeps commented out) on the GPU, through the C ABI.

  * against the oracle (XorShift lens streams, detmath): rays and two SPPM passes bit for bit;
  * against the compiled reference (tests/golden/ref_64x48_dof.npz, reference-stream mode): the reference
    takes cosf/sinf of the lens angle in FLOAT from glibc, the GPU rounds detmath's double value to float;
    where the two floats differ by an ulp the ray moves by ~1e-7, so rays agree to 1e-6, nearly all of them
    bit for bit, and the image within north_star's 1e-3 rel-RMSE (asserted much tighter).
"""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import sppm
from raytracer_amd.scene import CORNELL_SPHERES

pytestmark = pytest.mark.gpu
LENS = dict(camera_model=1, aperture=4.0, focus_dist=180.0, seed_lens=3000)


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    ok = (a.view(np.uint64) == b.view(np.uint64)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


def test_thin_lens_matches_oracle_bit_for_bit():
    W, H, M = 64, 48, 20000
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, **LENS))
    par = ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M, **LENS)
    orc = ol.Oracle(par, ol.make_spheres(CORNELL_SPHERES))
    rays = ctx.camera_rays().cpu().numpy()
    _same(rays, ol.camera_rays(par), "DoFCamera::get_ray")
    assert np.abs(rays[:, :3] - np.array([50.0, 52.0, 220.0])).max() > 1.0, "ray origins lie on the lens, not at the pinhole"
    for i in range(2):
        ctx.eye_pass(i)
        orc.eye_pass(i)
        s, o = ctx.hitpoints(), orc.state()
        for k in ("valid", "obj", "hp_f", "ld"):
            _same(s[k], o[k], f"pass {i} eye {k}")
        ctx.photon_pass(i, M)
        orc.photon_pass(i, M)
        s, o = ctx.hitpoints(), orc.state()
        for k in ("n", "r2", "flux"):
            _same(s[k], o[k], f"pass {i} state {k}")
        _same(ctx.resolve(i, M).cpu().numpy(), orc.resolve(i, M), f"pass {i} image")
    ctx.check()
    ctx.close()


def test_thin_lens_renders_the_reference_image(golden):
    g = golden("64x48_dof")
    w, h, photons, passes, traced, k, se, sp = [int(x) for x in g["meta"]]
    st = int(g["stride"][0])
    lens = dict(camera_model=1, aperture=float(g["dof"][0]), focus_dist=float(g["dof"][1]), seed_lens=int(g["dof"][2]))
    ctx = sppm.SppmContext(sppm.make_config(w, h, gather_radius2=float(g["gather_r2"][0]), gather_max=k, seed_eye=se, seed_photon=sp,
                                            photons_per_pass=photons, sampler=1, **lens))
    rays = ctx.camera_rays().cpu().numpy()
    np.testing.assert_allclose(rays, g["dof_rays"], rtol=0, atol=1e-6, err_msg="DoFCamera::get_ray")
    exact = (rays.view(np.uint64) == np.ascontiguousarray(g["dof_rays"]).view(np.uint64)).all(axis=1).mean()
    assert exact > 0.8, f"only {exact:.3f} of the rays are bit-identical"
    for i in range(passes):
        ctx.eye_pass(i)
        ctx.check()
        s = ctx.hitpoints()
        v = s["valid"] > 0
        assert np.array_equal(v, g[f"hp_valid_pass{i}"] > 0), f"pass {i}: which pixels own a hitpoint"
        assert (s["obj"].astype(np.int8)[v] == g[f"hp_obj_pass{i}"][v]).mean() > 0.999, f"pass {i}: hitpoint object ids"
        ctx.photon_pass(i, photons)
        ctx.end_pass()
        ctx.check()
        s = ctx.hitpoints()
        same_n = (s["n"][v] == g[f"state_n_pass{i}"][v]).mean()
        assert same_n > 0.98, f"pass {i}: only {same_n:.4f} of the photon counts are identical"
        img = ctx.resolve(i, photons).cpu().numpy()
        err = float(np.sqrt(np.mean((img[::st] - g[f"image_pass{i}"]) ** 2)) / np.sqrt(np.mean(g[f"image_pass{i}"] ** 2)))
        assert err < 1e-4, f"pass {i} image rel-RMSE {err:.3g}"
    ctx.close()


def test_thin_lens_needs_sppm():
    from raytracer_amd import capi
    with pytest.raises(capi.SppmError):
        sppm.SppmContext(sppm.make_config(32, 24, algorithm=1, **LENS))
    with pytest.raises(capi.SppmError):
        sppm.SppmContext(sppm.make_config(32, 24, camera_model=1, aperture=4.0, focus_dist=0.0))
