"""Classic progressive photon mapping (sppm_config.algorithm = 1; src/progressive_photonmapping.h, BASELINE
config 4) on the GPU, through the C ABI.

  * against the oracle (XorShift streams, detmath): everything bit for bit -- the hitpoint list, the
    photon-hit records, the accepted (record, hitpoint) pairs ("bit-exact gather index check"), radii,
    counts, flux, the image;
  * in reference-stream mode against the fixtures made by the compiled, unmodified reference
    (tests/golden/ref_ppm_*.npz): discrete state identical, values to rounding (glibc vs detmath sin/cos/x^5),
    the .hdr file byte for byte;
  * at 1080p through size-independent properties.
"""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import sppm
from raytracer_amd.scene import CORNELL_SPHERES

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} {b.shape}"
    ok = (a.view(np.uint64) == b.view(np.uint64)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


def _rel_rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def _pair(w, h, photons, ss, r2, samples=1):
    cfg = sppm.make_config(w, h, samples=samples, supersamples=ss, photons_per_pass=photons, gather_radius2=r2, initial_radius2=r2, algorithm=1)
    par = ol.make_params(w, h, samples=samples, supersamples=ss, sampler=1, math=1, photons_per_pass=photons, gather_radius2=r2, initial_radius2=r2)
    return sppm.SppmContext(cfg), ol.PpmOracle(par, ol.make_spheres(CORNELL_SPHERES))


def test_classic_ppm_repeated_samples():
    """samples > 1 repeats one ray per sub-pixel (src/progressive_photonmapping.h:774-788; src/main.cpp:74 runs 100): where no
    draw steers the path the hitpoints coincide, each with a state of its own, and the image divides by spp."""
    w, h, photons = 40, 30, 15000
    ctx, orc = _pair(w, h, photons, 1, 25.0, samples=3)
    assert ctx.n_hitpoints == orc.eye_pass() and ctx.n_hitpoints >= 3 * w * h * 0.9
    ctx.eye_pass(0)
    ctx.photon_pass(0, photons)
    orc.photon_pass(0, photons)
    ctx.check()
    s, o = ctx.hitpoints(), orc.state()
    _same(s["hp_f"][:, :8], o["hp_f"][:, :8], "hitpoint list")
    dup = (o["hp_f"][1:, 1:4] == o["hp_f"][:-1, 1:4]).all(axis=1).mean()
    assert dup > 0.5, "most consecutive hitpoints coincide"
    for k in ("n", "r2", "flux"):
        _same(s[k], o[k], f"state {k}")
    _same(ctx.resolve(0, photons).cpu().numpy(), orc.resolve(photons, order=1), "image")
    ctx.close()


@pytest.mark.parametrize("w,h,ss,r2,photons", [(64, 48, 1, 25.0, 20000), (48, 36, 2, 5.0, 30000)])
def test_classic_ppm_matches_oracle_bit_for_bit(w, h, ss, r2, photons, monkeypatch):
    ctx, orc = _pair(w, h, photons, ss, r2)
    nh = orc.eye_pass()
    assert ctx.n_hitpoints == nh, "length of the hitpoint list (known from sppm_create on)"
    ctx.eye_pass(0)
    s, o = ctx.hitpoints(), orc.state()
    _same(ctx.hitpoint_pixels(), o["pix"], "image index of every hitpoint, creation order")
    _same(s["obj"], o["obj"], "hitpoint object ids")
    _same(s["hp_f"][:, :8], o["hp_f"][:, :8], "hitpoint records: distance, position, normal, weight")
    assert (s["valid"] == 1).all()
    # the photons in one batch: records, then the accepted pairs of the gather
    ctx.trace(0, 0, photons)
    key, f = ctx.download_records()
    tap = orc.photon_pass(0, photons, tap=True, acc_cap=60_000_000)
    okey = (tap["rec_photon"].astype(np.uint64) << np.uint64(sppm.KEY_SHIFT)) | (tap["rec_depth"].astype(np.uint64) << np.uint64(1))
    okey |= (tap["rec_f"][:, 3:6] != tap["rec_f"][:, 6:9]).any(axis=1).astype(np.uint64)
    assert tap["rec_depth"].min() == 1, "photons deposit from their first hit on"
    _same(key, okey, "record keys")
    _same(f[:, 0:6], tap["rec_f"][:, 0:6], "record position / normal")
    _same(f[:, 6:9], tap["rec_f"][:, 9:12], "record flux = multiply(obj.color, current_flux)")
    ctx.accept_tap(60_000_000)
    ctx.gather()
    akey, ahp = ctx.accepted()
    ctx.accept_tap(0)
    ok = okey[tap["acc_rec"]]
    order = np.lexsort((tap["acc_pix"], ok))
    _same(akey, ok[order], "accepted pairs: record keys")
    _same(ahp, tap["acc_pix"][order], "accepted pairs: hitpoint ids")
    s, o = ctx.hitpoints(), orc.state()
    for k in ("n", "r2", "flux"):
        _same(s[k], o[k], f"state {k}")
    img = ctx.resolve(0, photons).cpu().numpy()
    _same(img, orc.resolve(photons, order=1), "image (per-pixel sums in creation order)")
    # the reference's own summation order (its kd-sorted vector) differs by the rounding of a few-term sum
    ref_order = orc.resolve(photons, order=0)
    assert np.abs(img - ref_order).max() <= 4e-16 * np.abs(ref_order).max()
    ctx.check()
    assert ctx.counters()["updates"] == orc.counters()["updates"]
    ctx.close()
    # the same run in small batches through sppm_photon_pass (pipelined tracer / binning / gather), split over
    # two calls, and once more with the simple per-hitpoint gather: identical state
    for env in ({"SPPM_BATCH_PHOTONS": "4096"}, {"SPPM_GATHER_TILES": "0", "SPPM_BATCH_PHOTONS": "8192"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c2 = sppm.SppmContext(sppm.make_config(w, h, supersamples=ss, photons_per_pass=photons, gather_radius2=r2, initial_radius2=r2, algorithm=1))
        c2.eye_pass(0)
        c2.photon_pass(0, photons // 3)
        c2.photon_pass(0, photons - photons // 3, first=photons // 3)
        c2.check()
        s2 = c2.hitpoints()
        for k in ("n", "r2", "flux"):
            _same(s2[k], o[k], f"{env}: state {k}")
        _same(c2.resolve(0, photons).cpu().numpy(), img, f"{env}: image")
        c2.close()
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("tag", ["64x48", "48x36_ss2_r5", "32x24_s2_k6"])
def test_classic_ppm_renders_the_reference_image(golden, tag, tmp_path):
    """(32x24_s2_k6: two identical samples per pixel and K = 6 -- duplicate hitpoints, equal keys in the tree, equal
    distances in the heap, the cap binding: which twin the cap drops is the reference's choice)"""
    g = golden("ppm_" + tag)
    w, h, photons, traced, k, se, sp, ss = [int(x) for x in g["meta"]]
    r2 = float(g["gather_r2"][0])
    cfg = sppm.make_config(w, h, samples=int(g["samples"][0]) if "samples" in g else 1, supersamples=ss, gather_radius2=r2, initial_radius2=r2, gather_max=k, seed_eye=se, seed_photon=sp,
                           photons_per_pass=photons, sampler=1, algorithm=1)
    ctx = sppm.SppmContext(cfg)
    assert ctx.n_hitpoints == len(g["hp_pix"]), "length of the hitpoint list"
    ctx.eye_pass(0)
    ctx.check()
    s = ctx.hitpoints()
    assert np.array_equal(ctx.hitpoint_pixels(), g["hp_pix"]), "image index of every hitpoint"
    assert np.array_equal(s["obj"].astype(np.int8), g["hp_obj"]), "hitpoint object ids"
    # distance, position, normal, weight: paths through glass carry x^5 in the weight
    np.testing.assert_allclose(s["hp_f"][:, :8], g["hp_f"][:, :8], rtol=1e-12, atol=1e-12, err_msg="hitpoint records")
    exact = (s["hp_f"][:, :8].view(np.uint64) == np.ascontiguousarray(g["hp_f"][:, :8]).view(np.uint64)).mean()
    assert exact > 0.9, f"only {exact:.3f} of the hitpoint record words are bit-identical"
    ctx.photon_pass(0, photons)
    ctx.check()
    s = ctx.hitpoints()
    assert np.array_equal(s["n"], g["state_n"]), "photon counts per hitpoint"
    assert np.array_equal(s["r2"].view(np.uint64), np.ascontiguousarray(g["state_f"][:, 0]).view(np.uint64)), "radii (a function of the counts)"
    np.testing.assert_allclose(s["flux"], g["state_f"][:, 1:], rtol=1e-10, atol=1e-12, err_msg="accumulated flux")
    img = ctx.resolve(0, photons).cpu().numpy()
    assert _rel_rmse(img, g["image"]) < 1e-12, "image"
    path = str(tmp_path / "out.hdr")
    sppm.save_hdr_file(path, img, w, h)
    mine = np.frombuffer(open(path, "rb").read(), np.uint8)
    ref = g["final_hdr"]
    assert mine.size == ref.size
    # last-bit differences may cross an RGBE rounding step in a handful of bytes; never more than one step
    diff = mine != ref
    assert diff.sum() <= 4 and (np.abs(mine.astype(int) - ref.astype(int))[diff] <= 1).all(), f"{diff.sum()} bytes of the .hdr differ"
    ctx.close()


def test_classic_ppm_mirror_of_render(tmp_path):
    """progressive_render, the mirror of photonmap::progressive::render, writes the pair of files."""
    name = str(tmp_path / "image_ppm")
    img = sppm.progressive_render(name, 64, 48, 1, 1, 20000, 25.0, 100000)
    orc = ol.PpmOracle(ol.make_params(64, 48, sampler=1, math=1, photons_per_pass=20000), ol.make_spheres(CORNELL_SPHERES))
    orc.eye_pass()
    orc.photon_pass(0, 20000)
    _same(img, orc.resolve(20000, order=1), "image")
    assert open(name + ".hdr", "rb").read() == ol.hdr_file(img, 64, 48)
    assert open(name + ".ppm", "rb").read() == ol.ppm_file(img, 64, 48)


def test_classic_ppm_rejects_what_it_does_not_do():
    from raytracer_amd import capi
    with pytest.raises(capi.SppmError):
        sppm.SppmContext(sppm.make_config(32, 24, algorithm=1, update_rule=1))
    with pytest.raises(capi.SppmError):
        sppm.SppmContext(sppm.make_config(32, 24, algorithm=1, tile_first=0, tile_stride=2))
    ctx = sppm.SppmContext(sppm.make_config(32, 24, algorithm=1, photons_per_pass=1000))
    with pytest.raises(capi.SppmError):
        ctx.eye_pass(1)
    ctx.close()


def test_classic_ppm_full_size_invariants():
    """BASELINE config 4 geometry (1080p), 2M photons, where the oracle would take many minutes:
      * r^2 is a function of the photon count alone (src/progressive_photonmapping.h:369-380), bit for bit;
      * sum of N over the hitpoint list == accepted updates counted by the kernel;
      * every pixel's hitpoints lie in consecutive slots, in render order;
      * a second run gives identical bits."""
    W, H, M = 1920, 1080, 2_000_000
    outs = []
    for rep_ in range(2):
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, gather_radius2=5.0, initial_radius2=5.0, algorithm=1))
        ctx.eye_pass(0)
        ctx.photon_pass(0, M)
        ctx.check()
        s = ctx.hitpoints()
        img = ctx.resolve(0, M).cpu().numpy()
        outs.append((s, img, ctx.counters(), ctx.hitpoint_pixels()))
        ctx.close()
    s, img, cnt, pix = outs[0]
    assert len(pix) >= W * H * 0.9 and (s["valid"] == 1).all()
    assert int(s["n"].sum()) == cnt["updates"]
    t = (H - 1 - pix // W) * W + pix % W   # place in the render order
    assert (np.diff(t) >= 0).all()
    nmax = int(s["n"].max())
    alpha, r2 = 0.7, 5.0
    table = np.empty(nmax + 1)
    table[0] = r2
    for k in range(nmax):
        g = 1.0 if k == 0 else (k * alpha + alpha) / (k * alpha + 1.0)
        r2 = r2 * g
        table[k + 1] = r2
    _same(s["r2"], table[s["n"]], "r2 as a function of N")
    assert np.isfinite(img).all() and (img >= 0).all()
    for k in ("n", "r2", "flux"):
        _same(outs[1][0][k], s[k], f"run-to-run {k}")
    _same(outs[1][1], img, "run-to-run image")


def test_classic_ppm_three_ranks_equal_one_context():
    """The hitpoint list shared out over three 'ranks' (contexts on one GPU) by ranges of its Hilbert order, records
    exchanged by hand (pack / append), owned state summed: equal to one context that did everything, bit for bit."""
    import torch
    W, H, M, N = 64, 48, 30000, 3
    kw = dict(supersamples=2, photons_per_pass=M, gather_radius2=5.0, initial_radius2=5.0, algorithm=1)
    full = sppm.SppmContext(sppm.make_config(W, H, **kw))
    ranks = [sppm.SppmContext(sppm.make_config(W, H, tile_first=r, tile_stride=N, ownership=1, **kw)) for r in range(N)]
    rows = [torch.empty((8 * M, 10), dtype=torch.float64, device="cuda") for _ in range(N)]
    full.eye_pass(0)
    full.photon_pass(0, M)
    for c in ranks:
        c.eye_pass(0)
    for lo in range(0, M, M // 2):        # two batches
        shard = M // 2 // N
        cnt = []
        for r, c in enumerate(ranks):
            c.trace(0, lo + r * shard, shard)
            cnt.append(c.pack_records(rows[r]))
        for c in ranks:
            for r in range(N):
                c.append_records(rows[r], cnt[r], lo + r * shard, lo + (r + 1) * shard)
            c.gather()
    masks = [c.owned_mask().cpu().numpy().astype(bool) for c in ranks]
    assert (np.sum(masks, axis=0) == 1).all() and len(masks[0]) == full.n_hitpoints
    state = sum(c.export_owned_state() for c in ranks)
    ref = full.hitpoints()
    st = state.cpu().numpy()
    _same(st[:, 0], ref["r2"], "r2 summed over ranks")
    _same(st[:, 1:4], ref["flux"], "flux summed over ranks")
    _same(st[:, 4].astype(np.int32), ref["n"], "N summed over ranks")
    ref_img = full.resolve(0, M).cpu().numpy()
    for c in ranks:
        c.import_state(state)
        _same(c.resolve(0, M).cpu().numpy(), ref_img, "image on a rank")
        c.check()
    full.check()
    for c in [full] + ranks:
        c.close()
