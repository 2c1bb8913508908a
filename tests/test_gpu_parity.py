"""GPU parity: the HIP path (through the C ABI) against the oracle and the reference's golden vectors.

Bars (north_star): first-hit ids, photon-hit records and accepted (gathered) index sets bit-exact;
this build goes further -- every double is compared bit-for-bit, because the kernels and the oracle
(sampler = XorShift streams, math = detmath) perform the same IEEE operations in the same order.
"""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import sppm
from raytracer_amd.scene import CORNELL_SPHERES

pytestmark = pytest.mark.gpu


def _pair(w, h, photons, **kw):
    cfg = sppm.make_config(w, h, photons_per_pass=photons, **kw)
    okw = dict(kw)
    if "update_rule" in okw:
        okw["rule"] = okw.pop("update_rule")
    par = ol.make_params(w, h, sampler=1, math=1, photons_per_pass=photons, **okw)
    return sppm.SppmContext(cfg), ol.Oracle(par, ol.make_spheres(CORNELL_SPHERES))


def _same(a, b, what):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    if a.dtype.kind == "f":
        ok = a.view(np.uint64) == b.view(np.uint64)
    else:
        ok = a == b
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


def test_camera_rays_and_first_hits_match_reference(golden):
    """PinholeCamera::get_ray and intersect_scene against the compiled reference, bit for bit."""
    import torch
    g = golden("64x48")
    ctx = sppm.SppmContext(sppm.make_config(64, 48))
    _same(ctx.camera_rays().cpu().numpy(), g["cam_rays"], "camera rays")
    for tag in ("cam", "rnd"):
        ids, hit = ctx.intersect(torch.from_numpy(g[f"{tag}_rays"]))
        ids, hit = ids.cpu().numpy(), hit.cpu().numpy()
        _same(ids, g[f"{tag}_hit_id"], f"{tag} first-hit ids")
        m = ids >= 0
        _same(hit[m], g[f"{tag}_hit_f"][m], f"{tag} hit records")
        assert (hit[~m, 0] == 1e128).all()
    ctx.close()


def test_intersect_empty_batch():
    import torch
    ctx = sppm.SppmContext(sppm.make_config(8, 8))
    ids, hit = ctx.intersect(torch.zeros((0, 6), dtype=torch.float64))
    assert ids.numel() == 0 and hit.numel() == 0
    ctx.close()


def test_hdr_pixels_match_reference(golden):
    import torch
    g = golden("64x48")
    ctx = sppm.SppmContext(sppm.make_config(8, 8))
    x = g["encode_in"]
    keep = (x >= 0).all(axis=1)  # a negative component is undefined behaviour in the reference's uchar conversion
    out = ctx.hdr_pixels(torch.from_numpy(x)).cpu().numpy()
    _same(out[keep], g["encode_rgbe"][keep], "rgbe")
    ctx.close()


@pytest.mark.parametrize("r2,passes", [(25.0, 3), (5.0, 2)])
def test_passes_match_oracle_bit_for_bit(r2, passes):
    """64x48: eye pass, photon-hit records, accepted sets, per-hitpoint state and image, every pass."""
    W, H, M = 64, 48, 20000
    ctx, orc = _pair(W, H, M, gather_radius2=r2)
    for i in range(passes):
        ctx.eye_pass(i)
        orc.eye_pass(i)
        s, o = ctx.hitpoints(), orc.state()
        for k in ("valid", "obj", "hp_f", "ld"):
            _same(s[k], o[k], f"pass {i} eye {k}")
        # records of the whole pass in one batch
        ctx.trace(i, 0, M)
        key, f = ctx.download_records()
        tap = orc.photon_pass(i, M, tap=True)
        okey = (tap["rec_photon"].astype(np.uint64) << np.uint64(16)) | (tap["rec_depth"].astype(np.uint64) << np.uint64(1))
        flip = (tap["rec_f"][:, 3:6] != tap["rec_f"][:, 6:9]).any(axis=1)
        okey |= flip.astype(np.uint64)
        _same(key, okey, f"pass {i} record keys")
        _same(f[:, 0:6], tap["rec_f"][:, 0:6], f"pass {i} record position/normal")
        _same(f[:, 6:9], tap["rec_f"][:, 9:12], f"pass {i} record flux")
        ctx.accept_tap(4_000_000)
        ctx.gather()
        akey, apix = ctx.accepted()
        ctx.accept_tap(0)
        ok = okey[tap["acc_rec"]]
        order = np.lexsort((tap["acc_pix"], ok))
        _same(akey, ok[order], f"pass {i} accepted keys")
        _same(apix, tap["acc_pix"][order], f"pass {i} accepted image indices")
        s, o = ctx.hitpoints(), orc.state()
        for k in ("n", "r2", "flux"):
            _same(s[k], o[k], f"pass {i} state {k}")
        img = ctx.resolve(i, M).cpu().numpy()
        _same(img, orc.resolve(i, M), f"pass {i} image")
    ctx.check()
    c, oc = ctx.counters(), orc.counters()
    assert c["updates"] == oc["updates"] and c["rays"] == oc["rays"] and c["records"] == sum([0]) + c["records"]
    ctx.close()


def test_gather_kernels_and_slab_lengths_agree(monkeypatch):
    """The ordered gather is exact, so how records are bucketed must not matter: the simple
    per-hitpoint kernel, and the tile kernel at three slab lengths (64 forces many slabs, 1<<20 forces
    the >64-candidates key-range halving), all give the same bits."""
    W, H, M = 64, 48, 40000
    outs = []
    for env in (dict(SPPM_GATHER_TILES="0"), dict(SPPM_SLAB_PHOTONS="64"), dict(SPPM_SLAB_PHOTONS="3000"),
                dict(SPPM_SLAB_PHOTONS="1048576"), dict(SPPM_CELL_FACTOR="0.13"), dict(SPPM_CELL_FACTOR="3.0")):
        for k in ("SPPM_GATHER_TILES", "SPPM_SLAB_PHOTONS", "SPPM_CELL_FACTOR"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
        for i in range(2):
            ctx.eye_pass(i)
            ctx.photon_pass(i, M)
        ctx.check()
        outs.append(ctx.hitpoints())
        ctx.close()
    for o in outs[1:]:
        for k in ("n", "r2", "flux", "ld"):
            _same(outs[0][k], o[k], k)


def test_tile_ownership_partitions_the_work():
    """Two contexts that own alternate tiles, fed the same photons, together equal one context."""
    W, H, M = 64, 48, 20000
    full = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
    parts = [sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, tile_first=r, tile_stride=2)) for r in range(2)]
    for c in [full] + parts:
        for i in range(2):
            c.eye_pass(i)
            c.photon_pass(i, M)
        c.check()
    from raytracer_amd.parallel import owner_of_pixels
    own = owner_of_pixels(W, H, 2)
    ref = full.hitpoints()
    got = [c.hitpoints() for c in parts]
    for k in ("n", "r2", "flux"):
        merged = np.where(own.reshape((-1,) + (1,) * (ref[k].ndim - 1)) == 0, got[0][k], got[1][k])
        _same(merged, ref[k], k)
    for c in [full] + parts:
        c.close()


def test_config1_320x240_matches_oracle():
    """BASELINE config 1 geometry (320x240), two passes x 100k photons: state and image bit for bit."""
    W, H, M = 320, 240, 100000
    ctx, orc = _pair(W, H, M)
    for i in range(2):
        ctx.eye_pass(i); orc.eye_pass(i)
        ctx.photon_pass(i, M); orc.photon_pass(i, M)
    ctx.check()
    s, o = ctx.hitpoints(), orc.state()
    for k in ("valid", "obj", "hp_f", "ld", "n", "r2", "flux"):
        _same(s[k], o[k], k)
    _same(ctx.resolve(1, M).cpu().numpy(), orc.resolve(1, M), "image")
    assert ctx.counters()["updates"] == orc.counters()["updates"]
    ctx.close()


def test_per_pass_rule_matches_oracle():
    """update_rule 1 (src/stochastic_ppm.h:721-741)."""
    W, H, M = 64, 48, 20000
    ctx, orc = _pair(W, H, M, update_rule=1)
    for i in range(3):
        ctx.eye_pass(i); orc.eye_pass(i)
        ctx.photon_pass(i, M); orc.photon_pass(i, M)
        ctx.end_pass(); orc.end_pass()
        s, o = ctx.hitpoints(), orc.state()
        for k in ("n", "r2", "flux", "ld"):
            _same(s[k], o[k], f"pass {i} {k}")
        _same(ctx.resolve(i, M).cpu().numpy(), orc.resolve(i, M), f"pass {i} image")
    ctx.check()
    ctx.close()
