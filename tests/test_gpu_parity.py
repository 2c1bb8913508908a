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
    n_records = 0
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
        n_records += len(tap["rec_photon"])
        okey = (tap["rec_photon"].astype(np.uint64) << np.uint64(sppm.KEY_SHIFT)) | (tap["rec_depth"].astype(np.uint64) << np.uint64(1))
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
    assert c["updates"] == oc["updates"] and c["rays"] == oc["rays"] and c["records"] == n_records
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


def test_record_exchange_equals_local_tracing():
    """The all-gather path of a 2-rank run, emulated with two contexts on one GPU: each 'rank' traces
    half of the photons, the packed records are exchanged through sppm_pack_records /
    sppm_append_records, every rank gathers for its own tiles.  Merged by owner the result equals a
    single context that did everything itself, bit for bit."""
    import torch
    from raytracer_amd.parallel import owner_of_pixels
    W, H, M = 64, 48, 20000
    full = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
    ranks = [sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, tile_first=r, tile_stride=2)) for r in range(2)]
    rows = [torch.empty((8 * M, 10), dtype=torch.float64, device="cuda") for _ in range(2)]
    half = M // 2
    for i in range(2):
        full.eye_pass(i)
        full.photon_pass(i, M)
        cnt = []
        for r, c in enumerate(ranks):
            c.eye_pass(i)
            c.trace(i, r * half, half)
            cnt.append(c.pack_records(rows[r]))
        assert sum(cnt) > 0
        for c in ranks:
            for r in range(2):
                c.append_records(rows[r], cnt[r], r * half, (r + 1) * half)
            c.gather()
    for c in [full] + ranks:
        c.check()
    own = owner_of_pixels(W, H, 2)
    ref = full.hitpoints()
    got = [c.hitpoints() for c in ranks]
    for k in ("n", "r2", "flux"):
        merged = np.where(own.reshape((-1,) + (1,) * (ref[k].ndim - 1)) == 0, got[0][k], got[1][k])
        _same(merged, ref[k], k)
    for c in [full] + ranks:
        c.close()


def test_spatial_ownership_three_ranks_equal_one_context():
    """ownership = 1 (every rank keeps a range of the Hilbert order of the first eye pass's hitpoints), three 'ranks'
    emulated on one GPU with the all-gather flow: the masks partition the pixels, and merged by mask the hitpoint
    state and the image equal a single context that did everything itself, bit for bit.  The binning of these
    ranks drops the records far from their region; nothing near it may be lost."""
    import torch
    W, H, M, N = 96, 64, 30000, 3
    full = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
    ranks = [sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, tile_first=r, tile_stride=N, ownership=1)) for r in range(N)]
    rows = [torch.empty((8 * M, 10), dtype=torch.float64, device="cuda") for _ in range(N)]
    shard = M // N
    for i in range(3):
        full.eye_pass(i)
        full.photon_pass(i, M)
        cnt = []
        for r, c in enumerate(ranks):
            c.eye_pass(i)
            c.trace(i, r * shard, shard)
            cnt.append(c.pack_records(rows[r]))
        for c in ranks:
            for r in range(N):
                c.append_records(rows[r], cnt[r], r * shard, (r + 1) * shard)
            c.gather()
    for c in [full] + ranks:
        c.check()
    masks = [c.owned_mask().cpu().numpy().astype(bool) for c in ranks]
    assert (np.sum(masks, axis=0) == 1).all(), "the ranks' masks must partition the pixels"
    sizes = [int(m.sum()) for m in masks]
    assert max(sizes) - min(sizes) <= 1
    ref = full.hitpoints()
    got = [c.hitpoints() for c in ranks]
    for k in ("n", "r2", "flux"):
        merged = np.zeros_like(ref[k])
        for m, g in zip(masks, got):
            merged[m] = g[k][m]
        _same(merged, ref[k], k)
    img = sum(c.resolve(2, M).cpu().numpy() * m[:, None] for c, m in zip(ranks, masks))
    _same(img, full.resolve(2, M).cpu().numpy(), "image summed over ranks")
    for c in [full] + ranks:
        c.close()


def test_pack_reports_a_short_row_buffer():
    import torch
    from raytracer_amd import capi
    ctx = sppm.SppmContext(sppm.make_config(64, 48, photons_per_pass=4096))
    ctx.eye_pass(0)
    ctx.trace(0, 0, 4096)
    small = torch.empty((8, 10), dtype=torch.float64, device="cuda")
    with pytest.raises(capi.SppmError):
        ctx.pack_records(small)
    big = torch.empty((40000, 10), dtype=torch.float64, device="cuda")
    n = ctx.pack_records(big)          # nothing was lost by the failed call
    key, f = ctx.download_records()
    assert n > 0 and len(key) == 0     # packed -> the buffer is empty now
    ctx.close()


@pytest.mark.parametrize("wh", [(1920, 1080)])
def test_full_size_invariants(wh):
    """BASELINE config 2 geometry (1080p, 1M photons per pass), where the oracle would take minutes:
    size-independent properties of the reference's update rule.
      * r^2 is a function of the photon count alone: r2 = r0^2 * prod_{k=1}^{N-1} (k a + a)/(k a + 1)
        evaluated in the reference's order (src/sppm2.h:398-407) -- checked bit for bit;
      * sum of N over hitpoints == accepted updates counted by the kernel;
      * a second run gives identical bits (no atomics on hitpoint data, order fixed by keys);
      * the image is finite and non-negative."""
    W, H = wh
    M = 1_000_000
    outs = []
    for rep_ in range(2):
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
        for i in range(2):
            ctx.eye_pass(i)
            ctx.photon_pass(i, M)
        ctx.check()
        s = ctx.hitpoints()
        img = ctx.resolve(1, M).cpu().numpy()
        cnt = ctx.counters()
        ctx.close()
        outs.append((s, img, cnt))
    s, img, cnt = outs[0]
    v = s["valid"] > 0
    assert int(s["n"][v].sum()) == cnt["updates"]
    nmax = int(s["n"].max())
    alpha = 0.7
    table = np.empty(nmax + 1)
    r2 = 25.0
    table[0] = r2
    for k in range(nmax):       # state after k+1 accepted photons
        g = 1.0 if k == 0 else (k * alpha + alpha) / (k * alpha + 1.0)
        r2 = r2 * g
        table[k + 1] = r2
    _same(s["r2"][v], table[s["n"][v]], "r2 as a function of N")
    assert np.isfinite(img).all() and (img >= 0).all()
    for k in ("n", "r2", "flux", "ld"):
        _same(outs[1][0][k], s[k], f"run-to-run {k}")
    _same(outs[1][1], img, "run-to-run image")


def test_k_cap_check_counts_every_querys_candidates():
    """check_gather_max: the largest number of hitpoints that pass the search filter of src/kdtree.h:123-127 for any
    photon hit, against a brute-force count in numpy; sppm_check reports when the reference's K cap would have acted."""
    from raytracer_amd import capi
    W, H, M = 64, 48, 4000
    par = ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M)
    orc = ol.Oracle(par, ol.make_spheres(CORNELL_SPHERES))
    orc.eye_pass(0)
    tap = orc.photon_pass(0, M, tap=True)
    o = orc.state()
    hp = o["hp_f"][o["valid"] > 0, 1:4]
    q, on = tap["rec_f"][:, 0:3], tap["rec_f"][:, 6:9]
    best = 0
    for a in range(0, len(q), 512):
        d = hp[None, :, :] - q[a:a + 512, None, :]
        d2 = (d * d).sum(-1)
        with np.errstate(invalid="ignore", divide="ignore"):
            dt = np.abs((on[a:a + 512, None, :] * (d / np.sqrt(d2)[..., None])).sum(-1))
        best = max(best, int(((d2 < 25.0) & (dt <= 0.25)).sum(1).max()))
    assert best > 4
    for k, ok in ((best, True), (best - 1, False)):
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, gather_max=k, check_gather_max=1))
        ctx.eye_pass(0)
        ctx.photon_pass(0, M)
        if ok:
            ctx.check()
            assert ctx.counters()["max_query_candidates"] == best
            s = ctx.hitpoints()
            orc_s = orc.state()
            for name in ("n", "r2", "flux"):
                _same(s[name], orc_s[name], f"state {name}")
        else:
            with pytest.raises(capi.SppmError, match="gather_max"):
                ctx.check()
        ctx.close()


def test_several_eye_paths_per_pixel_match_oracle():
    """samples = 2, supersamples = 2 (the loops of src/sppm2.h:628-643): the last path's hitpoint stays, the direct
    terms of all paths add up."""
    W, H, M = 48, 36, 15000
    ctx, orc = _pair(W, H, M, samples=2, supersamples=2)
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
