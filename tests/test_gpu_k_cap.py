"""The K cap of the reference's kd search (src/kdtree.h:126-133) when it BINDS, through the C ABI.

The oracle is pinned for this by the fixtures ref_64x48_k3 / ref_64x48_k8 / ref_320x240_k100 that the compiled reference wrote
with K = 3, 8 and 100 (tests/test_oracle_vs_reference.py); here the GPU path must reproduce
  * the oracle bit for bit (XorShift streams, detmath: accepted sets, counts, radii, flux, image), for the tile gather
    and for the simple gather, one batch or several, classic PPM too;
  * the reference's own run in reference-stream mode (same mt19937 streams): counts and radii identical.
check_gather_max = 2 switches the cap off again (the uncapped set, which must then differ)."""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import capi, sppm
from raytracer_amd.scene import CORNELL_SPHERES

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    ok = a.view(np.uint64) == b.view(np.uint64) if a.dtype.kind == "f" else a == b
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


def _okeys(tap):
    k = (tap["rec_photon"].astype(np.uint64) << np.uint64(sppm.KEY_SHIFT)) | (tap["rec_depth"].astype(np.uint64) << np.uint64(1))
    return k | (tap["rec_f"][:, 3:6] != tap["rec_f"][:, 6:9]).any(axis=1).astype(np.uint64)


@pytest.mark.parametrize("K,batches,mode", [(3, 1, "tiles"), (8, 3, "tiles"), (3, 2, "simple"), (1, 1, "tiles")])
def test_capped_search_matches_oracle(K, batches, mode, monkeypatch):
    W, H, M, passes = 64, 48, 20000, 3
    if mode == "simple":
        monkeypatch.setenv("SPPM_GATHER_TILES", "0")
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, gather_max=K))
    orc = ol.Oracle(ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M, gather_max=K), ol.make_spheres(CORNELL_SPHERES))
    for i in range(passes):
        ctx.eye_pass(i); orc.eye_pass(i)
        tap = orc.photon_pass(i, M, tap=True, acc_cap=8_000_000)
        okey = _okeys(tap)
        ok = okey[tap["acc_rec"]]
        order = np.lexsort((tap["acc_pix"], ok))
        got_k, got_p = [], []
        per = M // batches
        for b in range(batches):
            first, count = b * per, (per if b + 1 < batches else M - b * per)
            ctx.trace(i, first, count)
            ctx.accept_tap(8_000_000)
            ctx.gather()
            ak, ap = ctx.accepted()
            ctx.accept_tap(0)
            got_k.append(ak); got_p.append(ap)
        ak, ap = np.concatenate(got_k), np.concatenate(got_p)
        o2 = np.lexsort((ap, ak))
        _same(ak[o2], ok[order], f"pass {i} accepted keys")
        _same(ap[o2], tap["acc_pix"][order], f"pass {i} accepted image indices")
        ctx.end_pass()
        s, o = ctx.hitpoints(), orc.state()
        for k in ("n", "r2", "flux"):
            _same(s[k], o[k], f"pass {i} state {k}")
        _same(ctx.resolve(i, M).cpu().numpy(), orc.resolve(i, M), f"pass {i} image")
    ctx.check()
    c = ctx.counters()
    assert c["cap_queries"] > 0 and c["cap_dropped_pairs"] > 0, c   # the cap acted
    ctx.close()


def test_classic_ppm_with_binding_cap():
    """src/progressive_photonmapping.h takes the same K through the same KDTree (main.cpp:76-87 passes 10000)."""
    w, h, photons, K = 48, 36, 20000, 10
    ctx = sppm.SppmContext(sppm.make_config(w, h, supersamples=2, photons_per_pass=photons, gather_radius2=25.0, initial_radius2=25.0, algorithm=1, gather_max=K))
    orc = ol.PpmOracle(ol.make_params(w, h, supersamples=2, sampler=1, math=1, photons_per_pass=photons, gather_radius2=25.0, initial_radius2=25.0, gather_max=K),
                       ol.make_spheres(CORNELL_SPHERES))
    assert ctx.n_hitpoints == orc.eye_pass()
    ctx.eye_pass(0)
    for first in (0, photons // 2):   # two calls: the tree is built once, the cap acts in both
        ctx.photon_pass(0, photons // 2, first=first)
    orc.photon_pass(0, photons)
    ctx.check()
    s, o = ctx.hitpoints(), orc.state()
    for k in ("n", "r2", "flux"):
        _same(s[k], o[k], f"state {k}")
    _same(ctx.resolve(0, photons).cpu().numpy(), orc.resolve(photons, order=1), "image")
    assert ctx.counters()["cap_dropped_pairs"] > 0
    ctx.close()


def test_classic_ppm_identical_samples_tie_in_the_tree():
    """samples = 2 puts two hitpoints at EVERY position (src/main.cpp:79 runs 100 identical samples): every range of the
    tree holds equal keys, whose order std::sort decides, and the heap holds equal distances, of which libstdc++'s
    pop_heap decides which one the cap drops.  Equal to the oracle, which the compiled reference's run with the same
    set-up pins (ref_ppm_32x24_s2_k6: the GPU reproduces that run too, tests/test_gpu_classic_ppm.py)."""
    w, h, photons, K = 32, 24, 10000, 6
    ctx = sppm.SppmContext(sppm.make_config(w, h, samples=2, photons_per_pass=photons, gather_radius2=25.0, initial_radius2=25.0, algorithm=1, gather_max=K))
    orc = ol.PpmOracle(ol.make_params(w, h, samples=2, sampler=1, math=1, photons_per_pass=photons, gather_radius2=25.0, initial_radius2=25.0, gather_max=K),
                       ol.make_spheres(CORNELL_SPHERES))
    assert ctx.n_hitpoints == orc.eye_pass()
    ctx.eye_pass(0); ctx.photon_pass(0, photons); orc.photon_pass(0, photons)
    ctx.check()
    s, o = ctx.hitpoints(), orc.state()
    for k in ("n", "r2", "flux"):
        _same(s[k], o[k], f"state {k}")
    assert ctx.counters()["cap_dropped_pairs"] > 0
    ctx.close()


def test_cap_switched_off_is_the_uncapped_set():
    W, H, M = 64, 48, 20000
    res = {}
    for tag, kw in (("capped", dict(gather_max=5)), ("ignored", dict(gather_max=5, check_gather_max=2)), ("uncapped", dict())):
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, **kw))
        ctx.eye_pass(0); ctx.photon_pass(0, M); ctx.end_pass(); ctx.check()
        res[tag] = ctx.hitpoints()
        ctx.close()
    _same(res["ignored"]["n"], res["uncapped"]["n"], "check_gather_max = 2 runs the uncapped search")
    _same(res["ignored"]["flux"], res["uncapped"]["flux"], "check_gather_max = 2 runs the uncapped search")
    assert res["capped"]["n"].sum() < res["uncapped"]["n"].sum()


@pytest.mark.parametrize("K,batches,mode", [(3, 1, "tiles"), (8, 3, "tiles"), (3, 2, "simple")])
def test_capped_search_under_the_per_pass_rule_matches_oracle(K, batches, mode, monkeypatch):
    """src/stochastic_ppm.h:425-427 hand gahter_max_photon_num to the same SearchNearest: update_rule 1 with a binding K.
    Accepted pairs, fixed-point deposits, state and image equal the oracle's bit for bit (the oracle's capped per-pass
    rule is pinned by the compiled file's runs ref_pp_64x48_k8 / ref_pp_320x240_k100)."""
    W, H, M, passes = 64, 48, 20000, 3
    if mode == "simple":
        monkeypatch.setenv("SPPM_GATHER_TILES", "0")
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, gather_max=K, update_rule=1))
    orc = ol.Oracle(ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M, gather_max=K, rule=1), ol.make_spheres(CORNELL_SPHERES))
    for i in range(passes):
        ctx.eye_pass(i); orc.eye_pass(i)
        tap = orc.photon_pass(i, M, tap=True, acc_cap=8_000_000)
        ok = _okeys(tap)[tap["acc_rec"]]
        order = np.lexsort((tap["acc_pix"], ok))
        got_k, got_p = [], []
        per = M // batches
        for b in range(batches):
            first, count = b * per, (per if b + 1 < batches else M - b * per)
            ctx.trace(i, first, count)
            ctx.accept_tap(8_000_000)
            ctx.gather()
            ak, ap = ctx.accepted()
            ctx.accept_tap(0)
            got_k.append(ak); got_p.append(ap)
        ak, ap = np.concatenate(got_k), np.concatenate(got_p)
        o2 = np.lexsort((ap, ak))
        _same(ak[o2], ok[order], f"pass {i} accepted keys")
        _same(ap[o2], tap["acc_pix"][order], f"pass {i} accepted image indices")
        dep, flag = orc.deposits()
        assert flag == 0
        _same(ctx.export_deposits().cpu().numpy().T, dep, f"pass {i} deposits")
        ctx.end_pass(); orc.end_pass()
        s, o = ctx.hitpoints(), orc.state()
        for k in ("n", "r2", "flux"):
            _same(s[k], o[k], f"pass {i} state {k}")
        _same(ctx.resolve(i, M).cpu().numpy(), orc.resolve(i, M), f"pass {i} image")
    ctx.check()
    c = ctx.counters()
    assert c["cap_queries"] > 0 and c["cap_dropped_pairs"] > 0, c   # the cap acted
    ctx.close()


def test_cap_filter_skips_the_search_where_no_query_can_overflow():
    """K below the number of hitpoints, but above what any photon hit can see within the search radius: the count grid of
    sppm_cap.hpp (k_cap_filter) lets no photon hit through, no tree is walked, and the result is the uncapped one."""
    W, H, M = 64, 48, 20000
    res = {}
    for tag, kw in (("filtered", dict(gather_max=2000)), ("uncapped", dict(gather_max=1 << 40))):
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, gather_radius2=4.0, **kw))
        for i in range(2):
            ctx.eye_pass(i); ctx.photon_pass(i, M)
        ctx.check()
        res[tag] = (ctx.hitpoints(), ctx.counters())
        ctx.close()
    assert res["filtered"][1]["cap_queries"] == 0 and res["filtered"][1]["cap_dropped_pairs"] == 0
    for k in ("n", "r2", "flux"):
        _same(res["filtered"][0][k], res["uncapped"][0][k], k)


@pytest.mark.parametrize("tag", ["64x48_k3", "64x48_k8", "320x240_k100"])
def test_reference_run_with_binding_cap(golden, tag):
    """The compiled reference's run with a binding K, its own mt19937 streams: photon counts and radii of every
    hitpoint identical after every pass, flux and image to rounding (sin/cos/x^5 come from detmath here)."""
    g = golden(tag)
    w, h, photons, passes, traced, k, se, sp = [int(x) for x in g["meta"]]
    st = int(g["stride"][0])
    ctx = sppm.SppmContext(sppm.make_config(w, h, gather_radius2=float(g["gather_r2"][0]), gather_max=k, seed_eye=se, seed_photon=sp,
                                            photons_per_pass=photons, sampler=1))
    for i in range(passes):
        ctx.eye_pass(i)
        v = ctx.hitpoints()["valid"] > 0
        assert np.array_equal(v, g[f"hp_valid_pass{i}"] > 0)
        ctx.photon_pass(i, photons); ctx.end_pass(); ctx.check()
        s = ctx.hitpoints()
        vs = v[::st]
        assert np.array_equal(s["n"][v], g[f"state_n_pass{i}"][v]), f"pass {i}: photon counts per hitpoint"
        ref_state = g[f"state_f_pass{i}"]
        assert np.array_equal(s["r2"][::st][vs].view(np.uint64), ref_state[:, 0][vs].view(np.uint64)), f"pass {i}: radii"
        np.testing.assert_allclose(s["flux"][::st][vs], ref_state[:, 1:][vs], rtol=1e-10, atol=1e-12, err_msg=f"pass {i} flux")
        img = ctx.resolve(i, photons).cpu().numpy()
        rel = float(np.sqrt(np.mean((img[::st] - g[f"image_pass{i}"]) ** 2)) / np.sqrt(np.mean(g[f"image_pass{i}"] ** 2)))
        assert rel < 1e-12, f"pass {i} image: rel-RMSE {rel}"
    ctx.close()
