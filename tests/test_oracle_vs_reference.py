"""The oracle (oracle/sppm_oracle.c, sampler = MT19937, math = libm) against golden vectors made by
the compiled, unmodified reference (tests/golden/make_golden.py).  Everything is compared bit for
bit: this is what pins the oracle.  CPU only."""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd.scene import CORNELL_SPHERES


def _bits(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    ok = (a.view(np.uint64) == b.view(np.uint64)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ"


def test_stateless_pieces(golden):
    g = golden("64x48")
    sph = ol.make_spheres(CORNELL_SPHERES)
    p = ol.make_params(64, 48, sampler=0, math=0)
    _bits(ol.camera_rays(p), g["cam_rays"], "PinholeCamera::get_ray")
    for tag in ("cam", "rnd"):
        ids, f = ol.intersect(sph, g[f"{tag}_rays"])
        _bits(ids, g[f"{tag}_hit_id"], f"{tag} first-hit ids")
        m = ids >= 0
        _bits(f[m], g[f"{tag}_hit_f"][m], f"{tag} hit records")
    _bits(ol.mt_samples(4242, 4096), g["mt_samples_seed4242"], "ValueSampler<double> sequence")
    for i, s in enumerate((0, 1, 12345)):
        u, f01 = ol.xorshift(s, 16)
        _bits(u, g["xorshift_u32"][i], "XorShift::next")
        _bits(f01, g["xorshift_01"][i], "XorShift::next01")
    _bits(ol.hdr_pixels(g["encode_in"]), g["encode_rgbe"], "get_hdr_pixel")
    _bits(ol.to_int(g["encode_in"]).reshape(-1, 3), g["encode_toint"], "to_int")


@pytest.mark.parametrize("tag", ["64x48", "64x48_r5", "320x240", "64x48_dof", "64x48_spp", "64x48_k3", "64x48_k8", "320x240_k100"])
def test_sppm_passes(golden, tag):
    g = golden(tag)
    w, h, photons, passes, traced, k, se, sp = [int(x) for x in g["meta"]]
    st = int(g["stride"][0])
    cam = {}
    if "spp" in g:
        cam = dict(samples=int(g["spp"][0]), supersamples=int(g["spp"][1]))
    if "dof" in g:  # eye rays through DoFCamera::get_ray (src/cameras.h:123-139), lens draws from their own engine
        cam = dict(camera_model=1, aperture=float(g["dof"][0]), focus_dist=float(g["dof"][1]), seed_lens=int(g["dof"][2]))
        _bits(ol.camera_rays(ol.make_params(w, h, sampler=0, math=0, **cam)), g["dof_rays"], "DoFCamera::get_ray")
    p = ol.make_params(w, h, sampler=0, math=0, gather_radius2=float(g["gather_r2"][0]), gather_max=k, seed_eye=se,
                       seed_photon=sp, photons_per_pass=photons, **cam)
    o = ol.Oracle(p, ol.make_spheres(CORNELL_SPHERES))
    img = None
    for i in range(passes):
        o.eye_pass(i)
        s = o.state()
        _bits(s["valid"] > 0, g[f"hp_valid_pass{i}"] > 0, f"pass {i} which pixels own a hitpoint")
        _bits(s["obj"].astype(np.int8), g[f"hp_obj_pass{i}"], f"pass {i} hitpoint object ids")
        _bits(s["hp_f"][::st], g[f"hp_f_pass{i}"][:, :10], f"pass {i} hitpoint records")
        _bits(s["ld"][::st], g[f"ld_pass{i}"], f"pass {i} direct term (incl. the double first-sight sample)")
        assert o.counters()["eye_draws"] * 2 == int(g[f"eye_words_pass{i}"][0]), "eye stream position"
        t = o.photon_pass(i, photons, tap=True, acc_cap=40_000_000)
        s = o.state()
        v = s["valid"] > 0
        _bits(s["n"][v], g[f"state_n_pass{i}"][v], f"pass {i} photon counts")
        _bits(s["r2"][::st], g[f"state_f_pass{i}"][:, 0], f"pass {i} radii")
        _bits(s["flux"][::st], g[f"state_f_pass{i}"][:, 1:], f"pass {i} flux")
        nw = len(g[f"ph_words_pass{i}"])
        _bits(t["draws"][:nw] * 2, g[f"ph_words_pass{i}"], f"pass {i} engine words per photon")
        # accepted hitpoint sets of the first nw photons
        ph = t["rec_photon"][t["acc_rec"]]
        sel = ph < nw
        order = np.lexsort((t["acc_pix"][sel], ph[sel]))
        ph_s, px_s = ph[sel][order], t["acc_pix"][sel][order]
        pairs, cnt = np.unique(np.stack([ph_s, px_s], 1), axis=0, return_counts=True)
        off = g[f"ph_acc_off_pass{i}"]
        ref_ph = np.repeat(np.arange(nw), np.diff(off))
        _bits(pairs[:, 0], ref_ph, f"pass {i} accepted sets (photon)")
        _bits(pairs[:, 1], g[f"ph_acc_pix_pass{i}"], f"pass {i} accepted sets (image index)")
        _bits(cnt.astype(np.int8), g[f"ph_acc_cnt_pass{i}"], f"pass {i} accepted multiplicity")
        img = o.resolve(i, photons)
        _bits(img[::st], g[f"image_pass{i}"], f"pass {i} image")
        assert img.sum() == g[f"sums_pass{i}"][3]
    assert ol.hdr_file(img, w, h) == g["final_hdr"].tobytes(), "save_hdr_file bytes"
    if "final_ppm" in g:
        assert ol.ppm_file(img, w, h) == g["final_ppm"].tobytes(), "save_ppm_file bytes"
    else:
        import hashlib
        assert hashlib.sha256(ol.ppm_file(img, w, h)).digest() == g["final_ppm_sha256"].tobytes()


@pytest.mark.parametrize("tag", ["64x48", "48x36_ss2_r5", "32x24_s2_k6"])
def test_classic_ppm(golden, tag):
    """src/progressive_photonmapping.h: create_point's hitpoint list, the kd-sorted order render() sums in, the
    photons' accepted hitpoint sets, radii/counts/flux after all photons, the image and the .hdr bytes."""
    g = golden("ppm_" + tag)
    w, h, photons, traced, k, se, sp, ss = [int(x) for x in g["meta"]]
    r2 = float(g["gather_r2"][0])
    samples = int(g["samples"][0]) if "samples" in g else 1
    p = ol.make_params(w, h, samples=samples, supersamples=ss, sampler=0, math=0, gather_radius2=r2, initial_radius2=r2, gather_max=k,
                       seed_eye=se, seed_photon=sp, photons_per_pass=photons)
    o = ol.PpmOracle(p, ol.make_spheres(CORNELL_SPHERES))
    nh = o.eye_pass()
    assert nh == len(g["hp_pix"]), "number of hitpoints"
    s = o.state()
    _bits(s["pix"], g["hp_pix"], "hitpoint image indices, creation order")
    _bits(s["obj"].astype(np.int8), g["hp_obj"], "hitpoint object ids")
    _bits(s["hp_f"], g["hp_f"], "hitpoint records (distance, position, normal, weight, emission)")
    assert o.counters()["eye_draws"] * 2 == int(g["eye_words"][0]), "eye stream position"
    _bits(o.kd_order(), g["kd_order"], "CreateKDtree's vector order (std::sort's tie order included)")
    t = o.photon_pass(0, photons, tap=True, acc_cap=60_000_000)
    s = o.state()
    _bits(s["n"], g["state_n"], "photon counts")
    _bits(s["r2"], g["state_f"][:, 0], "radii")
    _bits(s["flux"], g["state_f"][:, 1:], "accumulated flux")
    nw = len(g["ph_words"])
    _bits(t["draws"][:nw] * 2, g["ph_words"], "engine words per photon")
    ph = t["rec_photon"][t["acc_rec"]]
    sel = ph < nw
    order = np.lexsort((t["acc_pix"][sel], ph[sel]))
    pairs, cnt = np.unique(np.stack([ph[sel][order], t["acc_pix"][sel][order]], 1), axis=0, return_counts=True)
    _bits(pairs[:, 0], np.repeat(np.arange(nw), np.diff(g["acc_off"])), "accepted sets (photon)")
    _bits(pairs[:, 1], g["acc_hp"], "accepted sets (hitpoint id)")
    _bits(cnt.astype(np.int8), g["acc_cnt"], "accepted multiplicity")
    img = o.resolve(photons, order=0)
    _bits(img, g["image"], "image, summed in the reference's vector order")
    assert ol.hdr_file(img, w, h) == g["final_hdr"].tobytes(), "save_hdr_file bytes"
    import hashlib
    assert hashlib.sha256(ol.ppm_file(img, w, h)).digest() == g["final_ppm_sha256"].tobytes()
    # creation-order sums (what the GPU path does) differ only by the rounding of a few-term sum
    img1 = o.resolve(photons, order=1)
    assert np.abs(img1 - img).max() <= 4e-16 * np.abs(img).max()


@pytest.mark.parametrize("rule", [0, 1])
def test_all_core_photon_pass_equals_serial(rule):
    """orc_photon_pass_mt (bench.py's all-core CPU baseline) against the serial pass: identical state and image for
    3 threads, for the per-photon rule (hitpoints cut into one kd-tree per thread, every thread replays all photon
    hits in order) and the per-pass rule (atomic integer deposits)."""
    import oracle_lib as ol
    from raytracer_amd.scene import CORNELL_SPHERES
    W, H, M = 48, 36, 6000
    outs = []
    for mt in (False, True):
        o = ol.Oracle(ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M, rule=rule), ol.make_spheres(CORNELL_SPHERES))
        for i in range(2):
            o.eye_pass(i)
            if mt:
                o.photon_pass_mt(i, M // 2, first=0, threads=3)
                o.photon_pass_mt(i, M - M // 2, first=M // 2, threads=3)
            else:
                o.photon_pass(i, M)
            o.end_pass()
        outs.append((o.state(), o.resolve(1, M), o.counters()))
        o.close()
    (s0, img0, c0), (s1, img1, c1) = outs
    for k in ("valid", "n", "r2", "flux"):
        assert np.array_equal(np.ascontiguousarray(s0[k]).view(np.uint8), np.ascontiguousarray(s1[k]).view(np.uint8)), k
    assert np.array_equal(img0.view(np.uint64), img1.view(np.uint64))
    assert c0["updates"] == c1["updates"] and c0["rays"] == c1["rays"]


def test_all_core_classic_ppm_equals_serial():
    import oracle_lib as ol
    from raytracer_amd.scene import CORNELL_SPHERES
    W, H, M = 48, 36, 6000
    outs = []
    for mt in (False, True):
        o = ol.PpmOracle(ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M, gather_radius2=5.0, initial_radius2=5.0), ol.make_spheres(CORNELL_SPHERES))
        o.eye_pass()
        if mt:
            o.photon_pass_mt(0, M, threads=4)
        else:
            o.photon_pass(0, M)
        outs.append((o.state(), o.counters()))
        o.close()
    for k in ("n", "r2", "flux"):
        assert np.array_equal(np.ascontiguousarray(outs[0][0][k]).view(np.uint8), np.ascontiguousarray(outs[1][0][k]).view(np.uint8)), k
    assert outs[0][1]["updates"] == outs[1][1]["updates"]


@pytest.mark.parametrize("tag", ["64x48", "64x48_l1", "320x240", "64x48_k8", "320x240_k100"])
def test_per_pass_rule_pinned(golden, tag):
    """The per-pass rule against the compiled, unmodified src/stochastic_ppm.h (oracle/ref_driver_pp.cpp drives its
    create_point_loop and create_photon and applies the update of :721-741).  The oracle with that file's tracer
    (tracer=1), its double sums in arrival order (accum=1) and, for _l1, render()'s own stream structure reproduces
    every tap bit for bit: hitpoints, direct term, stream positions, engine words and accepted hitpoints per photon, M
    and the double sums before the update, N / r^2 / tau after it, the image and the writers' bytes.  _k8 / _k100: the
    K cap of the kd search binds (:425-427 hand gahter_max_photon_num to the same SearchNearest)."""
    g = golden("pp_" + tag)
    w, h, photons, passes, traced, k, se, sp, layout = [int(x) for x in g["meta"]]
    st = int(g["stride"][0])
    o = None
    img = None
    for i in range(passes):
        r2s, r2max = g[f"search_r2_pass{i}"]
        # update_photons_with_render searches with the running maximum of the hitpoint radii (:577-586): as long as one
        # hitpoint has accepted nothing that is initial_radius_2, the constant this build is given as gather_radius2
        assert r2s == r2max == 25.0
        if o is None:
            p = ol.make_params(w, h, sampler=0, math=0, rule=1, tracer=1, accum=1, stream_layout=layout, gather_radius2=float(r2s),
                               gather_max=k, seed_eye=se, seed_photon=sp, photons_per_pass=photons)
            o = ol.Oracle(p, ol.make_spheres(CORNELL_SPHERES))
        o.eye_pass(i)
        s = o.state()
        _bits(s["valid"] > 0, g[f"hp_valid_pass{i}"] > 0, f"pass {i} which pixels own a hitpoint")
        _bits(s["obj"].astype(np.int8), g[f"hp_obj_pass{i}"], f"pass {i} hitpoint object ids")
        _bits(s["hp_f"][::st], g[f"hp_f_pass{i}"], f"pass {i} hitpoint records")
        _bits(s["ld"][::st], g[f"ld_pass{i}"], f"pass {i} direct term")
        assert o.counters()["eye_draws"] * 2 == int(g[f"eye_words_pass{i}"][0]), "eye stream position"
        t = o.photon_pass(i, photons, tap=True, acc_cap=40_000_000)
        if layout == 1:
            assert o.counters()["eye_draws"] * 2 == int(g[f"stream_words_pass{i}"][0]), "stream position after the photons"
        v = s["valid"] > 0
        dep, _ = o.deposits()
        _bits(dep[v, 3].astype(np.int32), g[f"dep_m_pass{i}"][v], f"pass {i} M of every hitpoint")
        _bits(o.deposits_double()[::st], g[f"dep_f_pass{i}"], f"pass {i} double sums of :452-454")
        nw = len(g[f"ph_words_pass{i}"])
        _bits(t["draws"][:nw] * 2, g[f"ph_words_pass{i}"], f"pass {i} engine words per photon")
        ph = t["rec_photon"][t["acc_rec"]]
        sel = ph < nw
        order = np.lexsort((t["acc_pix"][sel], ph[sel]))
        off = g[f"ph_acc_off_pass{i}"]
        _bits(ph[sel][order], np.repeat(np.arange(nw), np.diff(off)), f"pass {i} accepted sets (photon)")
        _bits(t["acc_pix"][sel][order], g[f"ph_acc_pix_pass{i}"], f"pass {i} accepted sets (image index)")
        o.end_pass()
        s = o.state()
        _bits(o_nd(o)[v], g[f"state_n_pass{i}"][v], f"pass {i} N")
        _bits(s["r2"][v], g[f"state_r2_pass{i}"][v], f"pass {i} r^2")
        _bits(s["flux"][::st], g[f"tau_pass{i}"], f"pass {i} tau")
        img = o.resolve(i, photons)
        _bits(img[::st], g[f"image_pass{i}"], f"pass {i} image")
        assert img.sum() == g[f"sums_pass{i}"][4]
    assert ol.hdr_file(img, w, h) == g["final_hdr"].tobytes(), "save_hdr_file bytes"
    import hashlib
    assert hashlib.sha256(ol.ppm_file(img, w, h)).digest() == g["final_ppm_sha256"].tobytes()


def o_nd(o):
    """N of the per-pass rule is a double (photon_count of SPPMPixel, src/stochastic_ppm.h:127)"""
    return o.state_nd()


@pytest.mark.parametrize("tracer", [0, 1])
def test_fixed_point_sums_against_the_references_double_sums(tracer):
    """What the GPU does (accum 0: photon flux summed in 2^-s fixed point, colour/pi applied once per pass) against the
    reference's own sum (accum 1: multiply(weight, colour/pi) added in double per deposit, :449-453), on the same
    photons.  M, hence N and r^2, are identical; a fixed-point sum is within M/2 units of the last place of the exact
    sum (round to nearest per deposit), i.e. within M * 2^-s / 2 * colour/pi of the reference's, plus double rounding."""
    W, H, M = 64, 48, 30000
    runs = []
    for accum in (0, 1):
        o = ol.Oracle(ol.make_params(W, H, sampler=1, math=1, rule=1, photons_per_pass=M, tracer=tracer, accum=accum), ol.make_spheres(CORNELL_SPHERES))
        taps = []
        for i in range(3):
            o.eye_pass(i)
            o.photon_pass(i, M)
            dep, flag = o.deposits()
            assert flag == 0
            taps.append((dep.copy(), o.deposits_double().copy(), o.state()["obj"].copy()))
            o.end_pass()
        runs.append((taps, o.state(), o.state_nd(), o.resolve(2, M), o.fixed_scale()))
        o.close()
    (t0, s0, n0, img0, scale), (t1, s1, n1, img1, _) = runs
    colors = np.array([c[3] for c in CORNELL_SPHERES])
    for (dep, _, obj), (dep1, dsum, _) in zip(t0, t1):
        assert np.array_equal(dep[:, 3], dep1[:, 3]), "M"
        f = colors[np.maximum(obj, 0)] / np.pi
        fixed = dep[:, :3] / scale * f
        m = dep[:, 3:4].astype(np.float64)
        bound = (0.5 * m / scale) * f + 4e-16 * (m + 1) * np.abs(dsum)
        assert (np.abs(fixed - dsum) <= bound).all(), float(np.max(np.abs(fixed - dsum) - bound))
    assert np.array_equal(n0, n1) and np.array_equal(s0["r2"].view(np.uint64), s1["r2"].view(np.uint64)), "N and r^2 do not depend on how flux is summed"
    tau0, tau1 = s0["flux"], s1["flux"]
    nz = np.abs(tau1) > 0
    assert np.max(np.abs(tau0 - tau1)[nz] / np.abs(tau1)[nz]) <= 1e-7
    assert np.sqrt(np.mean((img0 - img1) ** 2)) / np.sqrt(np.mean(img1 ** 2)) <= 1e-9
