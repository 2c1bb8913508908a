"""Scenes of thousands of spheres (BASELINE config 5): above 512 spheres the closest hit walks a uniform grid
over the small spheres instead of testing every sphere.  The reference's intersect_scene (src/scene.h:53-76)
is a minimum over all spheres with the lowest index winning ties; the grid only prunes spheres the ray
cannot hit, so everything must stay bit-identical to the brute-force oracle: first hits, then whole passes.
The scene has no reference counterpart (the reference's scene is a 10-sphere constant); it comes
from raytracer_amd.scene.synthetic_spheres (fixed-seed XorShift placement)."""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import capi, scene, sppm
from raytracer_amd.scene import CORNELL_SPHERES

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    ok = (a.view(np.uint64) == b.view(np.uint64)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


def _random_rays(n, seed):
    rs = np.random.RandomState(seed)
    org = np.stack([rs.uniform(2, 98, n), rs.uniform(1, 80, n), rs.uniform(1, 240, n)], 1)
    d = rs.normal(size=(n, 3))
    # a share of axis-parallel rays and of rays starting on cell-aligned coordinates
    d[: n // 16] = np.eye(3)[rs.randint(0, 3, n // 16)] * rs.choice([-1.0, 1.0], (n // 16, 1))
    org[n // 16: n // 8] = np.round(org[n // 16: n // 8])
    d /= np.sqrt((d * d).sum(1))[:, None]
    return np.concatenate([org, d], 1)


def _first_hits(ctx, table, rays, what):
    import torch
    ids, hit = ctx.intersect(torch.from_numpy(rays))
    ids, hit = ids.cpu().numpy(), hit.cpu().numpy()
    oid, of = ol.intersect(ol.make_spheres(table), rays)
    _same(ids, oid, f"{what}: first-hit ids")
    m = oid >= 0
    _same(hit[m], of[m], f"{what}: hit records")
    return oid


def _passes(ctx, orc, M, passes, what):
    for i in range(passes):
        ctx.eye_pass(i)
        orc.eye_pass(i)
        s, o = ctx.hitpoints(), orc.state()
        for k in ("valid", "obj", "hp_f", "ld"):
            _same(s[k], o[k], f"{what} pass {i} eye {k}")
        ctx.photon_pass(i, M)
        orc.photon_pass(i, M)
        s, o = ctx.hitpoints(), orc.state()
        v = o["valid"] > 0   # a pixel without a hitpoint yet may already hold the next eye pass's initial radius (it runs ahead)
        for k in ("n", "r2", "flux"):
            _same(s[k][v], o[k][v], f"{what} pass {i} state {k}")
        _same(ctx.resolve(i, M).cpu().numpy(), orc.resolve(i, M), f"{what} pass {i} image")
    ctx.check()
    assert ctx.counters()["updates"] == orc.counters()["updates"]  # (the ray counter also holds the batch traced ahead for the next pass)


def test_cornell_scene_through_the_grid(golden, monkeypatch):
    """The reference's own scene forced through the grid path: walls are 'big', four spheres are gridded."""
    monkeypatch.setenv("SPPM_SCENE_GRID", "1")
    W, H, M = 64, 48, 20000
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
    g = golden("64x48")
    for tag in ("cam", "rnd"):
        ids = _first_hits(ctx, CORNELL_SPHERES, g[f"{tag}_rays"], tag)
        _same(ids, g[f"{tag}_hit_id"], f"{tag}: the compiled reference's first-hit ids")
    _first_hits(ctx, CORNELL_SPHERES, _random_rays(200_000, 5), "random rays")
    orc = ol.Oracle(ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M), ol.make_spheres(CORNELL_SPHERES))
    _passes(ctx, orc, M, 2, "cornell/grid")
    ctx.close()


@pytest.mark.parametrize("n", [600, 10000])
def test_many_spheres_match_brute_force(n):
    table = scene.synthetic_spheres(n)
    W, H, M = 64, 48, 20000
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M), spheres=table)
    ids = _first_hits(ctx, table, _random_rays(100_000, n), f"{n} spheres")
    assert len(np.unique(ids)) > min(n, 2000) // 2, "the rays reach a large share of the spheres"
    orc = ol.Oracle(ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M), ol.make_spheres(table))
    _passes(ctx, orc, M, 2, f"{n} spheres")
    ctx.close()


def test_many_spheres_classic_ppm_and_reference_stream():
    """The other two eye kernels and the stream tracer over the grid: classic PPM, and the reference-stream sampler."""
    table = scene.synthetic_spheres(1500)
    W, H, M = 48, 36, 8000
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, gather_radius2=9.0, initial_radius2=9.0, algorithm=1), spheres=table)
    orc = ol.PpmOracle(ol.make_params(W, H, sampler=1, math=1, photons_per_pass=M, gather_radius2=9.0, initial_radius2=9.0), ol.make_spheres(table))
    assert ctx.n_hitpoints == orc.eye_pass()
    ctx.eye_pass(0)
    ctx.photon_pass(0, M)
    orc.photon_pass(0, M)
    s, o = ctx.hitpoints(), orc.state()
    _same(s["hp_f"][:, :8], o["hp_f"][:, :8], "hitpoint list")
    for k in ("n", "r2", "flux"):
        _same(s[k], o[k], f"state {k}")
    _same(ctx.resolve(0, M).cpu().numpy(), orc.resolve(M, order=1), "image")
    ctx.check()
    ctx.close()
    # sampler 1: the GPU reads mt19937 streams; the oracle (sampler 0, libm) differs from it only through sin/cos/x^5
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, sampler=1), spheres=table)
    orc = ol.Oracle(ol.make_params(W, H, sampler=0, math=0, photons_per_pass=M), ol.make_spheres(table))
    ctx.eye_pass(0)
    orc.eye_pass(0)
    ctx.photon_pass(0, M)
    orc.photon_pass(0, M)
    ctx.check()
    s, o = ctx.hitpoints(), orc.state()
    assert np.array_equal(s["valid"], o["valid"])
    assert (s["n"] == o["n"]).mean() > 0.995
    img, oimg = ctx.resolve(0, M).cpu().numpy(), orc.resolve(0, M)
    assert float(np.sqrt(np.mean((img - oimg) ** 2)) / np.sqrt(np.mean(oimg ** 2))) < 1e-3
    ctx.close()


def test_sphere_count_limit():
    with pytest.raises(capi.SppmError):
        sppm.SppmContext(sppm.make_config(8, 8), spheres=[CORNELL_SPHERES[0]] * ((1 << 20) + 1))


def test_grid_without_walls_and_empty_views():
    """Edge cases: a scene of equal spheres and nothing else (no 'big' sphere, most rays leave the grid and hit
    nothing), and cameras that see nothing at all (SPPM: no pixel owns a hitpoint; classic PPM: an empty list)."""
    table = [t for t in scene.synthetic_spheres(700, radius=(0.8, 0.8)) if t[0] < 100.0]  # drops the walls, keeps the light
    ctx = sppm.SppmContext(sppm.make_config(32, 24, photons_per_pass=2000), spheres=table)
    ids = _first_hits(ctx, table, _random_rays(100_000, 3), "no walls")
    assert (ids < 0).mean() > 0.2 and (ids >= 0).mean() > 0.02
    orc = ol.Oracle(ol.make_params(32, 24, sampler=1, math=1, photons_per_pass=2000), ol.make_spheres(table))
    _passes(ctx, orc, 2000, 2, "no walls")
    ctx.close()
    lonely = [(1.0, (50.0, 52.0, 400.0), (10.0, 10.0, 10.0), (0.9, 0.9, 0.9), scene.DIFFUSE)]  # one light behind the camera
    for algo in (0, 1):
        ctx = sppm.SppmContext(sppm.make_config(32, 24, photons_per_pass=1000, algorithm=algo), spheres=lonely)
        assert ctx.n_hitpoints == (0 if algo == 1 else 32 * 24)
        ctx.eye_pass(0)
        ctx.photon_pass(0, 1000)
        ctx.end_pass()
        img = ctx.resolve(0, 1000).cpu().numpy()
        ctx.check()
        assert (img == 0).all() and ctx.counters()["updates"] == 0
        if algo == 0:
            assert (ctx.hitpoints()["valid"] == 0).all()
        ctx.close()
