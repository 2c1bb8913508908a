"""Edge cases of the hot path against the oracle, bit for bit: image sizes that are no multiple of the 8x8 / 64-hitpoint
tiles (down to one pixel), passes of a handful of photons cut into ragged calls, empty calls, a camera that sees
nothing diffuse.  Both update rules."""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import sppm
from raytracer_amd.scene import CORNELL_SPHERES

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    ok = a.view(np.uint64) == b.view(np.uint64) if a.dtype.kind == "f" else a == b
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


def _state_keys(rule):
    return ("n", "r2", "flux") if rule == 0 else ("r2", "flux")


@pytest.mark.parametrize("w,h,rule", [(37, 23, 0), (37, 23, 1), (1, 1, 0), (1, 1, 1), (65, 1, 0), (3, 129, 1)])
def test_ragged_image_sizes(w, h, rule):
    M, passes = 3000, 3
    ctx = sppm.SppmContext(sppm.make_config(w, h, photons_per_pass=M, update_rule=rule))
    orc = ol.Oracle(ol.make_params(w, h, sampler=1, math=1, photons_per_pass=M, rule=rule), ol.make_spheres(CORNELL_SPHERES))
    for i in range(passes):
        ctx.eye_pass(i); orc.eye_pass(i)
        s, o = ctx.hitpoints(), orc.state()
        for k in ("valid", "obj", "hp_f", "ld"):
            _same(s[k], o[k], f"pass {i} eye {k}")
        ctx.photon_pass(i, M); ctx.end_pass()
        orc.photon_pass(i, M); orc.end_pass()
        s, o = ctx.hitpoints(), orc.state()
        for k in _state_keys(rule):
            _same(s[k], o[k], f"pass {i} state {k}")
        _same(ctx.resolve(i, M).cpu().numpy(), orc.resolve(i, M), f"pass {i} image")
    ctx.check()
    assert ctx.counters()["updates"] == orc.counters()["updates"]
    ctx.close()


@pytest.mark.parametrize("rule", [0, 1])
def test_ragged_photon_calls_and_empty_calls(rule):
    """a pass fed as calls of 0, 1, 7, 0, 640 and the remaining photons equals the pass in one call"""
    w, h, M = 48, 32, 2500
    cuts = [0, 1, 7, 0, 640]
    res = []
    for split in (False, True):
        ctx = sppm.SppmContext(sppm.make_config(w, h, photons_per_pass=M, update_rule=rule))
        for i in range(2):
            ctx.eye_pass(i)
            if split:
                first = 0
                for c in cuts:
                    ctx.photon_pass(i, c, first=first)
                    first += c
                ctx.photon_pass(i, M - first, first=first)
            else:
                ctx.photon_pass(i, M)
            ctx.end_pass()
        ctx.check()
        res.append((ctx.hitpoints(), ctx.resolve(1, M).cpu().numpy(), ctx.counters()["updates"]))
        ctx.close()
    for k in _state_keys(rule):
        _same(res[0][0][k], res[1][0][k], f"state {k}")
    _same(res[0][1], res[1][1], "image")
    assert res[0][2] == res[1][2]
    orc = ol.Oracle(ol.make_params(w, h, sampler=1, math=1, photons_per_pass=M, rule=rule), ol.make_spheres(CORNELL_SPHERES))
    for i in range(2):
        orc.eye_pass(i); orc.photon_pass(i, M); orc.end_pass()
    _same(res[0][1], orc.resolve(1, M), "image against the oracle")


@pytest.mark.parametrize("rule", [0, 1])
def test_camera_that_sees_no_diffuse_surface(rule):
    """every eye path leaves through a scene of one light and nothing else: no hitpoint exists, the photon passes run
    and change nothing, the image is the direct term (zero)"""
    light_only = [CORNELL_SPHERES[0]]
    w, h, M = 16, 16, 500
    ctx = sppm.SppmContext(sppm.make_config(w, h, photons_per_pass=M, update_rule=rule), spheres=light_only)
    orc = ol.Oracle(ol.make_params(w, h, sampler=1, math=1, photons_per_pass=M, rule=rule), ol.make_spheres(light_only))
    for i in range(2):
        ctx.eye_pass(i); orc.eye_pass(i)
        ctx.photon_pass(i, M); ctx.end_pass()
        orc.photon_pass(i, M); orc.end_pass()
    ctx.check()
    s, o = ctx.hitpoints(), orc.state()
    _same(s["valid"], o["valid"], "valid flags")
    assert ctx.counters()["updates"] == orc.counters()["updates"]
    _same(ctx.resolve(1, M).cpu().numpy(), orc.resolve(1, M), "image")
    ctx.close()


def test_per_pass_rule_with_hitpoint_ownership_is_refused():
    """update_rule 1 shards photons, not hitpoints: a share of the hitpoints per rank (tile_stride > 1) would leave the
    imported sums of the hitpoints a rank does not own in place for the next pass -- refused at creation."""
    from raytracer_amd import capi
    with pytest.raises(capi.SppmError, match="update_rule 1 keeps all hitpoints"):
        sppm.SppmContext(sppm.make_config(64, 48, photons_per_pass=1000, update_rule=1, tile_first=0, tile_stride=2))
