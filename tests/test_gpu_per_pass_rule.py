"""update_rule 1, the per-pass (N, tau) rule of src/stochastic_ppm.h:721-741 with deposits in 64-bit fixed point
(north_star's multi-GPU formulation: photons sharded, hitpoints replicated, ONE all-reduce of the deposits per pass).

The oracle's restatement of this rule is pinned bit for bit by runs of the compiled src/stochastic_ppm.h
(tests/test_oracle_vs_reference.py::test_per_pass_rule_pinned); here the GPU must equal the oracle bit for bit.
tests/test_gpu_per_pass_reference.py holds the GPU against those compiled-reference runs directly."""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import sppm
from raytracer_amd.scene import CORNELL_SPHERES

pytestmark = pytest.mark.gpu


def _pair(w, h, photons, **kw):
    cfg = sppm.make_config(w, h, photons_per_pass=photons, update_rule=1, **kw)
    par = ol.make_params(w, h, sampler=1, math=1, photons_per_pass=photons, rule=1, **kw)
    return sppm.SppmContext(cfg), ol.Oracle(par, ol.make_spheres(CORNELL_SPHERES))


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    ok = (a.view(np.uint64) == b.view(np.uint64)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


@pytest.mark.parametrize("r2", [25.0, 5.0])
def test_per_pass_rule_matches_oracle(r2):
    """Accepted sets, the fixed-point deposits, the state after the update and the image, every pass."""
    W, H, M = 64, 48, 20000
    ctx, orc = _pair(W, H, M, gather_radius2=r2)
    assert ctx.fixed_point_scale() == orc.fixed_scale()
    for i in range(3):
        ctx.eye_pass(i); orc.eye_pass(i)
        ctx.accept_tap(6_000_000)
        ctx.photon_pass(i, M)
        akey, apix = ctx.accepted()
        ctx.accept_tap(0)
        tap = orc.photon_pass(i, M, tap=True, acc_cap=6_000_000)
        okey = (tap["rec_photon"].astype(np.uint64) << np.uint64(sppm.KEY_SHIFT)) | (tap["rec_depth"].astype(np.uint64) << np.uint64(1))
        okey |= (tap["rec_f"][:, 3:6] != tap["rec_f"][:, 6:9]).any(axis=1).astype(np.uint64)
        ok = okey[tap["acc_rec"]]
        order = np.lexsort((tap["acc_pix"], ok))
        _same(akey, ok[order], f"pass {i} accepted keys")
        _same(apix, tap["acc_pix"][order], f"pass {i} accepted image indices")
        dep, flag = orc.deposits()
        assert flag == 0
        _same(ctx.export_deposits().cpu().numpy().T, dep, f"pass {i} deposits")
        ctx.end_pass(); orc.end_pass()
        assert not ctx.export_deposits().any()
        s, o = ctx.hitpoints(), orc.state()
        for k in ("valid", "n", "r2", "flux", "ld", "hp_f"):
            _same(s[k], o[k], f"pass {i} {k}")
        _same(ctx.resolve(i, M).cpu().numpy(), orc.resolve(i, M), f"pass {i} image")
    ctx.check()
    assert ctx.counters()["updates"] == orc.counters()["updates"]
    ctx.close()


def test_deposits_do_not_depend_on_how_the_work_is_cut(monkeypatch):
    """Integer deposits are exact: tiny chunks (many items and atomics per hitpoint), the simple per-hitpoint kernel,
    several batches per pass and a pass deposited in two halves all give the same bits."""
    W, H, M = 96, 64, 30000

    def run(env, halves=False):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
        for i in range(2):
            ctx.eye_pass(i)
            if halves:
                ctx.photon_pass(i, M // 3, first=0)
                ctx.photon_pass(i, M - M // 3, first=M // 3)
            else:
                ctx.photon_pass(i, M)
            ctx.end_pass()
        ctx.check()
        for k in env:  # (most switches are read when the context is made, SPPM_DEP_WG at every deposit)
            monkeypatch.delenv(k)
        s = ctx.hitpoints()
        img = ctx.resolve(1, M).cpu().numpy()
        ctx.close()
        return s, img
    ref_s, ref_img = run({})
    for env, halves in (({"SPPM_DEP_CHUNK": "64"}, False), ({"SPPM_GATHER_TILES": "0"}, False), ({"SPPM_BATCH_PHOTONS": "4096"}, False),
                        ({}, True), ({"SPPM_DEP_CELL_FACTOR": "0.6"}, False), ({"SPPM_PREFETCH": "0"}, False),
                        # the workgroup form with the distance test on the matrix pipe (v_mfma_f64_16x16x4_f64), an experiment kept in the tree
                        ({"SPPM_DEP_WG": "1"}, False), ({"SPPM_DEP_WG": "1", "SPPM_DEP_CHUNK": "64"}, False), ({"SPPM_DEP_WG": "1", "SPPM_BATCH_PHOTONS": "4096"}, False)):
        s, img = run(env, halves)
        for k in ("n", "r2", "flux"):
            _same(s[k], ref_s[k], f"{env} {k}")
        _same(img, ref_img, f"{env} image")


def test_two_emulated_ranks_sum_to_one_gpu():
    """north_star's split on one GPU: two contexts hold all hitpoints, each deposits half of the pass's photons
    (streams keyed by global photon index), the deposits are summed (what the all-reduce does), both apply the update.
    State and image equal a single context that traced everything, bit for bit."""
    W, H, M = 64, 48, 20000
    full = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    ranks = [sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1)) for _ in range(2)]
    for i in range(3):
        full.eye_pass(i); full.photon_pass(i, M); full.end_pass()
        total = None
        for r, ctx in enumerate(ranks):
            ctx.eye_pass(i)
            ctx.photon_pass(i, M // 2, first=r * (M // 2))
            d = ctx.export_deposits()
            total = d if total is None else total + d
        for ctx in ranks:
            ctx.import_deposits(total.clone())
            ctx.end_pass()
        ref = full.hitpoints()
        for ctx in ranks:
            s = ctx.hitpoints()
            for k in ("n", "r2", "flux"):
                _same(s[k], ref[k], f"pass {i} {k}")
            _same(ctx.resolve(i, M).cpu().numpy(), full.resolve(i, M).cpu().numpy(), f"pass {i} image")
    for ctx in ranks + [full]:
        ctx.check(); ctx.close()


def test_per_pass_rule_1080p_properties():
    """Full size (1920x1080, 1M photons): run-to-run identical bits, sum of M = accepted updates, radii shrink by the
    formula (r'^2 = r^2 (N + alpha M) / (N + M)), and a pass deposited in two halves on a second context equals it."""
    import torch
    W, H, M = 1920, 1080, 1_000_000
    a = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    b = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    for i in range(2):
        a.eye_pass(i); b.eye_pass(i)
        c0 = a.counters()["updates"]
        a.photon_pass(i, M)
        b.photon_pass(i, M // 2, first=0); b.photon_pass(i, M // 2, first=M // 2)
        da, db = a.export_deposits(), b.export_deposits()
        assert torch.equal(da, db)
        assert int(da[3].sum().item()) == a.counters()["updates"] - c0
        before = a.hitpoints()
        a.end_pass(); b.end_pass()
        after = a.hitpoints()
        m = da[3].cpu().numpy()
        hit = (m > 0) & (before["valid"] != 0)
        nd0 = np.zeros(W * H) if i == 0 else nd_prev
        expect = before["r2"][hit] * (nd0[hit] + 0.7 * m[hit]) / (nd0[hit] + m[hit])
        assert np.array_equal(after["r2"][hit].view(np.uint64), expect.view(np.uint64))
        assert np.array_equal(after["r2"][~hit].view(np.uint64), before["r2"][~hit].view(np.uint64))
        nd_prev = nd0 + 0.7 * m * (before["valid"] != 0)
    sa, sb = a.hitpoints(), b.hitpoints()
    for k in ("n", "r2", "flux"):
        _same(sa[k], sb[k], k)
    a.check(); b.check(); a.close(); b.close()


@pytest.mark.parametrize("rule", [1, 0])
def test_fused_end_of_pass_equals_the_two_calls(rule):
    """sppm_end_pass_resolve == sppm_end_pass + sppm_resolve: state and image bit for bit, pass after pass"""
    W, H, M = 96, 64, 20000
    a = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=rule))
    b = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=rule))
    for i in range(3):
        a.eye_pass(i); a.photon_pass(i, M); a.end_pass(); ia = a.resolve(i, M).cpu().numpy()
        b.eye_pass(i); b.photon_pass(i, M); ib = b.end_pass_resolve(i, M).cpu().numpy()
        _same(ib, ia, f"pass {i} image")
        sa, sb = a.hitpoints(), b.hitpoints()
        for k in ("r2", "flux", "n"):
            _same(sb[k], sa[k], f"pass {i} state {k}")
    a.check(); b.check()
    a.close(); b.close()


def test_end_of_pass_by_slices_equals_the_whole():
    """sppm_export_deposits_range / sppm_import_deposits_range / sppm_end_pass_resolve_range over ragged slices of the
    image index (what the sliced all-reduce of ShardedSppm does per pass) == the whole-array calls: the exported slices
    are the whole export's columns, state and image bit for bit, pass after pass; a range past the end is refused."""
    import torch
    W, H, M = 96, 64, 20000
    P = W * H
    a = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    b = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    bounds = [0, 1, 1000, 1000, 4097, P]   # a single hitpoint, an empty slice, ragged ones
    for i in range(3):
        a.eye_pass(i); a.photon_pass(i, M); ia = a.end_pass_resolve(i, M).cpu().numpy()
        b.eye_pass(i); b.photon_pass(i, M)
        whole = b.export_deposits().clone()
        img = torch.empty((P, 3), dtype=torch.float64, device=b.device)
        parts = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            d = torch.empty((4, hi - lo), dtype=torch.int64, device=b.device)
            b.export_deposits_range(d, lo, hi - lo)
            parts.append(d)
        assert torch.equal(torch.cat(parts, dim=1), whole), "the slices are the whole export's columns"
        for (lo, hi), d in zip(zip(bounds[:-1], bounds[1:]), parts):
            b.import_deposits_range(d * 2, lo, hi - lo)      # (something else than what is there: the import must land)
            b.import_deposits_range(d, lo, hi - lo)
            b.end_pass_resolve_range(i, M, img, lo, hi - lo)
        _same(img.cpu().numpy(), ia, f"pass {i} image")
        sa, sb = a.hitpoints(), b.hitpoints()
        for k in ("r2", "flux", "n"):
            _same(sb[k], sa[k], f"pass {i} state {k}")
        assert int(b.export_deposits().abs().sum().item()) == 0, "the end of the pass clears the deposits"
    with pytest.raises(Exception):
        b.export_deposits_range(torch.empty((4, 2), dtype=torch.int64, device=b.device), P - 1, 2)
    a.check(); b.check()
    a.close(); b.close()
