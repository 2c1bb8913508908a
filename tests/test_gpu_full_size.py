"""Full-size runs of the BASELINE configurations that the oracle cannot reach in test time, through properties that
do not depend on the size (run-to-run bits, r^2 as a function of N, sum of N = accepted updates, shares of a pass that
merge bit-exactly into the whole):
  * config 5: 3840x2160 over the 10 007-sphere synthetic scene (grid closest hit), one GPU;
  * config 3 (1080p, 8M photons per pass over 8 GPUs), one rank's share rehearsed on one GPU, for both update rules."""
import numpy as np
import pytest

from raytracer_amd import scene, sppm

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    ok = (a.view(np.uint64) == b.view(np.uint64)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{what}: {(~ok).sum()} of {ok.size} differ, first at {np.argwhere(~ok)[0]}"


def _r2_table(nmax, r0=25.0, alpha=0.7):
    """r^2 after N accepted photons, in the reference's operation order (src/sppm2.h:398-407)"""
    table = np.empty(nmax + 1)
    r2 = r0
    table[0] = r2
    for k in range(nmax):
        g = 1.0 if k == 0 else (k * alpha + alpha) / (k * alpha + 1.0)
        r2 = r2 * g
        table[k + 1] = r2
    return table


def test_config5_4k_10k_spheres_properties():
    """BASELINE config 5 on one GPU: 3840x2160, 10 007 spheres, 1M photons per pass, two passes, per-photon rule."""
    W, H, M = 3840, 2160, 1_000_000
    table = scene.synthetic_spheres(10000)
    assert len(table) == 10007
    outs = []
    for _ in range(2):
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M), table)
        for i in range(2):
            ctx.eye_pass(i)
            ctx.photon_pass(i, M)
        ctx.check()
        s, img, cnt = ctx.hitpoints(), ctx.resolve(1, M).cpu().numpy(), ctx.counters()
        ctx.close()
        outs.append((s, img, cnt))
    s, img, cnt = outs[0]
    v = s["valid"] > 0
    assert v.sum() > 0.9 * W * H
    assert int(s["n"][v].sum()) == cnt["updates"] > 0
    _same(s["r2"][v], _r2_table(int(s["n"].max()))[s["n"][v]], "r2 as a function of N")
    assert np.isfinite(img).all() and (img >= 0).all()
    assert len(np.unique(s["obj"][v])) > 1000, "the eye paths must reach many of the small spheres"
    for k in ("valid", "obj", "n", "r2", "flux", "ld"):
        _same(outs[1][0][k], s[k], f"run-to-run {k}")
    _same(outs[1][1], img, "run-to-run image")


def test_config5_4k_10k_spheres_per_pass_rule_run_to_run():
    """the same scene under the per-pass rule: two renders of three passes agree in every bit of the hitpoint state (this
    kind of test is what caught the raw-asm register pin of DESIGN.md section 10 in the ordered gather)"""
    W, H, M = 3840, 2160, 1_000_000
    table = scene.synthetic_spheres(10000)
    outs = []
    for _ in range(2):
        ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1), table)
        for i in range(3):
            ctx.eye_pass(i)
            ctx.photon_pass(i, M)
            ctx.end_pass()
        ctx.check()
        outs.append((ctx.hitpoints(), ctx.resolve(2, M).cpu().numpy(), ctx.counters()))
        ctx.close()
    assert outs[0][2]["updates"] == outs[1][2]["updates"] > 0
    for k in ("valid", "obj", "r2", "flux", "ld"):
        _same(outs[1][0][k], outs[0][0][k], f"run-to-run {k}")
    _same(outs[1][1], outs[0][1], "run-to-run image")


def test_config3_per_pass_rule_rank_shares_sum_to_the_pass():
    """north_star's split at full size: a pass of 8M photons at 1080p; every one of 8 'ranks' (one context, one after
    the other) deposits photons [r M/8, (r+1) M/8) into all hitpoints; the integer sum of the eight deposit arrays
    (what the all-reduce computes) equals the deposits of one context that traced the whole pass, and so does the state
    after the end-of-pass update."""
    W, H, N = 1920, 1080, 8
    Mr = 1_000_000
    M = Mr * N
    whole = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    whole.eye_pass(0)
    whole.photon_pass(0, M)
    ref = whole.export_deposits().clone()
    whole.end_pass()
    ref_state = whole.hitpoints()
    whole.check(); whole.close()
    total = None
    rank_ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    rank_ctx.eye_pass(0)
    zero = None
    for r in range(N):
        rank_ctx.photon_pass(0, Mr, first=r * Mr)
        d = rank_ctx.export_deposits().clone()
        if zero is None:
            zero = d * 0
        rank_ctx.import_deposits(zero.clone())   # the next 'rank' starts from empty accumulators
        total = d if total is None else total + d
    import torch
    assert torch.equal(total, ref)
    assert int(ref[3].max().item()) < (1 << 21), "deposits per hitpoint stay inside the fixed-point budget"
    rank_ctx.import_deposits(total)
    rank_ctx.end_pass()
    s = rank_ctx.hitpoints()
    rank_ctx.check(); rank_ctx.close()
    for k in ("valid", "n", "r2", "flux"):
        _same(s[k], ref_state[k], k)


def test_config3_per_photon_rule_one_rank_of_eight():
    """The exact rule over 8 GPUs, one rank's share at full size: rank 3 owns an eighth of the hitpoints (a range of
    their Hilbert order), traces its block of every batch and receives the other ranks' records (here: from a helper
    context, through the pack/append rows the exchange carries).  Its hitpoints end up exactly as in one context that
    did the whole 8M-photon pass."""
    import torch
    W, H, N, rank = 1920, 1080, 8, 3
    Mr = 1_000_000
    M = Mr * N
    whole = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
    whole.eye_pass(0)
    whole.photon_pass(0, M)
    ref = whole.hitpoints()
    whole.check(); whole.close()
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, tile_first=rank, tile_stride=N, ownership=1))
    helper = sppm.SppmContext(sppm.make_config(64, 48, photons_per_pass=M))
    rows = torch.empty((4 * Mr + 4096, 10), dtype=torch.float64, device="cuda")
    own_rows = torch.empty_like(rows)
    ctx.eye_pass(0)
    done, applied = 0, 0
    while done < M:
        batch = min((1 << 20) * N, max(32768, applied // 2))   # the batch rule of raytracer_amd.parallel
        n = min(max(batch // N, 1024), (M - done) // N)
        ctx.trace(0, done + rank * n, n)          # this rank's block first: packing empties the record buffer
        own_cnt = ctx.pack_records(own_rows)
        for r in range(N):
            if r == rank:
                ctx.append_records(own_rows, own_cnt, done + r * n, done + (r + 1) * n)
                continue
            helper.trace(0, done + r * n, n)
            cnt = helper.pack_records(rows)
            ctx.append_records(rows, cnt, done + r * n, done + (r + 1) * n)
        ctx.gather()
        applied += n * N
        done += n * N
    ctx.check()
    s = ctx.hitpoints()
    mask = ctx.owned_mask().cpu().numpy().astype(bool)
    ctx.close(); helper.close()
    assert abs(int(mask.sum()) - W * H // N) <= 1
    for k in ("n", "r2", "flux"):
        _same(s[k][mask], ref[k][mask], f"owned {k}")
    assert int(s["n"][mask].sum()) > 0
