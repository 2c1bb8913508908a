#!/usr/bin/env python3
"""bench.py -- throughput of the SPPM hot path on MI355X (contract: see the task statement).

A "step" is one SPPM pass at BASELINE config 2 geometry (Cornell box, 1920x1080, samples = supersamples = 1,
alpha 0.7, initial r^2 = 25 = search r^2): eye pass over every pixel, 1M photons per GPU traced and applied to the
hitpoints, the progressive update, image resolve.  The timed region is a FRESH render: passes 0..K-1 (default
K = 256, the whole of config 2: "1M photons/pass x 256 passes on one MI355X"); the W warmup passes run before it on a
scratch context of the same size (clocks, first touch), so they are not part of the render.  Pass cost falls as the
radii shrink (pass 0 is ~2.5x a late pass), so K matters and is reported in config.workload.  Nothing is uploaded
inside the timed region (scene and state are resident in HBM).

    python bench.py --gpus 1 --steps K --warmup W [--rule {1,0}]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

--rule picks the update rule of the primary line:
  1 (default)  the per-pass (N, tau) update of src/stochastic_ppm.h:721-741 -- north_star's formulation.  N > 1: every
               rank keeps all hitpoints, traces photons [rank M, (rank+1) M) of a pass of N x M, deposits them in 64-bit
               fixed point, ONE all-reduce of 32 bytes per hitpoint per pass; bit-identical for every N.
  0            the reference's live per-photon update (src/sppm2.h:391-421), ordered replay, exact for every N.  N > 1:
               hitpoints partitioned by ranges of their Hilbert order, photon-hit records sorted by destination and
               exchanged with one all-to-all per batch.
At N = 1 the other rule is measured in the same run and reported beside the primary line (--one-rule skips it), with
the relative RMSE between the two images; `k_cap` reports two passes with the reference's K = 10000 cap enforced
exactly (the timed renders search uncapped: config.gather_max); `cpu_baseline` times the oracle's all-core photon pass
on a bounded sample.  `roofline` is the gather kernel of the primary rule: algorithmic FP64 operations over HIP-event
kernel time against the vector FP64 peak, with the committed rocprofv3 --pmc counters (profiles/r03_pmc_rule<R>.json).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(width, height, photons, rule, seconds_target=12.0, table=None):
    """The oracle (a port of the reference's kd-tree SPPM; the reference itself is single-threaded) on this
    box's host cores, same workload geometry and update rule: eye pass 0 + kd build (untimed), then a bounded number
    of photons of pass 0 on ALL cores the OpenMP runtime offers (orc_photon_pass_mt: per-pass rule = photons shared
    out, atomic integer deposits; per-photon rule = hitpoints cut into one kd-tree per thread, every thread replays
    all photon hits in order -- both identical to the serial pass, tests/test_oracle_vs_reference.py), and on one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from raytracer_amd.scene import CORNELL_SPHERES
    spheres = ol.make_spheres(table if table is not None else CORNELL_SPHERES)
    cores = int(ol.lib().orc_max_threads())

    def sample(threads, budget):
        par = ol.make_params(width, height, sampler=1, math=1, photons_per_pass=photons, rule=rule)
        orc = ol.Oracle(par, spheres)
        t0 = time.perf_counter()
        orc.eye_pass(0)
        t_eye = time.perf_counter() - t0
        n, done, t_ph = (2000 if table is None else 100) * threads, 0, 0.0
        while t_ph < budget and done < photons:
            n = min(n, photons - done)
            if threads > 1:
                t_ph += orc.photon_pass_mt(0, n, first=done, threads=threads)
            else:
                t0 = time.perf_counter()
                orc.photon_pass(0, n, first=done)
                t_ph += time.perf_counter() - t0
            done += n
            n *= 2
        upd = orc.counters()["updates"]
        orc.close()
        return done, t_ph, upd, t_eye
    done, t_ph, upd, t_eye = sample(cores, seconds_target)
    out = {"value": done / t_ph, "unit": "photons/s", "cores": cores, "kind": "port",
           "sample": f"{width}x{height} pass 0, {'per-pass' if rule == 1 else 'per-photon'} rule: eye pass + kd-tree build {t_eye:.1f} s (one core, not in value), "
                     f"then the first {done} photons on {cores} cores (OpenMP) in {t_ph:.1f} s, {upd} hitpoint updates ({upd / t_ph:.3g} updates/s)"}
    if cores > 1:
        d1, t1, u1, _ = sample(1, seconds_target / 2)
        out["one_core"] = {"value": d1 / t1, "unit": "photons/s", "sample": f"the first {d1} photons of the same pass on one core in {t1:.1f} s ({u1 / t1:.3g} updates/s)"}
    return out


def reference_check(photons=20000):
    """The compiled, unmodified reference (oracle/_ref, built by oracle/Makefile from /root/reference where that exists)
    beside the port on a workload the reference can run: its eye pass searches the hitpoint list linearly for every
    pixel (src/sppm2.h:213-226), which is quadratic in the pixel count, so 1920x1080 is out of its reach; at 320x240
    the photon pass of both is timed on the same photons of pass 0.  None if the binary is not there."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver_320x240")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([exe, tmp, str(photons), "1", "0", "25.0", "10000"], env=dict(os.environ, REF_TIMING="1"),
                           capture_output=True, text=True, timeout=300)
    m = re.search(r"eye_s ([0-9.]+) kd_s ([0-9.]+) photons (\d+) photon_s ([0-9.]+)", r.stderr)
    if r.returncode != 0 or not m:
        return {"error": (r.stderr or "")[-200:]}
    import oracle_lib as ol
    from raytracer_amd.scene import CORNELL_SPHERES
    par = ol.make_params(320, 240, sampler=1, math=1, photons_per_pass=photons)
    orc = ol.Oracle(par, ol.make_spheres(CORNELL_SPHERES))
    orc.eye_pass(0)
    t0 = time.perf_counter()
    orc.photon_pass(0, photons, first=0)
    t_port = time.perf_counter() - t0
    orc.close()
    return {"workload": f"320x240, pass 0, {photons} photons, r^2 25, K 10000, one core each",
            "reference_photons_per_s": photons / float(m.group(4)), "port_photons_per_s": photons / t_port,
            "reference_eye_pass_s": float(m.group(1)), "reference_kd_build_s": float(m.group(2)),
            "note": "reference = oracle/_ref/ref_driver_320x240 (the unmodified headers compiled with g++ -O2); its eye pass "
                    "is quadratic in the pixel count (linear search of the hitpoint list per pixel), so it cannot run 1920x1080"}


def cpu_baseline_ppm(width, height, photons, r2, seconds_target=12.0):
    """Classic PPM on the host: the oracle's hitpoint list + kd build, then a bounded number of photons."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from raytracer_amd.scene import CORNELL_SPHERES
    par = ol.make_params(width, height, sampler=1, math=1, photons_per_pass=photons, gather_radius2=r2, initial_radius2=r2)
    orc = ol.PpmOracle(par, ol.make_spheres(CORNELL_SPHERES))
    t0 = time.perf_counter()
    nh = orc.eye_pass()
    t_eye = time.perf_counter() - t0
    n, done, t_ph = 2000, 0, 0.0
    while t_ph < seconds_target and done < photons:
        n = min(n, photons - done)
        t0 = time.perf_counter()
        orc.photon_pass(0, n, first=done)
        t_ph += time.perf_counter() - t0
        done += n
        n *= 2
    upd = orc.counters()["updates"]
    orc.close()
    return {"value": done / t_ph, "unit": "photons/s", "cores": 1, "kind": "port",
            "sample": f"{width}x{height}: {nh} hitpoints + kd-tree build {t_eye:.1f} s (not in value), then the first "
                      f"{done} photons in {t_ph:.1f} s, {upd} hitpoint updates ({upd / t_ph:.3g} updates/s)"}


def main_ppm(args, rank, world, barrier):
    """--workload ppm: BASELINE config 4, classic progressive photon mapping (src/progressive_photonmapping.h) at
    1080p.  A step is one batch of --photons photons per GPU applied to the fixed hitpoint list; the timed region
    is a whole render: the eye pass that stores the list, K batches, (N > 1: the all-reduce of the owned hitpoint
    state,) the image.  (sppm_create, outside the region as for the SPPM workload, traces the eye paths once to
    size the list: reported as create_ms.)  N > 1: the list is shared out by ranges of its Hilbert order, records
    by all-gather, as for SPPM (raytracer_amd.parallel.ShardedPpm)."""
    import torch
    import torch.distributed as dist
    from raytracer_amd.parallel import ShardedPpm
    W, H, M, K = args.width, args.height, args.photons, args.steps
    r2 = args.radius2
    total = M * K * world

    def make(steps):
        return ShardedPpm(W, H, M, steps, rank=rank, world=world, supersamples=args.supersamples, gather_radius2=r2, initial_radius2=r2)
    if args.warmup > 0:
        scratch = make(args.warmup)
        for i in range(args.warmup):
            scratch.step(i)
        scratch.finish()
        scratch.ctx.check()
        scratch.close()
    barrier()
    t0 = time.perf_counter()
    job = make(K)
    torch.cuda.synchronize()
    create_ms = 1e3 * (time.perf_counter() - t0)
    ctx = job.ctx
    ctx.set_timing(True)
    c0 = ctx.counters()
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        job.step(k)
    img = job.finish()
    ctx.check()
    barrier()
    dt = time.perf_counter() - t0
    c1 = ctx.counters()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    d = torch.tensor([c1[k] - c0[k] for k in ("records", "candidates", "updates")], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    records, _, upd_all = [float(x) for x in d.tolist()]
    if rank == 0:
        g_ns = c1["gather_ns"] - c0["gather_ns"]
        g_launch = max(1, c1["gather_launches"] - c0["gather_launches"])
        cand, upd = c1["candidates"] - c0["candidates"], c1["updates"] - c0["updates"]
        alg_bytes = 28.0 * cand + 32.0 * upd
        achieved = alg_bytes / max(g_ns, 1) if g_ns else 0.0
        out = {
            "metric": "photons traced/sec + hitpoint updates/sec (whole node), classic PPM Cornell 1080p",
            "value": total / dt, "unit": "photons/s", "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * dt / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Cornell-box classic PPM {W}x{H}, supersamples {args.supersamples}, {ctx.n_hitpoints} fixed hitpoints, "
                                   f"{K} batches of {M} photons per GPU, alpha 0.7, initial r^2 = search r^2 = {r2:g} (src/main.cpp:76-77), XorShift streams",
                       "photons": total, "width": W, "height": H, "hitpoints": ctx.n_hitpoints, "parallelism": job.describe()},
            "create_ms": create_ms, "hitpoint_updates_per_s": upd_all / dt, "updates_per_photon": upd_all / total,
            "records_per_photon": records / max(1.0, float(M) * K * (world if job.replicated_trace else 1)),
            "trace": {"kernel": "k_photon_trace", "launches": int(c1["trace_launches"] - c0["trace_launches"]),
                      "avg_launch_ms": (c1["trace_ns"] - c0["trace_ns"]) / 1e6 / max(1, c1["trace_launches"] - c0["trace_launches"]),
                      "note": "rank 0; on its own stream, overlapped with the gather of the batch before"},
            "roofline": {"bound": "hbm", "kernel": "k_gather_tiles", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "launches": int(g_launch), "avg_launch_ms": g_ns / 1e6 / g_launch, "alg_bytes_per_launch": alg_bytes / g_launch,
                         "note": "rank 0; algorithmic bytes = 28 B x (record, hitpoint) pairs distance-tested + 32 B x accepted updates (SURVEY 8d); launch durations from HIP events on the kernel's stream"},
        }
        assert bool(torch.isfinite(img).all())
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_ppm(W, H, total, r2)
        print(json.dumps(out), flush=True)
    job.close()
    if world > 1:
        dist.destroy_process_group()


def measured_stream_bandwidth(torch, gib=2):
    """What HBM delivers on THIS box to a plain device-to-device copy (bytes read + bytes written per second), beside the
    8 TB/s of the data sheet: the second denominator SURVEY 8(d) asks for."""
    n = gib * (1 << 30) // 8
    src = torch.ones(n, dtype=torch.float64, device="cuda")
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 8 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def measured_fma_peak(torch):
    """FP64 vector FMA rate this box sustains (TFLOP/s), beside the 78.6 of the data sheet."""
    import ctypes as C
    from raytracer_amd import capi
    v = C.c_double()
    capi.check(capi.lib().sppm_measure_fma_peak(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(v)))
    return v.value


def measured_counters(width, height, photons, rule, steps):
    """Per-launch hardware counters of the gather kernel from the committed rocprofv3 --pmc passes of this script
    (profiles/r03_pmc_rule<R>.json, made by tools/pmc_passes.sh + tools/rocpd_summary.py sq: HBM bytes = 2 x FETCH_SIZE
    (gfx950) + WRITE_SIZE, SQ busy / VALU / LDS cycles).  Counters cannot be read from inside a normal run; a file
    matches by (width, height, photons per pass, rule) -- it says which passes it was averaged over."""
    try:
        with open(os.path.join(ROOT, "profiles", f"r03_pmc_rule{rule}.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if [width, height, photons, rule] != [t.get("width"), t.get("height"), t.get("photons"), t.get("rule")]:
        return None
    return t


def k_cap_enforced(width, height, photons, table, torch, passes=3):
    """The timed renders run UNCAPPED (gather_max out of reach; config.gather_max says so).  BASELINE config 2 passes
    K = 10000 to the kd search, which acts on a few percent of the photon hits at 1080p: this measures the same render
    with the cap ENFORCED exactly (sppm_config.check_gather_max = 0, raytracer_amd/csrc/sppm_cap.hpp) over `passes`
    passes of the per-photon rule, and what the cap did."""
    from raytracer_amd import sppm
    ctx = sppm.SppmContext(sppm.make_config(width, height, photons_per_pass=photons, gather_max=10000), spheres=table)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_pass = []
    for i in range(passes):
        t1 = time.perf_counter()
        ctx.eye_pass(i); ctx.photon_pass(i, photons); ctx.end_pass()
        torch.cuda.synchronize()
        per_pass.append(1e3 * (time.perf_counter() - t1))
    ctx.check()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = ctx.counters()
    ctx.close()
    return {"gather_max": 10000, "passes": passes, "photons_per_s": passes * photons / dt, "ms_per_pass": 1e3 * dt / passes, "ms_each_pass": per_pass,
            "photon_hits": c["records"], "photon_hits_capped": c["cap_queries"], "pairs_withheld": c["cap_dropped_pairs"], "updates": c["updates"],
            "note": "per-photon rule with the reference's K cap enforced bit-exactly (tests/test_gpu_k_cap.py): host-built kd order (about 0.37 s of each pass, "
                    "std::sort as in the reference) + one search lane and one heap of K + 1 entries per photon hit that can see more than K hitpoints; the "
                    "first pass also allocates the heaps; a fidelity mode, not the timed path"}


def image_quality():
    """Both update rules at equal pass and photon counts (320x240, 200k photons per pass, after 16 and 256 passes) against
    the committed 65 536-pass render of the per-photon rule (tests/golden/converged_320x240_rule0.npz, made by
    tools/render_converged.py with seeds of its own): which rule's image is closer.  The primary line may be the
    per-pass rule only while its error is within 10 % of the per-photon rule's (tests/test_gpu_converged.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_converged import errors_vs_converged
    e = errors_vs_converged()
    return {"per_photon_rule": {str(k): v for k, v in e[0].items()}, "per_pass_rule": {str(k): v for k, v in e[1].items()},
            "workload": "Cornell-box SPPM 320x240, 200000 photons/pass, relative RMSE after 16 and 256 passes against 65536 passes of the per-photon rule "
                        "(different seeds); the long renders of the two rules differ by 0.0016 from each other"}


# algorithmic work of the gather kernels
PAIR_FLOP = 23.0      # FP64 operations of one (record, hitpoint) test: 3 sub, 5 |d|^2, 5 n.g, 5 g.d, 5 flatness with its band
UPDATE_FLOP = {0: 24.0, 1: 0.0}   # per accepted update: rule 0 = shrink factor, 3 x (w c f / pi), (acc + add) g; rule 1 = integer adds only
FP64_PEAK = 78.6      # TFLOP/s, vector FP64 with every operation an FMA (MI355X data sheet); 39.3 T operations/s without FMA


def run_render(args, rank, world, barrier, rule, table, torch, dist):
    """one timed render with the given update rule; returns the measurements of this rank plus the image"""
    from raytracer_amd.parallel import ShardedSppm, all_reduce_sum
    W, H, M = args.width, args.height, args.photons
    if args.warmup > 0:  # untimed: a scratch render of W passes
        scratch = ShardedSppm(W, H, photons_per_rank=M, rank=rank, world=world, spheres=table, rule=rule)
        scratch.total_passes = args.warmup
        for i in range(args.warmup):
            scratch.step(i)
        scratch.ctx.check()
        scratch.close()
    job = ShardedSppm(W, H, photons_per_rank=M, rank=rank, world=world, spheres=table, rule=rule)
    job.total_passes = args.steps
    job.time_stages = world > 1
    job.ctx.set_timing(True)
    c0 = job.ctx.counters()
    barrier()
    t0 = time.perf_counter()
    progress = os.environ.get("SPPM_BENCH_PROGRESS")
    for i in range(args.steps):
        job.step(i)
        if progress and rank == 0:
            torch.cuda.synchronize()
            print(f"[bench] pass {i} done at {time.perf_counter() - t0:.3f} s", file=sys.stderr, flush=True)
    img = job.gather_image() if world > 1 else job.image   # the final image on every rank
    job.ctx.check()   # drains the look-ahead streams too
    barrier()
    dt = time.perf_counter() - t0
    c1 = job.ctx.counters()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    # per-rank counter deltas summed over ranks
    d = torch.tensor([c1[k] - c0[k] for k in ("rays", "records", "candidates", "updates")], dtype=torch.float64, device="cuda")
    if world > 1:
        h = tmax.cpu() if dist.get_backend() == "gloo" else tmax
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        tmax = h
        all_reduce_sum(d)
    out = {"dt": float(tmax.item()), "c0": c0, "c1": c1, "sums": [float(x) for x in d.tolist()], "image": img, "describe": job.describe(),
           "replicated_trace": job.replicated_trace, "exchanged_bytes": job.exchanged_bytes,
           "exchange_ms": [a.elapsed_time(b) for _, a, b in job.stage_events], "exchange_names": [n for n, _, _ in job.stage_events]}
    job.close()
    return out


def report(args, world, rule, r, table_name):
    """the JSON fields of one render"""
    W, H, M = args.width, args.height, args.photons
    c0, c1 = r["c0"], r["c1"]
    dt = r["dt"]
    rays, records, cand, upd = r["sums"]
    photons_total = float(M) * world * args.steps
    g_ns = c1["gather_ns"] - c0["gather_ns"]
    g_launch = max(1, c1["gather_launches"] - c0["gather_launches"])
    cand0, upd0 = c1["candidates"] - c0["candidates"], c1["updates"] - c0["updates"]
    kernel = "k_gather_tiles" if rule == 0 else "k_deposit_tiles"
    flop = PAIR_FLOP * cand0 + UPDATE_FLOP[rule] * upd0
    tflops = flop / max(g_ns, 1) / 1e3 if g_ns else 0.0   # flop/ns = GFLOP/s
    alg_bytes = 28.0 * cand0 + 32.0 * upd0                  # 28 B per pair examined + 32 B per accepted update
    pm = measured_counters(W, H, M, rule, args.steps) if world == 1 else None
    rule_name = "per-pass update (src/stochastic_ppm.h:721-741), deposits in 64-bit fixed point" if rule == 1 else "per-photon update (src/sppm2.h:391-421)"
    out = {
        "value": photons_total / dt, "ms_per_step": 1e3 * dt / args.steps,
        "config": {"workload": f"{table_name} SPPM {W}x{H}, {M} photons/pass/GPU, passes 0..{args.steps - 1} of a fresh render, alpha 0.7, "
                               f"initial r^2 25, search r^2 25, {rule_name}, XorShift streams",
                   "update_rule": rule, "gather_max": "not enforced (uncapped search; K = 10000 enforced: see k_cap)", "photons_per_pass": M * world, "width": W, "height": H, "parallelism": r["describe"]},
        "hitpoint_updates_per_s": upd / dt, "rays_per_s": rays / dt,
        "records_per_photon": records / max(1.0, float(M) * args.steps * (world if r["replicated_trace"] else 1)),
        "updates_per_photon": upd / photons_total,
        "roofline": {
            "bound": "valu", "kernel": kernel, "achieved": tflops, "peak": FP64_PEAK, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK,
            "traffic": pm["hbm_bytes_per_launch"] if pm else None,
            "launches": int(g_launch), "avg_launch_ms": g_ns / 1e6 / g_launch,
            "flop_per_launch": flop / g_launch, "pairs_tested_per_launch": cand0 / g_launch, "updates_per_launch": upd0 / g_launch,
            "tests_per_update": cand0 / max(1.0, float(upd0)),
            # the byte accounting of SURVEY 8(d), kept beside the bound that the counters show
            "alg_bytes_per_launch": alg_bytes / g_launch, "alg_GBps": alg_bytes / max(g_ns, 1),
            "update_bytes_GBps_at_60B": 60.0 * upd0 / max(g_ns, 1), "update_bytes_frac_of_hbm_peak": 60.0 * upd0 / max(g_ns, 1) / 8000.0,
            "note": "rank 0; bound = the vector ALU: the committed SQ counters (profiles/r03_pmc_rule%d.json) show it as the busiest unit of the kernel (VALU busy %s "
                    "of the time every SIMD is occupied, LDS %s, HBM a few percent of 8 TB/s) and the pair loops are bound by the vector instructions they issue (ablation, DESIGN.md 3c), so the roofline is the FP64 vector peak -- the reference's arithmetic is "
                    "double precision; the kernels run the tests in packed single precision with error bands and an exact double-precision fallback.  Staging (a third of the kernel at pass 0, more late in a render) is chains of dependent LDS permutes, loads and classification.  achieved = 23 flop x "
                    "(record, hitpoint) pairs examined%s / kernel time from HIP events on the kernel's stream: ALGORITHMIC operations (the reference's test of a "
                    "pair); the kernels execute fewer%s.  peak 78.6 TFLOP/s counts every operation as an FMA -- the tests run with contraction off "
                    "(bit-exact parity), where the attainable rate is 39.3 T operations/s; traffic = HBM bytes per launch from rocprofv3 --pmc "
                    "(2 x FETCH_SIZE + WRITE_SIZE) of %s" % (
                        rule, ("%.0f %%" % (100.0 * pm["valu_active_share"])) if pm and pm.get("valu_active_share") else "most",
                        ("%.0f %%" % (100.0 * pm["lds_active_share"])) if pm and pm.get("lds_active_share") else "less",
                        " + 24 flop x accepted updates" if rule == 0 else "",
                        " (rule 1 decides the normal and flatness tests of ~80 % of the staged records once per cluster of 64 hitpoints and runs the distance test alone on them, three packed multiply-adds per two records in a form expanded about the cluster's centre)" if rule == 1 else "",
                        pm.get("run", "the committed run") if pm else "(no committed counters for this workload)"),
        },
    }
    if pm:
        out["roofline"]["traffic_GBps"] = pm["hbm_bytes_per_launch"] / (pm.get("avg_launch_ms", g_ns / 1e6 / g_launch) * 1e6)
        out["roofline"]["sq"] = {k: pm[k] for k in ("valu_active_share", "lds_active_share", "wait_inst_any_share", "valu_insts_per_launch", "lds_bank_conflict_share") if k in pm}
    t_ns = c1["trace_ns"] - c0["trace_ns"]
    t_launch = max(1, c1["trace_launches"] - c0["trace_launches"])
    p_rays = c1["photon_rays"] - c0["photon_rays"]
    t_tflops = 219.0 * p_rays / max(t_ns, 1) / 1e3 if t_ns else 0.0
    out["roofline_trace"] = {"bound": "valu_f64", "kernel": "k_photon_trace", "achieved": t_tflops, "peak": FP64_PEAK, "unit": "TFLOP/s",
                             "frac": t_tflops / FP64_PEAK, "launches": int(t_launch), "avg_launch_ms": t_ns / 1e6 / t_launch, "rays_per_launch": p_rays / t_launch,
                             "note": "rank 0; 219 flop per ray (20 x 10 spheres + 19, SURVEY 8d) x rays cast by photon tracing; the tracer runs on 128 of "
                                     "1024 workgroup slots, on its own streams, overlapped with the gather: a launch lasts as long as its longest photon path"}
    if args.spheres > 0:
        out["roofline_trace"]["note"] = "rank 0; rays cast by photon tracing; the flop figure prices a ray at the Cornell scene's 219 flop (a grid walk tests fewer spheres per ray than brute force, so this is not a flop count for this scene)"
    if world > 1:
        ex = r["exchange_ms"]
        import torch.distributed as dist
        out["rccl_ranks"] = dist.get_world_size()   # from the process group, not from the command line
        out["collective_backend"] = dist.get_backend()
        out["exchanged_bytes_per_pass_per_rank"] = r["exchanged_bytes"] / max(1, args.steps)
        names = r.get("exchange_names") or []
        per_slice = {}
        for n, ms in zip(names, ex):
            per_slice.setdefault(n, []).append(ms)
        out["stage_ms"] = {"trace_launch_avg": t_ns / 1e6 / t_launch, "exchange_avg": sum(ex) / len(ex) if ex else None, "gather_launch_avg": g_ns / 1e6 / g_launch,
                           "exchange_avg_by_slice": {n: sum(v) / len(v) for n, v in sorted(per_slice.items())} if rule == 1 else None,
                           "note": "rank 0; trace overlaps the gather on its own streams; exchange = " + ("per slice of the hitpoints (ShardedSppm.slices): from the end of the slice's export to the arrival of its sum on the context's stream -- the collectives of the slices run back to back on the process group's stream, so a later slice's time includes the earlier ones'; import and end of pass of a slice follow while the next collective runs" if rule == 1 else
                                    "counts + the record collective (all-to-all by destination, or all-gather), per batch, on a side stream beside the gather")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--photons", type=int, default=1_000_000, help="photons per pass per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-k-cap", action="store_true", help="skip the two passes with the reference's K cap enforced (N = 1)")
    ap.add_argument("--no-converged", action="store_true", help="skip the 320x240 renders of both rules against the committed long render (N = 1)")
    ap.add_argument("--workload", choices=("sppm", "ppm"), default="sppm", help="sppm: BASELINE config 2/3 (the metric's workload); ppm: config 4, classic PPM")
    ap.add_argument("--spheres", type=int, default=0, help="sppm workload: N > 0 replaces the Cornell scene by raytracer_amd.scene.synthetic_spheres(N) "
                    "(BASELINE config 5: --spheres 10000 --width 3840 --height 2160); rays then walk a grid over the spheres")
    ap.add_argument("--rule", type=int, choices=(0, 1), default=1,
                    help="sppm workload, the update rule of the primary line: 1 (default) = the per-pass (N, tau) update (src/stochastic_ppm.h:721-741), "
                         "north_star's formulation -- N > 1: photons sharded, hitpoints replicated, one all-reduce of the 64-bit fixed-point deposits per pass; "
                         "0 = the reference's live per-photon update (src/sppm2.h:391-421), exact for any N -- N > 1: hitpoints partitioned, photon records exchanged")
    ap.add_argument("--one-rule", action="store_true", help="measure only --rule (default at N = 1: the other rule is measured too and reported beside it)")
    ap.add_argument("--supersamples", type=int, default=1, help="--workload ppm only")
    ap.add_argument("--radius2", type=float, default=5.0, help="--workload ppm only: InitialRadius = gather radius (squared), src/main.cpp:76-77")
    args = ap.parse_args()

    if os.environ.get("SPPM_BENCH_PROGRESS"):  # diagnostics: where every rank is, every 30 s
        import faulthandler
        faulthandler.dump_traceback_later(30, repeat=True, file=sys.stderr)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:  # no retry, no re-exec: a mismatch is the launcher's to fix
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch N > 1 as `python -m torch.distributed.run --nnodes=1 "
                 f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 ... bench.py --gpus {args.gpus} ...` (one rank per GPU)")
    # SPPM_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("SPPM_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
        if rank == 0:
            print(f"[bench] process group: backend {dist.get_backend()}, {dist.get_world_size()} ranks, one per GPU", file=sys.stderr, flush=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "ppm":
        return main_ppm(args, rank, world, barrier)

    from raytracer_amd import scene
    table = scene.synthetic_spheres(args.spheres) if args.spheres > 0 else scene.CORNELL_SPHERES
    scene_name = f"{args.spheres}-sphere synthetic scene (scene.synthetic_spheres, seed 20260517) in the Cornell box" if args.spheres > 0 else "Cornell-box"

    primary = run_render(args, rank, world, barrier, args.rule, table, torch, dist)
    other = None
    if world == 1 and not args.one_rule:
        other = run_render(args, rank, world, barrier, 1 - args.rule, table, torch, dist)

    if rank == 0:
        body = report(args, world, args.rule, primary, scene_name)
        out = {"metric": "photons traced/sec + hitpoint updates/sec (whole node), SPPM Cornell 1080p",
               "value": body.pop("value"), "unit": "photons/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": body.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        out.update(body)
        assert bool(torch.isfinite(primary["image"]).all())
        if other is not None:
            ob = report(args, world, 1 - args.rule, other, scene_name)
            key = "per_photon_rule" if args.rule == 1 else "per_pass_rule"
            a, b = primary["image"], other["image"]
            rel_rmse = float(torch.sqrt(((a - b) ** 2).mean()) / torch.sqrt((b ** 2).mean()))
            out[key] = {"value": ob["value"], "unit": "photons/s", "ms_per_step": ob["ms_per_step"], "hitpoint_updates_per_s": ob["hitpoint_updates_per_s"],
                        "updates_per_photon": ob["updates_per_photon"], "roofline": ob["roofline"], "workload": ob["config"]["workload"],
                        "image_rel_rmse_between_the_rules": rel_rmse,
                        "note": "the same render with the other update rule, measured in the same run; the per-photon rule reproduces the reference's own image "
                                "(tests/test_gpu_reference_stream.py), the per-pass rule is another consistent estimator of the same radiance: after the same number of "
                                "passes the two images differ by the relative RMSE given here"}
        out["roofline"]["peak_measured"] = measured_fma_peak(torch)
        out["roofline"]["frac_of_measured_peak"] = out["roofline"]["achieved"] / out["roofline"]["peak_measured"]
        out["roofline"]["hbm_copy_GBps_measured"] = measured_stream_bandwidth(torch)
        if world == 1 and not args.no_converged and args.spheres <= 0:
            out["image_rel_rmse_vs_converged"] = image_quality()
        if world == 1 and not args.no_k_cap and args.spheres <= 0:
            out["k_cap"] = k_cap_enforced(args.width, args.height, args.photons, table, torch)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.width, args.height, args.photons, args.rule, table=table)
            if args.spheres <= 0:
                out["cpu_baseline"]["reference_check"] = reference_check()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
