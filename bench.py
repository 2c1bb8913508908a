#!/usr/bin/env python3
"""bench.py -- throughput of the SPPM hot path on MI355X (contract: see the task statement).

A "step" is one SPPM pass at BASELINE config 2 geometry (Cornell box, 1920x1080, samples =
supersamples = 1, alpha 0.7, initial r^2 = 25 = search r^2): eye pass over every pixel, 1M photons
per GPU traced and applied to the hitpoints with the reference's per-photon update, image resolve.
The timed region is a FRESH render: passes 0..K-1 (default K = 256, the whole of config 2: "1M
photons/pass x 256 passes on one MI355X"); the W warmup passes run before it on a scratch context of
the same size (clocks, first touch), so they are not part of the render.  Pass cost falls as the
radii shrink (pass 0 is ~10x a late pass), so K matters and is reported in config.workload.
Nothing is uploaded inside the timed region (scene and state are resident in HBM).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 (weak scaling, config 3): a pass has N x photons photons; every rank owns a compact region of the
hitpoints (a range of their Morton order), traces its block of every batch and receives the other ranks'
photon-hit records through one RCCL all-gather per batch; photon paths are keyed by global photon index
and every hitpoint replays its records in key order, so the result does not depend on N (tested with two
ranks against one GPU, bit for bit).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(width, height, photons, seconds_target=12.0, table=None):
    """The oracle (a faithful single-thread port of the reference's kd-tree SPPM) on this box's host
    cores: eye pass 0 + kd build + a bounded number of photons of pass 0, same workload geometry."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from raytracer_amd.scene import CORNELL_SPHERES
    par = ol.make_params(width, height, sampler=1, math=1, photons_per_pass=photons)
    orc = ol.Oracle(par, ol.make_spheres(table if table is not None else CORNELL_SPHERES))
    t0 = time.perf_counter()
    orc.eye_pass(0)
    t_eye = time.perf_counter() - t0
    n, done, t_ph = 2000 if table is None else 100, 0, 0.0
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
            "sample": f"{width}x{height} pass 0: eye pass + kd-tree build {t_eye:.1f} s (not in value), then the first "
                      f"{done} photons in {t_ph:.1f} s, {upd} hitpoint updates ({upd / t_ph:.3g} updates/s)"}


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
    size the list: reported as create_ms.)  N > 1: the list is shared out by ranges of its Morton order, records
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


def measured_traffic(width, height, photons, steps):
    """HBM bytes per k_gather_tiles launch from the committed PMC passes (profiles/r01_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs of this script, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read from inside a
    normal run, so the figure is only reported for the workload it was measured on; otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None
    if [width, height, photons, steps] != [t["width"], t["height"], t["photons"], t["steps"]]:
        return None
    return t["gather_bytes_per_launch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--photons", type=int, default=1_000_000, help="photons per pass per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=("sppm", "ppm"), default="sppm", help="sppm: BASELINE config 2/3 (the metric's workload); ppm: config 4, classic PPM")
    ap.add_argument("--spheres", type=int, default=0, help="sppm workload: N > 0 replaces the Cornell scene by raytracer_amd.scene.synthetic_spheres(N) "
                    "(BASELINE config 5: --spheres 10000 --width 3840 --height 2160); rays then walk a grid over the spheres")
    ap.add_argument("--rule", type=int, choices=(0, 1), default=0, help="sppm workload: 0 = the reference's per-photon update (src/sppm2.h:391-421; N > 1: hitpoints "
                    "partitioned, photon records exchanged); 1 = the per-pass update (src/stochastic_ppm.h:721-741; N > 1: photons sharded, hitpoints "
                    "replicated, one all-reduce of the fixed-point deposits per pass -- north_star's split)")
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
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N > 1 through torch.distributed.run (one rank per GPU)")
    # SPPM_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("SPPM_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "ppm":
        return main_ppm(args, rank, world, barrier)

    from raytracer_amd import sppm
    from raytracer_amd.parallel import ShardedSppm

    W, H, M = args.width, args.height, args.photons
    from raytracer_amd import scene
    table = scene.synthetic_spheres(args.spheres) if args.spheres > 0 else scene.CORNELL_SPHERES
    scene_name = f"{args.spheres}-sphere synthetic scene (scene.synthetic_spheres, seed 20260517) in the Cornell box" if args.spheres > 0 else "Cornell-box"

    if args.warmup > 0:  # untimed: a scratch render of W passes
        scratch = ShardedSppm(W, H, photons_per_rank=M, rank=rank, world=world, spheres=table, rule=args.rule)
        scratch.total_passes = args.warmup
        for i in range(args.warmup):
            scratch.step(i)
        scratch.ctx.check()
        scratch.close()
    job = ShardedSppm(W, H, photons_per_rank=M, rank=rank, world=world, spheres=table, rule=args.rule)
    job.total_passes = args.steps
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
    if world > 1:
        job.gather_image()   # the final image on every rank: one all-reduce of the ranks' masked images
    job.ctx.check()   # drains the look-ahead streams too
    barrier()
    dt = time.perf_counter() - t0
    c1 = job.ctx.counters()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    # per-rank counter deltas summed over ranks (each rank applies updates to its own hitpoints)
    d = torch.tensor([c1[k] - c0[k] for k in ("rays", "records", "candidates", "updates")], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    rays, records, cand, upd = [float(x) for x in d.tolist()]

    if rank == 0:
        photons_total = float(M) * world * args.steps
        g_ns = c1["gather_ns"] - c0["gather_ns"]
        g_launch = max(1, c1["gather_launches"] - c0["gather_launches"])
        cand0, upd0 = c1["candidates"] - c0["candidates"], c1["updates"] - c0["updates"]
        # 28 B per candidate hitpoint/record pair examined, +32 B per accepted update
        alg_bytes = 28.0 * cand0 + 32.0 * upd0
        achieved = alg_bytes / max(g_ns, 1) if g_ns else 0.0  # bytes/ns == GB/s
        t_ns = c1["trace_ns"] - c0["trace_ns"]
        t_launch = max(1, c1["trace_launches"] - c0["trace_launches"])
        p_rays = c1["photon_rays"] - c0["photon_rays"]
        tflops = 219.0 * p_rays / max(t_ns, 1) / 1e3 if t_ns else 0.0  # flop/ns = GFLOP/s -> TFLOP/s
        out = {
            "metric": "photons traced/sec + hitpoint updates/sec (whole node), SPPM Cornell 1080p",
            "value": photons_total / dt, "unit": "photons/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{scene_name} SPPM {W}x{H}, {M} photons/pass/GPU, passes 0..{args.steps - 1} "
                                   f"of a fresh render, alpha 0.7, initial r^2 25, search r^2 25, {'per-photon update (sppm2.h)' if args.rule == 0 else 'per-pass update (stochastic_ppm.h:721-741), deposits in 64-bit fixed point'}, "
                                   f"XorShift streams", "photons_per_pass": M * world, "width": W, "height": H,
                       "parallelism": job.describe()},
            "hitpoint_updates_per_s": upd / dt, "rays_per_s": rays / dt, "records_per_photon": records / max(1.0, float(M) * args.steps * (world if job.replicated_trace else 1)),
            "updates_per_photon": upd / photons_total,
            "roofline": {"bound": "hbm", "kernel": "k_gather_tiles" if args.rule == 0 else "k_deposit_tiles", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "launches": int(g_launch), "avg_launch_ms": g_ns / 1e6 / g_launch,
                         "alg_bytes_per_launch": alg_bytes / g_launch,
                         "note": "rank 0; algorithmic bytes = 28 B x (record, hitpoint) pairs distance-tested + 32 B x accepted updates (SURVEY 8d); launch durations from HIP events on the kernel's stream"},
        }
        # second fraction asked for by SURVEY 8(d): the intersection work of the photon tracer against the
        # FP64 vector peak.  The tracer runs on its own stream CONCURRENTLY with the gather, so its launch
        # durations include the time it shares the CUs.
        out["roofline_trace"] = {"bound": "valu_f64", "kernel": "k_photon_trace", "achieved": tflops, "peak": 78.6, "unit": "TFLOP/s",
                                 "frac": tflops / 78.6, "launches": int(t_launch), "avg_launch_ms": t_ns / 1e6 / t_launch,
                                 "rays_per_launch": p_rays / t_launch,
                                 "note": "rank 0; 219 flop per ray (20 x 9 spheres + 19, SURVEY 8d) x rays cast by photon tracing; overlapped with the gather"}
        out["roofline"]["traffic"] = measured_traffic(W, H, M, args.steps) if world == 1 else None  # the PMC passes were made at N = 1
        bw = measured_stream_bandwidth(torch)
        out["roofline"]["peak_measured"] = bw
        out["roofline"]["frac_of_measured"] = achieved / bw
        out["roofline"]["note"] += "; peak_measured = device-to-device copy of 2 GiB on this box (read + write bytes per second)"
        if args.spheres > 0:
            out["roofline_trace"]["note"] = "rank 0; rays cast by photon tracing; the flop figure prices a ray at the Cornell scene's 219 flop (a grid walk tests fewer spheres per ray than brute force, so this is not a flop count for this scene)"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, M, table=table)
            if args.spheres <= 0:
                out["cpu_baseline"]["reference_check"] = reference_check()
        print(json.dumps(out), flush=True)
    job.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
