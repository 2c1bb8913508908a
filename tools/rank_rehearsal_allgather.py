"""Compute side of ONE rank of an N-GPU weak-scaling run in all-gather mode, rehearsed on one GPU with the batch
layout of raytracer_amd/parallel.py: the other ranks' records come from a helper context (untimed); timed per
pass: eye pass, trace own blocks, pack, append N blocks, bin + gather, resolve.  The RCCL all-gather itself
(~126 MB per rank and pass over xGMI) is NOT in the time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import sppm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
own = int(sys.argv[4]) if len(sys.argv) > 4 else 1
Mr = 1_000_000
M = Mr * N
ctx = sppm.SppmContext(sppm.make_config(1920, 1080, photons_per_pass=M, tile_first=rank, tile_stride=N, ownership=own))
helper = sppm.SppmContext(sppm.make_config(64, 48, photons_per_pass=M))
rows = [torch.empty((4 * Mr + 4096, 10), dtype=torch.float64, device="cuda") for _ in range(N)]
ctx.set_timing(True)
ts, applied, kept = [], 0, []
for i in range(passes):
    tp = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.eye_pass(i)
    torch.cuda.synchronize(); tp += time.perf_counter() - t0
    done = 0
    while done < M:
        batch = min((1 << 20) * N, max(32768, applied // 2))
        n = min(max(batch // N, 1024), (M - done) // N)
        cnt = [0] * N
        for r in range(N):
            if r != rank:
                helper.trace(i, done + r * n, n)
                cnt[r] = helper.pack_records(rows[r])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.trace(i, done + rank * n, n)
        cnt[rank] = ctx.pack_records(rows[rank])
        for r in range(N):
            ctx.append_records(rows[r], cnt[r], done + r * n, done + (r + 1) * n)
        applied += n * N; done += n * N
        ni, ndone = (i, done) if done < M else (i + 1, 0)
        nn = min(max(min((1 << 20) * N, max(32768, applied // 2)) // N, 1024), (M - ndone) // N)
        if ni < passes:
            ctx.look_ahead(ni, ndone + rank * nn, nn)
        ctx.gather()
        torch.cuda.synchronize(); tp += time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.end_pass(); ctx.resolve(i, M)
    torch.cuda.synchronize(); tp += time.perf_counter() - t0
    ts.append(tp)
ctx.check()
c = ctx.counters()
late = ts[passes // 2:]
t = sum(late) / len(late)
print(f"N={N} rank {rank} ownership {own}: compute per pass (last {len(late)}): {1e3 * t:.2f} ms -> {M / t / 1e6:.0f} M photons/s for the node before the all-gather; "
      f"records kept {c['records'] / passes / 1e6:.2f} M of {1.58 * N:.1f} M per pass; gather kernel {c['gather_ns'] / max(1, c['gather_launches']) / 1e6:.2f} ms x {c['gather_launches']}, trace {c['trace_ns'] / max(1, c['trace_launches']) / 1e6:.2f} ms")
