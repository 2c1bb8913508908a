"""What ONE rank of an N-GPU weak-scaling run does (replicate mode has no data-path collective, so one GPU
can rehearse it): hitpoint tiles rank, rank+N, ...; N x 1M photons per pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import sppm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 48
M = 1_000_000 * N
ctx = sppm.SppmContext(sppm.make_config(1920, 1080, photons_per_pass=M, tile_first=0, tile_stride=N))
ctx.set_timing(True)
ts = []
for i in range(passes):
    t0 = time.perf_counter()
    ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass(); ctx.resolve(i, M)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ctx.check()
c = ctx.counters()
late = ts[passes // 2:]
print(f"N={N}: rank 0 pass time, last {len(late)} passes: {1e3 * sum(late) / len(late):.2f} ms -> {N * 1e6 * len(late) / sum(late) / 1e6:.1f} M photons/s for the node "
      f"(all {passes} passes: {1e3 * sum(ts) / passes:.2f} ms); gather {c['gather_ns'] / max(1, c['gather_launches']) / 1e6:.2f} ms x {c['gather_launches']}, trace {c['trace_ns'] / max(1, c['trace_launches']) / 1e6:.2f} ms x {c['trace_launches']}")
