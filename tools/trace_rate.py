"""Standalone rate of the photon tracer (no gather running beside it): photons/s and rays/s by batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SPPM_BATCH_PHOTONS", str(8 << 20))
import torch
from raytracer_amd import sppm
cfg = sppm.make_config(64, 48, photons_per_pass=1 << 30)
ctx = sppm.SppmContext(cfg)
ctx.set_timing(True)
ctx.eye_pass(0)
for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 23):
    for rep in range(2):
        c0 = ctx.counters()
        ctx.trace(1 + rep, 0, n)
        rows = torch.empty((4 * n + 4096, 10), dtype=torch.float64, device=ctx.device) if rep == 0 else rows
        cnt = ctx.pack_records(rows)
        c1 = ctx.counters()
    ns = c1["trace_ns"] - c0["trace_ns"]; rays = c1["photon_rays"] - c0["photon_rays"]
    print(f"{n:9d} photons: {ns/1e6:8.3f} ms  {n/ns*1e3:8.1f} Mphotons/s  {rays/ns*1e3:8.1f} Mrays/s  {rays/n:.2f} rays/photon {cnt/n:.2f} rec/photon  {219*rays/ns/1e3:.3f} TFLOP/s", flush=True)
    del rows
