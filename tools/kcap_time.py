"""Time of passes with the reference's K = 10000 enforced at 1080p (both rules) and what the filter lets through.
    SPPM_DEBUG=1 python tools/kcap_time.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd import sppm
W, H, M, passes = 1920, 1080, 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rule in (0, 1):
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, gather_max=10000, update_rule=rule))
    for i in range(passes):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass(); ctx.check()
        torch.cuda.synchronize()
        c = ctx.counters()
        print(f"rule {rule} pass {i}: {1e3 * (time.perf_counter() - t0):.0f} ms, records {c['records']}, capped queries {c['cap_queries']}, pairs withheld {c['cap_dropped_pairs']}", flush=True)
    ctx.close()
