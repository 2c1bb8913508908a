"""Probe of the K-capped search: counters and timing at a given size (diagnostic)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from raytracer_amd import sppm
w, h, M, K, passes = [int(x) for x in sys.argv[1:6]]
ctx = sppm.SppmContext(sppm.make_config(w, h, photons_per_pass=M, gather_max=K))
for i in range(passes):
    t0 = time.time()
    ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass(); ctx.check()
    c = ctx.counters()
    print(f"pass {i}: {time.time() - t0:.3f} s, capped queries {c['cap_queries']}, dropped pairs {c['cap_dropped_pairs']}, records {c['records']}, updates {c['updates']}", flush=True)
ctx.close()
