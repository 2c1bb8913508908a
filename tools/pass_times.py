"""Per-pass wall time of the bench workload (host clock; every pass ends in a stream sync inside the gather)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import sppm
M = 1_000_000
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = sppm.SppmContext(sppm.make_config(1920, 1080, photons_per_pass=M))
ts = []
for i in range(n):
    t0 = time.perf_counter()
    ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass(); img = ctx.resolve(i, M)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ctx.check()
ts = np.array(ts) * 1e3
print("mean %.3f ms  median %.3f  p90 %.3f  max %.3f" % (ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max()))
late = ts[64:]
print("passes 64..: mean %.3f median %.3f; passes slower than median+1ms: %d, their excess sum %.1f ms of %.1f ms" % (
    late.mean(), np.median(late), (late > np.median(late) + 1).sum(), (late[late > np.median(late) + 1] - np.median(late)).sum(), late.sum()))
print("first 12:", np.round(ts[:12], 2))
print("slowest:", np.round(np.sort(ts)[-12:], 2))
