import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import sppm
quiet = 40; M = 1_000_000
ctx = sppm.SppmContext(sppm.make_config(1920, 1080, photons_per_pass=M))
for i in range(quiet):
    ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass()
ctx.check()
os.environ["SPPM_DEBUG"] = "1"; os.environ["SPPM_STAMPS"] = "1"; os.environ["SPPM_ITEM_TIMES"] = "/tmp/item_times.bin"
ctx.eye_pass(quiet); ctx.photon_pass(quiet, M); ctx.end_pass(); ctx.check()
t = np.fromfile("/tmp/item_times.bin", np.uint64).reshape(-1, 2).astype(np.int64)
ok = t[:, 1] > 0
idx = np.nonzero(ok)[0]
t = t[ok]
# the clock is per XCD: group items by clock domain (start values far apart), align each group at its first start
order = np.argsort(t[:, 0]); ts = t[order, 0]
grp = np.concatenate([[0], np.cumsum(np.diff(ts) > 200_000_000)])
g = np.empty(len(t), np.int64); g[order] = grp
s = np.zeros(len(t), np.int64); e = np.zeros(len(t), np.int64)
for k in range(g.max() + 1):
    m = g == k
    base = t[m, 0].min()
    s[m] = t[m, 0] - base; e[m] = t[m, 1] - base
print("clock domains", g.max() + 1, "items per domain", np.bincount(g))
dur = e - s
T = e.max()
print("items", len(t), "span", T, "sum dur", dur.sum(), "=> mean concurrency %.1f waves of %d slots" % (dur.sum() / T, 1024 * 4))
for q in (0.5, 0.9, 0.99, 0.999, 1.0):
    print("dur quantile", q, int(np.quantile(dur, q)))
edges = np.linspace(0, T, 21)
for b in range(20):
    lo, hi = edges[b], edges[b + 1]
    ov = np.clip(np.minimum(e, hi) - np.maximum(s, lo), 0, None).sum() / (hi - lo)
    print("t %5.1f%%  running waves %7.0f" % (100 * lo / T, ov))
late = np.argsort(e)[-12:]
for i in late:
    print("item", idx[i], "domain", g[i], "start %.1f%% end %.1f%% dur %d" % (100 * s[i] / T, 100 * e[i] / T, dur[i]))
for k in range(g.max() + 1):
    m = g == k
    print("domain", k, "last end %.1f%%" % (100 * e[m].max() / T), "work", dur[m].sum())
