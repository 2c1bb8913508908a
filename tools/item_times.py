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
raw4 = np.fromfile("/tmp/item_times.bin", np.uint64).reshape(-1, 4)
t = raw4[:, :2].astype(np.int64)
ok = t[:, 1] > 0
idx = np.nonzero(ok)[0]
t = t[ok]
raw4 = raw4[ok]
# the clock is per XCD: the low 4 bits of the end word carry the XCC id; align each XCD at its first start
g = t[:, 1] & 15
t[:, 1] >>= 4
base = t[:, 0].min()
s = t[:, 0] - base; e = t[:, 1] - base   # 100 MHz device-wide clock: 10 ns ticks
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
m = g == 0
ss = np.sort(s[m]); ee = np.sort(e[m])
print("XCD 0: k-th item start (cycles):", {k: int(ss[k]) for k in (0, 1, 3, 7, 15, 31, 63, 127, 255, 383, 511, 767, 1023, 2047) if k < len(ss)})
print("XCD 0: k-th item end   (cycles):", {k: int(ee[k]) for k in (0, 1, 3, 7, 15, 31, 63, 127, 255, 383, 511) if k < len(ee)})
first = np.argsort(s[m])[:16]
print("XCD 0 first items (index, start, dur):", [(int(idx[m][i]), int(s[m][i]), int(dur[m][i])) for i in first])
print("start time of the statically assigned first items (item index = global wave id), by index range:")
for lo in range(0, 4096, 256):
    m2 = (idx >= lo) & (idx < lo + 256)
    if m2.any():
        print("  waves %4d..%4d: n %3d start min %9d median %9d max %9d  XCDs %s" % (lo, lo + 255, m2.sum(), s[m2].min(), np.median(s[m2]), s[m2].max(), np.unique(g[m2]).tolist()))

item = (raw4[:, 2] & np.uint64(0xffffffff)).astype(np.int64); sel = (raw4[:, 2] >> np.uint64(32)).astype(np.int64)
rounds = (raw4[:, 3] >> np.uint64(32)).astype(np.int64); ncl = ((raw4[:, 3] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64); cand64 = (raw4[:, 3] & np.uint64(0xffff)).astype(np.int64)
print("longest items: (order index, tile, lane0, lanes, clusters, sel, rounds, cand/64, dur us)")
for i in np.argsort(dur)[-14:]:
    print("  ", idx[i], item[i] >> 12, (item[i] >> 6) & 63, (item[i] & 63) + 1, ncl[i], sel[i], rounds[i], cand64[i], dur[i] / 100.0)
print("corr(dur, sel) %.3f  corr(dur, rounds) %.3f  corr(dur, cand) %.3f corr(dur, clusters) %.3f" % (np.corrcoef(dur, sel)[0, 1], np.corrcoef(dur, rounds)[0, 1], np.corrcoef(dur, cand64)[0, 1], np.corrcoef(dur, ncl)[0, 1]))
A = np.stack([sel, rounds, cand64, ncl, np.ones_like(sel)], 1).astype(float)
coef, *_ = np.linalg.lstsq(A, dur.astype(float), rcond=None)
print("least squares dur[10ns] ~ %.2f*sel + %.2f*rounds + %.2f*cand64 + %.2f*clusters + %.1f" % tuple(coef))
