"""Determinism soak of the gathers: the same render three times, every bit of the hitpoint state compared.
    python tools/soak.py [width height passes spheres rule]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import scene, sppm
W, H, passes, ns, rule = (int(x) for x in (sys.argv[1:6] + ["1920", "1080", "24", "0", "1"][len(sys.argv) - 1:]))
M = 1_000_000
table = scene.synthetic_spheres(ns) if ns else None
outs = []
for run in range(3):
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=rule), table) if table is not None else sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=rule))
    for i in range(passes):
        ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass()
    ctx.check()
    s = ctx.hitpoints(); img = ctx.resolve(passes - 1, M).cpu().numpy(); ctx.close()
    outs.append((s, img))
bad = 0
for r in (1, 2):
    for k in ("valid", "obj", "r2", "flux", "ld") + (("n",) if rule == 0 else ()):
        a, b = np.ascontiguousarray(outs[0][0][k]), np.ascontiguousarray(outs[r][0][k])
        ne = (a.view(np.uint64) != b.view(np.uint64)).sum() if a.dtype.kind == "f" else (a != b).sum()
        if ne: print("run", r, k, "differs in", int(ne)); bad += 1
    if (outs[0][1].view(np.uint64) != outs[r][1].view(np.uint64)).any(): print("run", r, "image differs"); bad += 1
print("soak", W, H, passes, ns, "rule", rule, "OK" if not bad else "MISMATCH")
sys.exit(1 if bad else 0)
