import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import sppm
W, H, M = 1920, 1080, 1_000_000
def run(split):
    c = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    out = []
    for i in range(2):
        c.eye_pass(i)
        if split:
            c.photon_pass(i, M // 2, first=0); c.photon_pass(i, M // 2, first=M // 2)
        else:
            c.photon_pass(i, M)
        d = c.export_deposits().cpu().numpy()
        h0 = c.hitpoints()
        c.end_pass()
        h1 = c.hitpoints()
        out.append((d, h0, h1))
    c.check(); c.close()
    return out
ref = run(False)
for trial in range(3):
    got = run(True)
    for i in range(2):
        for name, x, y in (("dep", ref[i][0], got[i][0]),) + tuple((f"before.{k}", ref[i][1][k], got[i][1][k]) for k in ("valid", "r2", "n", "flux")) + tuple((f"after.{k}", ref[i][2][k], got[i][2][k]) for k in ("valid", "r2", "n", "flux")):
            bad = np.argwhere(np.ascontiguousarray(x).view(np.uint8).reshape(x.shape[0] if x.ndim == 1 else x.shape[0], -1).any(axis=1) != np.ascontiguousarray(y).view(np.uint8).reshape(y.shape[0], -1).any(axis=1)) if False else None
            ne = (x != y)
            if ne.any():
                idx = np.argwhere(ne)[:3]
                print(f"trial {trial} pass {i} {name}: {int(ne.sum())} differ at {idx.tolist()} ref {x[tuple(idx[0])]} got {y[tuple(idx[0])]}")
print("done", os.environ.get("SPPM_PREFETCH"), os.environ.get("SPPM_AHEAD"))
