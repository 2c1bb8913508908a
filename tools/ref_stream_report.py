"""How close the GPU's reference-stream mode comes to the reference's golden vectors (for DESIGN.md)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import sppm
for tag in ("64x48", "64x48_r5", "320x240"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"ref_{tag}.npz"))
    w, h, photons, passes, traced, k, se, sp = [int(x) for x in g["meta"]]
    st = int(g["stride"][0])
    ctx = sppm.SppmContext(sppm.make_config(w, h, gather_radius2=float(g["gather_r2"][0]), gather_max=k, seed_eye=se, seed_photon=sp,
                                            photons_per_pass=photons, sampler=1))
    for i in range(passes):
        ctx.eye_pass(i); ctx.photon_pass(i, photons); ctx.end_pass(); ctx.check()
        s = ctx.hitpoints(); v = s["valid"] > 0; vs = v[::st]
        img = ctx.resolve(i, photons).cpu().numpy()
        ref = g[f"image_pass{i}"]
        fx = s["flux"][::st][vs]; rf = g[f"state_f_pass{i}"][:, 1:][vs]
        print(tag, "pass", i, "n equal:", np.array_equal(s["n"][v], g[f"state_n_pass{i}"][v]),
              "flux bit-identical %.4f max rel %.2e" % ((fx.view(np.uint64) == rf.view(np.uint64)).mean(), np.max(np.abs(fx - rf) / np.maximum(np.abs(rf), 1e-300))),
              "image bit-identical %.4f rel-RMSE %.2e" % ((img[::st].view(np.uint64) == ref.view(np.uint64)).mean(),
                                                       np.sqrt(np.mean((img[::st] - ref) ** 2) / np.mean(ref ** 2))))
    sppm.save_hdr_file("/tmp/o.hdr", img, w, h)
    mine = np.frombuffer(open("/tmp/o.hdr", "rb").read(), np.uint8)
    print(tag, "hdr bytes differing:", int((mine != g["final_hdr"]).sum()), "of", mine.size)
    ctx.close()
