#!/usr/bin/env python3
"""Render a long SPPM run on the GPU and store its image as a fixture: the yardstick both update rules are held to
(tests/test_gpu_converged.py, bench.py `image_rel_rmse_vs_converged`).

    python tools/render_converged.py --passes 8192 --out gpurun_out/converged_320x240_rule0.npz

Per-photon rule (the reference's live src/sppm2.h:391-421, the rule whose image is pinned against the compiled
reference), 320x240, 200k photons per pass, seeds DIFFERENT from the ones the tests render with, so that no test render
is a prefix of this one.  The image is stored as float32 (its own noise is orders above float32 rounding)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import sppm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--photons", type=int, default=200_000)
    ap.add_argument("--passes", type=int, default=8192)
    ap.add_argument("--rule", type=int, default=0)
    ap.add_argument("--seed-eye", type=int, default=777)
    ap.add_argument("--seed-photon", type=int, default=888)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    cfg = sppm.make_config(a.width, a.height, photons_per_pass=a.photons, update_rule=a.rule, seed_eye=a.seed_eye, seed_photon=a.seed_photon)
    ctx = sppm.SppmContext(cfg)
    t0 = time.time()
    snaps = {}
    for i in range(a.passes):
        ctx.eye_pass(i)
        ctx.photon_pass(i, a.photons)
        ctx.end_pass()
        if (i + 1) in (a.passes // 4, a.passes // 2, a.passes):
            snaps[i + 1] = ctx.resolve(i, a.photons).cpu().numpy()
            ctx.check()
            print(f"pass {i + 1}: {time.time() - t0:.1f} s", flush=True)
    img = snaps[a.passes]
    rr = lambda x, y: float(np.sqrt(np.mean((x - y) ** 2)) / np.sqrt(np.mean(y ** 2)))
    self_err = {str(k): rr(v, img) for k, v in snaps.items() if k != a.passes}
    print("rel-RMSE of the earlier snapshots against the last:", self_err)
    np.savez_compressed(a.out, image=img.astype(np.float32),
                        meta=np.array([a.width, a.height, a.photons, a.passes, a.rule, a.seed_eye, a.seed_photon], np.int64),
                        half_vs_full=np.array([rr(snaps[a.passes // 2], img)]), quarter_vs_full=np.array([rr(snaps[a.passes // 4], img)]))
    print("wrote", a.out, os.path.getsize(a.out), "bytes")
    ctx.close()


if __name__ == "__main__":
    main()
