#!/usr/bin/env python3
"""Largest number of candidate hitpoints any photon hit holds (the quantity the reference's K cap acts on,
src/kdtree.h:128-133), counted exactly by the library's check mode.  Usage: kcap_report.py [photons]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracer_amd import sppm  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
for name, w, h, kw in (("SPPM 320x240 r2=25 (config 1)", 320, 240, {}),
                       ("SPPM 1920x1080 r2=25 (config 2)", 1920, 1080, {}),
                       ("classic PPM 1920x1080 r2=5, 1 sample (config 4)", 1920, 1080, dict(algorithm=1, gather_radius2=5.0, initial_radius2=5.0))):
    ctx = sppm.SppmContext(sppm.make_config(w, h, photons_per_pass=M, gather_max=10000, check_gather_max=1, **kw))
    ctx.eye_pass(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.photon_pass(0, M)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    try:
        ctx.check()
        verdict = "<= K = 10000: the uncapped set is the reference's set"
    except Exception as e:  # noqa: BLE001
        verdict = "EXCEEDS K = 10000: the reference's cap acts here"
    print(f"{name}: {ctx.n_hitpoints} hitpoints, {M} photons: max candidates per photon hit = {ctx.counters()['max_query_candidates']} ({verdict}); checked in {dt:.1f} s", flush=True)
    ctx.close()
