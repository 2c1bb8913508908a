"""per-gather timings of the per-photon rule's first passes (SPPM_DEBUG lines) -- usage: SPPM_DEBUG=1 SPPM_BATCH_MIN=8192 python tools/r0_hole.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import sppm
ctx = sppm.SppmContext(sppm.make_config(1920, 1080, photons_per_pass=1_000_000, gather_max=1 << 40))
for i in range(4):
    ctx.eye_pass(i); ctx.photon_pass(i, 1_000_000)
ctx.check(); ctx.close()
