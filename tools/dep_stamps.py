"""Diagnostic: SPPM_STAMPS report of the deposit kernel over the first passes of the 1080p render (rule 1).
    SPPM_STAMPS=1 python tools/dep_stamps.py [passes]"""
import sys
sys.path.insert(0, ".")
from raytracer_amd import sppm
W, H, M, passes = 1920, 1080, 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
for i in range(passes):
    ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass()
ctx.check(); ctx.close()
