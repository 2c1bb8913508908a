"""SPPM_STAMPS=1 python tools/dep_classes.py PASSES -- the deposit kernel's diagnostic lines (shares of wave cycles, staged records by class) per pass"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import sppm
ctx = sppm.SppmContext(sppm.make_config(1920, 1080, photons_per_pass=1_000_000, gather_max=1 << 40, update_rule=1))
for i in range(int(sys.argv[1])):
    ctx.eye_pass(i); ctx.photon_pass(i, 1_000_000); ctx.end_pass()
ctx.check(); ctx.close()
