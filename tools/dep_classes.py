"""SPPM_STAMPS=1 python tools/dep_classes.py PASSES [spheres width height] -- the deposit kernel's diagnostic lines (shares of wave
cycles, staged records by class) per pass"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import scene, sppm
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
cfg = sppm.make_config(W, H, photons_per_pass=1_000_000, gather_max=1 << 40, update_rule=1)
ctx = sppm.SppmContext(cfg, scene.synthetic_spheres(ns)) if ns else sppm.SppmContext(cfg)
for i in range(int(sys.argv[1])):
    ctx.eye_pass(i); ctx.photon_pass(i, 1_000_000); ctx.end_pass()
ctx.check(); ctx.close()
