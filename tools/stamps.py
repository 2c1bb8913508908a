"""Per-phase cycle shares of k_gather_tiles (SPPM_DEBUG=1 SPPM_STAMPS=1): runs PASSES passes of the bench workload quietly,
then a few with the report on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import sppm
quiet = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = 1_000_000
ctx = sppm.SppmContext(sppm.make_config(1920, 1080, photons_per_pass=M))
for i in range(quiet):
    ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass()
ctx.check()
os.environ["SPPM_DEBUG"] = "1"; os.environ["SPPM_STAMPS"] = "1"
for i in range(quiet, quiet + 3):
    ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass()
ctx.check()
