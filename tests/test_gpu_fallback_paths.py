"""The exact fallbacks of the single-precision pair tests.

The deposit kernel (per-pass rule) and phase 1 of the ordered gather (per-photon rule) test a (record, hitpoint) pair in
single precision and trust the result only outside an error band; a pair inside a band is decided again in double
precision in the reference's operation order.  With the shipped bands that happens for about one row in six thousand, so
the ordinary parity tests barely touch those paths.  `raytracer_amd/libsppm_hip_wide.so` is the same source built with
the bands 3000 x wider (`make -C raytracer_amd/csrc wide`, part of `__graft_entry__.build()`): every other full-test
row and millions of distance-only groups per 1080p launch go through the fallbacks.  The parity tests must pass
unchanged -- bit for bit against the oracle -- with that library too."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDE = os.path.join(ROOT, "raytracer_amd", "libsppm_hip_wide.so")
FILES = ["tests/test_gpu_per_pass_rule.py", "tests/test_gpu_parity.py", "tests/test_gpu_k_cap.py", "tests/test_gpu_classic_ppm.py"]


@pytest.mark.gpu
def test_parity_holds_when_the_fallbacks_decide_most_pairs():
    assert os.path.exists(WIDE), "raytracer_amd/libsppm_hip_wide.so is missing: run __graft_entry__.build()"
    env = dict(os.environ, SPPM_HIP_LIB=WIDE)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "300", "--timeout-method", "thread"] + FILES,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
