"""The reference's call sites (src/main.cpp:87-93) compiled against the C++ drop-in headers (include/sppm2_hip.hpp,
include/progressive_photonmapping_hip.hpp; examples/*.cpp, built by __graft_entry__.build()) and RUN: the files they
write equal, byte for byte, what the Python mirror of the same interface writes for the same arguments."""
import os
import subprocess

import pytest

from raytracer_amd import sppm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe(name):
    path = os.path.join(ROOT, "examples", name)
    assert os.path.exists(path), f"{path} is missing: __graft_entry__.build() compiles it"
    return path


def _bytes(path):
    with open(path, "rb") as f:
        return f.read()


def test_render_sppm_example_equals_python_mirror(tmp_path):
    passes, photons = 2, 30000
    out = str(tmp_path / "cpp_sppm")
    r = subprocess.run([_exe("render_sppm"), str(passes), str(photons), out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    py = sppm.StochasticPpm(320, 240, 1, 1)
    ref = str(tmp_path / "py_sppm")
    assert py.render(ref, 320, 240, 1, 1, photons, 25, 10000, passes, write_pass_files=False) == 0
    for ext in (".hdr", ".ppm"):
        assert _bytes(out + ext) == _bytes(ref + ext), ext
    assert len(_bytes(out + ".hdr")) > 320 * 240 * 4


def test_render_ppm_example_equals_python_mirror(tmp_path):
    supersamples, photons = 1, 40000
    out = str(tmp_path / "cpp_ppm")
    r = subprocess.run([_exe("render_ppm"), str(supersamples), str(photons), out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ref = str(tmp_path / "py_ppm")
    sppm.progressive_render(ref, 320, 240, 1, supersamples, photons, 5, 10000)
    for ext in (".hdr", ".ppm"):
        assert _bytes(out + ext) == _bytes(ref + ext), ext


def test_first_hits_example_against_the_compiled_reference(golden):
    """examples/first_hits.cpp: edupt::PinholeCamera::get_ray and edupt::intersect_scene by the reference's names
    (include/sppm2_hip.hpp) against what the compiled reference produced for the same calls (tests/golden/ref_64x48.npz,
    made by oracle/ref_driver.cpp: cam_rays, cam_hit_id, cam_hit_f) -- the reference's data, not the Python mirror."""
    import numpy as np
    g = golden("64x48")
    r = subprocess.run([_exe("first_hits"), "64", "48"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    rows = [line.split() for line in r.stdout.strip().splitlines()]
    assert len(rows) == 64 * 48
    ids = np.array([int(x[0]) for x in rows], np.int32)
    dirs = np.array([[float.fromhex(v) for v in x[1:4]] for x in rows])
    dist = np.array([float.fromhex(x[4]) for x in rows])
    assert np.array_equal(ids, g["cam_hit_id"]), "first-hit sphere ids"
    assert np.array_equal(dirs.view(np.uint64), g["cam_rays"][:, 3:6].view(np.uint64)), "PinholeCamera::get_ray directions, bit for bit"
    hit = ids >= 0
    assert np.array_equal(dist[hit].view(np.uint64), np.ascontiguousarray(g["cam_hit_f"][hit, 0]).view(np.uint64)), "hit distances, bit for bit"
