"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol, the
host-side writers reproduce the reference's bytes, and nothing silently falls back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracer_amd import capi, parallel, sppm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sppm_hip.h")).read()
    declared = set(re.findall(r"\b(sppm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes structures against what a C compiler makes of include/sppm_hip.h: sizes and every field's offset."""
    import subprocess
    fields = lambda st: [f[0] for f in st._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "sppm_hip.h"', 'int main(void) {',
           'printf("%zu %zu\\n", sizeof(sppm_sphere), sizeof(sppm_config));']
    for name, st in (("sppm_sphere", capi.SppmSphere), ("sppm_config", capi.SppmConfig)):
        for f in fields(st):
            src.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    src += ['return 0; }']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    assert [int(x) for x in out[0].split()] == [C.sizeof(capi.SppmSphere), C.sizeof(capi.SppmConfig)]
    assert C.sizeof(capi.SppmSphere) == 88  # edupt::Sphere, src/sphere.h:15-27
    for line in out[1:]:
        if line:
            name, off = line.split()
            st = capi.SppmSphere if name.startswith("sppm_sphere.") else capi.SppmConfig
            assert getattr(st, name.split(".")[1]).offset == int(off), name


def test_writers_reproduce_reference_bytes(golden, tmp_path):
    g = golden("64x48")
    img = g["image_pass2"]
    sppm.save_hdr_file(str(tmp_path / "a.hdr"), img, 64, 48)
    sppm.save_ppm_file(str(tmp_path / "a.ppm"), img, 64, 48)
    assert open(tmp_path / "a.hdr", "rb").read() == g["final_hdr"].tobytes()
    assert open(tmp_path / "a.ppm", "rb").read() == g["final_ppm"].tobytes()


def test_writer_errors_are_reported(tmp_path):
    with pytest.raises(capi.SppmError):
        sppm.save_hdr_file(str(tmp_path / "no_such_dir" / "x.hdr"), np.zeros((4, 3)), 2, 2)


def test_no_cpu_fallback():
    L = capi.lib()
    if L.sppm_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = sppm.make_config(8, 8)
    sph = sppm._sphere_array(sppm.scene.CORNELL_SPHERES)
    h = C.c_void_p()
    rc = L.sppm_create(C.byref(cfg), sph, len(sph), None, C.byref(h))
    assert rc == -2 and b"no HIP device" in L.sppm_last_error()
    with pytest.raises(capi.SppmError):
        sppm.SppmContext(cfg)


def test_create_rejects_bad_arguments():
    L = capi.lib()
    sph = sppm._sphere_array(sppm.scene.CORNELL_SPHERES)
    h = C.c_void_p()
    for kw in (dict(gather_radius2=0.0), dict(alpha=-1.0), dict(update_rule=7), dict(samples=0)):
        cfg = sppm.make_config(8, 8, **kw)
        assert L.sppm_create(C.byref(cfg), sph, len(sph), None, C.byref(h)) == -1, kw
    cfg = sppm.make_config(8, 8)
    assert L.sppm_create(C.byref(cfg), sph, 0, None, C.byref(h)) == -1


def test_tile_ownership_partitions_the_image():
    for (w, h) in ((8, 8), (64, 48), (30, 21), (1920, 1080)):
        for world in (1, 2, 3, 8):
            own = parallel.owner_of_pixels(w, h, world)
            assert own.shape == (w * h,) and own.min() >= 0 and own.max() < world
            # an 8x8 block of pixels never straddles two ranks
            img = own.reshape(h, w)
            for r in range(0, h, 8):
                for c in range(0, w, 8):
                    assert len(np.unique(img[r:r + 8, c:c + 8])) == 1


def test_divide_by_pi_without_the_divider_is_exact():
    """raytracer_amd/csrc/detmath.h dm_div_pi (used by the HIP gather in place of `/ PI`, src/sppm2.h:413) returns the
    correctly rounded quotient: compared with the division on random arguments over 60 decades and on arguments whose
    quotient lies next to a rounding midpoint."""
    import ctypes as C
    import oracle_lib as ol
    L = ol.lib()
    L.orc_div_pi.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.orc_div_pi.restype = None
    rng = np.random.default_rng(7)

    def mismatches(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        a, b = np.empty_like(x), np.empty_like(x)
        L.orc_div_pi(x.ctypes.data, x.size, a.ctypes.data, b.ctypes.data)
        return int((a.view(np.uint64) != b.view(np.uint64)).sum())

    assert mismatches(rng.random(2_000_000) * 10.0 ** rng.integers(-30, 30, 2_000_000)) == 0
    kk = rng.integers(2 ** 52, 2 ** 53, 1_000_000)
    ld = np.longdouble
    xm = ((kk.astype(ld) + ld(0.5)) * ld(np.pi)).astype(np.float64) * 2.0 ** -52   # quotients next to a midpoint
    for x in (xm, np.nextafter(xm, np.inf), np.nextafter(xm, -np.inf)):
        assert mismatches(x) == 0
    assert mismatches(np.array([0.0, 1.0, np.pi, 1e-300, 1e300])) == 0
