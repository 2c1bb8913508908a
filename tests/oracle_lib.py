"""ctypes binding of oracle/liborc.so (the CPU checker).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class OrcSphere(C.Structure):
    _fields_ = [("radius", C.c_double), ("pos", C.c_double * 3), ("emission", C.c_double * 3),
                ("color", C.c_double * 3), ("type", C.c_int32), ("pad", C.c_int32)]


class OrcParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("supersamples", C.c_int32),
                ("cam_pos", C.c_double * 3), ("cam_dir", C.c_double * 3), ("cam_up", C.c_double * 3),
                ("screen_height", C.c_double), ("screen_dist", C.c_double),
                ("alpha", C.c_double), ("initial_radius2", C.c_double),
                ("gather_radius2", C.c_double), ("gather_max", C.c_int64),
                ("sampler", C.c_int32), ("math", C.c_int32), ("rule", C.c_int32), ("double_nee", C.c_int32),
                ("seed_eye", C.c_uint32), ("seed_photon", C.c_uint32),
                ("photons_per_pass", C.c_int64),
                ("camera_model", C.c_int32), ("seed_lens", C.c_uint32), ("aperture", C.c_double), ("focus_dist", C.c_double),
                ("tracer", C.c_int32), ("accum", C.c_int32), ("stream_layout", C.c_int32), ("pad2", C.c_int32)]


class OrcTap(C.Structure):
    _fields_ = [("rec_cap", C.c_int64), ("rec_n", C.c_int64),
                ("rec_photon", C.c_void_p), ("rec_depth", C.c_void_p), ("rec_obj", C.c_void_p), ("rec_f", C.c_void_p),
                ("acc_cap", C.c_int64), ("acc_n", C.c_int64), ("acc_rec", C.c_void_p), ("acc_pix", C.c_void_p),
                ("draws_cap", C.c_int64), ("draws", C.c_void_p), ("overflow", C.c_int64)]


def lib():
    """Load liborc.so, building it first if the source is newer (gcc only, a second or two)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "liborc.so")
    src = os.path.join(ORACLE_DIR, "sppm_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liborc.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcSphere), C.c_int]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_eye_pass.argtypes = [C.c_void_p, C.c_int]
    L.orc_photon_pass.restype = C.c_int
    L.orc_photon_pass.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.POINTER(OrcTap)]
    L.orc_end_pass.argtypes = [C.c_void_p]
    L.orc_resolve.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    L.orc_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    L.orc_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_get_deposits.restype = C.c_int
    L.orc_get_deposits.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_fixed_scale.restype = C.c_double
    L.orc_fixed_scale.argtypes = [C.c_void_p]
    L.orc_get_deposits_d.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_get_nd.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_photon_pass_mt.restype = C.c_int
    L.orc_photon_pass_mt.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.orc_max_threads.restype = C.c_int
    L.orc_mt_samples.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
    L.orc_xorshift.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_xs_stream.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]
    L.orc_camera_rays.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_int, C.c_void_p]
    L.orc_intersect.argtypes = [C.POINTER(OrcSphere), C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.orc_hdr_pixels.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.orc_to_int.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.orc_hdr_file.restype = C.c_int64
    L.orc_hdr_file.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    L.orc_ppm_file.restype = C.c_int64
    L.orc_ppm_file.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    L.orc_ppm_create.restype = C.c_void_p
    L.orc_ppm_create.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcSphere), C.c_int, C.c_int]
    L.orc_ppm_eye_pass.restype = C.c_int
    L.orc_ppm_eye_pass.argtypes = [C.c_void_p]
    L.orc_ppm_kd_order.restype = C.c_int
    L.orc_ppm_kd_order.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_ppm_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.orc_ppm_resolve.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_spheres(table):
    """table: iterable of (radius, pos, emission, color, type)."""
    arr = (OrcSphere * len(table))()
    for s, (r, p, e, c, t) in zip(arr, table):
        s.radius = r
        s.pos[:] = p
        s.emission[:] = e
        s.color[:] = c
        s.type = t
    return arr


def make_params(width, height, *, samples=1, supersamples=1, cam_pos=(50.0, 52.0, 220.0), cam_dir=None,
                cam_up=(0.0, 1.0, 0.0), screen_height=30.0, screen_dist=40.0, alpha=0.7, initial_radius2=25.0,
                gather_radius2=25.0, gather_max=1 << 40, sampler=1, math=1, rule=0, double_nee=1,
                seed_eye=1000, seed_photon=2000, photons_per_pass=0, camera_model=0, aperture=4.0, focus_dist=180.0, seed_lens=3000,
                tracer=0, accum=0, stream_layout=0):
    p = OrcParams()
    p.width, p.height, p.samples, p.supersamples = width, height, samples, supersamples
    if cam_dir is None:
        # normalize(Vec(0, -0.04, -1)) exactly as vec.h:19 evaluates it
        v = np.array([0.0, -0.04, -1.0])
        inv = 1.0 / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        cam_dir = tuple(v * inv)
    p.cam_pos[:] = cam_pos
    p.cam_dir[:] = cam_dir
    p.cam_up[:] = cam_up
    p.screen_height, p.screen_dist = screen_height, screen_dist
    p.alpha, p.initial_radius2 = alpha, initial_radius2
    p.gather_radius2, p.gather_max = gather_radius2, gather_max
    p.sampler, p.math, p.rule, p.double_nee = sampler, math, rule, double_nee
    p.seed_eye, p.seed_photon = seed_eye, seed_photon
    p.photons_per_pass = photons_per_pass
    p.camera_model, p.seed_lens, p.aperture, p.focus_dist = camera_model, seed_lens, aperture, focus_dist
    # tracer 1 / accum 1 / stream_layout 1: src/stochastic_ppm.h as shipped (oracle/sppm_oracle.c, "Modes")
    p.tracer, p.accum, p.stream_layout = tracer, accum, stream_layout
    return p


class Oracle:
    def __init__(self, params, spheres):
        self.L = lib()
        self.params = params
        self.spheres = spheres
        self.P = params.width * params.height
        self.h = self.L.orc_create(C.byref(params), spheres, len(spheres))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def eye_pass(self, i):
        self.L.orc_eye_pass(self.h, i)

    def photon_pass(self, i, count, first=0, tap=False, rec_cap=None, acc_cap=None):
        if not tap:
            rc = self.L.orc_photon_pass(self.h, i, first, count, None)
            assert rc == 0
            return None
        rec_cap = rec_cap or 8 * count + 64
        acc_cap = acc_cap or 4_000_000
        t = OrcTap()
        out = dict(rec_photon=np.zeros(rec_cap, np.int32), rec_depth=np.zeros(rec_cap, np.int32),
                   rec_obj=np.zeros(rec_cap, np.int32), rec_f=np.zeros((rec_cap, 12)),
                   acc_rec=np.zeros(acc_cap, np.int64), acc_pix=np.zeros(acc_cap, np.int32),
                   draws=np.zeros(count, np.int32))
        t.rec_cap, t.acc_cap, t.draws_cap = rec_cap, acc_cap, count
        for k in ("rec_photon", "rec_depth", "rec_obj", "rec_f", "acc_rec", "acc_pix", "draws"):
            setattr(t, k, out[k].ctypes.data)
        rc = self.L.orc_photon_pass(self.h, i, first, count, C.byref(t))
        assert rc == 0 and t.overflow == 0, (rc, t.overflow)
        for k in ("rec_photon", "rec_depth", "rec_obj", "rec_f"):
            out[k] = out[k][:t.rec_n]
        out["acc_rec"] = out["acc_rec"][:t.acc_n]
        out["acc_pix"] = out["acc_pix"][:t.acc_n]
        return out

    def photon_pass_mt(self, i, count, first=0, threads=None):
        """the same pass on `threads` host cores (OpenMP; default: all the runtime offers); returns the seconds spent"""
        sec = C.c_double()
        rc = self.L.orc_photon_pass_mt(self.h, i, first, count, threads or self.L.orc_max_threads(), C.byref(sec))
        assert rc == 0
        return sec.value

    def end_pass(self):
        self.L.orc_end_pass(self.h)

    def resolve(self, i, photon_num):
        img = np.zeros((self.P, 3))
        self.L.orc_resolve(self.h, i, photon_num, _p(img))
        return img

    def state(self):
        P = self.P
        s = dict(valid=np.zeros(P, np.int32), obj=np.zeros(P, np.int32), hp_f=np.zeros((P, 10)), r2=np.zeros(P),
                 n=np.zeros(P, np.int32), flux=np.zeros((P, 3)), ld=np.zeros((P, 3)))
        self.L.orc_get_state(self.h, _p(s["valid"]), _p(s["obj"]), _p(s["hp_f"]), _p(s["r2"]), _p(s["n"]),
                             _p(s["flux"]), _p(s["ld"]))
        return s

    def deposits(self):
        """rule 1: (dep[P][4] int64 = fixed-point flux sums and M of the pass so far, the clamp flag)"""
        d = np.zeros((self.P, 4), np.int64)
        flag = self.L.orc_get_deposits(self.h, _p(d))
        return d, flag

    def deposits_double(self):
        """rule 1 with accum=1: the pass's double sums so far (accumulated_flux of stochastic_ppm.h:453), [P][3]"""
        f = np.zeros((self.P, 3))
        self.L.orc_get_deposits_d(self.h, _p(f))
        return f

    def state_nd(self):
        """rule 1: N of every hitpoint as a double"""
        nd = np.zeros(self.P)
        self.L.orc_get_nd(self.h, _p(nd))
        return nd

    def fixed_scale(self):
        return self.L.orc_fixed_scale(self.h)

    def counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.orc_counters(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(eye_draws=a.value, updates=b.value, rays=c.value)


class PpmOracle(Oracle):
    """Classic progressive photon mapping (src/progressive_photonmapping.h): a LIST of hitpoints in creation
    order, several per pixel; photon_pass(0, count, ...) of the base class runs its photons (acc_pix of the tap
    then holds hitpoint ids)."""

    def __init__(self, params, spheres, cap=None):
        self.L = lib()
        self.params = params
        self.spheres = spheres
        self.P = params.width * params.height
        self.cap = cap or 16 * self.P * params.samples * params.supersamples ** 2
        self.h = self.L.orc_ppm_create(C.byref(params), spheres, len(spheres), self.cap)
        self.nh = 0

    def eye_pass(self, i=0):
        assert i == 0, "classic PPM traces its hitpoints once"
        self.nh = self.L.orc_ppm_eye_pass(self.h)
        assert self.nh >= 0, "hitpoint capacity too small"
        return self.nh

    def kd_order(self):
        o = np.zeros(self.nh, np.int32)
        fallback = self.L.orc_ppm_kd_order(self.h, _p(o))
        assert fallback == 0, "std::sort's heapsort fallback was reached: the vector order is not pinned"
        return o

    def state(self):
        n = self.nh
        s = dict(pix=np.zeros(n, np.int32), obj=np.zeros(n, np.int32), hp_f=np.zeros((n, 11)), r2=np.zeros(n),
                 n=np.zeros(n, np.int32), flux=np.zeros((n, 3)))
        self.L.orc_ppm_get_state(self.h, _p(s["pix"]), _p(s["obj"]), _p(s["hp_f"]), _p(s["r2"]), _p(s["n"]), _p(s["flux"]))
        return s

    def end_pass(self):
        pass

    def resolve(self, photon_num, order=1):
        """order 0: the reference's kd-sorted vector order; 1: creation order (per-pixel sums on the GPU)."""
        img = np.zeros((self.P, 3))
        self.L.orc_ppm_resolve(self.h, photon_num, order, _p(img))
        return img


def mt_samples(seed, n):
    out = np.zeros(n)
    lib().orc_mt_samples(seed, n, _p(out))
    return out


def xorshift(seed, n):
    u = np.zeros(n, np.uint32)
    f = np.zeros(n)
    lib().orc_xorshift(seed, n, _p(u), _p(f))
    return u, f


def xs_stream(base, key, n):
    f = np.zeros(n)
    lib().orc_xs_stream(base, key, n, _p(f))
    return f


def camera_rays(params, sx=0, sy=0):
    out = np.zeros((params.width * params.height, 6))
    lib().orc_camera_rays(C.byref(params), sx, sy, _p(out))
    return out


def intersect(spheres, rays):
    rays = np.ascontiguousarray(rays, dtype=np.float64)
    n = rays.shape[0]
    ids = np.zeros(n, np.int32)
    f = np.zeros((n, 7))
    lib().orc_intersect(spheres, len(spheres), _p(rays), n, _p(ids), _p(f))
    return ids, f


def hdr_pixels(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.zeros((rgb.shape[0], 4), np.uint8)
    lib().orc_hdr_pixels(_p(rgb), rgb.shape[0], _p(out))
    return out


def to_int(x):
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.zeros(x.shape[0], np.int32)
    lib().orc_to_int(_p(x), x.shape[0], _p(out))
    return out


def hdr_file(image, w, h):
    image = np.ascontiguousarray(image, dtype=np.float64)
    cap = 4096 + h * (4 + 4 * (w + w // 127 + 2))
    buf = np.zeros(cap, np.uint8)
    n = lib().orc_hdr_file(_p(image), w, h, _p(buf), cap)
    assert n >= 0
    return bytes(buf[:n])


def ppm_file(image, w, h):
    image = np.ascontiguousarray(image, dtype=np.float64)
    cap = 64 + 16 * w * h
    buf = np.zeros(cap, np.uint8)
    n = lib().orc_ppm_file(_p(image), w, h, _p(buf), cap)
    assert n >= 0
    return bytes(buf[:n])
