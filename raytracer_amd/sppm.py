"""Host-side mirror of the reference's SPPM interface on top of the C ABI.

`StochasticPpm(width, height, samples, supersamples).render(...)` takes the arguments of
photonmap::sppm2::StochasticPpm<W,H>::render (/root/reference/src/sppm2.h:562-564) with the same
meaning, writes the same per-pass `.ppm`/`.hdr` files (:687-694) and returns 0 like the reference.
`SppmContext` exposes the individual stages and the parity taps.  PyTorch is used only to own device
buffers and the HIP stream; all compute is in libsppm_hip.so.
"""
import ctypes as C

import numpy as np

from . import capi, scene


KEY_SHIFT = 20  # a photon-hit record's key: photon << KEY_SHIFT | depth << 1 | orient (raytracer_amd/csrc/sppm_trace.hpp)


def _sphere_array(table):
    arr = (capi.SppmSphere * len(table))()
    for s, (r, p, e, c, t) in zip(arr, table):
        s.radius = r
        s.position[:] = p
        s.emission[:] = e
        s.color[:] = c
        s.reflection_type = t
    return arr


def make_config(width, height, *, samples=1, supersamples=1, camera_position=scene.CAMERA_POSITION, camera_dir=None,
                camera_up=scene.CAMERA_UP, screen_height=scene.SCREEN_HEIGHT, screen_dist=scene.SCREEN_DIST, alpha=0.7,
                initial_radius2=25.0, gather_radius2=25.0, gather_max=1 << 40, update_rule=0, double_nee=1,
                seed_eye=1000, seed_photon=2000, photons_per_pass=0, tile_first=0, tile_stride=0, sampler=0, ownership=0, algorithm=0,
                camera_model=0, aperture=4.0, focus_dist=180.0, seed_lens=3000, check_gather_max=0, tracer_variant=0):
    cfg = capi.SppmConfig()
    cfg.width, cfg.height, cfg.samples, cfg.supersamples = width, height, samples, supersamples
    cfg.camera_position[:] = camera_position
    cfg.camera_dir[:] = camera_dir if camera_dir is not None else scene.camera_direction()
    cfg.camera_up[:] = camera_up
    cfg.screen_height, cfg.screen_dist = screen_height, screen_dist
    cfg.alpha, cfg.initial_radius2 = alpha, initial_radius2
    cfg.gather_radius2, cfg.gather_max = gather_radius2, gather_max
    cfg.update_rule, cfg.double_nee = update_rule, double_nee
    cfg.seed_eye, cfg.seed_photon = seed_eye, seed_photon
    cfg.photons_per_pass = photons_per_pass
    cfg.tile_first, cfg.tile_stride = tile_first, tile_stride
    cfg.sampler = sampler  # 0: XorShift streams; 1: the reference's sequential mt19937 streams
    # 0: PinholeCamera; 1: thin lens (DoFCamera, src/cameras.h:66-139; src/sppm2.h:73-79 has aperture 4, focus 180)
    cfg.camera_model, cfg.aperture, cfg.focus_dist, cfg.seed_lens = camera_model, aperture, focus_dist, seed_lens
    # 0: the reference's K cap is enforced exactly whenever it could act (gather_max below the number of hitpoints);
    # 1: only count every query's candidates and report if any exceeds gather_max (brute force); 2: ignore the cap
    cfg.check_gather_max = check_gather_max
    cfg.tracer_variant = tracer_variant  # 1 (with sampler 1): create_photon of src/stochastic_ppm.h as shipped
    cfg.algorithm = algorithm  # 0: sppm2.h (stochastic); 1: progressive_photonmapping.h (classic PPM, a list of hitpoints)
    cfg.ownership = ownership  # with tile_stride > 1 -- 0: 8x8 screen tiles round-robin; 1: ranges of the hitpoints' Hilbert order
    return cfg


class SppmContext:
    """One GPU's SPPM state: scene, camera, hitpoints, record buffers (sppm_create ... sppm_destroy)."""

    def __init__(self, cfg, spheres=scene.CORNELL_SPHERES, device=None):
        import torch

        self.L = capi.lib()
        if not torch.cuda.is_available() or self.L.sppm_device_count() < 1:
            raise capi.SppmError("raytracer_amd needs a HIP device: there is no CPU path")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        torch.cuda.set_device(self.device)
        self.cfg = cfg
        self.P = cfg.width * cfg.height  # image indices
        self._spheres = _sphere_array(spheres)
        self.h = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(self.L.sppm_create(C.byref(cfg), self._spheres, len(self._spheres), C.c_void_p(stream), C.byref(self.h)))
        n = C.c_int64()
        capi.check(self.L.sppm_hitpoint_count(self.h, C.byref(n)))
        self.n_hitpoints = n.value  # == P for sppm2.h; the length of the hitpoint list for classic PPM

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.sppm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stages -------------------------------------------------------------------------------
    def camera_rays(self, sx=0, sy=0):
        out = self.torch.empty((self.P, 6), dtype=self.torch.float64, device=self.device)
        capi.check(self.L.sppm_camera_rays(self.h, sx, sy, out.data_ptr()))
        return out

    def intersect(self, rays):
        t = self.torch
        rays = rays.to(device=self.device, dtype=t.float64).contiguous()
        n = rays.shape[0]
        ids = t.empty((n,), dtype=t.int32, device=self.device)
        hit = t.empty((n, 7), dtype=t.float64, device=self.device)
        capi.check(self.L.sppm_intersect(self.h, rays.data_ptr(), n, ids.data_ptr(), hit.data_ptr()))
        return ids, hit

    def eye_pass(self, i):
        capi.check(self.L.sppm_eye_pass(self.h, i))

    def photon_pass(self, i, count, first=0):
        capi.check(self.L.sppm_photon_pass(self.h, i, first, count))

    def trace(self, i, first, count):
        capi.check(self.L.sppm_trace(self.h, i, first, count))

    def look_ahead(self, i, first, count):
        capi.check(self.L.sppm_look_ahead(self.h, i, first, count))

    def look_ahead_rows(self, i, first, count, rows):
        capi.check(self.L.sppm_look_ahead_rows(self.h, i, first, count, rows.data_ptr(), rows.shape[0]))

    def look_ahead_rows_to(self, i, first, count, rows):
        """rows: device tensor [world][cap][10]; block d gets the records rank d can need (sppm_look_ahead_rows_to)"""
        capi.check(self.L.sppm_look_ahead_rows_to(self.h, i, first, count, rows.data_ptr(), rows.shape[1]))

    def rows_ready_to(self):
        n = (C.c_int64 * max(1, int(self.cfg.tile_stride)))()
        capi.check(self.L.sppm_rows_ready_to(self.h, n))
        return [int(x) for x in n]

    def rows_ready(self):
        n = C.c_int64()
        capi.check(self.L.sppm_rows_ready(self.h, C.byref(n)))
        return n.value

    def gather(self):
        capi.check(self.L.sppm_gather(self.h))

    def end_pass(self):
        capi.check(self.L.sppm_end_pass(self.h))

    def check(self):
        capi.check(self.L.sppm_check(self.h))

    def resolve(self, i, photon_num):
        img = self.torch.empty((self.P, 3), dtype=self.torch.float64, device=self.device)
        capi.check(self.L.sppm_resolve(self.h, i, photon_num, img.data_ptr()))
        return img

    def end_pass_resolve(self, i, photon_num):
        """end_pass() then resolve(): under the per-pass rule one sweep over the hitpoints (sppm_end_pass_resolve)"""
        img = self.torch.empty((self.P, 3), dtype=self.torch.float64, device=self.device)
        capi.check(self.L.sppm_end_pass_resolve(self.h, i, photon_num, img.data_ptr()))
        return img

    def hdr_pixels(self, rgb):
        t = self.torch
        rgb = rgb.to(device=self.device, dtype=t.float64).contiguous()
        out = t.empty((rgb.shape[0], 4), dtype=t.uint8, device=self.device)
        capi.check(self.L.sppm_hdr_pixels(self.h, rgb.data_ptr(), rgb.shape[0], out.data_ptr()))
        return out

    # ---- records (multi-GPU exchange) ------------------------------------------------------------
    def pack_records(self, rows):
        """Move the buffer's records into the torch tensor rows[cap][10] (float64, this device);
        returns the count."""
        n = C.c_int64()
        capi.check(self.L.sppm_pack_records(self.h, rows.data_ptr(), rows.shape[0], C.byref(n)))
        return n.value

    def append_records(self, rows, n, photon_first, photon_end):
        capi.check(self.L.sppm_append_records(self.h, rows.data_ptr(), n, photon_first, photon_end))

    def owned_mask(self):
        """uint8 tensor [hitpoints] on this device: 1 where this rank owns the hitpoint (for sppm2.h: the pixel's)."""
        m = self.torch.empty((self.n_hitpoints,), dtype=self.torch.uint8, device=self.device)
        capi.check(self.L.sppm_owned_mask(self.h, m.data_ptr()))
        return m

    def export_owned_state(self):
        """float64 tensor [hitpoints][5]: (r^2, flux, N) where this rank owns the hitpoint, zeros elsewhere."""
        t = self.torch.empty((self.n_hitpoints, 5), dtype=self.torch.float64, device=self.device)
        capi.check(self.L.sppm_export_owned_state(self.h, t.data_ptr()))
        return t

    def import_state(self, state):
        state = state.to(device=self.device, dtype=self.torch.float64).contiguous()
        assert tuple(state.shape) == (self.n_hitpoints, 5)
        capi.check(self.L.sppm_import_state(self.h, state.data_ptr()))
        self._keep = state  # the copy is asynchronous

    def export_deposits(self, out=None):
        """update_rule 1: int64 tensor [4][hitpoints] = this rank's deposits of the pass so far (fixed-point flux, M)."""
        if out is None:
            out = self.torch.empty((4, self.n_hitpoints), dtype=self.torch.int64, device=self.device)
        capi.check(self.L.sppm_export_deposits(self.h, out.data_ptr()))
        return out

    def import_deposits(self, dep):
        assert dep.dtype == self.torch.int64 and tuple(dep.shape) == (4, self.n_hitpoints) and dep.is_contiguous() and dep.device == self.device
        capi.check(self.L.sppm_import_deposits(self.h, dep.data_ptr()))
        self._keep_dep = dep  # the copy is asynchronous

    def export_deposits_range(self, out, first, count):
        """the deposits of hitpoints [first, first + count) into out[4][count] (int64, contiguous)"""
        assert out.dtype == self.torch.int64 and tuple(out.shape) == (4, count) and out.is_contiguous()
        capi.check(self.L.sppm_export_deposits_range(self.h, out.data_ptr(), first, count))
        return out

    def import_deposits_range(self, dep, first, count):
        assert dep.dtype == self.torch.int64 and tuple(dep.shape) == (4, count) and dep.is_contiguous() and dep.device == self.device
        capi.check(self.L.sppm_import_deposits_range(self.h, dep.data_ptr(), first, count))

    def end_pass_resolve_range(self, i, photon_num, image, first, count):
        """end of the pass for hitpoints [first, first + count); image: the whole [P][3] tensor or None"""
        capi.check(self.L.sppm_end_pass_resolve_range(self.h, i, photon_num, image.data_ptr() if image is not None else None, first, count))

    def fixed_point_scale(self):
        v = C.c_double()
        capi.check(self.L.sppm_fixed_point_scale(self.h, C.byref(v)))
        return v.value

    # ---- taps ------------------------------------------------------------------------------------
    def hitpoint_pixels(self):
        """classic PPM: image index of every hitpoint of the list, in creation order."""
        out = np.zeros(self.n_hitpoints, np.int32)
        capi.check(self.L.sppm_download_hitpoint_pixels(self.h, out.ctypes.data))
        return out

    def hitpoints(self):
        P = self.n_hitpoints
        s = dict(valid=np.zeros(P, np.int32), obj=np.zeros(P, np.int32), hp_f=np.zeros((P, 10)), r2=np.zeros(P),
                 n=np.zeros(P, np.int32), flux=np.zeros((P, 3)), ld=np.zeros((P, 3)))
        capi.check(self.L.sppm_download_hitpoints(self.h, *[s[k].ctypes.data for k in ("valid", "obj", "hp_f", "r2", "n", "flux", "ld")]))
        return s

    def download_records(self):
        n = C.c_int64()
        capi.check(self.L.sppm_download_records(self.h, None, None, C.byref(n)))
        key = np.zeros(n.value, np.uint64)
        f = np.zeros((n.value, 9))
        capi.check(self.L.sppm_download_records(self.h, key.ctypes.data, f.ctypes.data, C.byref(n)))
        return key, f

    def tile_order(self):
        n = C.c_int64()
        capi.check(self.L.sppm_download_tile_order(self.h, None, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        capi.check(self.L.sppm_download_tile_order(self.h, out.ctypes.data, C.byref(n)))
        return out

    def accept_tap(self, cap):
        capi.check(self.L.sppm_accept_tap(self.h, cap))

    def accepted(self):
        n = C.c_int64()
        capi.check(self.L.sppm_download_accepted(self.h, None, None, C.byref(n)))
        key = np.zeros(n.value, np.uint64)
        pix = np.zeros(n.value, np.int32)
        capi.check(self.L.sppm_download_accepted(self.h, key.ctypes.data, pix.ctypes.data, C.byref(n)))
        return key, pix

    def counters(self):
        out = np.zeros(16, np.uint64)
        capi.check(self.L.sppm_counters(self.h, out.ctypes.data))
        names = ("rays", "photons", "records", "candidates", "updates", "gather_batches", "gather_ns", "gather_launches",
                 "selected", "rounds", "tile_launches", "trace_ns", "trace_launches", "photon_rays", "max_query_candidates")
        d = {k: int(v) for k, v in zip(names, out)}
        # [14] is the largest candidate count under check_gather_max = 1, else the queries the K cap acted on; [15] the pairs it dropped
        d["cap_queries"], d["cap_dropped_pairs"] = int(out[14]), int(out[15])
        return d

    def set_timing(self, on=True):
        capi.check(self.L.sppm_set_timing(self.h, 1 if on else 0))


def save_hdr_file(filename, image, width, height):
    """photonmap::utility::save_hdr_file, src/hdr.h:48 (image: host array [H*W][3], row-major)."""
    img = np.ascontiguousarray(image, dtype=np.float64)
    capi.check(capi.lib().sppm_save_hdr_file(filename.encode(), img.ctypes.data, width, height))


def save_ppm_file(filename, image, width, height):
    """edupt::save_ppm_file, src/ppm.h:22."""
    img = np.ascontiguousarray(image, dtype=np.float64)
    capi.check(capi.lib().sppm_save_ppm_file(filename.encode(), img.ctypes.data, width, height))


def progressive_render(filename, width, height, samples, supersamples, photon_num, gather_photon_radius,
                       gahter_max_photon_num, *, initial_radius=None, alpha=0.7, spheres=scene.CORNELL_SPHERES,
                       seed_eye=1000, seed_photon=2000, sampler=0, write_files=True):
    """Mirror of photonmap::progressive::render (src/progressive_photonmapping.h:709-821), the classic
    progressive photon mapping that src/main.cpp:87 runs: one eye pass that leaves a list of hitpoints, then
    photon_num photons, then the image.  initial_radius is progressive::InitialRadius (src/main.cpp:77 sets it
    to gather_photon_radius, the default here; like the search radius it is used as a SQUARED length).
    The reference also writes an image every few photons (:681-704); this mirror writes the final pair only.
    Returns the image [H*W][3] (host array)."""
    r2 = float(gather_photon_radius)
    cfg = make_config(width, height, samples=samples, supersamples=supersamples, alpha=alpha,
                      initial_radius2=r2 if initial_radius is None else float(initial_radius), gather_radius2=r2,
                      gather_max=int(gahter_max_photon_num), seed_eye=seed_eye, seed_photon=seed_photon,
                      photons_per_pass=photon_num, sampler=sampler, algorithm=1)
    ctx = SppmContext(cfg, spheres)
    try:
        ctx.eye_pass(0)
        ctx.photon_pass(0, photon_num)
        image = ctx.resolve(0, photon_num).cpu().numpy()
        ctx.check()
        if write_files:
            save_ppm_file(filename + ".ppm", image, width, height)
            save_hdr_file(filename + ".hdr", image, width, height)
    finally:
        ctx.close()
    return image


class StochasticPpm:
    """Mirror of photonmap::sppm2::StochasticPpm<Width, Height> (src/sppm2.h:33-697).

    Width/Height are constructor arguments here (template parameters in the reference).  The
    tunables are the reference's public members (src/sppm2.h:38-44).
    """

    def __init__(self, width, height, samples, supersamples, spheres=scene.CORNELL_SPHERES):
        self.width, self.height = width, height
        self.samples, self.supersamples = samples, supersamples
        self.alpha = 0.7
        self.initial_radius_2 = 25.0
        self.spheres = spheres
        self.seed_eye, self.seed_photon = 1000, 2000
        self.update_rule = 0
        self.camera_model, self.aperture, self.focus_dist, self.seed_lens = 0, 4.0, 180.0, 3000  # 1: DoFCamera (src/sppm2.h:73-79)
        self.last_image = None

    def render(self, filename, width, height, samples, supersamples, photon_num, gather_photon_radius,
               gahter_max_photon_num, eye_pass_count=100, write_pass_files=True):
        if (width, height) != (self.width, self.height):
            raise ValueError("width/height must match the constructor's (template parameters in the reference)")
        cfg = make_config(width, height, samples=samples, supersamples=supersamples, alpha=self.alpha,
                          initial_radius2=self.initial_radius_2, gather_radius2=float(gather_photon_radius),
                          gather_max=int(gahter_max_photon_num), update_rule=self.update_rule,
                          seed_eye=self.seed_eye, seed_photon=self.seed_photon, photons_per_pass=photon_num,
                          camera_model=self.camera_model, aperture=self.aperture, focus_dist=self.focus_dist, seed_lens=self.seed_lens)
        ctx = SppmContext(cfg, self.spheres)
        try:
            image = None
            for i in range(eye_pass_count):
                ctx.eye_pass(i)
                ctx.photon_pass(i, photon_num)
                ctx.end_pass()
                if write_pass_files or i == eye_pass_count - 1:
                    image = ctx.resolve(i, photon_num).cpu().numpy()
                    ctx.check()
                if write_pass_files:
                    save_ppm_file(f"{filename}_{i}.ppm", image, width, height)
                    save_hdr_file(f"{filename}_{i}.hdr", image, width, height)
            ctx.check()
            if image is not None:
                save_ppm_file(filename + ".ppm", image, width, height)
                save_hdr_file(filename + ".hdr", image, width, height)
            self.last_image = image
        finally:
            ctx.close()
        return 0
