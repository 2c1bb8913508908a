"""ctypes binding of raytracer_amd/libsppm_hip.so (include/sppm_hip.h).

Loading fails loudly if the library is missing: there is no CPU fallback in the product.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPPM_HIP_LIB") or os.path.join(_HERE, "libsppm_hip.so")  # SPPM_HIP_LIB: another build of the same library (tuning experiments)

SYMBOLS = [
    "sppm_last_error", "sppm_device_count", "sppm_create", "sppm_destroy", "sppm_camera_rays", "sppm_intersect",
    "sppm_eye_pass", "sppm_photon_pass", "sppm_trace", "sppm_pack_records", "sppm_append_records", "sppm_gather",
    "sppm_end_pass", "sppm_resolve", "sppm_end_pass_resolve", "sppm_hdr_pixels", "sppm_save_hdr_file", "sppm_save_ppm_file", "sppm_check",
    "sppm_download_hitpoints", "sppm_download_records", "sppm_owned_mask", "sppm_look_ahead", "sppm_look_ahead_rows", "sppm_rows_ready", "sppm_look_ahead_rows_to", "sppm_rows_ready_to", "sppm_download_tile_order", "sppm_accept_tap", "sppm_download_accepted", "sppm_counters",
    "sppm_set_timing", "sppm_measure_fma_peak", "sppm_intersect_static", "sppm_export_deposits", "sppm_import_deposits", "sppm_fixed_point_scale", "sppm_hitpoint_count", "sppm_download_hitpoint_pixels", "sppm_export_owned_state", "sppm_import_state",
    "sppm_export_deposits_range", "sppm_import_deposits_range", "sppm_end_pass_resolve_range",
]


class SppmSphere(C.Structure):
    _fields_ = [("radius", C.c_double), ("position", C.c_double * 3), ("emission", C.c_double * 3),
                ("color", C.c_double * 3), ("reflection_type", C.c_int32), ("pad_", C.c_int32)]


class SppmConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("supersamples", C.c_int32),
                ("camera_position", C.c_double * 3), ("camera_dir", C.c_double * 3), ("camera_up", C.c_double * 3),
                ("screen_height", C.c_double), ("screen_dist", C.c_double),
                ("alpha", C.c_double), ("initial_radius2", C.c_double),
                ("gather_radius2", C.c_double), ("gather_max", C.c_int64),
                ("update_rule", C.c_int32), ("double_nee", C.c_int32),
                ("seed_eye", C.c_uint32), ("seed_photon", C.c_uint32),
                ("photons_per_pass", C.c_int64),
                ("tile_first", C.c_int32), ("tile_stride", C.c_int32),
                ("check_gather_max", C.c_int32), ("sampler", C.c_int32), ("ownership", C.c_int32), ("algorithm", C.c_int32),
                ("camera_model", C.c_int32), ("seed_lens", C.c_uint32), ("aperture", C.c_double), ("focus_dist", C.c_double),
                ("tracer_variant", C.c_int32), ("pad2_", C.c_int32)]


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` (or __graft_entry__.build()). "
            "raytracer_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.sppm_last_error.restype = C.c_char_p
    L.sppm_device_count.restype = C.c_int
    L.sppm_create.argtypes = [C.POINTER(SppmConfig), C.POINTER(SppmSphere), i32, vp, C.POINTER(vp)]
    L.sppm_destroy.argtypes = [vp]
    L.sppm_camera_rays.argtypes = [vp, i32, i32, vp]
    L.sppm_intersect.argtypes = [vp, vp, i64, vp, vp]
    L.sppm_eye_pass.argtypes = [vp, i32]
    L.sppm_photon_pass.argtypes = [vp, i32, i64, i64]
    L.sppm_trace.argtypes = [vp, i32, i64, i64]
    L.sppm_pack_records.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.sppm_append_records.argtypes = [vp, vp, i64, i64, i64]
    L.sppm_gather.argtypes = [vp]
    L.sppm_end_pass.argtypes = [vp]
    L.sppm_resolve.argtypes = [vp, i32, i64, vp]
    L.sppm_end_pass_resolve.argtypes = [vp, i32, i64, vp]
    L.sppm_hdr_pixels.argtypes = [vp, vp, i64, vp]
    L.sppm_save_hdr_file.argtypes = [C.c_char_p, vp, i32, i32]
    L.sppm_save_ppm_file.argtypes = [C.c_char_p, vp, i32, i32]
    L.sppm_check.argtypes = [vp]
    L.sppm_download_hitpoints.argtypes = [vp] + [vp] * 7
    L.sppm_download_records.argtypes = [vp, vp, vp, C.POINTER(i64)]
    L.sppm_hitpoint_count.argtypes = [vp, C.POINTER(i64)]
    L.sppm_download_hitpoint_pixels.argtypes = [vp, vp]
    L.sppm_export_owned_state.argtypes = [vp, vp]
    L.sppm_import_state.argtypes = [vp, vp]
    L.sppm_download_tile_order.argtypes = [vp, vp, C.POINTER(i64)]
    L.sppm_owned_mask.argtypes = [vp, vp]
    L.sppm_look_ahead.argtypes = [vp, i32, i64, i64]
    L.sppm_look_ahead_rows.argtypes = [vp, i32, i64, i64, vp, i64]
    L.sppm_rows_ready.argtypes = [vp, C.POINTER(i64)]
    L.sppm_look_ahead_rows_to.argtypes = [vp, i32, i64, i64, vp, i64]
    L.sppm_rows_ready_to.argtypes = [vp, C.POINTER(i64)]
    L.sppm_accept_tap.argtypes = [vp, i64]
    L.sppm_download_accepted.argtypes = [vp, vp, vp, C.POINTER(i64)]
    L.sppm_counters.argtypes = [vp, vp]
    L.sppm_set_timing.argtypes = [vp, i32]
    L.sppm_measure_fma_peak.argtypes = [vp, C.POINTER(C.c_double)]
    L.sppm_export_deposits.argtypes = [vp, vp]
    L.sppm_import_deposits.argtypes = [vp, vp]
    L.sppm_export_deposits_range.argtypes = [vp, vp, i64, i64]
    L.sppm_import_deposits_range.argtypes = [vp, vp, i64, i64]
    L.sppm_end_pass_resolve_range.argtypes = [vp, i32, i64, vp, i64, i64]
    L.sppm_fixed_point_scale.argtypes = [vp, C.POINTER(C.c_double)]
    L.sppm_intersect_static.argtypes = [C.POINTER(SppmSphere), i32, vp, i64, vp, vp, vp, vp]
    for name in SYMBOLS:
        if name not in ("sppm_last_error",):
            getattr(L, name).restype = C.c_int
    _LIB = L
    return L


class SppmError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise SppmError(f"libsppm_hip status {rc}: {lib().sppm_last_error().decode()}")
