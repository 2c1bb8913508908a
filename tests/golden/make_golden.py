#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz from the compiled, unmodified reference.

Runs only where /root/reference exists (the build container).  It builds oracle/_ref/ref_driver_*
(oracle/Makefile target `ref`: g++ on oracle/ref_driver.cpp with -I/root/reference/src), runs it,
and packs its raw dumps into small .npz fixtures.  The fixtures are DATA (inputs and the reference's
outputs); no reference source is copied.  Commit the .npz files together with this script.

    python tests/golden/make_golden.py            # regenerates every fixture (about a minute)
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ORACLE = os.path.join(ROOT, "oracle")


def rd(d, name, dt):
    return np.fromfile(os.path.join(d, name), dtype=dt)


def run(tag, w, h, photons, passes, traced, r2, k, seed_eye, seed_photon, full, dof=None, spp=None):
    exe = os.path.join(ORACLE, "_ref", f"ref_driver_{w}x{h}")
    with tempfile.TemporaryDirectory() as d:
        cmd = [exe, d, str(photons), str(passes), str(traced), repr(float(r2)), str(k), str(seed_eye), str(seed_photon)]
        if dof or spp:  # thin-lens camera: aperture (0 = pinhole), focus distance, seed of the lens engine
            cmd += [repr(float(dof[0])), repr(float(dof[1])), str(dof[2])] if dof else ["0", "180", "3000"]
        if spp:  # samples, supersamples
            cmd += [str(spp[0]), str(spp[1])]
        subprocess.check_call(cmd)
        P = w * h
        out = dict(meta=np.array([w, h, photons, passes, traced, k, seed_eye, seed_photon], np.int64),
                   gather_r2=np.array([r2], np.float64))
        if spp:
            out["spp"] = np.array(spp, np.int64)
        if dof:
            out["dof"] = np.array(dof, np.float64)
            out["dof_rays"] = rd(d, "dof_rays.bin", np.float64).reshape(P, 6)
        if full:
            out["xorshift_u32"] = rd(d, "xorshift_u32.bin", np.uint32).reshape(3, 16)
            out["xorshift_01"] = rd(d, "xorshift_01.bin", np.float64).reshape(3, 16)
            out["mt_samples_seed4242"] = rd(d, "mt_samples_seed4242.bin", np.float64)
            out["encode_in"] = rd(d, "encode_in.bin", np.float64).reshape(-1, 3)
            out["encode_rgbe"] = rd(d, "encode_rgbe.bin", np.uint8).reshape(-1, 4)
            out["encode_toint"] = rd(d, "encode_toint.bin", np.int32).reshape(-1, 3)
            out["cam_rays"] = rd(d, "cam_rays.bin", np.float64).reshape(P, 6)
            out["cam_hit_id"] = rd(d, "cam_hit_id.bin", np.int32)
            out["cam_hit_f"] = rd(d, "cam_hit_f.bin", np.float64).reshape(P, 7)
            out["rnd_rays"] = rd(d, "rnd_rays.bin", np.float64).reshape(-1, 6)
            out["rnd_hit_id"] = rd(d, "rnd_hit_id.bin", np.int32)
            out["rnd_hit_f"] = rd(d, "rnd_hit_f.bin", np.float64).reshape(-1, 7)
        stride = 1 if full else 16
        for i in range(passes):
            hp_i = rd(d, f"eye_hp_i_pass{i}.bin", np.int32).reshape(P, 2)
            hp_f = rd(d, f"eye_hp_f_pass{i}.bin", np.float64).reshape(P, 13)
            ld = rd(d, f"eye_ld_pass{i}.bin", np.float64).reshape(P, 3)
            st_n = rd(d, f"state_n_pass{i}.bin", np.int32)
            st_f = rd(d, f"state_f_pass{i}.bin", np.float64).reshape(P, 4)
            img = rd(d, f"image_pass{i}.bin", np.float64).reshape(P, 3)
            out[f"eye_words_pass{i}"] = rd(d, f"eye_words_pass{i}.bin", np.uint64)
            out[f"hp_valid_pass{i}"] = hp_i[:, 0].astype(np.int8)
            out[f"hp_obj_pass{i}"] = hp_i[:, 1].astype(np.int8)
            out[f"state_n_pass{i}"] = st_n
            # sums in a fixed order pin the arrays that are only stored strided
            out[f"sums_pass{i}"] = np.array([hp_f.sum(), ld.sum(), st_f.sum(), img.sum()])
            out[f"hp_f_pass{i}"] = hp_f[::stride]
            out[f"ld_pass{i}"] = ld[::stride]
            out[f"state_f_pass{i}"] = st_f[::stride]
            out[f"image_pass{i}"] = img[::stride]
            out[f"ph_words_pass{i}"] = rd(d, f"ph_words_pass{i}.bin", np.int32)
            out[f"ph_acc_off_pass{i}"] = rd(d, f"ph_acc_off_pass{i}.bin", np.int32)
            out[f"ph_acc_pix_pass{i}"] = rd(d, f"ph_acc_pix_pass{i}.bin", np.int32)
            out[f"ph_acc_cnt_pass{i}"] = rd(d, f"ph_acc_cnt_pass{i}.bin", np.int32).astype(np.int8)
        out["stride"] = np.array([stride])
        hdr = open(os.path.join(d, "final.hdr"), "rb").read()
        ppm = open(os.path.join(d, "final.ppm"), "rb").read()
        out["final_hdr"] = np.frombuffer(hdr, np.uint8)
        if full:
            out["final_ppm"] = np.frombuffer(ppm, np.uint8)
        else:
            out["final_ppm_sha256"] = np.frombuffer(hashlib.sha256(ppm).digest(), np.uint8)
        path = os.path.join(HERE, f"ref_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB")


def run_ppm(tag, w, h, photons, traced, r2, k, seed_eye, seed_photon, supersamples, samples=1):
    """Classic progressive photon mapping (src/progressive_photonmapping.h) through oracle/ref_driver_ppm.cpp."""
    exe = os.path.join(ORACLE, "_ref", "ref_driver_ppm")
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([exe, d, str(w), str(h), str(photons), str(traced), repr(float(r2)), str(k),
                               str(seed_eye), str(seed_photon), str(supersamples), str(samples)])
        P = w * h
        hp_i = rd(d, "ppm_hp_i.bin", np.int32).reshape(-1, 2)
        out = dict(meta=np.array([w, h, photons, traced, k, seed_eye, seed_photon, supersamples], np.int64), samples=np.array([samples], np.int64),
                   gather_r2=np.array([r2], np.float64),
                   hp_pix=hp_i[:, 0].copy(), hp_obj=hp_i[:, 1].astype(np.int8),
                   hp_f=rd(d, "ppm_hp_f.bin", np.float64).reshape(-1, 11),
                   eye_words=rd(d, "ppm_eye_words.bin", np.uint64),
                   kd_order=rd(d, "ppm_kd_order.bin", np.int32),
                   ph_words=rd(d, "ppm_ph_words.bin", np.int32),
                   acc_off=rd(d, "ppm_acc_off.bin", np.int32), acc_hp=rd(d, "ppm_acc_hp.bin", np.int32),
                   acc_cnt=rd(d, "ppm_acc_cnt.bin", np.int32).astype(np.int8),
                   state_n=rd(d, "ppm_state_n.bin", np.int32),
                   state_f=rd(d, "ppm_state_f.bin", np.float64).reshape(-1, 4),
                   image=rd(d, "ppm_image.bin", np.float64).reshape(P, 3),
                   final_hdr=np.frombuffer(open(os.path.join(d, "ppm_final.hdr"), "rb").read(), np.uint8),
                   final_ppm_sha256=np.frombuffer(hashlib.sha256(open(os.path.join(d, "ppm_final.ppm"), "rb").read()).digest(), np.uint8))
        path = os.path.join(HERE, f"ref_ppm_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(hp_i)} hitpoints")


def run_pp(tag, w, h, photons, passes, traced, r2, k, seed_eye, seed_photon, layout, stride):
    """The per-pass rule: src/stochastic_ppm.h (namespace sppm_miss) through oracle/ref_driver_pp.cpp.
    r2 <= 0: the search radius is the running maximum of the hitpoint radii (:577-581)."""
    exe = os.path.join(ORACLE, "_ref", f"ref_driver_pp_{w}x{h}")
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([exe, d, str(photons), str(passes), str(traced), repr(float(r2)), str(k), str(seed_eye), str(seed_photon), str(layout)])
        P = w * h
        out = dict(meta=np.array([w, h, photons, passes, traced, k, seed_eye, seed_photon, layout], np.int64), stride=np.array([stride]))
        for i in range(passes):
            hp_i = rd(d, f"eye_hp_i_pass{i}.bin", np.int32).reshape(P, 2)
            hp_f = rd(d, f"eye_hp_f_pass{i}.bin", np.float64).reshape(P, 10)
            ld = rd(d, f"eye_ld_pass{i}.bin", np.float64).reshape(P, 3)
            dep_f = rd(d, f"dep_f_pass{i}.bin", np.float64).reshape(P, 3)
            st_f = rd(d, f"state_f_pass{i}.bin", np.float64).reshape(P, 5)
            img = rd(d, f"image_pass{i}.bin", np.float64).reshape(P, 3)
            out[f"eye_words_pass{i}"] = rd(d, f"eye_words_pass{i}.bin", np.uint64)
            if layout == 1:
                out[f"stream_words_pass{i}"] = rd(d, f"stream_words_pass{i}.bin", np.uint64)
            out[f"search_r2_pass{i}"] = rd(d, f"search_r2_pass{i}.bin", np.float64)
            out[f"hp_valid_pass{i}"] = hp_i[:, 0].astype(np.int8)
            out[f"hp_obj_pass{i}"] = hp_i[:, 1].astype(np.int8)
            out[f"dep_m_pass{i}"] = rd(d, f"dep_m_pass{i}.bin", np.int32)     # M of every hitpoint, before the update
            out[f"state_n_pass{i}"] = st_f[:, 0].copy()                       # N (a double under this rule) and r^2 of EVERY hitpoint
            out[f"state_r2_pass{i}"] = st_f[:, 1].copy()
            out[f"sums_pass{i}"] = np.array([hp_f.sum(), ld.sum(), dep_f.sum(), st_f.sum(), img.sum()])
            out[f"hp_f_pass{i}"] = hp_f[::stride]
            out[f"ld_pass{i}"] = ld[::stride]
            out[f"dep_f_pass{i}"] = dep_f[::stride]                          # the double sums of :452-454
            out[f"tau_pass{i}"] = st_f[::stride, 2:]
            out[f"image_pass{i}"] = img[::stride]
            out[f"ph_words_pass{i}"] = rd(d, f"ph_words_pass{i}.bin", np.int32)
            out[f"ph_acc_off_pass{i}"] = rd(d, f"ph_acc_off_pass{i}.bin", np.int32)
            out[f"ph_acc_pix_pass{i}"] = rd(d, f"ph_acc_pix_pass{i}.bin", np.int32)
        out["final_hdr"] = np.frombuffer(open(os.path.join(d, "final.hdr"), "rb").read(), np.uint8)
        out["final_ppm_sha256"] = np.frombuffer(hashlib.sha256(open(os.path.join(d, "final.ppm"), "rb").read()).digest(), np.uint8)
        path = os.path.join(HERE, f"ref_pp_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    if not os.path.isdir("/root/reference/src"):
        sys.exit("the reference is not present here; fixtures can only be regenerated in the build container")
    subprocess.check_call(["make", "-C", ORACLE, "ref"])
    # tag, W, H, photons/pass, passes, per-photon-traced, search r^2, K cap, seeds, full dump
    run("64x48", 64, 48, 20000, 3, 4096, 25.0, 100000, 1000, 2000, True)
    run("64x48_r5", 64, 48, 20000, 2, 1024, 5.0, 10000, 31, 77, False)   # main.cpp:76-80's r=5, K=10000
    run("320x240", 320, 240, 200000, 2, 1024, 25.0, 100000, 1000, 2000, False)
    # eye rays through DoFCamera::get_ray (aperture 4, focus 180 as in the commented-out member of src/sppm2.h:73-79)
    run("64x48_dof", 64, 48, 20000, 2, 1024, 25.0, 100000, 1000, 2000, False, dof=(4.0, 180.0, 3000))
    # several eye paths per pixel: samples = 2, supersamples = 2 (the last path's hitpoint stays, the direct terms add up)
    run("64x48_spp", 64, 48, 20000, 2, 1024, 25.0, 100000, 1000, 2000, False, spp=(2, 2))
    # the K cap of the kd search BINDS (at 64x48 a query sees ~25 hitpoints, at 320x240 ~600): src/kdtree.h:126-133 drops
    # the farthest entry and tightens both tests for the rest of the walk, so the accepted sets depend on the tree's
    # visiting order
    run("64x48_k3", 64, 48, 20000, 3, 4096, 25.0, 3, 1000, 2000, False)
    run("64x48_k8", 64, 48, 20000, 2, 2048, 25.0, 8, 31, 77, False)
    run("320x240_k100", 320, 240, 50000, 2, 512, 25.0, 100, 1000, 2000, False)
    # tag, W, H, photons, per-photon-traced, search r^2 (= InitialRadius, main.cpp:76-77), K cap, seeds, supersamples
    run_ppm("64x48", 64, 48, 20000, 4096, 25.0, 100000, 1000, 2000, 1)
    run_ppm("48x36_ss2_r5", 48, 36, 30000, 2048, 5.0, 100000, 31, 77, 2)
    # identical samples per pixel (src/main.cpp:79 runs 100): duplicate hitpoints, equal keys in every range of the tree, and a binding K
    run_ppm("32x24_s2_k6", 32, 24, 10000, 10000, 25.0, 6, 1000, 2000, 1, samples=2)
    # the per-pass rule (src/stochastic_ppm.h): tag, W, H, photons/pass, passes, per-photon-traced, search r^2 (0 = running
    # max of the hitpoint radii), K cap, seeds, stream layout (0 = sppm2's streams, 1 = render()'s own), stride
    run_pp("64x48", 64, 48, 20000, 3, 4096, 0.0, 100000, 1000, 2000, 0, 1)
    run_pp("64x48_l1", 64, 48, 20000, 2, 1024, 0.0, 100000, 31, 77, 1, 1)
    run_pp("320x240", 320, 240, 100000, 2, 1024, 0.0, 100000, 1000, 2000, 0, 16)
    run_pp("64x48_k8", 64, 48, 20000, 2, 2048, 0.0, 8, 31, 77, 0, 1)      # a BINDING K cap (:425-427 pass K to the same search)
    run_pp("320x240_k100", 320, 240, 50000, 2, 512, 0.0, 100, 1000, 2000, 0, 16)


if __name__ == "__main__":
    main()
