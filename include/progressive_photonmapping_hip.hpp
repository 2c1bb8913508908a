// progressive_photonmapping_hip.hpp -- the reference's classic progressive photon mapping entry
//     photonmap::progressive::render(name, width, height, samples, supersamples, photon_num,
//                                    gather_photon_radius, gahter_max_photon_num);
// (src/progressive_photonmapping.h:709-821; the live call of src/main.cpp:87-88) on top of the C ABI
// (include/sppm_hip.h, sppm_config.algorithm = 1).  A program written against the reference compiles
// against this header and links libsppm_hip.so instead of including the reference's header.
//
// Kept: the argument list and its meaning (gather_photon_radius is a SQUARED bound, src/kdtree.h:57), the
// namespace-level tunables InitialRadius and Alpha (:32-33; main.cpp:77 sets InitialRadius before the call),
// the camera of :713-721, the final name.ppm / name.hdr pair (:818-819).
// Different, deliberately: seeds are explicit (seed_eye, seed_photon) instead of
// std::random_device; photon_count starts at 0 (the constructor drops it, :46-55); the images the reference
// also writes every few photons (:681-704) are not written; a pixel's hitpoints are summed in creation order
// (the reference: in the order its kd build left the vector in), a last-bit difference.
#pragma once
#include "sppm2_hip.hpp"

namespace photonmap
{
    namespace progressive
    {
        using namespace edupt;

        inline double InitialRadius = 25;
        inline double Alpha = 0.7;
        inline unsigned seed_eye = 1000, seed_photon = 2000;
        inline std::vector<Color> image;  // the last image, row-major, y flipped as in the reference (:768)

        inline int render(const std::string& filename, const int width, const int height, const int samples, const int supersamples,
                          int photon_num, double gather_photon_radius, int gahter_max_photon_num)
        {
            sppm_config cfg = {};
            cfg.algorithm = 1;
            cfg.width = width; cfg.height = height; cfg.samples = samples; cfg.supersamples = supersamples;
            const Vec dir = normalize(Vec(0.0, -0.04, -1.0));
            cfg.camera_position[0] = 50.0; cfg.camera_position[1] = 52.0; cfg.camera_position[2] = 220.0;
            cfg.camera_dir[0] = dir.x; cfg.camera_dir[1] = dir.y; cfg.camera_dir[2] = dir.z;
            cfg.camera_up[1] = 1.0;
            cfg.screen_height = 30.0; cfg.screen_dist = 40.0;
            cfg.alpha = Alpha; cfg.initial_radius2 = InitialRadius;
            cfg.gather_radius2 = gather_photon_radius; cfg.gather_max = gahter_max_photon_num;
            cfg.seed_eye = seed_eye; cfg.seed_photon = seed_photon;
            cfg.photons_per_pass = photon_num;
            const std::vector<sppm_sphere> sp = c_scene();
            sppm_ctx* ctx = nullptr;
            int rc = sppm_create(&cfg, sp.data(), (int)sp.size(), nullptr, &ctx);
            if (rc) return rc;
            const size_t P = (size_t)width * height;
            image.assign(P, Color());
            double* d_image = nullptr;
            if (hipMalloc((void**)&d_image, P * 3 * sizeof(double)) != hipSuccess) { sppm_destroy(ctx); return SPPM_ERR_HIP; }
            rc = sppm_eye_pass(ctx, 0);                              // create_point loops + CreateKDtree, :760-797
            if (!rc) rc = sppm_photon_pass(ctx, 0, 0, photon_num);   // update_photons_with_render, :801
            if (!rc) rc = sppm_resolve(ctx, 0, photon_num, d_image); // :807-813
            if (!rc) rc = sppm_check(ctx);
            if (!rc && hipMemcpy(&image[0].x, d_image, P * 3 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = SPPM_ERR_HIP;
            if (!rc) {
                save_ppm_file(filename + ".ppm", image.data(), width, height);
                utility::save_hdr_file(filename + ".hdr", image.data(), width, height);
            }
            (void)hipFree(d_image);
            sppm_destroy(ctx);
            return rc;
        }
    }  // namespace progressive
}  // namespace photonmap
