// sppm2_hip.hpp -- C++ host surface that keeps the reference's names on top of the C ABI
// (include/sppm_hip.h).  A program written against the reference's
//     photonmap::sppm2::StochasticPpm<W,H> sppm(samples, supersamples);
//     sppm.render(name, W, H, samples, supersamples, photon_num, gather_r2, gather_max, passes);
// (src/sppm2.h:90-97, :562-564, call site src/main.cpp:91-93) compiles against this header and links
// libsppm_hip.so instead of including the reference's headers.  Host-only: needs hip/hip_runtime_api.h
// for hipMalloc/hipMemcpy of the image buffer, nothing else.
//
// Differences from the reference, all deliberate:
//   * the scene is run-time data (`edupt::spheres` is a mutable std::vector initialised to the
//     reference's table, src/scene.h:12-24; `edupt::scene()` returns the same object) instead of a const global
//     compiled into every TU;
//   * random numbers come from per-(pass,pixel) and per-photon XorShift streams (src/random.h)
//     seeded from `seed_eye` / `seed_photon`, not from std::random_device, so a render is reproducible;
//   * photon_count starts at 0 (the reference leaves it uninitialised, src/sppm2.h:108-116);
//   * render() returns 0 on success like the reference, and a negative sppm_status instead of
//     undefined behaviour when something fails; sppm_last_error() has the text.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "sppm_hip.h"

namespace edupt
{
    // src/vec.h:7-25
    struct Vec {
        double x, y, z;
        Vec(const double x_ = 0, const double y_ = 0, const double z_ = 0) : x(x_), y(y_), z(z_) {}
        Vec operator+(const Vec& b) const { return Vec(x + b.x, y + b.y, z + b.z); }
        Vec operator-(const Vec& b) const { return Vec(x - b.x, y - b.y, z - b.z); }
        Vec operator*(const double b) const { return Vec(x * b, y * b, z * b); }
        Vec operator/(const double b) const { return Vec(x / b, y / b, z / b); }
        double length_squared() const { return x * x + y * y + z * z; }
        double length() const { return std::sqrt(length_squared()); }
    };
    inline Vec operator*(double f, const Vec& v) { return v * f; }
    inline Vec normalize(const Vec& v) { return v * (1.0 / v.length()); }
    inline const Vec multiply(const Vec& v1, const Vec& v2) { return Vec(v1.x * v2.x, v1.y * v2.y, v1.z * v2.z); }
    inline double dot(const Vec& v1, const Vec& v2) { return v1.x * v2.x + v1.y * v2.y + v1.z * v2.z; }
    inline const Vec cross(const Vec& v1, const Vec& v2)
    {
        return Vec((v1.y * v2.z) - (v1.z * v2.y), (v1.z * v2.x) - (v1.x * v2.z), (v1.x * v2.y) - (v1.y * v2.x));
    }
    typedef Vec Color;  // src/material.h:8

    // src/constant.h:6-8
    const double kPI = 3.14159265358979323846;
    const double kINF = 1e128;
    const double kEPS = 1e-6;

    // src/ray.h:8-12
    struct Ray {
        Vec org, dir;
        Ray(const Vec& org_, const Vec& dir_) : org(org_), dir(dir_) {}
    };

    // src/intersection.h:9-24
    struct Hitpoint {
        double distance;
        Vec normal;
        Vec position;
        Hitpoint() : distance(kINF), normal(), position() {}
    };
    struct Intersection {
        Hitpoint hitpoint;
        int object_id;
        Intersection() : object_id(-1) {}
    };

    // src/material.h:10-17
    enum ReflectionType { REFLECTION_TYPE_DIFFUSE, REFLECTION_TYPE_SPECULAR, REFLECTION_TYPE_REFRACTION };

    // src/sphere.h:15-27 (constructor argument order kept)
    struct Sphere {
        double radius;
        Vec position;
        Color emission, color;
        ReflectionType reflection_type;
        Sphere(const double radius_, const Vec& position_, const Color& emission_, const Color& color_, const ReflectionType type_)
            : radius(radius_), position(position_), emission(emission_), color(color_), reflection_type(type_) {}
    };

    // src/scene.h:12-24, as run-time data; sphere 0 is the light (LightID, src/sppm2.h:132)
    inline std::vector<Sphere>& scene()
    {
        static std::vector<Sphere> s = {
            Sphere(5, Vec(50.0, 75.0, 81.6), Color(36, 36, 36), Color(0.999, 0.999, 0.999), REFLECTION_TYPE_DIFFUSE),
            Sphere(1e5, Vec(1e5 + 1, 40.8, 81.6), Color(), Color(0.75, 0.25, 0.25), REFLECTION_TYPE_DIFFUSE),
            Sphere(1e5, Vec(-1e5 + 99, 40.8, 81.6), Color(), Color(0.25, 0.25, 0.75), REFLECTION_TYPE_DIFFUSE),
            Sphere(1e5, Vec(50, 40.8, 1e5), Color(), Color(0.75, 0.75, 0.75), REFLECTION_TYPE_DIFFUSE),
            Sphere(1e5, Vec(50, 40.8, -1e5 + 250), Color(), Color(), REFLECTION_TYPE_DIFFUSE),
            Sphere(1e5, Vec(50, 1e5, 81.6), Color(), Color(0.75, 0.75, 0.75), REFLECTION_TYPE_DIFFUSE),
            Sphere(1e5, Vec(50, -1e5 + 81.6, 81.6), Color(), Color(0.75, 0.75, 0.75), REFLECTION_TYPE_DIFFUSE),
            Sphere(20, Vec(65, 20, 20), Color(), Color(0.25, 0.75, 0.25), REFLECTION_TYPE_DIFFUSE),
            Sphere(16.5, Vec(27, 16.5, 47), Color(), Color(0.99, 0.99, 0.99), REFLECTION_TYPE_SPECULAR),
            Sphere(16.5, Vec(77, 16.5, 78), Color(), Color(0.99, 0.99, 0.99), REFLECTION_TYPE_REFRACTION),
        };
        return s;
    }

    // the reference's name for the table (src/scene.h:12): `edupt::spheres[i]`, `edupt::spheres.size()`
    inline std::vector<Sphere>& spheres = scene();

    // the scene as the C ABI takes it
    inline std::vector<sppm_sphere> c_scene()
    {
        std::vector<sppm_sphere> sp;
        for (const Sphere& s : scene()) {
            sppm_sphere o = {};
            o.radius = s.radius;
            o.position[0] = s.position.x; o.position[1] = s.position.y; o.position[2] = s.position.z;
            o.emission[0] = s.emission.x; o.emission[1] = s.emission.y; o.emission[2] = s.emission.z;
            o.color[0] = s.color.x; o.color[1] = s.color.y; o.color[2] = s.color.z;
            o.reflection_type = (int)s.reflection_type;
            sp.push_back(o);
        }
        return sp;
    }

    // src/cameras.h:11-65: the same members, the same expressions in the same order (host arithmetic; the kernels' camera
    // is set up by sppm_create from the same numbers and gives the same rays bit for bit: tests/test_gpu_parity.py)
    class PinholeCamera
    {
    public:
        const double image_width;
        const double image_height;
        const double screen_width;
        const double screen_height;
        const double screen_dist = 40.0;
        const Vec camera_position;
        const Vec camera_dir;
        const Vec camera_up;
        double supersamples;
        const Vec screen_x;
        const Vec screen_y;
        const Vec screen_center;

        PinholeCamera(const double width, const double height, const double screen_height_, const double screen_dist_, const Vec& camera_position_,
                      const Vec& camera_dir_, const Vec& camera_up_, const double supersamples_)
            : image_width(width), image_height(height), screen_width(screen_height_ * width / height), screen_height(screen_height_),
              screen_dist(screen_dist_), camera_position(camera_position_), camera_dir(camera_dir_), camera_up(camera_up_),
              supersamples(supersamples_), screen_x(normalize(cross(camera_dir, camera_up)) * screen_width),
              screen_y(normalize(cross(screen_x, camera_dir)) * screen_height), screen_center(camera_position + camera_dir * screen_dist)
        {
        }

        Ray get_ray(const double x, const double y, const int supersample_x = 0, const int supersample_y = 0)
        {
            const double rate = (1.0 / supersamples);
            const double r1 = supersample_x * rate + rate / 2.0;
            const double r2 = supersample_y * rate + rate / 2.0;
            const Vec screen_position = screen_center + screen_x * ((r1 + x) / image_width - 0.5) + screen_y * ((r2 + y) / image_height - 0.5);
            const Vec dir = normalize(screen_position - camera_position);
            return Ray(camera_position, dir);
        }
    };

    // intersect_scene (src/scene.h:53-76) for a batch of rays, on the GPU: the closest hit over `spheres`, lowest index on
    // ties, through sppm_intersect_static (no context; at most 512 spheres -- larger scenes go through sppm_create /
    // sppm_intersect, which walk a grid).  Returns 0 or a negative sppm_status.
    inline int intersect_scene(const std::vector<Ray>& rays, std::vector<Intersection>* out)
    {
        const size_t n = rays.size();
        out->assign(n, Intersection());
        if (n == 0) return 0;
        std::vector<double> h_rays(6 * n), h_hit(7 * n);
        std::vector<int32_t> h_id(n);
        for (size_t i = 0; i < n; i++) {
            double* r = &h_rays[6 * i];
            r[0] = rays[i].org.x; r[1] = rays[i].org.y; r[2] = rays[i].org.z; r[3] = rays[i].dir.x; r[4] = rays[i].dir.y; r[5] = rays[i].dir.z;
        }
        double *d_rays = nullptr, *d_hit = nullptr;
        int32_t* d_id = nullptr;
        int rc = SPPM_ERR_HIP;
        if (hipMalloc((void**)&d_rays, 6 * n * sizeof(double)) == hipSuccess && hipMalloc((void**)&d_hit, 7 * n * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&d_id, n * sizeof(int32_t)) == hipSuccess &&
            hipMemcpy(d_rays, h_rays.data(), 6 * n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess) {
            const std::vector<sppm_sphere> sp = c_scene();
            rc = sppm_intersect_static(sp.data(), (int32_t)sp.size(), d_rays, (int64_t)n, d_id, d_hit, nullptr, nullptr);
            if (!rc && (hipMemcpy(h_id.data(), d_id, n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(h_hit.data(), d_hit, 7 * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) rc = SPPM_ERR_HIP;
        }
        (void)hipFree(d_rays); (void)hipFree(d_hit); (void)hipFree(d_id);
        if (rc) return rc;
        for (size_t i = 0; i < n; i++) {
            Intersection& is = (*out)[i];
            is.object_id = h_id[i];
            const double* f = &h_hit[7 * i];
            is.hitpoint.distance = f[0];
            if (is.object_id >= 0) { is.hitpoint.position = Vec(f[1], f[2], f[3]); is.hitpoint.normal = Vec(f[4], f[5], f[6]); }
        }
        return 0;
    }
    // the reference's signature (src/scene.h:53): one ray.  A batch of one on the GPU -- for drop-in use; callers with
    // many rays use the batch form above.
    inline bool intersect_scene(const Ray& ray, Intersection* intersection)
    {
        std::vector<Intersection> out;
        if (intersect_scene(std::vector<Ray>(1, ray), &out) != 0) { *intersection = Intersection(); return false; }
        *intersection = out[0];
        return intersection->object_id != -1;
    }

    // src/ppm.h:22 and src/hdr.h:48 (the reference puts save_hdr_file in photonmap::utility)
    inline void save_ppm_file(const std::string& filename, const Color* image, const int width, const int height)
    {
        sppm_save_ppm_file(filename.c_str(), &image[0].x, width, height);
    }
}  // namespace edupt

namespace photonmap
{
    namespace utility
    {
        inline void save_hdr_file(const std::string& filename, const edupt::Color* image, const int width, const int height)
        {
            sppm_save_hdr_file(filename.c_str(), &image[0].x, width, height);
        }
    }  // namespace utility

    namespace sppm2
    {
        using namespace edupt;

        template <int Width, int Height>
        class StochasticPpm
        {
        public:
            // the reference's public tunables, src/sppm2.h:38-56
            double alpha = 0.7;
            double initial_radius_2 = 25;
            Vec camera_position = Vec(50.0, 52.0, 220.0);
            Vec camera_dir = normalize(Vec(0.0, -0.04, -1.0));
            Vec camera_up = Vec(0.0, 1.0, 0.0);
            double screen_height = 30.0;
            double screen_dist = 40.0;
            // additions
            unsigned seed_eye = 1000, seed_photon = 2000;
            int camera_model = 0;        // 1 = the thin-lens DoFCamera the reference keeps commented out (src/sppm2.h:73-79)
            double aperture = 4, focus_dist = 180;
            unsigned seed_lens = 3000;
            int update_rule = 0;         // 1 = per-pass (N, tau) rule of src/stochastic_ppm.h:721-741
            bool write_pass_files = true;  // the reference writes name_<i>.ppm/.hdr every pass (src/sppm2.h:687-688)
            std::vector<Color> image;      // last image, row-major, y flipped as in the reference (src/sppm2.h:632)

            int samples, supersamples;
            StochasticPpm(const int samples_, const int supersamples_) : samples(samples_), supersamples(supersamples_) {}

            int render(const std::string& filename, const int width, const int height, const int samples_, const int supersamples_,
                       int photon_num, double gather_photon_radius, int gahter_max_photon_num, int eye_pass_count = 100)
            {
                if (width != Width || height != Height) return SPPM_ERR_INVALID;
                sppm_config cfg = {};
                cfg.width = width; cfg.height = height; cfg.samples = samples_; cfg.supersamples = supersamples_;
                const Vec* v[3] = {&camera_position, &camera_dir, &camera_up};
                double* dst[3] = {cfg.camera_position, cfg.camera_dir, cfg.camera_up};
                for (int k = 0; k < 3; k++) { dst[k][0] = v[k]->x; dst[k][1] = v[k]->y; dst[k][2] = v[k]->z; }
                cfg.screen_height = screen_height; cfg.screen_dist = screen_dist;
                cfg.alpha = alpha; cfg.initial_radius2 = initial_radius_2;
                cfg.gather_radius2 = gather_photon_radius; cfg.gather_max = gahter_max_photon_num;
                cfg.update_rule = update_rule; cfg.double_nee = 1;
                cfg.seed_eye = seed_eye; cfg.seed_photon = seed_photon;
                cfg.photons_per_pass = photon_num;
                cfg.camera_model = camera_model; cfg.aperture = aperture; cfg.focus_dist = focus_dist; cfg.seed_lens = seed_lens;
                const std::vector<sppm_sphere> sp = c_scene();
                sppm_ctx* ctx = nullptr;
                int rc = sppm_create(&cfg, sp.data(), (int)sp.size(), nullptr, &ctx);
                if (rc) return rc;
                const size_t P = (size_t)width * height;
                image.assign(P, Color());
                double* d_image = nullptr;
                if (hipMalloc((void**)&d_image, P * 3 * sizeof(double)) != hipSuccess) { sppm_destroy(ctx); return SPPM_ERR_HIP; }
                for (int i = 0; i < eye_pass_count && rc == 0; i++) {
                    rc = sppm_eye_pass(ctx, i);
                    if (!rc) rc = sppm_photon_pass(ctx, i, 0, photon_num);
                    if (!rc) rc = sppm_end_pass(ctx);
                    if (!rc && (write_pass_files || i == eye_pass_count - 1)) {
                        rc = sppm_resolve(ctx, i, photon_num, d_image);
                        if (!rc) rc = sppm_check(ctx);
                        if (!rc && hipMemcpy(&image[0].x, d_image, P * 3 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = SPPM_ERR_HIP;
                        if (!rc && write_pass_files) {
                            save_ppm_file(filename + "_" + std::to_string(i) + ".ppm", image.data(), width, height);
                            utility::save_hdr_file(filename + "_" + std::to_string(i) + ".hdr", image.data(), width, height);
                        }
                    }
                }
                if (!rc) rc = sppm_check(ctx);
                if (!rc) {
                    save_ppm_file(filename + ".ppm", image.data(), width, height);
                    utility::save_hdr_file(filename + ".hdr", image.data(), width, height);
                }
                (void)hipFree(d_image);
                sppm_destroy(ctx);
                return rc;
            }
        };
    }  // namespace sppm2
}  // namespace photonmap
