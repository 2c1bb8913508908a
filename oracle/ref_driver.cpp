// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Harness around the UNMODIFIED reference headers (compiled from where they lie,
// -I/root/reference/src; nothing of the reference is copied into this repository).
// It drives the public pieces of photonmap::sppm2::StochasticPpm<W,H> (src/sppm2.h)
// in the order render() uses them (src/sppm2.h:615-694) and dumps golden vectors as
// raw little-endian binaries.  tests/golden/make_golden.py packs them into .npz.
//
// Why a harness instead of calling render() directly:
//   * ProgressiveIntersection::photon_count is never initialised by its constructor
//     (src/sppm2.h:108-116) and is read at :398, so render() as written depends on
//     stack garbage.  The harness zeroes the field of every newly added point through
//     the public PointMap::GetData() before the photon pass.
//   * render() offers no tap for intermediate state.
// Determinism: std::random_device::_M_getval is interposed to return a counter, so the
// seed of every mt19937 the reference constructs is chosen here.
//
// Build: see oracle/Makefile (g++ -std=c++17 -O2, W/H are compile-time constants because
// the reference makes width/height template parameters, src/sppm2.h:33).

#include <algorithm>
#include <array>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <queue>
#include <random>
#include <sstream>
#include <string>
#include <vector>

static unsigned int g_seed_counter = 0;
unsigned int std::random_device::_M_getval() { return g_seed_counter++; }

// ValueSampler keeps its engine private (src/sampler/value_sampler.h:8-10); the harness
// needs to read the engine state to count how many 32-bit words a photon consumed.
// Every standard header the reference uses is already included above, so this only
// affects the reference's own classes.
#define private public
#include "sampler/value_sampler.h"
#undef private
#include "sppm2.h"

#ifndef REF_W
#define REF_W 64
#endif
#ifndef REF_H
#define REF_H 48
#endif

using Sppm = photonmap::sppm2::StochasticPpm<REF_W, REF_H>;
using edupt::Color;
using edupt::Intersection;
using edupt::Ray;

static std::string g_out;
// REF_TIMING=1: time the reference's own code (eye pass, kd-tree build, photon pass) and write no vectors; used by
// bench.py's cpu_baseline.reference_check with the 320x240 build (the eye pass is quadratic in the pixel count: 1080p is out of reach)
static bool g_timing = false;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename T>
static void dump(const std::string& name, const std::vector<T>& v)
{
    if (g_timing) return;
    std::string path = g_out + "/" + name;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}

static void push3(std::vector<double>& v, const Vec& a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }

static std::vector<unsigned char> slurp(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s outdir photons passes traced_photons gather_r2 gather_k [seed_eye seed_photon aperture focus_dist seed_lens samples supersamples]\n", argv[0]);
        return 2;
    }
    g_out = argv[1];
    g_timing = getenv("REF_TIMING") != nullptr;
    const int photon_num = atoi(argv[2]);
    const int passes = atoi(argv[3]);
    const int traced = atoi(argv[4]);
    const double gather_r2 = atof(argv[5]);
    const int gather_k = atoi(argv[6]);
    const unsigned seed_eye = argc > 7 ? (unsigned)strtoul(argv[7], 0, 10) : 1000u;
    const unsigned seed_photon = argc > 8 ? (unsigned)strtoul(argv[8], 0, 10) : 2000u;
    // thin-lens camera (src/cameras.h:66-139): aperture > 0 sends the eye rays through DoFCamera::get_ray, the
    // way the commented-out member of src/sppm2.h:73-79 would (aperture 4, focus 180 there); the lens draws come
    // from an engine of their own (src/sppm2.h:70-71), seeded seed_lens here
    const double aperture = argc > 9 ? atof(argv[9]) : 0.0;
    const double focus_dist = argc > 10 ? atof(argv[10]) : 180.0;
    const unsigned seed_lens = argc > 11 ? (unsigned)strtoul(argv[11], 0, 10) : 3000u;
    const int W = REF_W, H = REF_H, P = W * H;
    const int samples = argc > 12 ? atoi(argv[12]) : 1, supersamples = argc > 13 ? atoi(argv[13]) : 1;

    std::streambuf* cout_buf = std::cout.rdbuf();
    std::ofstream devnull("/dev/null");
    std::cout.rdbuf(devnull.rdbuf());  // the reference prints a line per image row (src/sppm2.h:627)

    // ---- (10) XorShift known answers (src/random.h:9-33) --------------------------------
    {
        std::vector<uint32_t> xs;
        std::vector<double> xs01;
        const unsigned seeds[3] = {0u, 1u, 12345u};
        for (unsigned s : seeds) {
            edupt::XorShift a(s), b(s);
            for (int i = 0; i < 16; i++) xs.push_back(a.next());
            for (int i = 0; i < 16; i++) xs01.push_back(b.next01());
        }
        dump("xorshift_u32.bin", xs);
        dump("xorshift_01.bin", xs01);
    }

    // ---- (3) raw sample() sequence of ValueSampler<double>(0,1) ---------------------------
    {
        g_seed_counter = 4242u;
        ValueSampler<double> s(0, 1);
        std::vector<double> v;
        for (int i = 0; i < 4096; i++) v.push_back(s.sample());
        dump("mt_samples_seed4242.bin", v);
    }

    // ---- (11) output encoders on a value sweep (src/hdr.h:37-45, src/ppm.h:13-20) ---------
    {
        std::vector<double> in;
        std::vector<uint8_t> rgbe;
        std::vector<int32_t> ti;
        std::mt19937 e(99);
        std::uniform_real_distribution<double> u(0, 1);
        for (int i = 0; i < 2048; i++) {
            double scale = std::pow(10.0, -36.0 + 44.0 * u(e));
            Color c(u(e) * scale, u(e) * scale, u(e) * scale);
            if (i % 64 == 0) c = Color(0, 0, 0);
            if (i % 64 == 1) c = Color(1e-33, 0, 0);
            if (i % 64 == 2) c = Color(1.0, 0.5, 0.25);
            if (i % 64 == 3) c = Color(-0.5, 2.0, 0.999999);
            push3(in, c);
            auto p = photonmap::utility::get_hdr_pixel(c);
            rgbe.push_back(p.r); rgbe.push_back(p.g); rgbe.push_back(p.b); rgbe.push_back(p.e);
            ti.push_back(edupt::to_int(c.x)); ti.push_back(edupt::to_int(c.y)); ti.push_back(edupt::to_int(c.z));
        }
        dump("encode_in.bin", in);
        dump("encode_rgbe.bin", rgbe);
        dump("encode_toint.bin", ti);
    }

    g_seed_counter = 1u;  // consumed by the StochasticPpm::engine_ member (src/sppm2.h:70-71); unused afterwards
    static Sppm sppm(samples, supersamples);

    // ---- (1) camera rays, (2) closest hits -------------------------------------------------
    {
        std::vector<double> rays, hit_f;
        std::vector<int32_t> hit_id;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                Ray r = sppm.camera.get_ray(x, y, 0, 0);
                push3(rays, r.org); push3(rays, r.dir);
                Intersection is;
                bool hit = edupt::intersect_scene(r, &is);
                hit_id.push_back(hit ? is.object_id : -1);
                hit_f.push_back(is.hitpoint.distance);
                push3(hit_f, is.hitpoint.position); push3(hit_f, is.hitpoint.normal);
            }
        dump("cam_rays.bin", rays);
        dump("cam_hit_id.bin", hit_id);
        dump("cam_hit_f.bin", hit_f);

        std::mt19937 e(777);
        std::uniform_real_distribution<double> u(0, 1);
        std::vector<double> rr, rf;
        std::vector<int32_t> rid;
        for (int i = 0; i < 4096; i++) {
            Vec o(1 + 98 * u(e), 81.6 * u(e), 250 * u(e));
            double z = 1 - 2 * u(e), ph = 2 * M_PI * u(e), s = std::sqrt(1 - z * z);
            Vec d = normalize(Vec(s * std::cos(ph), s * std::sin(ph), z));
            if (i % 16 == 0) o = Vec(50, 75, 81.6) + d * (5 + 1e-6);  // starts on the light, as photons do
            Ray r(o, d);
            push3(rr, r.org); push3(rr, r.dir);
            Intersection is;
            bool hit = edupt::intersect_scene(r, &is);
            rid.push_back(hit ? is.object_id : -1);
            rf.push_back(is.hitpoint.distance);
            push3(rf, is.hitpoint.position); push3(rf, is.hitpoint.normal);
        }
        dump("rnd_rays.bin", rr);
        dump("rnd_hit_id.bin", rid);
        dump("rnd_hit_f.bin", rf);
    }

    // ---- passes, in render()'s order (src/sppm2.h:615-690) ---------------------------------
    sppm.imageidx_pointidx = new int[P];
    for (int i = 0; i < P; i++) sppm.imageidx_pointidx[i] = -1;
    for (int i = 0; i < P; i++) sppm.ld_container[i] = Color();  // std::array member of a static: already zero
    Sppm::PointMap point_map;
    std::vector<Color> image(P);

    std::mt19937 lens_engine(seed_lens);
    edupt::DoFCamera dof(W, H, 30.0, 40.0, Vec(50.0, 52.0, 220.0), normalize(Vec(0.0, -0.04, -1.0)), Vec(0.0, 1.0, 0.0), supersamples,
                         aperture > 0.0 ? aperture : 4.0, focus_dist, lens_engine);
    if (aperture > 0.0) {  // known answers of get_ray: the first P rays of a camera with a lens engine of its own
        std::mt19937 e2(seed_lens);
        edupt::DoFCamera cam2(W, H, 30.0, 40.0, Vec(50.0, 52.0, 220.0), normalize(Vec(0.0, -0.04, -1.0)), Vec(0.0, 1.0, 0.0), supersamples,
                              aperture, focus_dist, e2);
        std::vector<double> rays;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) { Ray r = cam2.get_ray(x, y, 0, 0); push3(rays, r.org); push3(rays, r.dir); }
        dump("dof_rays.bin", rays);
    }
    g_seed_counter = seed_eye;
    ValueSampler<double> eye_sampler(0, 1);  // one stream for every eye pass (src/sppm2.h:621)

    for (int pass = 0; pass < passes; pass++) {
        const size_t n_before = point_map.Size();
        const double t_eye0 = now_s();
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int image_index = (H - y - 1) * W + x;
                for (int sy = 0; sy < supersamples; sy++)
                    for (int sx = 0; sx < supersamples; sx++)
                        for (int s = 0; s < samples; s++)
                            sppm.create_point_loop(aperture > 0.0 ? dof.get_ray(x, y, sx, sy) : sppm.camera.get_ray(x, y, sx, sy), &eye_sampler, &point_map, image_index);
            }
        const double t_eye = now_s() - t_eye0;
        {   // B1: the constructor leaves photon_count unset; its call site passes 0 (src/sppm2.h:208)
            auto& pts = point_map.GetData();
            for (size_t k = n_before; k < pts.size(); k++) {
                pts[k].photon_count = 0;
                pts[k].accumulated_flux = Color();
            }
        }
        if (!g_timing) {   // (4) hitpoint records and (5) direct term after this eye pass, by pixel index
            std::vector<int32_t> hp_i(2 * (size_t)P, -1);
            std::vector<double> hp_f(13 * (size_t)P, 0.0), ld;
            for (auto& p : point_map.GetData()) {
                hp_i[2 * p.index] = 1;
                hp_i[2 * p.index + 1] = p.intersection.object_id;
                double* f = &hp_f[13 * (size_t)p.index];
                f[0] = p.intersection.hitpoint.distance;
                f[1] = p.position.x; f[2] = p.position.y; f[3] = p.position.z;
                f[4] = p.intersection.hitpoint.normal.x; f[5] = p.intersection.hitpoint.normal.y; f[6] = p.intersection.hitpoint.normal.z;
                f[7] = p.weight.x; f[8] = p.weight.y; f[9] = p.weight.z;
                f[10] = p.intersection.hitpoint.position.x; f[11] = p.intersection.hitpoint.position.y; f[12] = p.intersection.hitpoint.position.z;
            }
            for (int i = 0; i < P; i++) push3(ld, sppm.ld_container[i]);
            dump("eye_hp_i_pass" + std::to_string(pass) + ".bin", hp_i);
            dump("eye_hp_f_pass" + std::to_string(pass) + ".bin", hp_f);
            dump("eye_ld_pass" + std::to_string(pass) + ".bin", ld);
            // words of the eye stream consumed so far
            std::mt19937 shadow(seed_eye);
            uint64_t words = 0;
            while (!(shadow == eye_sampler.engine_)) { shadow(); words++; if (words > (1ull << 34)) break; }
            std::vector<uint64_t> w1{words};
            dump("eye_words_pass" + std::to_string(pass) + ".bin", w1);
        }

        const double t_kd0 = now_s();
        point_map.CreateKDtree();
        const double t_kd = now_s() - t_kd0;

        {   // photon pass = update_photons (src/sppm2.h:527-558) minus its progress printing
            g_seed_counter = seed_photon + pass;
            ValueSampler<double> ph_sampler(0, 1);
            auto& pts = point_map.GetData();
            std::vector<int32_t> prev(P, 0);
            for (auto& p : pts) prev[p.index] = p.photon_count;
            std::vector<int32_t> words_per_photon, acc_off{0}, acc_pix, acc_cnt;
            const int ntr = std::min(traced, photon_num);
            const double t_ph0 = now_s();
            for (int i = 0; i < photon_num; i++) {
                if (i < ntr) {
                    std::mt19937 before = ph_sampler.engine_;
                    sppm.create_photon(edupt::spheres[0], ph_sampler, &point_map, gather_r2, gather_k);
                    int words = 0;
                    while (!(before == ph_sampler.engine_)) { before(); words++; }
                    words_per_photon.push_back(words);
                    std::vector<std::pair<int, int>> acc;
                    for (auto& p : pts)
                        if (p.photon_count != prev[p.index]) {
                            acc.push_back({p.index, p.photon_count - prev[p.index]});
                            prev[p.index] = p.photon_count;
                        }
                    std::sort(acc.begin(), acc.end());
                    for (auto& a : acc) { acc_pix.push_back(a.first); acc_cnt.push_back(a.second); }
                    acc_off.push_back((int32_t)acc_pix.size());
                } else {
                    sppm.create_photon(edupt::spheres[0], ph_sampler, &point_map, gather_r2, gather_k);
                }
            }
            if (g_timing)
                fprintf(stderr, "[ref-timing] pass %d eye_s %.6f kd_s %.6f photons %d photon_s %.6f\n", pass, t_eye, t_kd, photon_num, now_s() - t_ph0);
            dump("ph_words_pass" + std::to_string(pass) + ".bin", words_per_photon);
            dump("ph_acc_off_pass" + std::to_string(pass) + ".bin", acc_off);
            dump("ph_acc_pix_pass" + std::to_string(pass) + ".bin", acc_pix);
            dump("ph_acc_cnt_pass" + std::to_string(pass) + ".bin", acc_cnt);
        }
        {   // (8) per-hitpoint state after the pass
            std::vector<int32_t> st_n(P, -1);
            std::vector<double> st_f(4 * (size_t)P, 0.0);
            for (auto& p : point_map.GetData()) {
                st_n[p.index] = p.photon_count;
                st_f[4 * (size_t)p.index] = p.photon_radius_2;
                st_f[4 * (size_t)p.index + 1] = p.accumulated_flux.x;
                st_f[4 * (size_t)p.index + 2] = p.accumulated_flux.y;
                st_f[4 * (size_t)p.index + 3] = p.accumulated_flux.z;
            }
            dump("state_n_pass" + std::to_string(pass) + ".bin", st_n);
            dump("state_f_pass" + std::to_string(pass) + ".bin", st_f);
        }
        // image assembly exactly as src/sppm2.h:658-670
        for (int p_it = 0; p_it < P; p_it++) image[p_it] = sppm.ld_container[p_it] / ((pass + 1));
        for (auto node : point_map.GetData())
            image[node.index] = image[node.index] +
                                node.accumulated_flux * (1.0 / (M_PI * node.photon_radius_2 * (pass + 1) * photon_num));
        {
            std::vector<double> img;
            for (int i = 0; i < P; i++) push3(img, image[i]);
            dump("image_pass" + std::to_string(pass) + ".bin", img);
        }
        fprintf(stderr, "[ref] pass %d done, %zu hitpoints\n", pass, point_map.Size());
    }
    // (9) the reference's own writers on the final image (src/hdr.h:48, src/ppm.h:22)
    if (!g_timing) photonmap::utility::save_hdr_file(g_out + "/final.hdr", image.data(), W, H);
    if (!g_timing) edupt::save_ppm_file(g_out + "/final.ppm", image.data(), W, H);
    std::cout.rdbuf(cout_buf);
    return 0;
}
