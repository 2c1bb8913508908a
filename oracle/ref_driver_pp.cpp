// oracle/ref_driver_pp.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Harness around the UNMODIFIED src/stochastic_ppm.h (namespace photonmap::sppm_miss), compiled from where it lies
// (-I/root/reference/src; nothing of the reference is copied into this repository).  That file holds the PER-PASS
// progressive update: photons of a pass only deposit (create_photon, :371-572, deposit at :439-456) and the
// (N, r^2, tau) update runs once per pass (:721-741).  It is what update_rule = 1 of this build follows, and
// this harness is what pins that rule against the compiled reference.
//
// The header compiles and its class instantiates (g++ -std=c++17 -O2).  What cannot be used is its render()
// (:590-789): it reads max_radius_2 uninitialised (:577-581), runs 10 hard-coded passes (:659), keeps
// Color images[10][PixelCount] on the stack (:657) and saves an `image` buffer it never writes (:609, :778-787).
// So, as oracle/ref_driver.cpp does for sppm2.h, the harness drives the public pieces in render()'s order:
//
//     create_point_loop (:186-369)  per pixel        -- called as it is
//     PointMap::CreateKDtree        (:697)           -- called as it is
//     create_photon (:371-572)      per photon       -- called as it is
//     the per-pass update           (:721-741)       -- INLINE in render(); applied here on the public SPPMPixel
//                                                       fields and the public tau_container, same expressions
//     image assembly                (:743-765)       -- inline in render() too; same expressions here
//
// What the harness decides, because render() is unusable:
//   * uninitialised fields.  SPPMPixel's constructor stores neither photon_count nor photon_add_count (:136-144);
//     both are read (:454, :726-729).  Every newly added point gets 0 through the public GetData() (the call site
//     passes 0 for the count, :238) -- the same repair as bug B1 of sppm2.h.
//   * the search radius.  update_photons_with_render takes the running maximum of the hitpoint radii from an
//     uninitialised start (:577-581) and passes it to every create_photon of the pass (:586).  With gather_r2 <= 0
//     on the command line the harness does that from 0.0; it is dumped per pass (it stays at initial_radius_2 as
//     long as one hitpoint has never accepted a photon).
//   * the streams.  layout 0: one ValueSampler for all eye passes and a fresh one per photon pass -- the stream
//     structure of the live sppm2.h (:621, :530), which is what the GPU's reference-stream mode generates.
//     layout 1: render()'s own structure -- ONE sampler for everything, one draw before the first pass (:651-652),
//     two draws per pixel for the subpixel (:685-686; with supersamples = 1 both give 0), photons from the same
//     sampler (:700-701).
// Determinism: std::random_device::_M_getval is interposed to return a counter.

#include <algorithm>
#include <array>
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

// ValueSampler keeps its engine private (src/sampler/value_sampler.h:8-10); the harness reads the engine state to
// count the 32-bit words a photon consumed.  Every standard header is already included above.
#define private public
#include "sampler/value_sampler.h"
#undef private
#include "stochastic_ppm.h"

#ifndef REF_W
#define REF_W 64
#endif
#ifndef REF_H
#define REF_H 48
#endif

using Sppm = photonmap::sppm_miss::StochasticPpm<REF_W, REF_H>;
using edupt::Color;
using edupt::Ray;

static std::string g_out;
template <typename T>
static void dump(const std::string& name, const std::vector<T>& v)
{
    std::string path = g_out + "/" + name;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}
static void push3(std::vector<double>& v, const Vec& a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }
static uint64_t words_since(std::mt19937 from, const std::mt19937& to)
{
    uint64_t w = 0;
    while (!(from == to)) { from(); w++; if (w > (1ull << 36)) break; }
    return w;
}

int main(int argc, char** argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s outdir photons passes traced_photons gather_r2(<=0: running max) gather_k [seed_eye seed_photon layout]\n", argv[0]);
        return 2;
    }
    g_out = argv[1];
    const int photon_num = atoi(argv[2]);
    const int passes = atoi(argv[3]);
    const int traced = atoi(argv[4]);
    const double gather_r2_arg = atof(argv[5]);
    const int gather_k = atoi(argv[6]);
    const unsigned seed_eye = argc > 7 ? (unsigned)strtoul(argv[7], 0, 10) : 1000u;
    const unsigned seed_photon = argc > 8 ? (unsigned)strtoul(argv[8], 0, 10) : 2000u;
    const int layout = argc > 9 ? atoi(argv[9]) : 0;
    const int W = REF_W, H = REF_H, P = W * H;
    const int samples = 1, supersamples = 1;

    std::streambuf* cout_buf = std::cout.rdbuf();
    std::ofstream devnull("/dev/null");
    std::cout.rdbuf(devnull.rdbuf());  // create_point_loop prints the map's size for every new point (:236)

    g_seed_counter = 1u;  // consumed by the engine_ member (:70-71), which nothing uses afterwards
    static Sppm sppm(samples, supersamples);
    sppm.photon_num = photon_num;
    sppm.imageidx_pointidx = new int[P];
    for (int i = 0; i < P; i++) sppm.imageidx_pointidx[i] = -1;  // :610-614
    for (int i = 0; i < P; i++) { sppm.ld_container[i] = Color(); sppm.tau_container[i] = Color(); }
    Sppm::PointMap point_map;
    std::vector<Color> image(P);

    g_seed_counter = seed_eye;
    ValueSampler<double> eye_sampler(0, 1);
    if (layout == 1) eye_sampler.sample();  // :652
    std::mt19937 eye_mark(seed_eye);        // where the dumped word counts are measured from

    for (int pass = 0; pass < passes; pass++) {
        const size_t n_before = point_map.Size();
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int image_index = (H - y - 1) * W + x;
                int sx = 0, sy = 0;
                if (layout == 1) {  // :685-686
                    sx = eye_sampler.sample() * supersamples;
                    sy = eye_sampler.sample() * supersamples;
                }
                for (int s = 0; s < samples; s++)
                    sppm.create_point_loop(sppm.camera.get_ray(x, y, sx, sy), &eye_sampler, &point_map, image_index);  // :691
            }
        {   // the fields the constructor leaves unset (:136-144)
            auto& pts = point_map.GetData();
            for (size_t k = n_before; k < pts.size(); k++) {
                pts[k].photon_count = 0;
                pts[k].photon_add_count = 0;
                pts[k].accumulated_flux = Color();
            }
        }
        {   // hitpoint records and direct term after this eye pass, by image index
            std::vector<int32_t> hp_i(2 * (size_t)P, -1);
            std::vector<double> hp_f(10 * (size_t)P, 0.0), ld;
            for (auto& p : point_map.GetData()) {
                hp_i[2 * p.index] = 1;
                hp_i[2 * p.index + 1] = p.intersection.object_id;
                double* f = &hp_f[10 * (size_t)p.index];
                f[0] = p.intersection.hitpoint.distance;
                f[1] = p.position.x; f[2] = p.position.y; f[3] = p.position.z;
                f[4] = p.intersection.hitpoint.normal.x; f[5] = p.intersection.hitpoint.normal.y; f[6] = p.intersection.hitpoint.normal.z;
                f[7] = p.weight.x; f[8] = p.weight.y; f[9] = p.weight.z;
            }
            for (int i = 0; i < P; i++) push3(ld, sppm.ld_container[i]);
            dump("eye_hp_i_pass" + std::to_string(pass) + ".bin", hp_i);
            dump("eye_hp_f_pass" + std::to_string(pass) + ".bin", hp_f);
            dump("eye_ld_pass" + std::to_string(pass) + ".bin", ld);
            std::vector<uint64_t> w1{words_since(eye_mark, eye_sampler.engine_)};
            dump("eye_words_pass" + std::to_string(pass) + ".bin", w1);
        }

        point_map.CreateKDtree();  // :697

        auto& pts = point_map.GetData();
        // the search radius of the pass: update_photons_with_render's loop (:577-581), started from 0
        double max_radius_2 = 0.0;
        for (const auto& p : pts) max_radius_2 = std::max(max_radius_2, p.photon_radius_2);
        const double search_r2 = gather_r2_arg > 0.0 ? gather_r2_arg : max_radius_2;
        {
            std::vector<double> r{search_r2, max_radius_2};
            dump("search_r2_pass" + std::to_string(pass) + ".bin", r);
        }
        {   // photon pass: photon_num x create_photon (:584-587)
            g_seed_counter = seed_photon + pass;
            ValueSampler<double> fresh(0, 1);
            ValueSampler<double>& ph_sampler = layout == 1 ? eye_sampler : fresh;
            std::vector<int32_t> prev(P, 0);
            for (auto& p : pts) prev[p.index] = p.photon_add_count;
            std::vector<int32_t> words_per_photon, acc_off{0}, acc_pix;
            const int ntr = std::min(traced, photon_num);
            for (int i = 0; i < photon_num; i++) {
                if (i < ntr) {
                    std::mt19937 before = ph_sampler.engine_;
                    sppm.create_photon(edupt::spheres[0], ph_sampler, &point_map, search_r2, gather_k);
                    words_per_photon.push_back((int32_t)words_since(before, ph_sampler.engine_));
                    std::vector<std::pair<int, int>> acc;
                    for (auto& p : pts)
                        if (p.photon_add_count != prev[p.index]) {
                            // a photon deposits at several surfaces: a hitpoint can accept it more than once
                            for (int r = 0; r < p.photon_add_count - prev[p.index]; r++) acc.push_back({p.index, r});
                            prev[p.index] = p.photon_add_count;
                        }
                    std::sort(acc.begin(), acc.end());
                    for (auto& a : acc) acc_pix.push_back(a.first);
                    acc_off.push_back((int32_t)acc_pix.size());
                } else {
                    sppm.create_photon(edupt::spheres[0], ph_sampler, &point_map, search_r2, gather_k);
                }
            }
            dump("ph_words_pass" + std::to_string(pass) + ".bin", words_per_photon);
            dump("ph_acc_off_pass" + std::to_string(pass) + ".bin", acc_off);
            dump("ph_acc_pix_pass" + std::to_string(pass) + ".bin", acc_pix);
            if (layout == 1) {
                std::vector<uint64_t> w1{words_since(eye_mark, eye_sampler.engine_)};
                dump("stream_words_pass" + std::to_string(pass) + ".bin", w1);
            }
        }
        {   // the pass's deposits, before the update: M and the double sums of :452-454
            std::vector<int32_t> dep_m(P, -1);
            std::vector<double> dep_f(3 * (size_t)P, 0.0);
            for (auto& p : pts) {
                dep_m[p.index] = p.photon_add_count;
                dep_f[3 * (size_t)p.index] = p.accumulated_flux.x;
                dep_f[3 * (size_t)p.index + 1] = p.accumulated_flux.y;
                dep_f[3 * (size_t)p.index + 2] = p.accumulated_flux.z;
            }
            dump("dep_m_pass" + std::to_string(pass) + ".bin", dep_m);
            dump("dep_f_pass" + std::to_string(pass) + ".bin", dep_f);
        }
        // ---- the per-pass update, :721-741 (inline in render(); the same expressions on the same fields) ----
        const auto Np = (pass + 1) * photon_num;  // :721
        for (auto& node : pts) {
            const int p_it = node.index;
            if (node.photon_add_count > 0) {
                const double Nnew = node.photon_count + sppm.alpha * node.photon_add_count;
                const double Rnew2 = node.photon_radius_2 * Nnew / (node.photon_count + node.photon_add_count);
                sppm.tau_container[p_it] = (sppm.tau_container[p_it] + multiply(node.accumulated_flux, node.weight)) * Rnew2 / node.photon_radius_2;
                node.photon_count = Nnew;
                node.photon_radius_2 = Rnew2;
                node.photon_add_count = 0;
                node.accumulated_flux = Color();
            }
            node.weight = Color();
        }
        {   // per-hitpoint state after the update
            std::vector<double> st_f(5 * (size_t)P, 0.0);
            for (auto& p : pts) {
                double* f = &st_f[5 * (size_t)p.index];
                f[0] = p.photon_count; f[1] = p.photon_radius_2;
                f[2] = sppm.tau_container[p.index].x; f[3] = sppm.tau_container[p.index].y; f[4] = sppm.tau_container[p.index].z;
            }
            dump("state_f_pass" + std::to_string(pass) + ".bin", st_f);
        }
        // ---- image assembly, :743-765 ----
        for (int p_it = 0; p_it < P; p_it++) image[p_it] = sppm.ld_container[p_it] / ((pass + 1));
        for (auto& node : pts) {
            const int p_it = node.index;
            image[p_it] = image[p_it] + sppm.tau_container[p_it] * (1.0 / (M_PI * node.photon_radius_2 * Np));
        }
        {
            std::vector<double> img;
            for (int i = 0; i < P; i++) push3(img, image[i]);
            dump("image_pass" + std::to_string(pass) + ".bin", img);
        }
        fprintf(stderr, "[ref-pp] pass %d done, %zu hitpoints, search r2 %.17g\n", pass, point_map.Size(), search_r2);
    }
    photonmap::utility::save_hdr_file(g_out + "/final.hdr", image.data(), W, H);
    edupt::save_ppm_file(g_out + "/final.ppm", image.data(), W, H);
    std::cout.rdbuf(cout_buf);
    return 0;
}
