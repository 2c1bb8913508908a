// oracle/ref_driver_ppm.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Harness around the UNMODIFIED reference header src/progressive_photonmapping.h (classic progressive
// photon mapping, BASELINE config 4), compiled from where it lies (-I/root/reference/src; nothing of the
// reference is copied into this repository).  It calls the reference's own create_point / create_photon /
// CreateKDtree in the order render() does (src/progressive_photonmapping.h:709-821) and dumps golden vectors
// as raw little-endian binaries; tests/golden/make_golden.py packs them into .npz.
//
// Why not render() itself: ProgressiveIntersection::photon_count is never initialised by its constructor
// (src/progressive_photonmapping.h:45-54: the parameter is dropped) and is read at :376, so render() as
// written depends on heap garbage; the harness zeroes it through PointMap::GetData() before the photon
// pass (as for sppm2.h).  render() also writes image files every few photons and offers no tap.
// Determinism: std::random_device::_M_getval is interposed to return a counter (seed of every mt19937).
//
// usage: ref_driver_ppm outdir W H photons traced gather_r2 gather_k [seed_eye seed_photon supersamples samples]
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <queue>
#include <random>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

static unsigned int g_seed_counter = 0;
unsigned int std::random_device::_M_getval() { return g_seed_counter++; }

#define private public
#include "sampler/value_sampler.h"
#undef private
#include "progressive_photonmapping.h"

namespace ppm = photonmap::progressive;
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


int main(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s outdir W H photons traced gather_r2 gather_k [seed_eye seed_photon supersamples samples]\n", argv[0]); return 2; }
    g_out = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]), P = W * H;
    const int photon_num = atoi(argv[4]), traced = atoi(argv[5]);
    const double gather_r2 = atof(argv[6]);
    const int gather_k = atoi(argv[7]);
    const unsigned seed_eye = argc > 8 ? (unsigned)strtoul(argv[8], 0, 10) : 1000u;
    const unsigned seed_photon = argc > 9 ? (unsigned)strtoul(argv[9], 0, 10) : 2000u;
    // samples > 1 repeats one ray (src/progressive_photonmapping.h:741-757): identical hitpoints, as src/main.cpp:79 runs it
    const int samples = argc > 11 ? atoi(argv[11]) : 1, supersamples = argc > 10 ? atoi(argv[10]) : 1, spp = samples * supersamples * supersamples;

    std::streambuf* cout_buf = std::cout.rdbuf();
    std::ofstream devnull("/dev/null");
    std::cout.rdbuf(devnull.rdbuf());

    // the camera set-up of render() (src/progressive_photonmapping.h:712-721)
    const Vec camera_position = Vec(50.0, 52.0, 220.0);
    const Vec camera_dir = normalize(Vec(0.0, -0.04, -1.0));
    const Vec camera_up = Vec(0.0, 1.0, 0.0);
    const double screen_width = 30.0 * W / H, screen_height = 30.0, screen_dist = 40.0;
    const Vec screen_x = normalize(cross(camera_dir, camera_up)) * screen_width;
    const Vec screen_y = normalize(cross(screen_x, camera_dir)) * screen_height;
    const Vec screen_center = camera_position + camera_dir * screen_dist;

    ppm::InitialRadius = gather_r2;  // as its caller does, src/main.cpp:76-77
    ppm::PointMap point_map;
    g_seed_counter = seed_eye;
    ValueSampler<double> eye_sampler(0, 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int image_index = (H - y - 1) * W + x;
            for (int sy = 0; sy < supersamples; sy++)
                for (int sx = 0; sx < supersamples; sx++)
                    for (int s = 0; s < samples; s++) {
                        const double rate = (1.0 / supersamples);
                        const double r1 = sx * rate + rate / 2.0, r2 = sy * rate + rate / 2.0;
                        const Vec screen_position = screen_center + screen_x * ((r1 + x) / W - 0.5) + screen_y * ((r2 + y) / H - 0.5);
                        const Vec dir = normalize(screen_position - camera_position);
                        ppm::create_point(Ray(camera_position, dir), eye_sampler, 0, 1.0, &point_map, image_index);
                    }
        }
    auto& pts = point_map.GetData();
    const size_t Hn = pts.size();
    std::vector<int> id_at(Hn);  // creation id of the hitpoint at each place of the vector once CreateKDtree has reordered it
    {   // hitpoints in creation order; B1: photon_count is unset, the call site passes 0 (:525)
        std::vector<int32_t> hp_i;
        std::vector<double> hp_f;
        for (size_t k = 0; k < Hn; k++) {
            auto& p = pts[k];
            p.photon_count = (int)k;  // the creation id rides through CreateKDtree in this field (identical samples make identical hitpoints); zeroed below
            p.accumulated_flux = Color();
            hp_i.push_back(p.index); hp_i.push_back(p.intersection.object_id);
            hp_f.push_back(p.intersection.hitpoint.distance);
            push3(hp_f, p.position); push3(hp_f, p.intersection.hitpoint.normal);
            hp_f.push_back(p.weight);
            push3(hp_f, p.emission);
        }
        dump("ppm_hp_i.bin", hp_i);
        dump("ppm_hp_f.bin", hp_f);
        std::mt19937 shadow(seed_eye);
        uint64_t words = 0;
        while (!(shadow == eye_sampler.engine_)) { shadow(); words++; if (words > (1ull << 34)) break; }
        dump("ppm_eye_words.bin", std::vector<uint64_t>{words});
    }
    point_map.CreateKDtree();
    {   // the order in which render() later walks the hitpoints (the vector was sorted in place)
        std::vector<int32_t> order;
        for (size_t j = 0; j < Hn; j++) { id_at[j] = pts[j].photon_count; pts[j].photon_count = 0; order.push_back(id_at[j]); }
        dump("ppm_kd_order.bin", order);
    }
    {   // photon pass = update_photons (:618-640) minus its printing
        g_seed_counter = seed_photon;
        ValueSampler<double> ph_sampler(0, 1);
        std::vector<int32_t> prev(Hn, 0), words_per_photon, acc_off{0}, acc_hp, acc_cnt;
        const int ntr = std::min(traced, photon_num);
        for (int i = 0; i < photon_num; i++) {
            if (i < ntr) {
                std::mt19937 before = ph_sampler.engine_;
                ppm::create_photon(edupt::spheres[ppm::LightID], ph_sampler, &point_map, gather_r2, gather_k);
                int words = 0;
                while (!(before == ph_sampler.engine_)) { before(); words++; }
                words_per_photon.push_back(words);
                std::vector<std::pair<int, int>> acc;
                for (size_t j = 0; j < Hn; j++) {
                    auto& p = pts[j];
                    const int id = id_at[j];
                    if (p.photon_count != prev[id]) { acc.push_back({id, p.photon_count - prev[id]}); prev[id] = p.photon_count; }
                }
                std::sort(acc.begin(), acc.end());
                for (auto& a : acc) { acc_hp.push_back(a.first); acc_cnt.push_back(a.second); }
                acc_off.push_back((int32_t)acc_hp.size());
            } else
                ppm::create_photon(edupt::spheres[ppm::LightID], ph_sampler, &point_map, gather_r2, gather_k);
        }
        dump("ppm_ph_words.bin", words_per_photon);
        dump("ppm_acc_off.bin", acc_off);
        dump("ppm_acc_hp.bin", acc_hp);
        dump("ppm_acc_cnt.bin", acc_cnt);
    }
    {   // per-hitpoint state after the photons, by creation id
        std::vector<int32_t> st_n(Hn, -1);
        std::vector<double> st_f(4 * Hn, 0.0);
        for (size_t j = 0; j < Hn; j++) {
            auto& p = pts[j];
            const int id = id_at[j];
            st_n[id] = p.photon_count;
            st_f[4 * (size_t)id] = p.photon_radius_2;
            st_f[4 * (size_t)id + 1] = p.accumulated_flux.x; st_f[4 * (size_t)id + 2] = p.accumulated_flux.y; st_f[4 * (size_t)id + 3] = p.accumulated_flux.z;
        }
        dump("ppm_state_n.bin", st_n);
        dump("ppm_state_f.bin", st_f);
    }
    // image assembly exactly as render() (:806-812)
    std::vector<Color> image(P);
    for (auto node : point_map.GetData())
        image[node.index] = image[node.index] + (node.emission + node.weight * node.accumulated_flux * (1.0 / (M_PI * node.photon_radius_2 * photon_num))) / spp;
    {
        std::vector<double> img;
        for (int i = 0; i < P; i++) push3(img, image[i]);
        dump("ppm_image.bin", img);
    }
    photonmap::utility::save_hdr_file(g_out + "/ppm_final.hdr", image.data(), W, H);
    edupt::save_ppm_file(g_out + "/ppm_final.ppm", image.data(), W, H);
    std::cout.rdbuf(cout_buf);
    fprintf(stderr, "[ref ppm] %zu hitpoints for %d pixels, %d photons\n", Hn, P, photon_num);
    return 0;
}
