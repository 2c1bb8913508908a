// The reference's live call site, src/main.cpp:74-88 (classic progressive photon mapping), against this build.
//   hipcc -std=c++17 -I include examples/render_ppm.cpp -L raytracer_amd -lsppm_hip -Wl,-rpath,$PWD/raytracer_amd -o render_ppm
#include <cstdio>
#include <cstdlib>
#include <sstream>

#include "progressive_photonmapping_hip.hpp"

int main(int argc, char** argv)
{
    const int width = 320, height = 240;
    const int samples = 1, supersamples = argc > 1 ? atoi(argv[1]) : 2;
    const int photon_num = argc > 2 ? atoi(argv[2]) : 10000000;
    const double gather_photon_radius = 5;
    photonmap::progressive::InitialRadius = gather_photon_radius;  // src/main.cpp:77
    const int gahter_max_photon_num = 10000;
    std::stringstream ss;
    ss << "image_ppm_" << width << "_" << height << "_" << samples << "_" << supersamples << "_" << photon_num << "_" << gather_photon_radius;
    const int rc = photonmap::progressive::render(argc > 3 ? std::string(argv[3]) : ss.str(), width, height, samples, supersamples, photon_num, gather_photon_radius, gahter_max_photon_num);
    if (rc) fprintf(stderr, "render failed: %s\n", sppm_last_error());
    return rc ? 1 : 0;
}
