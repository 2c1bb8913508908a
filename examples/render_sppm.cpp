// The reference's (dead) SPPM call site, src/main.cpp:91-93, against this build.
//   hipcc -std=c++17 -I include examples/render_sppm.cpp -L raytracer_amd -lsppm_hip -Wl,-rpath,$PWD/raytracer_amd -o render_sppm
#include <cstdio>
#include <cstdlib>

#include "sppm2_hip.hpp"

int main(int argc, char** argv)
{
    constexpr int width = 320, height = 240;
    const int passes = argc > 1 ? atoi(argv[1]) : 16;
    const int photon_num = argc > 2 ? atoi(argv[2]) : 200000;
    photonmap::sppm2::StochasticPpm<width, height> sppm(1, 1);
    sppm.write_pass_files = false;
    const char* name = argc > 3 ? argv[3] : "image_sppm";  // output prefix: <name>.ppm, <name>.hdr
    const int rc = sppm.render(name, width, height, 1, 1, photon_num, 25, 10000, passes);
    if (rc) fprintf(stderr, "render failed: %s\n", sppm_last_error());
    return rc ? 1 : 0;
}
