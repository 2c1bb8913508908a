// edupt::PinholeCamera::get_ray (src/cameras.h:34-64) and edupt::intersect_scene (src/scene.h:53-76) of the reference,
// written against include/sppm2_hip.hpp: the first-hit sphere of every pixel's camera ray, in the reference's render order.
//   hipcc -std=c++17 -I include examples/first_hits.cpp -L raytracer_amd -lsppm_hip -Wl,-rpath,$PWD/raytracer_amd -o first_hits
// Prints one line per pixel: object id, then the ray's direction and the hit distance as hexadecimal doubles.
#include <cstdio>
#include <cstdlib>

#include "sppm2_hip.hpp"

int main(int argc, char** argv)
{
    using namespace edupt;
    const int width = argc > 1 ? atoi(argv[1]) : 64, height = argc > 2 ? atoi(argv[2]) : 48;
    // the camera StochasticPpm<W,H> constructs, src/sppm2.h:48-56,81-86
    PinholeCamera camera(width, height, 30.0, 40.0, Vec(50.0, 52.0, 220.0), normalize(Vec(0.0, -0.04, -1.0)), Vec(0.0, 1.0, 0.0), 1);
    std::vector<Ray> rays;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) rays.push_back(camera.get_ray(x, y, 0, 0));
    std::vector<Intersection> hits;
    if (intersect_scene(rays, &hits) != 0) { fprintf(stderr, "intersect_scene failed: %s\n", sppm_last_error()); return 1; }
    for (size_t i = 0; i < rays.size(); i++)
        printf("%d %a %a %a %a\n", hits[i].object_id, rays[i].dir.x, rays[i].dir.y, rays[i].dir.z, hits[i].hitpoint.distance);
    // the single-ray form of the reference's signature, on the centre pixel
    Intersection one;
    const bool hit = intersect_scene(camera.get_ray(width / 2, height / 2), &one);
    const size_t c = (size_t)(height / 2) * width + width / 2;
    if (hit != (hits[c].object_id != -1) || one.object_id != hits[c].object_id || one.hitpoint.distance != hits[c].hitpoint.distance) { fprintf(stderr, "single-ray form disagrees\n"); return 2; }
    fprintf(stderr, "%zu spheres (edupt::spheres), %zu rays\n", spheres.size(), rays.size());
    return 0;
}
