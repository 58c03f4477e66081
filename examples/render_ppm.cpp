// examples/render_ppm.cpp — the C++ front end end to end: build a World with the reference's builder names,
// render it on the MI355X through the C ABI and write a PPM (the reference's examples end in Canvas::to_ppm / PNG).
//   g++ -std=c++17 -Iinclude examples/render_ppm.cpp -Lraytracing_amd -lrtmi -Wl,-rpath,$PWD/raytracing_amd -o render_ppm
#include <cmath>
#include <cstdio>
#include <fstream>

#include "rtmi.hpp"

using namespace rtmi;

int main(int argc, char **argv) {
    const uint32_t hsize = argc > 2 ? (uint32_t)atoi(argv[2]) : 640, vsize = hsize / 2;
    World w = World::empty();
    w.add_light(PointLight(point(-10, 10, -10), color(1, 1, 1)));
    w.add_item(plane().with_material(Phong().with_pattern(checkers_pattern(color(0.2, 0.2, 0.2), color(0.8, 0.8, 0.8))).with_reflective(0.2)));
    w.add_item(sphere().with_transform(translation(-0.5, 1, 0.5)).with_material(Phong().with_rgb(0.1, 1.0, 0.5).with_diffuse(0.7).with_specular(0.3)));
    w.add_item(glass_sphere().with_transform(translation(1.5, 0.5, -0.5) * scaling(0.5, 0.5, 0.5)));
    w.finalize_scene();
    Camera cam = Camera(hsize, vsize, M_PI / 3).with_view_transform(point(0, 1.5, -5), point(0, 1, 0), vector(0, 1, 0));
    cam.set_allowed_standard_error(1e-2);
    rtmi_render_opts opts{};
    opts.seed = 0x5EED;
    rtmi_stats stats{};
    try {
        Canvas canvas = cam.render(w, &opts, &stats); // throws std::runtime_error carrying rtmi_last_error()
        std::ofstream(argc > 1 ? argv[1] : "out.ppm") << canvas.to_ppm();
        double rays = (double)(stats.counter[RTMI_STAT_PRIMARY] + stats.counter[RTMI_STAT_REFLECTION] + stats.counter[RTMI_STAT_REFRACTION] +
                               stats.counter[RTMI_STAT_SHADOW]);
        std::printf("%ux%u  %.0f rays  %.3f ms  %.1f Mrays/s\n", hsize, vsize, rays, stats.kernel_ms, rays / stats.kernel_ms / 1e3);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "render failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
