// Command-line renderer with the interface of s2899221/Code/render_driver.cpp:11-44:
//   render_driver <scene.json> <camera_id> <output.ppm> [samples_per_pixel] [max_scatters]
// (max_scatters is an addition; the default 2 is RayTracer::render_unaccelerated_ppm's.)
#include <algorithm>
#include <exception>
#include <iostream>
#include <string>

#include "raytracer.h"

int main(int argc, char** argv) {
    if (argc < 4 || argc > 6) {
        std::cerr << "Usage: render_driver <scene.json> <camera_id> <output.ppm> [samples_per_pixel] [max_scatters]\n"
                  << "- scene.json: path to JSON scene file\n- camera_id: id string of camera to render\n"
                  << "- output.ppm: destination PPM file path\n- samples_per_pixel: optional integer >=1 (default 1)\n"
                  << "- max_scatters: optional mirror-bounce depth (default 2)\n";
        return 1;
    }
    try {
        const std::string scene = argv[1], camera = argv[2], output = argv[3];
        const int spp = (argc >= 5) ? std::max(1, std::stoi(argv[4])) : 1;
        const int scatters = (argc >= 6) ? std::max(0, std::stoi(argv[5])) : 2;
        RayTracer rt;
        if (!rt.load_from_json(scene)) { std::cerr << "Failed to load scene: " << scene << "\n"; return 1; }
        if (!rt.render_unaccelerated_ppm(camera, output, spp, scatters)) {
            std::cerr << "Render failed (bad camera id or write error)\n";
            return 1;
        }
        std::cout << "Wrote image to: " << output << "\n";
    } catch (const std::exception& e) {
        std::cerr << "render_driver: " << e.what() << "\n";
        return 2;
    }
    return 0;
}
