// Exercises the class API of rt_host.hpp the way a caller of the reference would and dumps what it sees as raw
// doubles, for the Python tests to compare with fixtures made by the real reference.
//   mirror_check host <scene.json> <camera_id> <out.bin>      no GPU needed
//   mirror_check gpu  <scene.json> <camera_id> <out.bin>      stride-31 sample of the primary rays
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "raytracer.h"

static int g_failed = 0;
#define CHECK(cond) do { if (!(cond)) { std::cerr << "CHECK failed line " << __LINE__ << ": " #cond "\n"; ++g_failed; } } while (0)

static void put(std::vector<double>& out, double v) { out.push_back(v); }

static void put_hit(std::vector<double>& out, const Hit& h, int index) {
    const bool hit = h.hasDistanceAlongRay();
    put(out, hit ? 1.0 : 0.0); put(out, hit ? index : -1.0); put(out, hit ? h.getDistanceAlongRay() : INFINITY);
    const HitVec3 z{ 0, 0, 0 };
    const HitVec3 p = hit ? h.getIntersectionPoint() : z, n = hit ? h.getSurfaceNormal() : z, r = hit ? h.getReflectedDirection() : z;
    put(out, p.x); put(out, p.y); put(out, p.z); put(out, n.x); put(out, n.y); put(out, n.z); put(out, r.x); put(out, r.y); put(out, r.z);
    put(out, hit && h.hasUV() ? h.getU() : 0.0); put(out, hit && h.hasUV() ? h.getV() : 0.0);
}

static int index_of(const std::vector<const Mesh*>& meshes, const Mesh* m) {
    for (size_t i = 0; i < meshes.size(); ++i) if (meshes[i] == m) return static_cast<int>(i);
    return -1;
}

template <class F> static bool throws_runtime_error(F f) {
    try { f(); } catch (const std::runtime_error&) { return true; } catch (...) { return false; }
    return false;
}

static void host_self_checks(const std::string& tmp_prefix) {
    // Hit: presence flags and throwing getters (utils.h:78-117)
    Hit h;
    CHECK(!h.hasDistanceAlongRay() && !h.hasMesh() && !h.hasUV());
    CHECK(throws_runtime_error([&] { (void)h.getDistanceAlongRay(); }));
    CHECK(throws_runtime_error([&] { (void)h.getIntersectionPoint(); }));
    CHECK(throws_runtime_error([&] { (void)h.getMesh(); }));
    CHECK(throws_runtime_error([&] { (void)h.getU(); }));
    h.setDistanceAlongRay(2.5); h.setUV(0.25, 0.75);
    CHECK(h.hasDistanceAlongRay() && h.getDistanceAlongRay() == 2.5 && h.getV() == 0.75);
    Hit full(HitVec3{ 1, 2, 3 }, 4.0, HitVec3{ 0, 0, 1 }, HitVec3{ 0, 1, 0 }, nullptr);
    CHECK(full.hasMesh() && full.getMesh() == nullptr && full.getIntersectionPoint().y == 2);

    // Image: clamp to maxval, P6 round trip, P3 read, error cases (utils.cpp:176-250)
    Image img(3, 2);
    img.setPixel(0, 0, 300, -5, 17); img.addToPixel(0, 0, 0, 10, 250);
    CHECK(img.getPixel(0, 0).r == 255 && img.getPixel(0, 0).g == 10 && img.getPixel(0, 0).b == 255);
    CHECK(throws_runtime_error([&] { img.setMaxValue(0); }));
    CHECK(throws_runtime_error([&] { img.setMaxValue(256); }));
    img.setPixel(2, 1, 1, 2, 3);
    const std::string p6 = tmp_prefix + "_p6.ppm", p3 = tmp_prefix + "_p3.ppm";
    img.write(p6);
    Image back(p6);
    CHECK(back.width() == 3 && back.height() == 2 && back.maxValue() == 255 && back.getPixel(2, 1).b == 3 && back.getPixel(0, 0).r == 255);
    { std::ofstream f(p3); f << "P3\n# a comment\n2 1\n# another\n100\n1 2 3  100 0 50\n"; }
    Image txt(p3);
    CHECK(txt.width() == 2 && txt.height() == 1 && txt.maxValue() == 100 && txt.getPixel(1, 0).r == 100 && txt.getPixel(1, 0).b == 50);
    { std::ofstream f(p3); f << "P3\n1 1\n255\n1 2 999\n"; }
    CHECK(throws_runtime_error([&] { Image bad(p3); }));
    { std::ofstream f(p3); f << "P5\n1 1\n255\n"; }
    CHECK(throws_runtime_error([&] { Image bad(p3); }));
    CHECK(throws_runtime_error([&] { Image missing(tmp_prefix + "_does_not_exist.ppm"); }));
    CHECK(throws_runtime_error([&] { img.write("/nonexistent_dir_for_test/x.ppm"); }));
    std::remove(p6.c_str()); std::remove(p3.c_str());

    // util_json (utils.cpp:11-69)
    const std::string doc = "{ \"A\": { \"s\": \"br}ace\", \"v\": { \"x\": 1.5, \"y\": -2e1, \"z\": 3 } }, \"L\": [ 1, [2], 3 ], \"res\": { \"x\": 640, \"y\": 480 }, \"n\": 7 }";
    std::string block; double x = 0, y = 0, z = 0; int ix = 0, iy = 0;
    CHECK(util_json::extract_object_block(doc, "A", block) && block.find("br}ace") != std::string::npos && block.find("\"z\": 3 }") != std::string::npos);
    CHECK(util_json::extract_object_block(doc, "L", block) && block == " 1, [2], 3 ");
    CHECK(!util_json::extract_object_block(doc, "n", block) && !util_json::extract_object_block(doc, "absent", block));
    CHECK(util_json::parse_vec3(doc, "v", x, y, z) && x == 1.5 && y == -20.0 && z == 3.0);
    CHECK(util_json::parse_number(doc, "n", x) && x == 7.0);
    CHECK(!util_json::parse_number(doc, "s", x));
    CHECK(util_json::parse_vec2i(doc, "res", ix, iy) && ix == 640 && iy == 480);
    CHECK(!util_json::parse_vec2i(doc, "v", ix, iy));

    // Camera: defaults, look-at, pixel <-> world (persp_camera.cpp:19-118)
    Camera cam;
    int w = 0, hh = 0; cam.getImageResolution(w, hh);
    CHECK(w == 1920 && hh == 1080);
    cam.setImageResolution(640, 360);
    cam.updateRotationMatrix(Vec3(1, -8, 3), Vec3(0.1, 1, -0.3));
    Vec3 o, d; cam.pixelToRay(100.5, 200.5, o, d);
    CHECK(std::fabs(d.length() - 1.0) < 1e-12 && o.x == 1 && o.y == -8 && o.z == 3);
    int px = -1, py = -1;
    CHECK(cam.worldToPixel(o + d * 5.0, px, py) && px == 100 && py == 200);
    CHECK(!cam.worldToPixel(o - d * 5.0, px, py));
    cam.updateRotationMatrix(Vec3(0, 0, 5), Vec3(0, 0, -1));                 // looking straight down: fallback up axis
    cam.pixelToRay(320, 180, o, d);
    CHECK(std::fabs(d.z + 1.0) < 1e-12);
    cam.updateRotationMatrix(Vec3(0, 0, 5), Vec3(0, 1, 0), Vec3(1, 0, 0.2)); // explicit up keeps the roll
    cam.pixelToRay(320, 0, o, d);
    CHECK(d.x > 0.0);
    CHECK(Vec3(0, 0, 0).normalized().length() == 0.0);

    // per-class readers on text blocks (mesh.cpp:52, 316, 544, 707) with defaults
    std::vector<Cube> cubes = Cube::read_from_json("\"1\": { \"translation\": {\"x\":1,\"y\":2,\"z\":3}, \"material\": {\"metallic\": 0.25} }, \"2\": { }");
    CHECK(cubes.size() == 2 && cubes[0].getMetallic() == 0.25 && cubes[0].getRoughness() == 0.5 && cubes[1].getIndexOfRefraction() == 1.5);
    Float3 lo, hi; cubes[0].compute_aabb(lo, hi);
    CHECK(lo.x == 0 && hi.x == 2 && lo.z == 2 && hi.z == 4);
    std::vector<Sphere> spheres = Sphere::read_from_json("\"7\": { \"location\": {\"x\":0,\"y\":0,\"z\":1}, \"scale\": {\"x\":2,\"y\":1,\"z\":1} }");
    CHECK(spheres.size() == 1);
    spheres[0].compute_aabb(lo, hi);
    CHECK(lo.x == -2 && hi.x == 2 && lo.z == 0 && hi.z == 2);
    std::vector<Plane> planes = Plane::read_from_json("\"3\": { \"corners\": [ {\"x\":0,\"y\":0,\"z\":0}, {\"x\":1,\"y\":0,\"z\":0}, {\"x\":1,\"y\":1,\"z\":0}, {\"x\":0,\"y\":1,\"z\":0.5} ] }");
    CHECK(planes.size() == 1);
    planes[0].compute_aabb(lo, hi);
    CHECK(lo.x == 0 && hi.x == 1 && hi.z == 0.5);
    CHECK(Cylinder::read_from_json("\"4\": { \"scale\": {\"x\":1,\"y\":1,\"z\":2} }").size() == 1);
    std::ostringstream os; cubes[0].write_to_console(os);
    CHECK(os.str().rfind("Cube(translation=[1, 2, 3]", 0) == 0);
    // evaluate_albedo: solid colour, then a 2x2 texture sampled nearest with wrap (mesh.cpp:21-45)
    cubes[0].setAlbedo(0.5, 2.0, -1.0);
    Hit uvh; uvh.setUV(1.75, 0.2);
    Pixel a = cubes[0].evaluate_albedo(uvh);
    CHECK(a.r == 128 && a.g == 255 && a.b == 0);
    std::unique_ptr<Image> tex(new Image(2, 2));
    tex->setPixel(0, 0, 10, 0, 0); tex->setPixel(1, 0, 20, 0, 0); tex->setPixel(0, 1, 30, 0, 0); tex->setPixel(1, 1, 40, 0, 0);
    cubes[0].setAlbedoTexture(std::move(tex));
    CHECK(cubes[0].evaluate_albedo(uvh).r == 10);       // u = .75 -> floor(.75 * 1) = 0, v = .2 -> 0
    uvh.setUV(1.0 - 1e-12, 0.9999999999999);            // floor(u * (w - 1)) only reaches the last texel at u == 1, which wraps to 0
    CHECK(cubes[0].evaluate_albedo(uvh).r == 10);

    // BoundingBox slab test (mesh.cpp:942-967): entry distance from outside, exit distance from inside
    BoundingBox box(Float3{ -1, -1, -1 }, Float3{ 1, 1, 1 });
    Hit bh;
    CHECK(box.intersect(Ray(RayVec3{ -3, 0, 0 }, RayVec3{ 1, 0, 0 }), bh) && bh.getDistanceAlongRay() == 2.0);
    CHECK(box.intersect(Ray(RayVec3{ 0, 0, 0 }, RayVec3{ 0, 0, 2 }), bh) && bh.getDistanceAlongRay() == 0.5);
    CHECK(!box.intersect(Ray(RayVec3{ -3, 2, 0 }, RayVec3{ 1, 0, 0 }), bh));
    CHECK(!box.intersect(Ray(RayVec3{ 3, 0, 0 }, RayVec3{ 1, 0, 0 }), bh));
    CHECK(box.asBoundingBox() == &box && cubes[0].asBoundingBox() == nullptr);
}

int main(int argc, char** argv) {
    if (argc != 5) { std::cerr << "usage: mirror_check host|gpu <scene.json> <camera_id> <out.bin>\n"; return 64; }
    const std::string mode = argv[1], scene = argv[2], camera = argv[3], out_path = argv[4];
    std::vector<double> out;
    try {
        RayTracer rt;
        CHECK(!rt.load_from_json(scene + ".does_not_exist"));
        CHECK(rt.load_from_json(scene));
        const std::vector<const Mesh*>& meshes = rt.get_scene_mesh_ptrs();
        CHECK(rt.get_one_ray_per_pixel("no such camera").empty());
        const std::vector<Ray> rays = rt.get_one_ray_per_pixel(camera);
        put(out, static_cast<double>(meshes.size())); put(out, static_cast<double>(rays.size()));
        for (const Mesh* m : meshes) {
            Float3 lo, hi; m->compute_aabb(lo, hi);
            put(out, lo.x); put(out, lo.y); put(out, lo.z); put(out, hi.x); put(out, hi.y); put(out, hi.z);
        }
        std::vector<Ray> sample;
        for (size_t i = 0; i < rays.size(); i += 31) sample.push_back(rays[i]);
        put(out, static_cast<double>(sample.size()));
        for (const Ray& r : sample) {
            put(out, r.getPosition().x); put(out, r.getPosition().y); put(out, r.getPosition().z);
            put(out, r.getDirection().x); put(out, r.getDirection().y); put(out, r.getDirection().z);
        }
        Camera cam;
        CHECK(read_camera_from_json(scene, camera, cam));
        CHECK(!read_camera_from_json(scene, "no such camera", cam));
        if (mode == "host") {
            host_self_checks(out_path);
        } else {
            // whole-scene closest hits, batch and single-ray forms
            const std::vector<Hit> hits = rt.one_pass_intersection(sample);
            for (const Hit& h : hits) put_hit(out, h, h.hasMesh() ? index_of(meshes, h.getMesh()) : -1);
            // the tree API: N - 1 boxes for N bounded objects, same hits
            std::vector<std::unique_ptr<BoundingBox>> store;
            const Mesh* root = get_BVH_tree(meshes, store);
            put(out, static_cast<double>(store.size()));
            const std::vector<Hit> fast = accelerated_one_pass_intersection(root, sample);
            for (const Hit& h : fast) put_hit(out, h, h.hasMesh() ? index_of(meshes, h.getMesh()) : -1);
            // linear shading of those hits
            const std::vector<Pixel> colours = rt.shade(hits, sample);
            for (const Pixel& p : colours) { put(out, p.r); put(out, p.g); put(out, p.b); }
            // single-ray entry points agree with the batch ones; the hit mesh alone gives the same record
            size_t n_single = 0;
            for (size_t i = 0; i < sample.size() && n_single < 24; i += 1 + sample.size() / 24) {
                Hit a, b;
                const bool ha = rt.one_pass_intersection(sample[i], a);
                const bool hb = accelerated_one_pass_intersection(root, sample[i], b);
                CHECK(ha == hits[i].hasDistanceAlongRay() && hb == ha);
                if (ha) {
                    CHECK(a.getDistanceAlongRay() == hits[i].getDistanceAlongRay() && a.getMesh() == hits[i].getMesh());
                    CHECK(b.getDistanceAlongRay() == a.getDistanceAlongRay() && b.getMesh() == a.getMesh());
                    Hit alone;
                    CHECK(a.getMesh()->intersect(sample[i], alone) && !alone.hasMesh());
                    CHECK(alone.getDistanceAlongRay() == a.getDistanceAlongRay() && alone.getSurfaceNormal().x == a.getSurfaceNormal().x);
                    const Pixel one = rt.shade(a, sample[i]);
                    CHECK(one.r == colours[i].r && one.g == colours[i].g && one.b == colours[i].b);
                    ++n_single;
                }
            }
            Hit none; none.setDistanceAlongRay(1.0);
            const Pixel black = rt.shade(none, sample[0]);               // no normal / point: black (raytracer.cpp:161)
            CHECK(black.r == 0 && black.g == 0 && black.b == 0);
            CHECK(!rt.render_unaccelerated_ppm("no such camera", out_path + ".ppm"));
            // one-object and empty trees (raytracer.cpp:455-456)
            std::vector<std::unique_ptr<BoundingBox>> small_store;
            CHECK(get_BVH_tree(std::vector<const Mesh*>{}, small_store) == nullptr);
            if (!meshes.empty()) {
                const Mesh* lone = get_BVH_tree(std::vector<const Mesh*>{ meshes[0] }, small_store);
                CHECK(lone == meshes[0] && small_store.empty());
                Hit lh; (void)accelerated_one_pass_intersection(lone, sample[0], lh);
            }
            Hit nh; CHECK(!accelerated_one_pass_intersection(nullptr, sample[0], nh));
        }
    } catch (const std::exception& e) {
        std::cerr << "mirror_check: exception: " << e.what() << "\n";
        return 2;
    }
    std::ofstream f(out_path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(out.data()), static_cast<std::streamsize>(out.size() * sizeof(double)));
    if (g_failed) { std::cerr << g_failed << " checks failed\n"; return 1; }
    std::cout << "ok " << out.size() << " values\n";
    return 0;
}
