// Implementation of the reference's class API on the C ABI of librt_hip.so (see rt_host.hpp).
#include "rt_host.hpp"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <mutex>
#include <sstream>

// ---- plumbing -------------------------------------------------------------------------------------------

namespace rt_host {

// Scenes up to this size keep the reference's every-primitive loop behind RayTracer::one_pass_intersection.
constexpr size_t kBruteForceMaxPrims = 4096;
// Brute-force mode of a context for the lifetime of the object (rt_ctx_set_brute_force).
struct BruteForce {
    rt_ctx* ctx; bool on;
    BruteForce(rt_ctx* c, bool enable) : ctx(c), on(enable) { if (on) rt_ctx_set_brute_force(ctx, 1); }
    ~BruteForce() { if (on) rt_ctx_set_brute_force(ctx, 0); }
    BruteForce(const BruteForce&) = delete;
    BruteForce& operator=(const BruteForce&) = delete;
};

[[noreturn]] static void gpu_fail(const char* what, int rc) {
    throw std::runtime_error(std::string(what) + ": " + rt_strerror(rc) + " (" + rt_last_error() + ")");
}

static int device_index() {
    const char* env = std::getenv("RT_DEVICE");
    return env ? std::atoi(env) : 0;
}

// RT_GPUS=n (n > 1): frames are rendered on GPUs device .. device + n - 1, image tiles dealt round-robin, the scene
// replicated, the tiles gathered to the first GPU over RCCL (rt_render_frame_multi_host).  Everything else - single
// rays, timing passes, multi-sample frames - stays on the first GPU.
static int gpu_count() {
    const char* env = std::getenv("RT_GPUS");
    const int n = env ? std::atoi(env) : 1;
    return n < 1 ? 1 : (n > 64 ? 64 : n);
}

// One context per scene and GPU: a context holds one uploaded scene at a time.
struct GpuScene {
    rt_scene* scene = nullptr;
    rt_ctx* ctx = nullptr;                       // the first GPU
    std::vector<rt_ctx*> all;                    // RT_GPUS > 1: every GPU's context, all[0] == ctx
    std::vector<rt_comm*> comms;
    GpuScene() = default;
    GpuScene(const GpuScene&) = delete;
    GpuScene& operator=(const GpuScene&) = delete;
    ~GpuScene() {
        for (rt_comm* c : comms) if (c) rt_comm_destroy(c);
        for (size_t k = 1; k < all.size(); ++k) if (all[k]) rt_ctx_destroy(all[k]);
        if (ctx) rt_ctx_destroy(ctx);
        if (scene) rt_scene_destroy(scene);
    }
    void upload() {
        if (ctx) return;
        if (int rc = rt_ctx_create(device_index(), &ctx)) { ctx = nullptr; gpu_fail("no GPU context", rc); }
        if (int rc = rt_scene_upload(ctx, scene)) gpu_fail("scene upload", rc);
    }
    // contexts on the other GPUs and the group over all of them, on first use
    void spread(int n) {
        if (!comms.empty() || n < 2) return;
        upload();
        all.assign(1, ctx);
        for (int k = 1; k < n; ++k) {
            rt_ctx* c = nullptr;
            if (int rc = rt_ctx_create(device_index() + k, &c)) gpu_fail("RT_GPUS: no context on a further GPU", rc);
            all.push_back(c);
            if (int rc = rt_scene_upload(c, scene)) gpu_fail("scene upload", rc);
        }
        comms.assign(n, nullptr);
        if (int rc = rt_comm_create_all(all.data(), n, comms.data())) { comms.clear(); gpu_fail("RT_GPUS: no RCCL group", rc); }
    }
};

// Context without a scene, for Mesh::intersect on a lone primitive.
static rt_ctx* bare_context() {
    static std::mutex mu;
    static rt_ctx* ctx = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ctx) if (int rc = rt_ctx_create(device_index(), &ctx)) { ctx = nullptr; gpu_fail("no GPU context", rc); }
    return ctx;
}

static void pack_rays(const std::vector<Ray>& rays, std::vector<double>& out) {
    out.resize(rays.size() * 6);
    for (size_t i = 0; i < rays.size(); ++i) {
        const RayVec3& o = rays[i].getPosition(); const RayVec3& d = rays[i].getDirection();
        double* r = &out[i * 6];
        r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
    }
}

// root node of a tree built by get_BVH_tree -> its device copy and the caller's object list
struct TreeEntry { std::unique_ptr<GpuScene> gpu; std::vector<const Mesh*> objects; };
static std::mutex g_tree_mu;
static std::unordered_map<const BoundingBox*, TreeEntry>& trees() {
    static std::unordered_map<const BoundingBox*, TreeEntry> t;
    return t;
}

} // namespace rt_host

using rt_host::GpuScene;

// ---- util_json --------------------------------------------------------------------------------------------

namespace util_json {

static bool is_space(char c) { return std::isspace(static_cast<unsigned char>(c)) != 0; }

// utils.cpp:11-51: first occurrence of "key", the next ':', then a brace or bracket block, string-aware matching.
bool extract_object_block(const std::string& src, const std::string& key, std::string& out_block) {
    const std::string quoted = "\"" + key + "\"";
    const size_t at = src.find(quoted);
    if (at == std::string::npos) return false;
    size_t i = src.find(':', at + quoted.size());
    if (i == std::string::npos) return false;
    for (++i; i < src.size() && is_space(src[i]); ++i) {}
    if (i >= src.size() || (src[i] != '{' && src[i] != '[')) return false;
    const char open = src[i], close = (open == '{') ? '}' : ']';
    int depth = 0;
    bool quoted_run = false, escaped = false;
    const size_t start = i;
    for (size_t j = start; j < src.size(); ++j) {
        const char c = src[j];
        if (quoted_run) {
            if (escaped) escaped = false;
            else if (c == '\\') escaped = true;
            else if (c == '"') quoted_run = false;
            continue;
        }
        if (c == '"') { quoted_run = true; continue; }
        if (c == open) ++depth;
        else if (c == close && --depth == 0) { out_block = src.substr(start + 1, j - start - 1); return true; }
    }
    return false;
}

static bool number_char(char c) { return c == '-' || c == '+' || c == 'e' || c == 'E' || c == '.' || (c >= '0' && c <= '9'); }

// after "key" at `at`: optional blanks, ':', optional blanks; returns the position after them or npos
static size_t after_colon(const std::string& s, size_t at) {
    while (at < s.size() && is_space(s[at])) ++at;
    if (at >= s.size() || s[at] != ':') return std::string::npos;
    for (++at; at < s.size() && is_space(s[at]); ++at) {}
    return at;
}

// utils.cpp:53-56: the first place where "key" : <run of [-+eE0-9.]> matches; std::stod on the run (and its exceptions).
bool parse_number(const std::string& src, const std::string& key, double& out) {
    const std::string quoted = "\"" + key + "\"";
    for (size_t at = src.find(quoted); at != std::string::npos; at = src.find(quoted, at + 1)) {
        const size_t v = after_colon(src, at + quoted.size());
        if (v == std::string::npos) continue;
        size_t e = v;
        while (e < src.size() && number_char(src[e])) ++e;
        if (e == v) continue;
        out = std::stod(src.substr(v, e - v));
        return true;
    }
    return false;
}

bool parse_vec3(const std::string& src, const std::string& key, double& x, double& y, double& z) {
    std::string block;
    if (!extract_object_block(src, key, block)) return false;
    double tx = 0, ty = 0, tz = 0;
    if (!parse_number(block, "x", tx) || !parse_number(block, "y", ty) || !parse_number(block, "z", tz)) return false;
    x = tx; y = ty; z = tz;
    return true;
}

static bool digits_at(const std::string& s, size_t v, size_t& end) {
    end = v;
    while (end < s.size() && s[end] >= '0' && s[end] <= '9') ++end;
    return end > v;
}

// utils.cpp:64-69: "x" : digits, at least one non-digit, "y" : digits, with only non-digits in between.
bool parse_vec2i(const std::string& src, const std::string& key, int& outx, int& outy) {
    std::string block;
    if (!extract_object_block(src, key, block)) return false;
    const std::string qx = "\"x\"", qy = "\"y\"";
    for (size_t at = block.find(qx); at != std::string::npos; at = block.find(qx, at + 1)) {
        const size_t v = after_colon(block, at + qx.size());
        size_t xe = 0;
        if (v == std::string::npos || !digits_at(block, v, xe)) continue;
        size_t run_end = xe;                          // the run of non-digits after the x value
        while (run_end < block.size() && !(block[run_end] >= '0' && block[run_end] <= '9')) ++run_end;
        for (size_t y_at = block.rfind(qy, run_end); y_at != std::string::npos && y_at > xe; y_at = (y_at ? block.rfind(qy, y_at - 1) : std::string::npos)) {
            const size_t yv = after_colon(block, y_at + qy.size());
            size_t ye = 0;
            if (yv == std::string::npos || yv > run_end || !digits_at(block, yv, ye)) continue;
            outx = std::stoi(block.substr(v, xe - v));
            outy = std::stoi(block.substr(yv, ye - yv));
            return true;
        }
    }
    return false;
}

} // namespace util_json

// ---- Hit ---------------------------------------------------------------------------------------------------

Hit::Hit(std::optional<HitVec3> p, std::optional<double> t, std::optional<HitVec3> n, std::optional<HitVec3> r, std::optional<const Mesh*> mesh) {
    if (p) setIntersectionPoint(*p);
    if (t) setDistanceAlongRay(*t);
    if (n) setSurfaceNormal(*n);
    if (r) setReflectedDirection(*r);
    if (mesh) setMesh(*mesh);
}

Hit Hit::from_record(const double* rec, const Mesh* mesh) {
    Hit h;
    if (rec[0] == 0.0) return h;
    h.setDistanceAlongRay(rec[2]);
    h.setIntersectionPoint(HitVec3{ rec[3], rec[4], rec[5] });
    h.setSurfaceNormal(HitVec3{ rec[6], rec[7], rec[8] });
    h.setReflectedDirection(HitVec3{ rec[9], rec[10], rec[11] });
    h.setUV(rec[12], rec[13]);
    if (mesh) h.setMesh(mesh);
    return h;
}

// ---- Image -------------------------------------------------------------------------------------------------

static unsigned char clamp_channel(int v, int maxv) { return static_cast<unsigned char>(v < 0 ? 0 : (v > maxv ? maxv : v)); }

Image::Image(int width, int height)
    : m_w(width), m_h(height), m_max(255), m_px(static_cast<size_t>(width) * static_cast<size_t>(height), Pixel{ 0, 0, 0 }) {}

// utils.cpp:147-154, 212-250: P3 or P6, '#' comments between header fields, maxval <= 255.
static void skip_blank_and_comments(std::istream& in) {
    for (int c = in.peek(); c != EOF && (std::isspace(c) || c == '#'); c = in.peek()) {
        if (c == '#') { std::string rest; std::getline(in, rest); } else in.get();
    }
}

Image::Image(const std::string& filename) : m_w(0), m_h(0), m_max(255) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) throw std::runtime_error("Failed to open image file: " + filename);
    std::string magic; in >> magic;
    if (magic != "P6" && magic != "P3") throw std::runtime_error("Unsupported PPM format (expected P3 or P6): " + magic);
    int maxval = 0;
    skip_blank_and_comments(in); in >> m_w;
    skip_blank_and_comments(in); in >> m_h;
    skip_blank_and_comments(in); in >> maxval; in.get();
    if (m_w <= 0 || m_h <= 0 || maxval <= 0) throw std::runtime_error("Invalid PPM header values");
    if (maxval > 255) throw std::runtime_error("Unsupported PPM max value (>255 not supported)");
    m_max = maxval;
    const size_t n = static_cast<size_t>(m_w) * static_cast<size_t>(m_h);
    m_px.resize(n);
    if (magic == "P6") {
        std::vector<unsigned char> raw(n * 3);
        in.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(raw.size()));
        if (static_cast<size_t>(in.gcount()) != raw.size()) throw std::runtime_error("Unexpected EOF while reading binary PPM data");
        assign_rgb(raw.data());
    } else {
        static const char* const names[3] = { "r", "g", "b" };
        for (size_t i = 0; i < n; ++i) {
            int c[3];
            for (int k = 0; k < 3; ++k) {
                skip_blank_and_comments(in);
                if (!(in >> c[k])) throw std::runtime_error(std::string("Unexpected EOF reading P3 data (") + names[k] + ")");
            }
            for (int k = 0; k < 3; ++k) if (c[k] < 0 || c[k] > maxval) throw std::runtime_error("P3 pixel out of range");
            m_px[i] = Pixel{ static_cast<unsigned char>(c[0]), static_cast<unsigned char>(c[1]), static_cast<unsigned char>(c[2]) };
        }
    }
}

Pixel Image::getPixel(int x, int y) const { return m_px[static_cast<size_t>(y) * static_cast<size_t>(m_w) + static_cast<size_t>(x)]; }

void Image::setPixel(int x, int y, int r, int g, int b) {
    m_px[static_cast<size_t>(y) * static_cast<size_t>(m_w) + static_cast<size_t>(x)] =
        Pixel{ clamp_channel(r, m_max), clamp_channel(g, m_max), clamp_channel(b, m_max) };
}

void Image::addToPixel(int x, int y, int dr, int dg, int db) {
    const Pixel p = getPixel(x, y);
    setPixel(x, y, p.r + dr, p.g + dg, p.b + db);
}

void Image::setMaxValue(int maxVal) {
    if (maxVal < 1 || maxVal > 255) throw std::runtime_error("Invalid PPM max value, must be [1,255].");
    m_max = maxVal;
}

void Image::assign_rgb(const unsigned char* rgb) {
    for (size_t i = 0; i < m_px.size(); ++i) m_px[i] = Pixel{ rgb[3*i], rgb[3*i + 1], rgb[3*i + 2] };
}

void Image::write(const std::string& filename) const {
    std::ofstream out(filename, std::ios::binary);
    if (!out) throw std::runtime_error("Failed to open file for writing: " + filename);
    out << "P6\n" << m_w << " " << m_h << "\n" << m_max << "\n";
    std::vector<unsigned char> raw(m_px.size() * 3);
    for (size_t i = 0; i < m_px.size(); ++i) {
        raw[3*i] = clamp_channel(m_px[i].r, m_max); raw[3*i + 1] = clamp_channel(m_px[i].g, m_max); raw[3*i + 2] = clamp_channel(m_px[i].b, m_max);
    }
    out.write(reinterpret_cast<const char*>(raw.data()), static_cast<std::streamsize>(raw.size()));
    if (!out) throw std::runtime_error("Failed while writing image data to: " + filename);
}

// ---- Vec3 / Camera --------------------------------------------------------------------------------------------

double Vec3::length() const { return std::sqrt(x*x + y*y + z*z); }
Vec3 Vec3::normalized() const { const double len = length(); return len == 0 ? Vec3(0, 0, 0) : Vec3(x/len, y/len, z/len); }

Camera::Camera() {
    std::memset(&m_cam, 0, sizeof(m_cam));
    m_cam.rot[0] = m_cam.rot[4] = m_cam.rot[8] = 1.0;
    m_cam.focal = 35.0; m_cam.sensor_w = 36.0; m_cam.sensor_h = 24.0; m_cam.width = 1920; m_cam.height = 1080;
}

void Camera::setPosition(const Vec3& pos) { m_cam.pos[0] = pos.x; m_cam.pos[1] = pos.y; m_cam.pos[2] = pos.z; }

void Camera::updateRotationMatrix(const Vec3& cam_pos, const Vec3& cam_dir) {
    const double p[3] = { cam_pos.x, cam_pos.y, cam_pos.z }, d[3] = { cam_dir.x, cam_dir.y, cam_dir.z };
    rt_camera_look(p, d, nullptr, &m_cam);
}

void Camera::updateRotationMatrix(const Vec3& cam_pos, const Vec3& cam_dir, const Vec3& cam_up) {
    const double p[3] = { cam_pos.x, cam_pos.y, cam_pos.z }, d[3] = { cam_dir.x, cam_dir.y, cam_dir.z }, u[3] = { cam_up.x, cam_up.y, cam_up.z };
    rt_camera_look(p, d, u, &m_cam);
}

// persp_camera.cpp:67-92: camera-space coordinates through the transposed basis, sensor-fit rule, truncation to int.
bool Camera::worldToPixel(const Vec3& w, int& pixel_x, int& pixel_y) {
    const double v[3] = { w.x - m_cam.pos[0], w.y - m_cam.pos[1], w.z - m_cam.pos[2] };
    const double* R = m_cam.rot;
    const double xc = R[0]*v[0] + R[3]*v[1] + R[6]*v[2];
    const double yc = R[1]*v[0] + R[4]*v[1] + R[7]*v[2];
    const double zc = R[2]*v[0] + R[5]*v[1] + R[8]*v[2];
    if (zc <= 0) return false;
    const double aspect = static_cast<double>(m_cam.width) / static_cast<double>(m_cam.height);
    const double sensor_aspect = (m_cam.sensor_h > 0.0) ? (m_cam.sensor_w / m_cam.sensor_h) : aspect;
    double ew, eh;
    if (aspect >= sensor_aspect) { ew = m_cam.sensor_w; eh = ew / aspect; } else { eh = m_cam.sensor_h; ew = eh * aspect; }
    const double xs = m_cam.focal * (xc / zc), ys = m_cam.focal * (yc / zc);
    pixel_x = static_cast<int>(((xs / ew) + 0.5) * m_cam.width);
    pixel_y = static_cast<int>((1.0 - ((ys / eh) + 0.5)) * m_cam.height);
    return true;
}

void Camera::pixelToRay(double pixel_x, double pixel_y, Vec3& o, Vec3& d) const {
    double ro[3], rd[3];
    rt_camera_pixel_to_ray(&m_cam, pixel_x, pixel_y, ro, rd);
    o = Vec3(ro[0], ro[1], ro[2]); d = Vec3(rd[0], rd[1], rd[2]);
}

bool read_camera_from_json(const std::string& json_filepath, const std::string& camera_id, Camera& camera) {
    rt_camera c;
    const int rc = rt_camera_from_json(json_filepath.c_str(), camera_id.c_str(), &c);
    if (rc == RT_ERR_IO) { std::cerr << "Failed to open JSON file: " << json_filepath << "\n"; return false; }
    if (rc == RT_ERR_PARSE) throw std::invalid_argument("stod");       // the reference lets std::stod's exception escape
    if (rc != RT_OK) return false;
    camera = Camera(c);
    return true;
}

// ---- Mesh and the primitive classes --------------------------------------------------------------------------------

static rt_prim_desc blank_desc(int kind) {
    rt_prim_desc d; std::memset(&d, 0, sizeof(d));
    d.kind = kind; d.alpha = 1.0; d.metallic = 0.0; d.roughness = 0.5; d.ior = 1.5;
    d.albedo[0] = d.albedo[1] = d.albedo[2] = 1.0;
    return d;
}

Mesh::Mesh() : m_desc(blank_desc(-1)) {}
Mesh::~Mesh() = default;
Mesh::Mesh(Mesh&&) noexcept = default;
Mesh& Mesh::operator=(Mesh&&) noexcept = default;
bool Mesh::intersect(const Ray&, Hit&) const { return false; }
void Mesh::setAlbedoTexture(std::unique_ptr<Image> texture) { m_albedoTexture = std::move(texture); }

// mesh.cpp:15-45
Pixel Mesh::evaluate_albedo(const Hit& hit) const {
    if (m_albedoTexture && hit.hasUV() && m_albedoTexture->width() > 0 && m_albedoTexture->height() > 0) {
        const int w = m_albedoTexture->width(), h = m_albedoTexture->height();
        double u = hit.getU(), v = hit.getV();
        u -= std::floor(u); v -= std::floor(v);
        int x = static_cast<int>(std::floor(u * static_cast<double>(w - 1)));
        int y = static_cast<int>(std::floor(v * static_cast<double>(h - 1)));
        x = std::min(std::max(x, 0), w - 1); y = std::min(std::max(y, 0), h - 1);
        return m_albedoTexture->getPixel(x, y);
    }
    auto q = [](double a) -> unsigned char { return a <= 0.0 ? 0 : (a >= 1.0 ? 255 : static_cast<unsigned char>(std::round(a * 255.0))); };
    return Pixel{ q(m_desc.albedo[0]), q(m_desc.albedo[1]), q(m_desc.albedo[2]) };
}

std::vector<Hit> Mesh::intersect(const std::vector<Ray>& rays) const {
    std::vector<Hit> out(rays.size());
    if (rays.empty() || m_desc.kind < 0) return out;
    std::vector<double> packed, rec(rays.size() * 16);
    rt_host::pack_rays(rays, packed);
    if (int rc = rt_intersect_prim_host(rt_host::bare_context(), &m_desc, packed.data(), rays.size(), rec.data()))
        rt_host::gpu_fail("Mesh::intersect", rc);
    for (size_t i = 0; i < rays.size(); ++i) out[i] = Hit::from_record(&rec[i * 16], nullptr);
    return out;
}

// The reference's intersectors fill point, distance, normal, reflected direction and UV, never the mesh.
bool Mesh::gpu_intersect(const Ray& ray, Hit& hit) const {
    double packed[6] = { ray.getPosition().x, ray.getPosition().y, ray.getPosition().z, ray.getDirection().x, ray.getDirection().y, ray.getDirection().z };
    double rec[16];
    if (int rc = rt_intersect_prim_host(rt_host::bare_context(), &m_desc, packed, 1, rec)) rt_host::gpu_fail("Mesh::intersect", rc);
    if (rec[0] == 0.0) return false;
    hit.setIntersectionPoint(HitVec3{ rec[3], rec[4], rec[5] });
    hit.setDistanceAlongRay(rec[2]);
    hit.setSurfaceNormal(HitVec3{ rec[6], rec[7], rec[8] });
    hit.setReflectedDirection(HitVec3{ rec[9], rec[10], rec[11] });
    hit.setUV(rec[12], rec[13]);
    return true;
}

void Mesh::gpu_aabb(Float3& outMin, Float3& outMax) const {
    double b[6];
    rt_prim_aabb(&m_desc, b);
    outMin = Float3{ b[0], b[1], b[2] }; outMax = Float3{ b[3], b[4], b[5] };
}

static void set_trs(rt_prim_desc& d, const Float3& t, const EulerAngles& r, const Float3& s) {
    d.q[0] = t.x; d.q[1] = t.y; d.q[2] = t.z; d.q[3] = r.roll; d.q[4] = r.pitch; d.q[5] = r.yaw; d.q[6] = s.x; d.q[7] = s.y; d.q[8] = s.z;
}

Cube::Cube() { m_desc = blank_desc(RT_CUBE); set_trs(m_desc, Float3{ 0, 0, 0 }, EulerAngles{ 0, 0, 0 }, Float3{ 1, 1, 1 }); }
Cube::Cube(const Float3& t, const EulerAngles& r, const Float3& s) { m_desc = blank_desc(RT_CUBE); set_trs(m_desc, t, r, s); }
Cylinder::Cylinder() { m_desc = blank_desc(RT_CYLINDER); set_trs(m_desc, Float3{ 0, 0, 0 }, EulerAngles{ 0, 0, 0 }, Float3{ 1, 1, 1 }); }
Cylinder::Cylinder(const Float3& t, const EulerAngles& r, const Float3& s) { m_desc = blank_desc(RT_CYLINDER); set_trs(m_desc, t, r, s); }

Sphere::Sphere() : Sphere(Float3{ 0, 0, 0 }, Float3{ 1, 1, 1 }) {}
Sphere::Sphere(const Float3& c, const Float3& s) {
    m_desc = blank_desc(RT_SPHERE);
    m_desc.q[0] = c.x; m_desc.q[1] = c.y; m_desc.q[2] = c.z; m_desc.q[3] = s.x; m_desc.q[4] = s.y; m_desc.q[5] = s.z;
}

Plane::Plane() { m_desc = blank_desc(RT_PLANE); m_desc.reserved = -1; }
Plane::Plane(const std::vector<Float3>& corners) {
    m_desc = blank_desc(RT_PLANE);
    m_n_corners = static_cast<int>(std::min<size_t>(corners.size(), 4));
    for (int k = 0; k < m_n_corners; ++k) { m_desc.q[3*k] = corners[k].x; m_desc.q[3*k + 1] = corners[k].y; m_desc.q[3*k + 2] = corners[k].z; }
    if (corners.size() < 4) m_desc.reserved = corners.empty() ? -1 : static_cast<int32_t>(corners.size());   // never hit (mesh.cpp:796)
}

// mesh.cpp:922-930: min / max over the stored corners; no corners leaves the outputs untouched at zero
void Plane::compute_aabb(Float3& outMin, Float3& outMax) const {
    if (m_n_corners == 0) { outMin = Float3{ 0, 0, 0 }; outMax = Float3{ 0, 0, 0 }; return; }
    outMin = outMax = Float3{ m_desc.q[0], m_desc.q[1], m_desc.q[2] };
    for (int k = 1; k < m_n_corners; ++k) {
        outMin.x = std::min(outMin.x, m_desc.q[3*k]); outMin.y = std::min(outMin.y, m_desc.q[3*k + 1]); outMin.z = std::min(outMin.z, m_desc.q[3*k + 2]);
        outMax.x = std::max(outMax.x, m_desc.q[3*k]); outMax.y = std::max(outMax.y, m_desc.q[3*k + 1]); outMax.z = std::max(outMax.z, m_desc.q[3*k + 2]);
    }
}

Triangle::Triangle() : Triangle(Float3{ 0, 0, 0 }, Float3{ 1, 0, 0 }, Float3{ 0, 1, 0 }) {}
Triangle::Triangle(const Float3& a, const Float3& b, const Float3& c) {
    m_desc = blank_desc(RT_TRIANGLE);
    const Float3 v[3] = { a, b, c };
    for (int k = 0; k < 3; ++k) { m_desc.q[3*k] = v[k].x; m_desc.q[3*k + 1] = v[k].y; m_desc.q[3*k + 2] = v[k].z; }
}

void Cube::write_to_console(std::ostream& out) const {
    const double* q = m_desc.q;
    out << "Cube(translation=[" << q[0] << ", " << q[1] << ", " << q[2] << "], rotation(rpy)=[" << q[3] << ", " << q[4] << ", " << q[5]
        << "], scale=[" << q[6] << ", " << q[7] << ", " << q[8] << "])\n";
}
void Cylinder::write_to_console(std::ostream& out) const {
    const double* q = m_desc.q;
    out << "Cylinder(translation=[" << q[0] << ", " << q[1] << ", " << q[2] << "], rotation(rpy)=[" << q[3] << ", " << q[4] << ", " << q[5]
        << "], scale=[" << q[6] << ", " << q[7] << ", " << q[8] << "])\n";
}
void Sphere::write_to_console(std::ostream& out) const {
    const double* q = m_desc.q;
    out << "Sphere(location=[" << q[0] << ", " << q[1] << ", " << q[2] << "], scale=[" << q[3] << ", " << q[4] << ", " << q[5] << "])\n";
}
void Plane::write_to_console(std::ostream& out) const {
    out << "Plane(corners=[";
    for (int k = 0; k < m_n_corners; ++k) out << (k ? ", " : "") << "[" << m_desc.q[3*k] << ", " << m_desc.q[3*k + 1] << ", " << m_desc.q[3*k + 2] << "]";
    out << "])\n";
}
void Triangle::write_to_console(std::ostream& out) const {
    out << "Triangle(corners=[";
    for (int k = 0; k < 3; ++k) out << (k ? ", " : "") << "[" << m_desc.q[3*k] << ", " << m_desc.q[3*k + 1] << ", " << m_desc.q[3*k + 2] << "]";
    out << "])\n";
}

// The per-class readers take the text between the braces of "Cube": { ... } (mesh.cpp:52, 316, 544, 707).  The block
// is wrapped back into a document and handed to the library's scene parser, which applies the reference's defaults.
template <class T, class Make>
static std::vector<T> read_class(const std::string& class_block, const char* class_name, int kind, Make make) {
    const std::string doc = std::string("{\"MESH\":{\"") + class_name + "\":{" + class_block + "}}}";
    rt_scene* s = nullptr;
    const int rc = rt_scene_load_json_text(doc.data(), doc.size(), nullptr, &s);
    if (rc == RT_ERR_PARSE) throw std::invalid_argument("stod");
    std::vector<T> out;
    if (rc != RT_OK) return out;
    const rt_prim_desc* prims = nullptr; uint64_t n = 0;
    rt_scene_get_prims(s, &prims, &n);
    for (uint64_t i = 0; i < n; ++i) {
        if (prims[i].kind != kind) continue;
        T m = make(prims[i]);
        m.setMaterial(prims[i].alpha, prims[i].metallic, prims[i].roughness);
        m.setIndexOfRefraction(prims[i].ior);
        m.setAlbedo(prims[i].albedo[0], prims[i].albedo[1], prims[i].albedo[2]);
        out.emplace_back(std::move(m));
    }
    rt_scene_destroy(s);
    return out;
}

static Cube make_cube(const rt_prim_desc& d) { return Cube(Float3{ d.q[0], d.q[1], d.q[2] }, EulerAngles{ d.q[3], d.q[4], d.q[5] }, Float3{ d.q[6], d.q[7], d.q[8] }); }
static Cylinder make_cylinder(const rt_prim_desc& d) { return Cylinder(Float3{ d.q[0], d.q[1], d.q[2] }, EulerAngles{ d.q[3], d.q[4], d.q[5] }, Float3{ d.q[6], d.q[7], d.q[8] }); }
static Sphere make_sphere(const rt_prim_desc& d) { return Sphere(Float3{ d.q[0], d.q[1], d.q[2] }, Float3{ d.q[3], d.q[4], d.q[5] }); }
static Plane make_plane(const rt_prim_desc& d) {
    std::vector<Float3> c;
    // a plane read with fewer than four corners keeps reserved != 0; its missing corners are zero and never used
    for (int k = 0; k < 4; ++k) c.push_back(Float3{ d.q[3*k], d.q[3*k + 1], d.q[3*k + 2] });
    if (d.reserved) c.resize(d.reserved > 0 && d.reserved < 4 ? d.reserved : 0);
    return Plane(c);
}
static Triangle make_triangle(const rt_prim_desc& d) {
    return Triangle(Float3{ d.q[0], d.q[1], d.q[2] }, Float3{ d.q[3], d.q[4], d.q[5] }, Float3{ d.q[6], d.q[7], d.q[8] });
}

std::vector<Cube> Cube::read_from_json(const std::string& b) { return read_class<Cube>(b, "Cube", RT_CUBE, make_cube); }
std::vector<Cylinder> Cylinder::read_from_json(const std::string& b) { return read_class<Cylinder>(b, "Cylinder", RT_CYLINDER, make_cylinder); }
std::vector<Sphere> Sphere::read_from_json(const std::string& b) { return read_class<Sphere>(b, "Sphere", RT_SPHERE, make_sphere); }
std::vector<Plane> Plane::read_from_json(const std::string& b) { return read_class<Plane>(b, "Plane", RT_PLANE, make_plane); }

// ---- BoundingBox ---------------------------------------------------------------------------------------------------------

BoundingBox::BoundingBox() : m_min{ 0, 0, 0 }, m_max{ 0, 0, 0 } {}
BoundingBox::BoundingBox(const Float3& bmin, const Float3& bmax) : m_min(bmin), m_max(bmax) {}

BoundingBox::~BoundingBox() {
    // the device copy of a tree lives exactly as long as its root node
    std::lock_guard<std::mutex> lock(rt_host::g_tree_mu);
    rt_host::trees().erase(this);
}

void BoundingBox::write_to_console(std::ostream& out) const {
    out << "BoundingBox(min=[" << m_min.x << ", " << m_min.y << ", " << m_min.z << "], max=[" << m_max.x << ", " << m_max.y << ", " << m_max.z << "])\n";
}

// mesh.cpp:942-967.  A box is not a primitive of any scene; this host-side slab test exists for callers that walk the
// returned tree themselves.  Sets only the distance: entry if the origin is outside, exit otherwise.
bool BoundingBox::intersect(const Ray& ray, Hit& hit) const {
    const double o[3] = { ray.getPosition().x, ray.getPosition().y, ray.getPosition().z };
    const double d[3] = { ray.getDirection().x, ray.getDirection().y, ray.getDirection().z };
    const double lo[3] = { m_min.x, m_min.y, m_min.z }, hi[3] = { m_max.x, m_max.y, m_max.z };
    double t_in = -std::numeric_limits<double>::infinity(), t_out = std::numeric_limits<double>::infinity();
    for (int k = 0; k < 3; ++k) {
        if (std::abs(d[k]) < 1e-12) { if (o[k] < lo[k] || o[k] > hi[k]) return false; continue; }
        double a = (lo[k] - o[k]) / d[k], b = (hi[k] - o[k]) / d[k];
        if (a > b) std::swap(a, b);
        if (a > t_in) t_in = a;
        if (b < t_out) t_out = b;
        if (!(t_in <= t_out)) return false;
    }
    if (t_out < 0.0) return false;
    hit.setDistanceAlongRay(t_in >= 0.0 ? t_in : t_out);
    return true;
}

// ---- RayTracer -------------------------------------------------------------------------------------------------------------

RayTracer::RayTracer() = default;
RayTracer::~RayTracer() = default;

// raytracer.cpp:20-27: cubes, planes, cylinders, spheres; triangles (extension) last
void RayTracer::rebuild_mesh_ptrs() {
    m_scene_mesh_ptrs.clear();
    for (const Cube& m : m_cubes) m_scene_mesh_ptrs.push_back(&m);
    for (const Plane& m : m_planes) m_scene_mesh_ptrs.push_back(&m);
    for (const Cylinder& m : m_cylinders) m_scene_mesh_ptrs.push_back(&m);
    for (const Sphere& m : m_spheres) m_scene_mesh_ptrs.push_back(&m);
    for (const Triangle& m : m_triangles) m_scene_mesh_ptrs.push_back(&m);
}

template <class T> static int index_in(const std::vector<T>& v, const Mesh* m, int base) {
    if (v.empty()) return -1;
    const uintptr_t first = reinterpret_cast<uintptr_t>(static_cast<const Mesh*>(v.data()));
    const uintptr_t p = reinterpret_cast<uintptr_t>(m);
    if (p < first || p >= first + v.size() * sizeof(T) || (p - first) % sizeof(T) != 0) return -1;
    return base + static_cast<int>((p - first) / sizeof(T));
}

int RayTracer::index_of(const Mesh* m) const {
    int base = 0, r;
    if ((r = index_in(m_cubes, m, base)) >= 0) return r;
    base += static_cast<int>(m_cubes.size());
    if ((r = index_in(m_planes, m, base)) >= 0) return r;
    base += static_cast<int>(m_planes.size());
    if ((r = index_in(m_cylinders, m, base)) >= 0) return r;
    base += static_cast<int>(m_cylinders.size());
    if ((r = index_in(m_spheres, m, base)) >= 0) return r;
    base += static_cast<int>(m_spheres.size());
    return index_in(m_triangles, m, base);
}

template <class T> static void carry_material(T& m, const rt_prim_desc& d) {
    m.setMaterial(d.alpha, d.metallic, d.roughness);
    m.setIndexOfRefraction(d.ior);
    m.setAlbedo(d.albedo[0], d.albedo[1], d.albedo[2]);
}

bool RayTracer::load_from_json(const std::string& json_filepath) {
    rt_scene* s = nullptr;
    const int rc = rt_scene_load_json(json_filepath.c_str(), nullptr, &s);
    if (rc == RT_ERR_IO) return false;                                   // raytracer.cpp:109
    if (rc == RT_ERR_PARSE) throw std::invalid_argument("stod");        // std::stod's exception escapes the reference's loader
    if (rc != RT_OK) rt_host::gpu_fail("load_from_json", rc);

    m_cubes.clear(); m_planes.clear(); m_cylinders.clear(); m_spheres.clear(); m_triangles.clear();
    m_cameras.clear(); m_point_lights.clear(); m_light_order.clear(); m_scene_mesh_ptrs.clear();
    m_gpu.reset(new GpuScene());
    m_gpu->scene = s;                   // already flattened and ordered: upload happens on first use

    const rt_prim_desc* prims = nullptr; uint64_t n = 0;
    rt_scene_get_prims(s, &prims, &n);
    for (uint64_t i = 0; i < n; ++i) {
        const rt_prim_desc& d = prims[i];
        switch (d.kind) {
            case RT_CUBE: m_cubes.emplace_back(make_cube(d)); carry_material(m_cubes.back(), d); break;
            case RT_PLANE: m_planes.emplace_back(make_plane(d)); carry_material(m_planes.back(), d); break;
            case RT_CYLINDER: m_cylinders.emplace_back(make_cylinder(d)); carry_material(m_cylinders.back(), d); break;
            case RT_SPHERE: m_spheres.emplace_back(make_sphere(d)); carry_material(m_spheres.back(), d); break;
            default: break;
        }
    }
    rebuild_mesh_ptrs();
    for (size_t i = 0; i < m_scene_mesh_ptrs.size(); ++i) {
        rt_texture t;
        if (rt_scene_get_texture(s, i, &t) == RT_OK && t.width > 0) {
            std::unique_ptr<Image> img(new Image(t.width, t.height));
            img->assign_rgb(t.rgb);
            const_cast<Mesh*>(m_scene_mesh_ptrs[i])->setAlbedoTexture(std::move(img));
        }
    }
    rt_scene_info info;
    rt_scene_get_info(s, &info);
    for (uint32_t k = 0; k < info.n_cameras; ++k) {
        rt_camera c; char id[64] = { 0 };
        if (rt_scene_camera(s, nullptr, static_cast<int32_t>(k), &c, id) == RT_OK) m_cameras[id] = Camera(c);
    }
    const rt_light* lights = nullptr; uint32_t nl = 0;
    rt_scene_get_lights(s, &lights, &nl);
    for (uint32_t k = 0; k < nl; ++k) {
        PointLight L{ RayVec3{ lights[k].pos[0], lights[k].pos[1], lights[k].pos[2] }, lights[k].intensity };
        m_light_order.push_back(L);
        m_point_lights[std::to_string(k)] = L;
    }
    rebuild_mesh_ptrs();
    return true;                                                         // even when nothing was parsed (raytracer.cpp:120)
}

void RayTracer::set_scene(std::vector<Cube> cubes, std::vector<Plane> planes, std::vector<Cylinder> cylinders, std::vector<Sphere> spheres,
                          std::vector<Triangle> triangles, const std::vector<PointLight>& lights) {
    m_gpu.reset();
    m_cubes = std::move(cubes); m_planes = std::move(planes); m_cylinders = std::move(cylinders);
    m_spheres = std::move(spheres); m_triangles = std::move(triangles);
    m_point_lights.clear(); m_light_order = lights;
    for (size_t k = 0; k < lights.size(); ++k) m_point_lights[std::to_string(k)] = lights[k];
    rebuild_mesh_ptrs();
}

GpuScene& RayTracer::gpu() const {
    if (!m_gpu) m_gpu.reset(new GpuScene());
    if (!m_gpu->scene) {
        std::vector<rt_prim_desc> descs;
        descs.reserve(m_scene_mesh_ptrs.size());
        for (const Mesh* m : m_scene_mesh_ptrs) descs.push_back(m->describe());
        std::vector<rt_light> lights;
        for (const PointLight& L : m_light_order) lights.push_back(rt_light{ { L.location.x, L.location.y, L.location.z }, L.radiant_intensity });
        if (int rc = rt_scene_create(descs.data(), descs.size(), lights.data(), static_cast<uint32_t>(lights.size()), nullptr, &m_gpu->scene))
            rt_host::gpu_fail("scene build", rc);
        // textures the caller attached to its meshes (Mesh::setAlbedoTexture)
        std::vector<rt_texture> tex; std::vector<std::vector<unsigned char>> pixels; std::vector<int32_t> prim_tex(descs.size(), -1);
        for (size_t i = 0; i < m_scene_mesh_ptrs.size(); ++i) {
            const Image* img = m_scene_mesh_ptrs[i]->albedoTexture();
            if (!img || img->width() <= 0 || img->height() <= 0) continue;
            pixels.emplace_back();
            for (const Pixel& p : img->pixels()) { pixels.back().push_back(p.r); pixels.back().push_back(p.g); pixels.back().push_back(p.b); }
            prim_tex[i] = static_cast<int32_t>(tex.size());
            tex.push_back(rt_texture{ img->width(), img->height(), nullptr });
        }
        for (size_t k = 0; k < tex.size(); ++k) tex[k].rgb = pixels[k].data();
        if (!tex.empty())
            if (int rc = rt_scene_set_textures(m_gpu->scene, tex.data(), static_cast<uint32_t>(tex.size()), prim_tex.data()))
                rt_host::gpu_fail("scene textures", rc);
    }
    m_gpu->upload();
    return *m_gpu;
}

// raytracer.cpp:123-141
std::vector<Ray> RayTracer::get_one_ray_per_pixel(const std::string& camera_id) const {
    std::vector<Ray> rays;
    auto it = m_cameras.find(camera_id);
    if (it == m_cameras.end()) return rays;
    int w = 0, h = 0; it->second.getImageResolution(w, h);
    if (w <= 0 || h <= 0) return rays;
    rays.reserve(static_cast<size_t>(w) * static_cast<size_t>(h));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double o[3], d[3];
            rt_camera_pixel_to_ray(&it->second.raw(), x + 0.5, y + 0.5, o, d);
            rays.emplace_back(RayVec3{ o[0], o[1], o[2] }, RayVec3{ d[0], d[1], d[2] });
        }
    return rays;
}

std::vector<Hit> RayTracer::one_pass_intersection(const std::vector<Ray>& rays) const {
    std::vector<Hit> out(rays.size());
    if (rays.empty() || m_scene_mesh_ptrs.empty()) return out;
    GpuScene& g = gpu();
    std::vector<double> packed, rec(rays.size() * 16);
    rt_host::pack_rays(rays, packed);
    // the reference's one_pass_intersection is the loop over all primitives (raytracer.cpp:143-157); so is this one while
    // the scene is small enough for that to be cheap - beyond, the hierarchy gives the same hits
    rt_host::BruteForce mode(g.ctx, m_scene_mesh_ptrs.size() <= rt_host::kBruteForceMaxPrims);
    if (int rc = rt_trace_rays_full_host(g.ctx, packed.data(), rays.size(), rec.data())) rt_host::gpu_fail("one_pass_intersection", rc);
    for (size_t i = 0; i < rays.size(); ++i) {
        const double* r = &rec[i * 16];
        if (r[0] != 0.0) out[i] = Hit::from_record(r, m_scene_mesh_ptrs[static_cast<size_t>(r[1])]);
    }
    return out;
}

bool RayTracer::one_pass_intersection(const Ray& ray, Hit& out_hit) const {
    if (m_scene_mesh_ptrs.empty()) return false;
    std::vector<Hit> h = one_pass_intersection(std::vector<Ray>{ ray });
    if (!h[0].hasDistanceAlongRay()) return false;
    out_hit = h[0];
    return true;
}

std::vector<Pixel> RayTracer::shade(const std::vector<Hit>& hits, const std::vector<Ray>& view_rays) const {
    if (hits.size() != view_rays.size()) throw std::invalid_argument("shade: one view ray per hit");
    std::vector<Pixel> out(hits.size(), Pixel{ 0, 0, 0 });
    if (hits.empty() || m_light_order.empty()) return out;               // raytracer.cpp:197-202
    std::vector<double> h8(hits.size() * 10, 0.0), v3(hits.size() * 3);
    for (size_t i = 0; i < hits.size(); ++i) {
        const Hit& h = hits[i];
        double* o = &h8[i * 10];
        o[6] = -1.0;                                                     // black unless the hit carries surface data (raytracer.cpp:161)
        if (h.hasDistanceAlongRay() && h.hasSurfaceNormal() && h.hasIntersectionPoint()) {
            const HitVec3& p = h.getIntersectionPoint(); const HitVec3& n = h.getSurfaceNormal();
            o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = n.x; o[4] = n.y; o[5] = n.z;
            const int id = h.hasMesh() ? index_of(h.getMesh()) : -1;
            o[6] = id >= 0 ? static_cast<double>(id) : -2.0;             // no (known) mesh: default material, white
            if (h.hasUV()) { o[7] = h.getU(); o[8] = h.getV(); o[9] = 1.0; }
        }
        const RayVec3& c = view_rays[i].getPosition();
        v3[i*3] = c.x; v3[i*3 + 1] = c.y; v3[i*3 + 2] = c.z;
    }
    GpuScene& g = gpu();
    std::vector<unsigned char> rgb(hits.size() * 3);
    if (int rc = rt_shade_hits_host(g.ctx, h8.data(), v3.data(), hits.size(), rgb.data())) rt_host::gpu_fail("shade", rc);
    for (size_t i = 0; i < hits.size(); ++i) out[i] = Pixel{ rgb[3*i], rgb[3*i + 1], rgb[3*i + 2] };
    return out;
}

Pixel RayTracer::shade(const Hit& hit, const Ray& view_ray) const {
    return shade(std::vector<Hit>{ hit }, std::vector<Ray>{ view_ray })[0];
}

bool RayTracer::trace_primary(const std::string& camera_id, std::vector<int>& hit_index, std::vector<double>& distance, bool brute_force) const {
    auto it = m_cameras.find(camera_id);
    if (it == m_cameras.end()) return false;
    const rt_camera& c = it->second.raw();
    if (c.width <= 0 || c.height <= 0) return false;
    const size_t n = static_cast<size_t>(c.width) * static_cast<size_t>(c.height);
    hit_index.assign(n, -1); distance.assign(n, std::numeric_limits<double>::infinity());
    if (m_scene_mesh_ptrs.empty()) return true;
    GpuScene& g = gpu();
    static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bits");
    rt_host::BruteForce mode(g.ctx, brute_force);
    if (int rc = rt_trace_primary_host(g.ctx, &c, hit_index.data(), distance.data())) rt_host::gpu_fail("trace_primary", rc);
    return true;
}

bool RayTracer::render_to_image(const std::string& camera_id, Image& out, int samples_per_pixel, int max_scatters, double min_scatter_intensity) const {
    auto it = m_cameras.find(camera_id);
    if (it == m_cameras.end()) return false;
    const rt_camera& c = it->second.raw();
    if (c.width <= 0 || c.height <= 0) return false;
    if (samples_per_pixel < 1) samples_per_pixel = 1;
    out = Image(c.width, c.height);
    out.setMaxValue(255);
    // no lights (raytracer.cpp:387-395) or nothing to hit: black frame, no GPU work needed
    if (m_light_order.empty() || m_scene_mesh_ptrs.empty()) return true;
    GpuScene& g = gpu();
    std::vector<unsigned char> rgb(static_cast<size_t>(c.width) * static_cast<size_t>(c.height) * 3);
    const int n_gpus = samples_per_pixel == 1 ? rt_host::gpu_count() : 1;
    if (n_gpus > 1) {
        g.spread(n_gpus);
        if (int rc = rt_render_frame_multi_host(g.comms.data(), n_gpus, &c, 64, max_scatters < 0 ? 0 : max_scatters, min_scatter_intensity, rgb.data()))
            rt_host::gpu_fail("render on several GPUs", rc);
    } else if (int rc = rt_render_samples_host(g.ctx, &c, samples_per_pixel, m_sample_seed, max_scatters, min_scatter_intensity, rgb.data()))
        rt_host::gpu_fail("render", rc);
    out.assign_rgb(rgb.data());
    return true;
}

// raytracer.cpp:373-442
bool RayTracer::render_unaccelerated_ppm(const std::string& camera_id, const std::string& output_filepath, int samples_per_pixel,
                                         int max_scatters, double min_scatter_intensity) const {
    Image img(0, 0);
    if (!render_to_image(camera_id, img, samples_per_pixel, max_scatters, min_scatter_intensity)) return false;
    img.write(output_filepath);
    return true;
}

// ---- free functions ----------------------------------------------------------------------------------------------------------

// raytracer.cpp:553-585.  Both time one closest-hit pass over all primary rays with the scene already resident on the
// device, as the reference times only its intersection loop: the first with every primitive tested against every ray
// (raytracer.cpp:553-565), the second through the hierarchy (raytracer.cpp:567-585).
static double time_one_pass(const std::string& json_filepath, const std::string& camera_id, bool brute_force) {
    RayTracer rt;
    if (!rt.load_from_json(json_filepath)) return 0.0;
    std::vector<int> ids; std::vector<double> dist;
    if (!rt.trace_primary(camera_id, ids, dist, brute_force)) return 0.0;          // first call: upload, allocation, warm-up
    const auto t0 = std::chrono::high_resolution_clock::now();
    rt.trace_primary(camera_id, ids, dist, brute_force);
    const auto t1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

double time_trace_all_one_pass(const std::string& json_filepath, const std::string& camera_id) { return time_one_pass(json_filepath, camera_id, true); }
double time_trace_all_accelerated_one_pass(const std::string& json_filepath, const std::string& camera_id) { return time_one_pass(json_filepath, camera_id, false); }

// raytracer.cpp:454-499 re-designed: the hierarchy is the library's flattened SAH tree; node_store receives one
// BoundingBox per internal node (N - 1 of them for N bounded objects with one object per leaf).
const Mesh* get_BVH_tree(const std::vector<const Mesh*>& objects, std::vector<std::unique_ptr<BoundingBox>>& node_store) {
    if (objects.empty()) return nullptr;                              // raytracer.cpp:455
    if (objects.size() == 1) return objects[0];                       // raytracer.cpp:456
    std::vector<rt_prim_desc> descs;
    descs.reserve(objects.size());
    for (const Mesh* m : objects) {
        if (m->describe().kind < 0) throw std::invalid_argument("get_BVH_tree: objects must be primitives");
        descs.push_back(m->describe());
    }
    std::unique_ptr<GpuScene> g(new GpuScene());
    rt_build_opts opts; std::memset(&opts, 0, sizeof(opts));
    opts.max_leaf = 1;
    if (int rc = rt_scene_create(descs.data(), descs.size(), nullptr, 0, &opts, &g->scene)) rt_host::gpu_fail("get_BVH_tree", rc);
    const rt_node32* nodes = nullptr; const uint32_t* slot_prim = nullptr; uint32_t n_nodes = 0; uint64_t n_slots = 0;
    rt_scene_get_bvh(g->scene, &nodes, &n_nodes, &slot_prim, &n_slots);
    rt_scene_info info; rt_scene_get_info(g->scene, &info);

    // mirror the flattened nodes: node k -> a BoundingBox (internal) or the object in its leaf
    const size_t store_base = node_store.size();
    std::vector<const Mesh*> as_mesh(n_nodes, nullptr);
    std::vector<size_t> box_of(n_nodes, static_cast<size_t>(-1));
    const Mesh* root = nullptr;
    if (info.n_slots > 0) {
        for (uint32_t k = 0; k < n_nodes; ++k) {
            if (k == 1) continue;                                     // unused slot next to the root
            const rt_node32& nd = nodes[k];
            if (nd.count > 0) { as_mesh[k] = objects[slot_prim[nd.first]]; continue; }
            node_store.emplace_back(new BoundingBox(Float3{ nd.lo[0], nd.lo[1], nd.lo[2] }, Float3{ nd.hi[0], nd.hi[1], nd.hi[2] }));
            box_of[k] = node_store.size() - 1;
            as_mesh[k] = node_store.back().get();
        }
        for (uint32_t k = 0; k < n_nodes; ++k)
            if (box_of[k] != static_cast<size_t>(-1)) node_store[box_of[k]]->setChildren(as_mesh[nodes[k].first], as_mesh[nodes[k].first + 1]);
        root = as_mesh[0];
    }
    // objects without finite bounds hang off unbounded boxes above the root
    const double inf = std::numeric_limits<double>::infinity();
    for (uint64_t s = info.n_slots; s < info.n_slots + info.n_loose; ++s) {
        const Mesh* loose = objects[slot_prim[s]];
        if (!root) { root = loose; continue; }
        node_store.emplace_back(new BoundingBox(Float3{ -inf, -inf, -inf }, Float3{ inf, inf, inf }));
        node_store.back()->setChildren(root, loose);
        root = node_store.back().get();
    }
    if (!root || !root->asBoundingBox()) {
        // everything degenerate, or a single usable object: keep a box on top so that the device copy has an owner
        node_store.emplace_back(new BoundingBox(Float3{ -inf, -inf, -inf }, Float3{ inf, inf, inf }));
        node_store.back()->setChildren(root, nullptr);
        root = node_store.back().get();
    }
    (void)store_base;
    std::lock_guard<std::mutex> lock(rt_host::g_tree_mu);
    rt_host::TreeEntry& e = rt_host::trees()[root->asBoundingBox()];
    e.gpu = std::move(g);
    e.objects = objects;
    return root;
}

std::vector<Hit> accelerated_one_pass_intersection(const Mesh* bvh_root, const std::vector<Ray>& rays) {
    std::vector<Hit> out(rays.size());
    if (!bvh_root || rays.empty()) return out;
    const BoundingBox* box = bvh_root->asBoundingBox();
    if (!box) {                                                       // a lone primitive is its own tree
        out = bvh_root->intersect(rays);
        for (Hit& h : out) if (h.hasDistanceAlongRay()) h.setMesh(bvh_root);
        return out;
    }
    GpuScene* g = nullptr; const std::vector<const Mesh*>* objects = nullptr;
    {
        std::lock_guard<std::mutex> lock(rt_host::g_tree_mu);
        auto it = rt_host::trees().find(box);
        if (it == rt_host::trees().end()) throw std::invalid_argument("accelerated_one_pass_intersection: not a root returned by get_BVH_tree");
        g = it->second.gpu.get(); objects = &it->second.objects;
    }
    g->upload();
    std::vector<double> packed, rec(rays.size() * 16);
    rt_host::pack_rays(rays, packed);
    if (int rc = rt_trace_rays_full_host(g->ctx, packed.data(), rays.size(), rec.data())) rt_host::gpu_fail("accelerated_one_pass_intersection", rc);
    for (size_t i = 0; i < rays.size(); ++i) {
        const double* r = &rec[i * 16];
        if (r[0] != 0.0) out[i] = Hit::from_record(r, (*objects)[static_cast<size_t>(r[1])]);
    }
    return out;
}

bool accelerated_one_pass_intersection(const Mesh* bvh_root, const Ray& ray, Hit& out_hit) {
    if (!bvh_root) return false;
    std::vector<Hit> h = accelerated_one_pass_intersection(bvh_root, std::vector<Ray>{ ray });
    if (!h[0].hasDistanceAlongRay()) return false;
    out_hit = h[0];
    return true;
}
