// Scene-file reader for the Blender-export format (producer: s2899221/Blend/Export.py; consumers:
// RayTracer::load_from_json s2899221/Code/raytracer.cpp:29-121 and the per-type read_from_json in
// s2899221/Code/mesh.cpp:52, 316, 544, 707).
//
// The reference scans the text with regular expressions and brace matching and copies the tail of the
// class block once per object (quadratic).  This reader tokenises the file once into a small ordered
// tree and then applies the same lookup rules on that tree:
//   * a "block" lookup takes the first occurrence of the key in document order, at any depth, and only
//     succeeds when its value is an object or array (extract_object_block, utils.cpp:11-51);
//   * a "number" lookup takes the first occurrence, at any depth, whose value is a bare number
//     (parse_number, utils.cpp:53-56);
//   * objects of a class are the members whose key is all digits and whose value is an object;
//   * defaults: translation/location 0, rotation 0, scale 1, material {1, 0, 0.5, 1.5, albedo 1},
//     camera focal 35, sensor 36x24, film 1920x1080, light intensity 1.
// Textures (albedo_texture_ppm) are not loaded: solid albedo only.
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "rt_scene.h"

namespace {

struct Node;
using Member = std::pair<std::string, std::unique_ptr<Node>>;
struct Node {
    enum Type { Null, Bool, Number, String, Object, Array } type = Null;
    double number = 0.0;
    bool unsigned_integer_text = false;        // token was digits only
    std::string text;                          // string value
    std::string raw;                           // string value exactly as written between the quotes
    std::vector<Member> members;               // object, in document order
    std::vector<std::unique_ptr<Node>> items;  // array
};

struct Parser {
    const char* p; const char* end; std::string err; int depth = 0;
    void ws() { while (p < end && std::isspace(static_cast<unsigned char>(*p))) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }

    bool string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p; out.clear();
        const char* first = p;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) { out.push_back(p[1]); p += 2; }   // escapes are kept verbatim minus the backslash
            else out.push_back(*p++);
        }
        if (p >= end) return fail("unterminated string");
        last_raw.assign(first, p);
        ++p; return true;
    }
    std::string last_raw;

    bool value(Node& n) {
        if (++depth > 200) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of file");
        bool ok = true;
        const char c = *p;
        if (c == '{') {
            n.type = Node::Object; ++p; ws();
            if (p < end && *p == '}') { ++p; }
            else for (;;) {
                ws(); std::string key;
                if (!string(key)) { ok = false; break; }
                ws(); if (p >= end || *p != ':') { ok = fail("expected ':'"); break; }
                ++p;
                std::unique_ptr<Node> child(new Node());
                if (!value(*child)) { ok = false; break; }
                n.members.emplace_back(std::move(key), std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = fail("expected ',' or '}'"); break;
            }
        } else if (c == '[') {
            n.type = Node::Array; ++p; ws();
            if (p < end && *p == ']') { ++p; }
            else for (;;) {
                std::unique_ptr<Node> child(new Node());
                if (!value(*child)) { ok = false; break; }
                n.items.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = fail("expected ',' or ']'"); break;
            }
        } else if (c == '"') {
            n.type = Node::String; ok = string(n.text); n.raw = last_raw;
        } else if (std::isdigit(static_cast<unsigned char>(c)) || c == '-' || c == '+' || c == '.') {
            // the reference accepts any run of [-+eE0-9.] and hands it to std::stod
            const char* s = p; bool digits_only = true;
            while (p < end && (std::isdigit(static_cast<unsigned char>(*p)) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) {
                if (!std::isdigit(static_cast<unsigned char>(*p))) digits_only = false;
                ++p;
            }
            std::string tok(s, p);
            errno = 0; char* e = nullptr;
            const double v = std::strtod(tok.c_str(), &e);
            if (e == tok.c_str()) ok = fail("malformed number");
            n.type = Node::Number; n.number = v; n.unsigned_integer_text = digits_only;
        } else if (end - p >= 4 && !std::memcmp(p, "true", 4))  { n.type = Node::Bool; n.number = 1; p += 4; }
        else if (end - p >= 5 && !std::memcmp(p, "false", 5))   { n.type = Node::Bool; p += 5; }
        else if (end - p >= 4 && !std::memcmp(p, "null", 4))    { n.type = Node::Null; p += 4; }
        else ok = fail("unexpected character");
        --depth;
        return ok;
    }
};

// first member named `key` in document order, any depth
const Node* find_key(const Node& n, const char* key) {
    if (n.type == Node::Object) {
        for (const Member& m : n.members) {
            if (m.first == key) return m.second.get();
            if (const Node* r = find_key(*m.second, key)) return r;
        }
    } else if (n.type == Node::Array) {
        for (const auto& it : n.items) if (const Node* r = find_key(*it, key)) return r;
    }
    return nullptr;
}

const Node* find_block(const Node& n, const char* key) {
    const Node* r = find_key(n, key);
    return (r && (r->type == Node::Object || r->type == Node::Array)) ? r : nullptr;
}

// first member named `key` whose value is a number, document order, any depth
const Node* find_number_node(const Node& n, const char* key) {
    if (n.type == Node::Object) {
        for (const Member& m : n.members) {
            if (m.first == key && m.second->type == Node::Number) return m.second.get();
            if (const Node* r = find_number_node(*m.second, key)) return r;
        }
    } else if (n.type == Node::Array) {
        for (const auto& it : n.items) if (const Node* r = find_number_node(*it, key)) return r;
    }
    return nullptr;
}

bool get_number(const Node& n, const char* key, double& out) {
    const Node* r = find_number_node(n, key);
    if (!r) return false;
    out = r->number; return true;
}

bool get_vec3(const Node& n, const char* key, double v[3]) {
    const Node* b = find_block(n, key);
    if (!b) return false;
    double x, y, z;
    if (!get_number(*b, "x", x) || !get_number(*b, "y", y) || !get_number(*b, "z", z)) return false;
    v[0] = x; v[1] = y; v[2] = z; return true;
}

// parse_vec2i (utils.cpp:64-69): "x": <digits> ... "y": <digits>, x textually before y.
bool get_vec2i(const Node& n, const char* key, int& ox, int& oy) {
    const Node* b = find_block(n, key);
    if (!b || b->type != Node::Object) return false;
    const Node* x = nullptr;
    for (const Member& m : b->members) {
        if (!x) { if (m.first == "x" && m.second->type == Node::Number && m.second->unsigned_integer_text) x = m.second.get(); }
        else if (m.first == "y" && m.second->type == Node::Number && m.second->unsigned_integer_text) {
            ox = static_cast<int>(x->number); oy = static_cast<int>(m.second->number); return true;
        }
    }
    return false;
}

bool all_digits(const std::string& s) {
    if (s.empty()) return false;
    for (char c : s) if (!std::isdigit(static_cast<unsigned char>(c))) return false;
    return true;
}

void default_material(rt_prim_desc& d) {
    d.alpha = 1.0; d.metallic = 0.0; d.roughness = 0.5; d.ior = 1.5;
    d.albedo[0] = d.albedo[1] = d.albedo[2] = 1.0;
}

// first member named `key` whose value is a string, document order, any depth
const Node* find_string_node(const Node& n, const char* key) {
    if (n.type == Node::Object) {
        for (const Member& m : n.members) {
            if (m.first == key && m.second->type == Node::String) return m.second.get();
            if (const Node* r = find_string_node(*m.second, key)) return r;
        }
    } else if (n.type == Node::Array) {
        for (const auto& it : n.items) if (const Node* r = find_string_node(*it, key)) return r;
    }
    return nullptr;
}

// The primitive being read leaves its texture path here (mesh.cpp:91-97: the text between the quotes, not unescaped).
thread_local std::string g_texture_path;

void read_material(const Node& obj, rt_prim_desc& d) {
    default_material(d);
    g_texture_path.clear();
    const Node* m = find_block(obj, "material");
    if (!m) return;
    if (const Node* t = find_string_node(*m, "albedo_texture_ppm")) g_texture_path = t->raw;
    get_number(*m, "alpha", d.alpha);
    get_number(*m, "metallic", d.metallic);
    get_number(*m, "roughness", d.roughness);
    get_number(*m, "ior", d.ior);
    get_vec3(*m, "albedo", d.albedo);
}

template <class Fn> void for_each_object(const Node& cls, Fn fn) {
    if (cls.type != Node::Object) return;
    for (const Member& m : cls.members)
        if (all_digits(m.first) && m.second->type == Node::Object) fn(m.first, *m.second);
}

void read_trs(const Node& obj, int kind, std::vector<rt_prim_desc>& out, std::vector<std::string>& tex) {
    rt_prim_desc d; std::memset(&d, 0, sizeof(d));
    d.kind = kind;
    double t[3] = { 0, 0, 0 }, s[3] = { 1, 1, 1 }, r = 0, p = 0, y = 0;
    get_vec3(obj, "translation", t);
    get_number(obj, "roll", r); get_number(obj, "pitch", p); get_number(obj, "yaw", y);
    get_vec3(obj, "scale", s);
    d.q[0] = t[0]; d.q[1] = t[1]; d.q[2] = t[2]; d.q[3] = r; d.q[4] = p; d.q[5] = y; d.q[6] = s[0]; d.q[7] = s[1]; d.q[8] = s[2];
    read_material(obj, d);
    out.push_back(d);
    tex.push_back(g_texture_path);
}

bool read_file(const char* path, std::string& text) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16]; size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    std::fclose(f);
    return true;
}

void camera_defaults(rt_camera& c) {
    std::memset(&c, 0, sizeof(c));
    c.focal = 35.0; c.sensor_w = 36.0; c.sensor_h = 24.0; c.width = 1920; c.height = 1080;
}

} // namespace

int rt_parse_scene_file(const char* path, std::vector<rt_prim_desc>& prims, std::vector<rt_light>& lights,
                        std::vector<std::pair<std::string, rt_camera>>& cameras, std::vector<std::string>& tex_paths, std::string& err) {
    std::string text;
    if (!read_file(path, text)) { err = std::string("cannot open ") + path; return RT_ERR_IO; }
    return rt_parse_scene_text(text.data(), text.size(), prims, lights, cameras, tex_paths, err);
}

int rt_parse_scene_text(const char* text, size_t len, std::vector<rt_prim_desc>& prims, std::vector<rt_light>& lights,
                        std::vector<std::pair<std::string, rt_camera>>& cameras, std::vector<std::string>& tex_paths, std::string& err) {
    Node root;
    Parser ps{ text, text + len, std::string() };
    if (!ps.value(root)) {
        // load_from_json reports success even when nothing parses (raytracer.cpp:120); a malformed number is
        // the one case where the reference does not return at all (std::stod throws).
        if (ps.err == "malformed number") { err = ps.err; return RT_ERR_PARSE; }
        root = Node();
    }

    prims.clear(); lights.clear(); cameras.clear(); tex_paths.clear();
    std::vector<std::string> light_ids;
    if (const Node* mesh = find_block(root, "MESH")) {
        // primitive-id order of rebuild_mesh_ptrs: cubes, planes, cylinders, spheres (raytracer.cpp:20-27)
        if (const Node* c = find_block(*mesh, "Cube"))
            for_each_object(*c, [&](const std::string&, const Node& o) { read_trs(o, RT_CUBE, prims, tex_paths); });
        if (const Node* c = find_block(*mesh, "Plane"))
            for_each_object(*c, [&](const std::string&, const Node& o) {
                const Node* corners = find_block(o, "corners");
                if (!corners) return;
                rt_prim_desc d; std::memset(&d, 0, sizeof(d));
                d.kind = RT_PLANE;
                int n = 0;
                auto take = [&](const Node& item) {
                    if (n == 4 || item.type != Node::Object) return;
                    double x = 0, y = 0, z = 0;
                    get_number(item, "x", x); get_number(item, "y", y); get_number(item, "z", z);
                    d.q[3*n] = x; d.q[3*n + 1] = y; d.q[3*n + 2] = z; ++n;
                };
                if (corners->type == Node::Array) for (const auto& it : corners->items) take(*it);
                else for (const Member& m : corners->members) take(*m.second);
                if (n == 0) return;
                if (n < 4) d.reserved = n;            // kept with the corners it has, never hit (mesh.cpp:796)
                read_material(o, d);
                prims.push_back(d);
                tex_paths.push_back(g_texture_path);
            });
        if (const Node* c = find_block(*mesh, "Cylinder"))
            for_each_object(*c, [&](const std::string&, const Node& o) { read_trs(o, RT_CYLINDER, prims, tex_paths); });
        if (const Node* c = find_block(*mesh, "Sphere"))
            for_each_object(*c, [&](const std::string&, const Node& o) {
                rt_prim_desc d; std::memset(&d, 0, sizeof(d));
                d.kind = RT_SPHERE;
                double l[3] = { 0, 0, 0 }, s[3] = { 1, 1, 1 };
                get_vec3(o, "location", l); get_vec3(o, "scale", s);
                for (int k = 0; k < 3; ++k) { d.q[k] = l[k]; d.q[3 + k] = s[k]; }
                read_material(o, d);
                prims.push_back(d);
                tex_paths.push_back(g_texture_path);
            });
    }
    if (const Node* cam = find_block(root, "CAMERA"))
        if (const Node* persp = find_block(*cam, "PERSP"))
            for_each_object(*persp, [&](const std::string& id, const Node& o) {
                double loc[3], dir[3], up[3];
                if (!get_vec3(o, "location", loc) || !get_vec3(o, "direction", dir)) return;
                rt_camera c; camera_defaults(c);
                get_number(o, "focal_length", c.focal);
                get_number(o, "sensor_width", c.sensor_w);
                get_number(o, "sensor_height", c.sensor_h);
                int rx = 1920, ry = 1080;
                get_vec2i(o, "film_resolution", rx, ry);
                c.width = rx; c.height = ry;
                rt_camera_basis(loc, dir, get_vec3(o, "up", up) ? up : nullptr, &c);
                for (auto& kv : cameras) if (kv.first == id) { kv.second = c; return; }
                cameras.emplace_back(id, c);
            });
    if (const Node* li = find_block(root, "LIGHT"))
        if (const Node* pt = find_block(*li, "POINT"))
            for_each_object(*pt, [&](const std::string& id, const Node& o) {
                rt_light L; L.intensity = 1.0;
                if (!get_vec3(o, "location", L.pos)) return;
                get_number(o, "radiant_intensity", L.intensity);
                // the reference keeps lights in a map keyed by the id text (raytracer.cpp:103): a repeated id replaces
                // the earlier light
                for (size_t k = 0; k < light_ids.size(); ++k) if (light_ids[k] == id) { lights[k] = L; return; }
                light_ids.push_back(id);
                lights.push_back(L);
            });
    return RT_OK;
}

// read_camera_from_json (persp_camera.cpp:120-168): every field is mandatory, world-up orientation only.
int rt_parse_camera_file(const char* path, const char* camera_id, rt_camera* out, std::string& err) {
    std::string text;
    if (!read_file(path, text)) { err = std::string("Failed to open JSON file: ") + path; return RT_ERR_IO; }
    Node root;
    Parser ps{ text.data(), text.data() + text.size(), std::string() };
    if (!ps.value(root)) { err = ps.err; return ps.err == "malformed number" ? RT_ERR_PARSE : RT_ERR_NOT_FOUND; }
    const Node* cam = find_block(root, "CAMERA");
    const Node* persp = cam ? find_block(*cam, "PERSP") : nullptr;
    const Node* o = persp ? find_block(*persp, camera_id) : nullptr;
    if (!o) { err = "camera not found"; return RT_ERR_NOT_FOUND; }
    double loc[3], dir[3];
    rt_camera c; camera_defaults(c);
    int rx = 0, ry = 0;
    bool ok = get_vec3(*o, "location", loc);
    ok &= get_vec3(*o, "direction", dir);
    ok &= get_number(*o, "focal_length", c.focal);
    ok &= get_number(*o, "sensor_width", c.sensor_w);
    ok &= get_number(*o, "sensor_height", c.sensor_h);
    ok &= get_vec2i(*o, "film_resolution", rx, ry);
    if (!ok) { err = "camera block incomplete"; return RT_ERR_NOT_FOUND; }
    c.width = rx; c.height = ry;
    rt_camera_basis(loc, dir, nullptr, &c);
    *out = c;
    return RT_OK;
}
