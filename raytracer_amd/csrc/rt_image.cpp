// PPM reading for albedo textures, and attaching textures to a scene's primitives.
#include <cctype>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>

#include "rt_scene.h"

namespace {

struct Reader {
    const std::vector<uint8_t>& buf; size_t pos = 0;
    explicit Reader(const std::vector<uint8_t>& b) : buf(b) {}
    int peek() const { return pos < buf.size() ? buf[pos] : EOF; }
    void skip_blank_and_comments() {           // utils.cpp:147-154
        for (int c = peek(); c != EOF && (std::isspace(c) || c == '#'); c = peek()) {
            if (c == '#') { while (pos < buf.size() && buf[pos] != '\n') ++pos; if (pos < buf.size()) ++pos; }
            else ++pos;
        }
    }
    // operator>>(int): leading blanks, optional sign, digits
    bool integer(int& out) {
        while (pos < buf.size() && std::isspace(buf[pos])) ++pos;
        size_t p = pos; bool neg = false;
        if (p < buf.size() && (buf[p] == '+' || buf[p] == '-')) { neg = buf[p] == '-'; ++p; }
        if (p >= buf.size() || !std::isdigit(buf[p])) return false;
        long long v = 0;
        while (p < buf.size() && std::isdigit(buf[p])) { v = v * 10 + (buf[p] - '0'); if (v > 2147483647LL) return false; ++p; }
        pos = p; out = static_cast<int>(neg ? -v : v);
        return true;
    }
};

} // namespace

// Image::loadFromFile, utils.cpp:212-250: "P6" or "P3", '#' comments between the header fields, one byte after
// maxval, maxval in [1, 255]; sample values are kept as stored (not rescaled to 255).
bool rt_load_ppm(const char* path, int32_t& width, int32_t& height, int32_t& maxval, std::vector<uint8_t>& rgb, std::string& err) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { err = std::string("Failed to open image file: ") + path; return false; }
    std::vector<uint8_t> buf; uint8_t chunk[1 << 16]; size_t n;
    while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    std::fclose(f);
    Reader r(buf);
    while (r.pos < buf.size() && std::isspace(buf[r.pos])) ++r.pos;
    std::string magic;
    while (r.pos < buf.size() && !std::isspace(buf[r.pos])) magic.push_back(static_cast<char>(buf[r.pos++]));
    if (magic != "P6" && magic != "P3") { err = "Unsupported PPM format (expected P3 or P6): " + magic; return false; }
    int w = 0, h = 0, mv = 0;
    r.skip_blank_and_comments(); (void)r.integer(w);
    r.skip_blank_and_comments(); (void)r.integer(h);
    r.skip_blank_and_comments(); (void)r.integer(mv);
    if (r.pos < buf.size()) ++r.pos;
    if (w <= 0 || h <= 0 || mv <= 0) { err = "Invalid PPM header values"; return false; }
    if (mv > 255) { err = "Unsupported PPM max value (>255 not supported)"; return false; }
    const size_t count = static_cast<size_t>(w) * static_cast<size_t>(h) * 3u;
    // the payload must be there before anything of that size is allocated (a P3 sample takes at least two bytes, the
    // last one at least one): a few header bytes must not be able to ask for gigabytes
    if (magic == "P6" ? buf.size() - r.pos < count : (buf.size() - r.pos + 1) / 2 < count) {
        err = magic == "P6" ? "Unexpected EOF while reading binary PPM data" : "Unexpected EOF reading P3 data";
        return false;
    }
    rgb.resize(count);
    if (magic == "P6") {
        std::memcpy(rgb.data(), buf.data() + r.pos, count);
    } else {
        for (size_t i = 0; i < count; i += 3) {
            int c[3];
            for (int k = 0; k < 3; ++k) {
                r.skip_blank_and_comments();
                if (!r.integer(c[k])) { err = "Unexpected EOF reading P3 data"; return false; }
            }
            for (int k = 0; k < 3; ++k) {
                if (c[k] < 0 || c[k] > mv) { err = "P3 pixel out of range"; return false; }
                rgb[i + k] = static_cast<uint8_t>(c[k]);
            }
        }
    }
    width = w; height = h; maxval = mv;
    return true;
}

// One texture per distinct path that loads; a path that does not load leaves its primitives with their solid colour
// (mesh.cpp:105-110 swallows the exception).
void rt_scene_attach_textures(rt_scene* s, const std::vector<std::string>& tex_paths) {
    s->texels.clear(); s->tex_table.clear();
    s->prim_tex.assign(s->prims.size(), -1);
    std::map<std::string, int32_t> seen;
    for (size_t i = 0; i < tex_paths.size() && i < s->prims.size(); ++i) {
        if (tex_paths[i].empty()) continue;
        auto it = seen.find(tex_paths[i]);
        if (it == seen.end()) {
            int32_t w = 0, h = 0, mv = 0; std::vector<uint8_t> rgb; std::string err;
            int32_t index = -1;
            try {       // whatever goes wrong with one texture, the primitive keeps its solid colour (mesh.cpp:105-110)
                if (rt_load_ppm(tex_paths[i].c_str(), w, h, mv, rgb, err)) {
                    const size_t before = s->texels.size();
                    s->texels.insert(s->texels.end(), rgb.begin(), rgb.end());
                    s->tex_table.push_back(rt_tex_rec{ before, w, h });
                    index = static_cast<int32_t>(s->tex_table.size()) - 1;
                }
            } catch (const std::exception&) { index = -1; }
            it = seen.emplace(tex_paths[i], index).first;
        }
        s->prim_tex[i] = it->second;
    }
    for (size_t i = 0; i < s->mats.size() && i < s->prim_tex.size(); ++i) s->mats[i].tex = s->prim_tex[i];
}
