// Host-side robustness driver, built with -fsanitize=address,undefined by tests/test_host_sanitized.py.
// Feeds the scene parser, the PPM reader and the BVH builder with well-formed, mutated and truncated input and checks
// structural invariants of whatever comes out.  No GPU, no HIP: only the host sources of the library are linked.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rt_scene.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {                    // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double uniform() { return static_cast<double>(next_u64() >> 11) * (1.0 / 9007199254740992.0); }

static int g_failed = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); ++g_failed; } } while (0)

static void check_scene(rt_scene& s) {
    if (rt_scene_finalize(&s, nullptr) != RT_OK) return;
    const size_t n_nodes = s.nodes.size();
    CHECK(s.recs.size() == s.n_slots + s.n_loose);
    CHECK(s.slot_prim.size() == s.recs.size());
    CHECK(s.info.depth <= 31);
    if (s.n_slots == 0) { CHECK(n_nodes == 0 || n_nodes == 2); return; }
    // every slot is reached exactly once, children sit at even indices, boxes contain their children's boxes
    std::vector<uint32_t> seen(s.n_slots, 0);
    std::vector<uint32_t> stack{ 0 };
    size_t visited = 0;
    while (!stack.empty()) {
        const uint32_t k = stack.back(); stack.pop_back();
        CHECK(k < n_nodes);
        if (k >= n_nodes) break;
        const rt_node32& nd = s.nodes[k];
        ++visited;
        if (nd.count) {
            CHECK(nd.first + nd.count <= s.n_slots);
            for (uint32_t i = 0; i < nd.count && nd.first + i < s.n_slots; ++i) ++seen[nd.first + i];
        } else {
            CHECK(nd.first % 2 == 0 && nd.first + 1 < n_nodes && nd.first >= 2);
            if (nd.first + 1 >= n_nodes) break;
            for (uint32_t c = nd.first; c < nd.first + 2; ++c)
                for (int a = 0; a < 3; ++a) CHECK(s.nodes[c].lo[a] >= nd.lo[a] && s.nodes[c].hi[a] <= nd.hi[a]);
            stack.push_back(nd.first); stack.push_back(nd.first + 1);
        }
    }
    CHECK(visited + 1 == n_nodes);              // node 1 is the unused slot beside the root
    for (uint32_t c : seen) CHECK(c == 1);
}

static std::string read_all(const char* path) {
    std::ifstream in(path, std::ios::binary);
    std::stringstream ss; ss << in.rdbuf();
    return ss.str();
}

static void parse_text(const std::string& text) {
    rt_scene s; std::vector<std::string> tex; std::string err;
    const int rc = rt_parse_scene_text(text.data(), text.size(), s.prims, s.lights, s.cameras, tex, err);
    CHECK(rc == RT_OK || rc == RT_ERR_PARSE);
    if (rc != RT_OK) return;
    CHECK(tex.size() == s.prims.size());
    check_scene(s);
}

int main(int argc, char** argv) {
    // 1. documents given on the command line: as they are, truncated at many points, and with random byte edits
    for (int i = 1; i < argc; ++i) {
        const std::string doc = read_all(argv[i]);
        parse_text(doc);
        const size_t cuts = doc.size() < 4000 ? doc.size() : 4000;
        for (size_t c = 0; c < cuts; ++c) parse_text(doc.substr(0, doc.size() * c / (cuts ? cuts : 1)));
        for (int m = 0; m < 300; ++m) {
            std::string mut = doc;
            const int edits = 1 + static_cast<int>(next_u64() % 4);
            static const char alphabet[] = "{}[]\":,.-+eE0123456789 \n\\x";
            for (int e = 0; e < edits && !mut.empty(); ++e) {
                const size_t at = next_u64() % mut.size();
                switch (next_u64() % 3) {
                    case 0: mut[at] = alphabet[next_u64() % (sizeof(alphabet) - 1)]; break;
                    case 1: mut.erase(at, 1 + next_u64() % 8); break;
                    default: mut.insert(at, 1, alphabet[next_u64() % (sizeof(alphabet) - 1)]); break;
                }
            }
            parse_text(mut);
        }
    }
    // 2. pathological nesting and sizes
    parse_text(std::string(100000, '['));
    parse_text(std::string(100000, '{'));
    parse_text("{\"MESH\":{\"Cube\":{\"1\":{\"scale\":{\"x\":1e999,\"y\":-1e999,\"z\":1e-999}}}}}");
    parse_text("{\"MESH\":{\"Sphere\":{\"1\":{\"location\":{\"x\":1e308,\"y\":1e308,\"z\":1e308},\"scale\":{\"x\":1e308,\"y\":1,\"z\":1}}}}}");
    parse_text("{\"MESH\":{\"Plane\":{\"1\":{\"corners\":[]},\"2\":{\"corners\":[{\"x\":1}]},\"3\":{\"corners\":{\"a\":{\"x\":1,\"y\":2,\"z\":3}}}}}}");
    parse_text(std::string("{\"a\":\"") + std::string(1 << 20, 'x'));
    // 3. builder: random boxes, duplicates, a sorted line (the adversarial case for depth), huge and tiny extents
    for (int round = 0; round < 30; ++round) {
        const uint32_t n = 1 + static_cast<uint32_t>(next_u64() % 3000);
        std::vector<rt_box64> boxes(n);
        for (uint32_t i = 0; i < n; ++i) {
            double c[3], e[3];
            for (int a = 0; a < 3; ++a) {
                switch (round % 5) {
                    case 0: c[a] = uniform() * 100 - 50; e[a] = uniform(); break;
                    case 1: c[a] = (a == 0) ? std::ldexp(1.0, static_cast<int>(i % 900)) : 0.0; e[a] = 0.0; break;   // exponential line
                    case 2: c[a] = 7.0; e[a] = 0.5; break;                                                        // all identical
                    case 3: c[a] = uniform() * 1e300; e[a] = uniform() * 1e-300; break;
                    default: c[a] = static_cast<double>(i) * (a == 1); e[a] = (i % 7 == 0) ? 1e6 : 0.25; break;
                }
            }
            for (int a = 0; a < 3; ++a) { boxes[i].lo[a] = c[a] - e[a]; boxes[i].hi[a] = c[a] + e[a]; }
        }
        for (uint32_t leaf = 1; leaf <= 8; leaf += 3) {
            rt_bvh_result out;
            rt_build_bvh(boxes, leaf, 4 + static_cast<uint32_t>(next_u64() % 40), out);
            CHECK(out.depth <= 31);
            CHECK(out.order.size() == n);
            std::vector<uint8_t> used(n, 0);
            for (uint32_t p : out.order) { CHECK(p < n); if (p < n) { CHECK(!used[p]); used[p] = 1; } }
        }
    }
    // 4. PPM reader: valid files, truncations and edits of them
    const char* tmp = std::getenv("RT_FUZZ_TMP");
    if (tmp) {
        const std::string path = std::string(tmp) + "/fuzz.ppm";
        const std::string good6 = std::string("P6\n# c\n3 2\n255\n") + std::string(18, '\x7f');
        const std::string good3 = "P3\n2 2\n# c\n100\n1 2 3 4 5 6\n7 8 9 10 11 100\n";
        for (const std::string& base : { good6, good3 }) {
            for (size_t cut = 0; cut <= base.size(); ++cut) {
                for (int variant = 0; variant < 3; ++variant) {
                    std::string data = base.substr(0, cut);
                    if (variant == 1 && !data.empty()) data[next_u64() % data.size()] = static_cast<char>(next_u64());
                    if (variant == 2) data += "999999999999999999999 -5 3";
                    { std::ofstream f(path, std::ios::binary); f << data; }
                    int32_t w = 0, h = 0, mv = 0; std::vector<uint8_t> rgb; std::string err;
                    if (rt_load_ppm(path.c_str(), w, h, mv, rgb, err)) {
                        CHECK(w > 0 && h > 0 && mv > 0 && mv <= 255);
                        CHECK(rgb.size() == static_cast<size_t>(w) * h * 3);
                    } else CHECK(!err.empty());
                }
            }
        }
        { std::ofstream f(path, std::ios::binary); f << "P6\n100000 100000\n255\n"; }      // header promises 30 GB
        int32_t w = 0, h = 0, mv = 0; std::vector<uint8_t> rgb; std::string err;
        CHECK(!rt_load_ppm(path.c_str(), w, h, mv, rgb, err));
        std::remove(path.c_str());
    }
    if (g_failed) { std::fprintf(stderr, "%d checks failed\n", g_failed); return 1; }
    std::puts("ok");
    return 0;
}
