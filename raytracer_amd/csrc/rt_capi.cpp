// C ABI of librt_hip.so (include/rt_hip.h): host-side scene handling and kernel launches.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_scene.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

#define RT_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

constexpr uint32_t kQueueSlots = 64 * RT_QUEUE_PARTS;   // RT_QUEUE_PARTS heads per persistent launch
constexpr size_t kCounterBytes = static_cast<size_t>(RT_CTR_SHARDS) * RT_CTR_STRIDE * sizeof(uint64_t);
static_assert(sizeof(rt_counters) == 8 * sizeof(uint64_t), "counter block");

template <class T> int upload(const std::vector<T>& v, T** dev) {
    *dev = nullptr;
    if (v.empty()) return RT_OK;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(dev), v.size() * sizeof(T)));
    RT_HIP(hipMemcpy(*dev, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

} // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool counting = false;
    bool have_scene = false;
    DevScene sc{};
    rt_node32* d_nodes = nullptr; rt_prim_rec* d_recs = nullptr; rt_mat_rec* d_mats = nullptr;
    rt_light* d_lights = nullptr; rt_counters* d_ctr = nullptr;
    uint8_t* d_texels = nullptr; rt_tex_rec* d_tex_table = nullptr;
    float* d_loose_boxes = nullptr;        // boxes of the primitives beside the tree (8 floats each)
    rt_node4* d_nodes4 = nullptr;          // wide nodes (host-built: collapsed on the host; device-built: written by the builder)
    uint32_t n_nodes4_dev = 0;
    bool dev_wide = false; uint32_t dev_root_ref = 0; float dev_root_lo[3] = {}, dev_root_hi[3] = {};   // device-built scenes
    rt_inst_rec* d_inst = nullptr;         // two-level scenes
    uint32_t* d_slot_prim = nullptr;       // slot -> primitive id, as uploaded or as the device build ordered it
    uint32_t n_nodes_dev = 0; uint64_t n_slots_dev = 0;
    double build_ms = 0.0;
    uint8_t* d_gamma = nullptr;            // 256 bytes of the display transfer table, then 256 bytes of identity
    unsigned int* d_queue = nullptr;       // kQueueSlots work-queue heads, one per launch of a pipeline call
    hipEvent_t switch_event = nullptr;     // orders a stream switch; created with the context so that a switch allocates nothing
    void* static_scratch = nullptr;        // this GPU's copy of the library's fixed workspace (rt_render_rect_static)
    size_t static_scratch_bytes = 0;
    uint32_t max_waves = 4096;             // persistent waves per launch: CUs x resident waves per CU
    uint32_t wave_div = 8;                 // a launch gets one wave per wave_div groups of 64 rays (RT_WAVE_DIV; experiments)
    int refill_env = 0;                    // RT_REFILL_MIN from the environment (experiments); 0: the defaults below
    // free lanes a persistent wave lets pile up before it swaps rays: walks through instances pay more per swap
    // (100M-triangle frame at 32 / 40 / 48: 22.2 / 21.7 / 21.4 ms; one-level scenes are best at 32)
    int refill() const { return refill_env ? refill_env : (sc.inst ? RT_REFILL_MIN_INST : RT_REFILL_MIN); }
    int refill_primary_env = 0;            // RT_REFILL_MIN_PRIMARY from the environment (experiments)
    // closest-hit launches of a plain scene: camera rays are cheap to set up and their walks end together more often
    int refill_primary() const { return refill_primary_env ? refill_primary_env : (refill_env ? refill_env : (sc.inst ? RT_REFILL_MIN_INST : RT_REFILL_MIN)); }   // 28-44 time the same: no value of its own
    // the resolve of a one-batch frame in two passes (mixed precision for every slot it decides, FP64 for the listed rest):
    // pays when few slots need the second pass, i.e. when few materials have highlights too sharp for the first
    bool two_pass_resolve = false;
    bool entry_search = true;              // start a tile's walks below the root where its beam allows (rt_ctx_set_entry_search)
    bool brute_force = false;              // closest-hit entry points test every primitive against every ray (rt_ctx_set_brute_force)
    // optional per-launch timing: (kind, start, stop) triples drawn from a pool of events
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_kind;             // one entry per recorded launch; events 2k and 2k+1 of the pool belong to it
    hipEvent_t timing_event(size_t i) {
        while (ev_pool.size() <= i) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev_pool.push_back(e); }
        return ev_pool[i];
    }
    void mark_begin(int kind) {
        if (!timing) return;
        if (hipEvent_t e = timing_event(2 * ev_kind.size())) { (void)hipEventRecord(e, stream); ev_kind.push_back(kind); }
    }
    void mark_end() {
        if (!timing || ev_kind.empty()) return;
        if (hipEvent_t e = timing_event(2 * ev_kind.size() - 1)) (void)hipEventRecord(e, stream);
    }
    // grow-only scratch of the *_host conveniences
    void* scratch[8] = {};    // 0..2 host conveniences, 3 frame working set, 4 one sample's colours, 5 sample sums, 6 bounce levels, 7 gathered tiles
    size_t scratch_bytes[8] = {};

    void free_scene() {
        if (d_nodes) (void)hipFree(d_nodes);
        if (d_recs) (void)hipFree(d_recs);
        if (d_mats) (void)hipFree(d_mats);
        if (d_lights) (void)hipFree(d_lights);
        if (d_texels) (void)hipFree(d_texels);
        if (d_tex_table) (void)hipFree(d_tex_table);
        if (d_slot_prim) (void)hipFree(d_slot_prim);
        if (d_inst) (void)hipFree(d_inst);
        if (d_nodes4) (void)hipFree(d_nodes4);
        if (d_loose_boxes) (void)hipFree(d_loose_boxes);
        d_nodes4 = nullptr; d_loose_boxes = nullptr;
        d_slot_prim = nullptr; d_inst = nullptr;
        d_nodes = nullptr; d_recs = nullptr; d_mats = nullptr; d_lights = nullptr; d_texels = nullptr; d_tex_table = nullptr;
        have_scene = false;
    }
    // Grow-only buffer `slot` of at least `bytes`, on THIS context's GPU whatever device the caller left current (a
    // process that drives several GPUs calls this for one context after working on another).
    int need(int slot, size_t bytes, void** out) {
        if (scratch_bytes[slot] < bytes) {
            RT_HIP(hipSetDevice(device));
            if (scratch[slot]) (void)hipFree(scratch[slot]);
            scratch[slot] = nullptr; scratch_bytes[slot] = 0;
            RT_HIP(hipMalloc(&scratch[slot], bytes));
            scratch_bytes[slot] = bytes;
        }
        *out = scratch[slot];
        return RT_OK;
    }
};

// for rt_comm.cpp
int rt_fail(int code, const std::string& msg) { return fail(code, msg); }
hipStream_t rt_ctx_stream(rt_ctx* ctx) { return ctx->stream; }
int rt_ctx_device(rt_ctx* ctx) { return ctx->device; }
int rt_ctx_scratch(rt_ctx* ctx, int slot, size_t bytes, void** out) { return ctx->need(slot, bytes, out); }   // slots 0..2 and 7
uint32_t rt_ctx_max_waves(rt_ctx* ctx) { return ctx->max_waves; }                                              // rt_probe.hip

namespace {

bool camera_ok(const rt_camera* cam) { return cam && cam->width > 0 && cam->height > 0; }

// Effective sensor size: the image-aspect / sensor-aspect branch of Camera::pixelToRay
// (s2899221/Code/persp_camera.cpp:98-108), evaluated once per launch.
void to_dev_camera(const rt_camera& c, DevCamera& d) {
    for (int k = 0; k < 3; ++k) d.pos[k] = c.pos[k];
    for (int k = 0; k < 9; ++k) d.rot[k] = c.rot[k];
    const double aspect_img = static_cast<double>(c.width) / static_cast<double>(c.height);
    const double aspect_sensor = (c.sensor_h > 0.0) ? (c.sensor_w / c.sensor_h) : aspect_img;
    if (aspect_img >= aspect_sensor) { d.eff_w = c.sensor_w; d.eff_h = d.eff_w / aspect_img; }
    else                             { d.eff_h = c.sensor_h; d.eff_w = d.eff_h * aspect_img; }
    d.focal = c.focal;
    d.width = static_cast<double>(c.width); d.height = static_cast<double>(c.height);
    d.jitter_seed = 0;
}

int rect_map(const rt_camera* cam, rt_rect r, PixelMap& pm, uint32_t& n_blocks) {
    if (r.w <= 0 || r.h <= 0 || r.x0 < 0 || r.y0 < 0 || r.x0 + static_cast<int64_t>(r.w) > cam->width ||
        r.y0 + static_cast<int64_t>(r.h) > cam->height)
        return fail(RT_ERR_INVALID, "rectangle outside the camera's film resolution");
    std::memset(&pm, 0, sizeof(pm));
    pm.mode = 0; pm.x0 = r.x0; pm.y0 = r.y0; pm.w = r.w; pm.h = r.h;
    pm.super_x = (static_cast<uint32_t>(r.w) + 127u) / 128u;
    const uint64_t nb = static_cast<uint64_t>(pm.super_x) * ((static_cast<uint32_t>(r.h) + 127u) / 128u) * 256u;
    if (nb * 64u * RT_LIGHT_BATCH >= 0xFFFFFFFFull) return fail(RT_ERR_LIMIT, "rectangle too large for one launch");
    n_blocks = static_cast<uint32_t>(nb);      // waves: one 8x8 tile each
    pixel_map_divisors(pm);
    return RT_OK;
}

int tiling_ok(rt_tiling t) {
    if (t.width <= 0 || t.height <= 0 || t.n_ranks <= 0 || t.tile < 16 || (t.tile % 16) != 0 || t.tile > 1024)
        return fail(RT_ERR_INVALID, "tiling: tile must be a multiple of 16 in [16, 1024], sizes and ranks positive");
    return RT_OK;
}

int use(rt_ctx* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    RT_HIP(hipSetDevice(ctx->device));
    return RT_OK;
}

int need_scene(rt_ctx* ctx) {
    if (int rc = use(ctx)) return rc;
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "no scene uploaded to this context");
    return RT_OK;
}

} // namespace

extern "C" {

// ---- scenes --------------------------------------------------------------------------------------

int rt_scene_create(const rt_prim_desc* prims, uint64_t n_prims, const rt_light* lights, uint32_t n_lights,
                    const rt_build_opts* opts, rt_scene** out) {
    if (!out || (n_prims && !prims) || (n_lights && !lights)) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_LIMIT, "out of memory");
    try {
        s->prims.assign(prims, prims + n_prims);
        s->lights.assign(lights, lights + n_lights);
        const int rc = rt_scene_finalize(s, opts);
        if (rc != RT_OK) { delete s; return fail(rc, "scene too large"); }
    } catch (const std::exception& e) { delete s; return fail(RT_ERR_LIMIT, e.what()); }
    *out = s;
    return RT_OK;
}

int rt_scene_create_instanced(const rt_prim_desc* prims, uint64_t n_prims, const rt_instance* instances, uint64_t n_instances,
                              const rt_light* lights, uint32_t n_lights, const rt_build_opts* opts, rt_scene** out) {
    if (!out || (n_prims && !prims) || (n_lights && !lights) || (n_instances && !instances)) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_prims >= (1ull << 28) || n_instances >= (1ull << 28)) return fail(RT_ERR_LIMIT, "scene too large");
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_LIMIT, "out of memory");
    try {
        s->prims.assign(prims, prims + n_prims);
        s->lights.assign(lights, lights + n_lights);
        s->instances.assign(instances, instances + n_instances);
        s->instanced = true;
        const int rc = rt_scene_finalize(s, opts);
        if (rc != RT_OK) {
            delete s;
            return fail(rc, rc == RT_ERR_LIMIT ? "scene too large" :
                        "instanced scene: a mesh run outside the primitive list, an unbounded mesh primitive, a transform that is not rigid, or an unsupported option");
        }
    } catch (const std::exception& e) { delete s; return fail(RT_ERR_LIMIT, e.what()); }
    *out = s;
    return RT_OK;
}

int rt_scene_load_json(const char* path, const rt_build_opts* opts, rt_scene** out) {
    if (!path || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_LIMIT, "out of memory");
    try {
        std::string err;
        std::vector<std::string> tex_paths;
        int rc = rt_parse_scene_file(path, s->prims, s->lights, s->cameras, tex_paths, err);
        if (rc == RT_OK) rc = rt_scene_finalize(s, opts);
        if (rc == RT_OK) rt_scene_attach_textures(s, tex_paths);
        if (rc != RT_OK) { delete s; return fail(rc, err.empty() ? "scene load failed" : err); }
    } catch (const std::exception& e) { delete s; return fail(RT_ERR_LIMIT, e.what()); }
    *out = s;
    return RT_OK;
}

int rt_scene_load_json_text(const char* text, uint64_t len, const rt_build_opts* opts, rt_scene** out) {
    if (!text || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_LIMIT, "out of memory");
    try {
        std::string err;
        std::vector<std::string> tex_paths;
        int rc = rt_parse_scene_text(text, static_cast<size_t>(len), s->prims, s->lights, s->cameras, tex_paths, err);
        if (rc == RT_OK) rc = rt_scene_finalize(s, opts);
        if (rc == RT_OK) rt_scene_attach_textures(s, tex_paths);
        if (rc != RT_OK) { delete s; return fail(rc, err.empty() ? "scene load failed" : err); }
    } catch (const std::exception& e) { delete s; return fail(RT_ERR_LIMIT, e.what()); }
    *out = s;
    return RT_OK;
}

int rt_scene_set_textures(rt_scene* scene, const rt_texture* textures, uint32_t n_textures, const int32_t* prim_texture) {
    if (!scene || (n_textures && (!textures || !prim_texture))) return fail(RT_ERR_INVALID, "null argument");
    if (scene->instanced) return fail(RT_ERR_INVALID, "textures are not available for instanced scenes");
    try {
        scene->texels.clear(); scene->tex_table.clear();
        scene->prim_tex.assign(scene->prims.size(), -1);
        for (uint32_t t = 0; t < n_textures; ++t) {
            if (textures[t].width <= 0 || textures[t].height <= 0 || !textures[t].rgb) return fail(RT_ERR_INVALID, "empty texture");
            const size_t bytes = static_cast<size_t>(textures[t].width) * static_cast<size_t>(textures[t].height) * 3u;
            scene->tex_table.push_back(rt_tex_rec{ scene->texels.size(), textures[t].width, textures[t].height });
            scene->texels.insert(scene->texels.end(), textures[t].rgb, textures[t].rgb + bytes);
        }
        for (size_t i = 0; i < scene->prims.size() && n_textures; ++i) {
            if (prim_texture[i] >= static_cast<int32_t>(n_textures)) return fail(RT_ERR_INVALID, "texture index out of range");
            scene->prim_tex[i] = prim_texture[i] < 0 ? -1 : prim_texture[i];
        }
        for (size_t i = 0; i < scene->mats.size(); ++i) scene->mats[i].tex = scene->prim_tex[i];
    } catch (const std::exception& e) { return fail(RT_ERR_LIMIT, e.what()); }
    return RT_OK;
}

int rt_scene_get_texture(const rt_scene* scene, uint64_t prim_id, rt_texture* out) {
    if (!scene || !out) return fail(RT_ERR_INVALID, "null argument");
    if (prim_id >= scene->prims.size()) return fail(RT_ERR_INVALID, "primitive id out of range");
    out->width = 0; out->height = 0; out->rgb = nullptr;
    const int32_t t = prim_id < scene->prim_tex.size() ? scene->prim_tex[prim_id] : -1;
    if (t >= 0) {
        out->width = scene->tex_table[t].width; out->height = scene->tex_table[t].height;
        out->rgb = scene->texels.data() + scene->tex_table[t].offset;
    }
    return RT_OK;
}

int rt_image_read_ppm(const char* path, int32_t* width, int32_t* height, int32_t* maxval, uint8_t* rgb_or_null, uint64_t capacity) {
    if (!path || !width || !height || !maxval) return fail(RT_ERR_INVALID, "null argument");
    try {
        std::vector<uint8_t> rgb; std::string err;
        if (!rt_load_ppm(path, *width, *height, *maxval, rgb, err))
            return fail(err.rfind("Failed to open", 0) == 0 ? RT_ERR_IO : RT_ERR_PARSE, err);
        if (rgb_or_null) {
            if (capacity < rgb.size()) return fail(RT_ERR_LIMIT, "pixel buffer too small");
            std::memcpy(rgb_or_null, rgb.data(), rgb.size());
        }
    } catch (const std::exception& e) { return fail(RT_ERR_LIMIT, e.what()); }
    return RT_OK;
}

int rt_scene_destroy(rt_scene* scene) { delete scene; return RT_OK; }

int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out) {
    if (!scene || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = scene->info; return RT_OK;
}

int rt_scene_get_prims(const rt_scene* scene, const rt_prim_desc** prims, uint64_t* n) {
    if (!scene || !prims || !n) return fail(RT_ERR_INVALID, "null argument");
    *prims = scene->prims.data(); *n = scene->prims.size(); return RT_OK;
}

int rt_scene_get_lights(const rt_scene* scene, const rt_light** lights, uint32_t* n) {
    if (!scene || !lights || !n) return fail(RT_ERR_INVALID, "null argument");
    *lights = scene->lights.data(); *n = static_cast<uint32_t>(scene->lights.size()); return RT_OK;
}

int rt_scene_get_bvh(const rt_scene* scene, const rt_node32** nodes, uint32_t* n_nodes, const uint32_t** slot_prim, uint64_t* n_slots) {
    if (!scene || !nodes || !n_nodes || !slot_prim || !n_slots) return fail(RT_ERR_INVALID, "null argument");
    if (scene->device_build) return fail(RT_ERR_NOT_FOUND, "this scene's hierarchy is built on the device: see rt_ctx_download_bvh");
    *nodes = scene->nodes.data(); *n_nodes = static_cast<uint32_t>(scene->nodes.size());
    *slot_prim = scene->slot_prim.data(); *n_slots = scene->n_slots;
    return RT_OK;
}

int rt_scene_get_loose_boxes(const rt_scene* scene, const float** boxes8, uint64_t* n) {
    if (!scene || !boxes8 || !n) return fail(RT_ERR_INVALID, "null argument");
    *boxes8 = scene->loose_boxes.empty() ? nullptr : scene->loose_boxes.data();
    *n = scene->loose_boxes.size() / 8;
    return RT_OK;
}

int rt_scene_get_wide_bvh(const rt_scene* scene, const rt_node4** nodes4, uint32_t* n_nodes4) {
    if (!scene) return fail(RT_ERR_INVALID, "null scene");
    if (nodes4) *nodes4 = scene->nodes4.empty() ? nullptr : scene->nodes4.data();
    if (n_nodes4) *n_nodes4 = static_cast<uint32_t>(scene->nodes4.size());
    return RT_OK;
}

int rt_scene_camera(const rt_scene* scene, const char* id, int32_t index, rt_camera* out, char* id_out) {
    if (!scene || !out) return fail(RT_ERR_INVALID, "null argument");
    const std::pair<std::string, rt_camera>* found = nullptr;
    if (id) { for (const auto& kv : scene->cameras) if (kv.first == id) { found = &kv; break; } }
    else if (index >= 0 && static_cast<size_t>(index) < scene->cameras.size()) found = &scene->cameras[index];
    if (!found) return fail(RT_ERR_NOT_FOUND, "camera not found");
    *out = found->second;
    if (id_out) { std::strncpy(id_out, found->first.c_str(), 31); id_out[31] = 0; }
    return RT_OK;
}

int rt_prim_aabb(const rt_prim_desc* prim, double out6[6]) {
    if (!prim || !out6) return fail(RT_ERR_INVALID, "null argument");
    rt_reference_aabb(*prim, out6); return RT_OK;
}

int rt_camera_look(const double pos[3], const double dir[3], const double* up, rt_camera* cam) {
    if (!pos || !dir || !cam) return fail(RT_ERR_INVALID, "null argument");
    rt_camera_basis(pos, dir, up, cam); return RT_OK;
}

int rt_camera_from_json(const char* path, const char* camera_id, rt_camera* out) {
    if (!path || !camera_id || !out) return fail(RT_ERR_INVALID, "null argument");
    std::string err;
    const int rc = rt_parse_camera_file(path, camera_id, out, err);
    return rc == RT_OK ? RT_OK : fail(rc, err);
}

int rt_camera_pixel_to_ray(const rt_camera* cam, double px, double py, double origin[3], double dir[3]) {
    if (!cam || !origin || !dir) return fail(RT_ERR_INVALID, "null argument");
    rt_pixel_to_ray(cam, px, py, origin, dir); return RT_OK;
}

int rt_gamma_lut(uint8_t lut[256]) { if (!lut) return fail(RT_ERR_INVALID, "null argument"); rt_gamma_table(lut); return RT_OK; }

// ---- contexts ------------------------------------------------------------------------------------

int rt_ctx_create(int device, rt_ctx** out) {
    if (!out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(RT_ERR_NO_DEVICE, "device index out of range");
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return fail(RT_ERR_LIMIT, "out of memory");
    c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->switch_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_ctr), kCounterBytes);
    if (e == hipSuccess) e = hipMemset(c->d_ctr, 0, kCounterBytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_queue), kQueueSlots * sizeof(unsigned int));
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) {
            int per_cu = 16;                                   // 4 waves per SIMD at <= 128 VGPRs
            if (const char* env = std::getenv("RT_WAVES_PER_CU")) { const int v = std::atoi(env); if (v >= 1 && v <= 32) per_cu = v; }
            c->max_waves = static_cast<uint32_t>(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * per_cu;
            if (const char* env = std::getenv("RT_MAX_WAVES")) { const long v = std::atol(env); if (v >= 64 && v <= (1l << 24)) c->max_waves = static_cast<uint32_t>(v); }
            if (const char* env = std::getenv("RT_WAVE_DIV")) { const int v = std::atoi(env); if (v >= 1 && v <= 1024) c->wave_div = static_cast<uint32_t>(v); }
            if (const char* env = std::getenv("RT_ENTRY_SEARCH")) c->entry_search = std::atoi(env) != 0;
            if (const char* env = std::getenv("RT_REFILL_MIN")) { const int v = std::atoi(env); if (v >= 1 && v <= 64) c->refill_env = v; }
            if (const char* env = std::getenv("RT_REFILL_MIN_PRIMARY")) { const int v = std::atoi(env); if (v >= 1 && v <= 64) c->refill_primary_env = v; }
        }
    }
    // this GPU's address of the library's fixed workspace, looked up once per device with the device current
    if (e == hipSuccess) e = rt_static_scratch(device, &c->static_scratch, &c->static_scratch_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_gamma), 512);
    if (e == hipSuccess) {
        uint8_t lut[512]; rt_gamma_table(lut);
        for (int k = 0; k < 256; ++k) lut[256 + k] = static_cast<uint8_t>(k);
        e = hipMemcpy(c->d_gamma, lut, 512, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        const std::string msg = std::string("context creation: ") + hipGetErrorString(e);
        rt_ctx_destroy(c);
        return fail(RT_ERR_NO_DEVICE, msg);
    }
    c->stream = c->own_stream;
    *out = c;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    ctx->free_scene();
    if (ctx->d_ctr) (void)hipFree(ctx->d_ctr);
    if (ctx->d_gamma) (void)hipFree(ctx->d_gamma);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    for (void* p : ctx->scratch) if (p) (void)hipFree(p);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->switch_event) (void)hipEventDestroy(ctx->switch_event);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RT_OK;
}

namespace {
// The frame working set, the queue heads and the counters are shared by whatever stream the context launches on: work
// queued on the new stream must start after what is still pending on the old one.  The event that orders the two is the
// context's own (nothing is allocated here).  A stream that is being captured cannot wait on an event recorded outside
// its capture; such a stream is taken as it is, and the caller orders it against the old stream's pending work (set the
// stream before the capture begins, or synchronise first).
int switch_stream(rt_ctx* ctx, hipStream_t to) {
    if (to == ctx->stream) return RT_OK;
    if (int rc = use(ctx)) return rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (to && hipStreamIsCapturing(to, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone) {
        hipError_t e = hipEventRecord(ctx->switch_event, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(to, ctx->switch_event, 0);
        if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("stream switch: ") + hipGetErrorString(e));
    }
    ctx->stream = to;
    return RT_OK;
}
} // namespace

int rt_ctx_set_stream(rt_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    return switch_stream(ctx, static_cast<hipStream_t>(hip_stream));
}

int rt_ctx_use_own_stream(rt_ctx* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    return switch_stream(ctx, ctx->own_stream);
}

int rt_ctx_synchronize(rt_ctx* ctx) {
    if (int rc = use(ctx)) return rc;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_ctx_set_timing(rt_ctx* ctx, int on) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->timing = on != 0;
    if (!ctx->timing) ctx->ev_kind.clear();
    return RT_OK;
}

int rt_timing_read(rt_ctx* ctx, rt_timing* out) {
    if (!out) return fail(RT_ERR_INVALID, "null argument");
    if (int rc = use(ctx)) return rc;
    std::memset(out, 0, sizeof(*out));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    double* ms[5] = { &out->primary_ms, &out->shadow_ms, &out->resolve_ms, &out->bounce_ms, &out->entry_ms };
    uint32_t* n[5] = { &out->primary_launches, &out->shadow_launches, &out->resolve_launches, &out->bounce_launches, &out->entry_launches };
    for (size_t k = 0; k < ctx->ev_kind.size(); ++k) {
        float t = 0.0f;
        RT_HIP(hipEventElapsedTime(&t, ctx->ev_pool[2*k], ctx->ev_pool[2*k + 1]));
        *ms[ctx->ev_kind[k]] += t; ++*n[ctx->ev_kind[k]];
    }
    ctx->ev_kind.clear();
    return RT_OK;
}

int rt_ctx_set_counting(rt_ctx* ctx, int on) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->counting = on != 0; return RT_OK;
}

int rt_ctx_set_entry_search(rt_ctx* ctx, int on) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->entry_search = on != 0; return RT_OK;
}

int rt_ctx_set_brute_force(rt_ctx* ctx, int on) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->brute_force = on != 0; return RT_OK;
}

int rt_counters_reset(rt_ctx* ctx) {
    if (int rc = use(ctx)) return rc;
    RT_HIP(hipMemsetAsync(ctx->d_ctr, 0, kCounterBytes, ctx->stream));
    return RT_OK;
}

int rt_counters_read(rt_ctx* ctx, rt_counters* out) {
    if (!out) return fail(RT_ERR_INVALID, "null argument");
    if (int rc = use(ctx)) return rc;
    uint64_t raw[RT_CTR_SHARDS * RT_CTR_STRIDE];
    RT_HIP(hipMemcpyAsync(raw, ctx->d_ctr, kCounterBytes, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t* o = reinterpret_cast<uint64_t*>(out);
    for (int f = 0; f < 8; ++f) {
        o[f] = 0;
        for (int s = 0; s < RT_CTR_SHARDS; ++s) o[f] += raw[s * RT_CTR_STRIDE + f];
    }
    return RT_OK;
}

namespace {

// RT_BUILD_ON_DEVICE: records and FP32 boxes go over in id order; the device sorts them and builds the tree.
int build_on_device(rt_ctx* ctx, const rt_scene* scene) {
    const uint32_t n = static_cast<uint32_t>(scene->n_slots);
    const size_t total = scene->recs.size();                 // bounded, then unbounded
    rt_prim_rec* d_in = nullptr; float* d_boxes = nullptr; uint32_t* d_order = nullptr; void* d_work = nullptr;
    uint8_t* d_need = nullptr;
    size_t work_bytes = 0;
    // wide nodes beside the binary ones (wide node i = inner node i of the binary tree) while their references fit
    const bool want_wide = n >= 2 && 2ull * n <= RT_WIDE_NODE_LIMIT;      // wide node i = inner node i: 2n places, addressed by 32-bit byte offsets
    ctx->dev_root_ref = 0; ctx->dev_wide = false;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int rc = RT_OK;
    auto check = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == RT_OK) rc = fail(RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
    check(hipMalloc(reinterpret_cast<void**>(&d_in), total * sizeof(rt_prim_rec)), "hipMalloc");
    if (rc == RT_OK) check(hipMalloc(reinterpret_cast<void**>(&d_boxes), scene->boxes_f32.size() * sizeof(float)), "hipMalloc");
    if (rc == RT_OK) check(hipMalloc(reinterpret_cast<void**>(&d_order), n * sizeof(uint32_t)), "hipMalloc");
    if (rc == RT_OK) check(rt_device_build_workspace(n, &work_bytes), "sort workspace");
    if (rc == RT_OK) check(hipMalloc(&d_work, work_bytes), "hipMalloc");
    if (rc == RT_OK) check(hipMalloc(reinterpret_cast<void**>(&ctx->d_nodes), 2ull * n * sizeof(rt_node32)), "hipMalloc");
    if (rc == RT_OK) check(hipMalloc(reinterpret_cast<void**>(&ctx->d_recs), total * sizeof(rt_prim_rec)), "hipMalloc");
    if (rc == RT_OK) check(hipMalloc(reinterpret_cast<void**>(&ctx->d_slot_prim), total * sizeof(uint32_t)), "hipMalloc");
    if (rc == RT_OK && want_wide) check(hipMalloc(reinterpret_cast<void**>(&ctx->d_nodes4), 2ull * n * sizeof(rt_node4)), "hipMalloc");
    if (rc == RT_OK && want_wide) check(hipMalloc(reinterpret_cast<void**>(&d_need), 2ull * n), "hipMalloc");
    if (rc == RT_OK && want_wide) check(hipMemsetAsync(d_need, 0, 2ull * n, ctx->stream), "hipMemsetAsync");
    if (rc == RT_OK) check(hipMemcpy(d_in, scene->recs.data(), total * sizeof(rt_prim_rec), hipMemcpyHostToDevice), "hipMemcpy");
    if (rc == RT_OK) check(hipMemcpy(d_boxes, scene->boxes_f32.data(), scene->boxes_f32.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy");
    if (rc == RT_OK) check(hipMemcpy(ctx->d_slot_prim, scene->slot_prim.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy");
    if (rc == RT_OK) {
        float bounds[6];
        for (int k = 0; k < 3; ++k) {
            bounds[k] = std::nextafterf(static_cast<float>(scene->info.bounds[k]), -INFINITY);
            bounds[3 + k] = std::nextafterf(static_cast<float>(scene->info.bounds[3 + k]), INFINITY);
        }
        check(hipEventCreate(&t0), "hipEventCreate"); check(hipEventCreate(&t1), "hipEventCreate");
        if (rc == RT_OK) {
            (void)hipEventRecord(t0, ctx->stream);
            check(rt_device_build(ctx->stream, n, d_boxes, bounds, d_in, ctx->d_nodes, ctx->d_recs, d_order, d_work, work_bytes,
                                  want_wide ? ctx->d_nodes4 : nullptr, d_need), "device build");
            // the unbounded primitives follow the sorted ones unchanged; slot -> id of the sorted part comes from the records
            if (rc == RT_OK && total > n)
                check(hipMemcpyAsync(ctx->d_recs + n, d_in + n, (total - n) * sizeof(rt_prim_rec), hipMemcpyDeviceToDevice, ctx->stream), "hipMemcpyAsync");
            if (rc == RT_OK) check(rt_launch_slot_ids(ctx->stream, ctx->d_recs, n, ctx->d_slot_prim), "slot ids");
            (void)hipEventRecord(t1, ctx->stream);
            check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
            float ms = 0.0f;
            if (rc == RT_OK && hipEventElapsedTime(&ms, t0, t1) == hipSuccess) ctx->build_ms = ms;
            // the root's box for the kernel arguments, and whether the wide walk fits the traversal stack
            rt_node32 root{};
            if (rc == RT_OK) check(hipMemcpy(&root, ctx->d_nodes, sizeof(root), hipMemcpyDeviceToHost), "hipMemcpy");
            uint8_t need0 = 255;
            if (rc == RT_OK && want_wide) check(hipMemcpy(&need0, d_need, 1, hipMemcpyDeviceToHost), "hipMemcpy");
            for (int k = 0; k < 3; ++k) { ctx->dev_root_lo[k] = root.lo[k]; ctx->dev_root_hi[k] = root.hi[k]; }
            ctx->dev_wide = rc == RT_OK && want_wide && root.count == 0 && need0 < RT_WIDE_STACK_LIMIT;
        }
    }
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    if (!ctx->dev_wide && ctx->d_nodes4) { (void)hipFree(ctx->d_nodes4); ctx->d_nodes4 = nullptr; }
    if (d_need) (void)hipFree(d_need);
    if (d_work) (void)hipFree(d_work);
    if (d_order) (void)hipFree(d_order);
    if (d_boxes) (void)hipFree(d_boxes);
    if (d_in) (void)hipFree(d_in);
    ctx->n_nodes_dev = 2u * n;
    return rc;
}

} // namespace

int rt_ctx_build_time(rt_ctx* ctx, double* ms_out) {
    if (!ctx || !ms_out) return fail(RT_ERR_INVALID, "null argument");
    *ms_out = ctx->build_ms;
    return RT_OK;
}

int rt_ctx_download_bvh(rt_ctx* ctx, rt_node32* nodes_out, uint32_t node_capacity, uint32_t* n_nodes_out,
                        uint32_t* slot_prim_out, uint64_t slot_capacity, uint64_t* n_slots_out) {
    if (int rc = need_scene(ctx)) return rc;
    if (n_nodes_out) *n_nodes_out = ctx->n_nodes_dev;
    if (n_slots_out) *n_slots_out = ctx->n_slots_dev;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    if (nodes_out && ctx->n_nodes_dev) {
        if (node_capacity < ctx->n_nodes_dev) return fail(RT_ERR_LIMIT, "node buffer too small");
        RT_HIP(hipMemcpy(nodes_out, ctx->d_nodes, ctx->n_nodes_dev * sizeof(rt_node32), hipMemcpyDeviceToHost));
    }
    if (slot_prim_out && ctx->n_slots_dev) {
        if (slot_capacity < ctx->n_slots_dev) return fail(RT_ERR_LIMIT, "slot buffer too small");
        RT_HIP(hipMemcpy(slot_prim_out, ctx->d_slot_prim, ctx->n_slots_dev * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

int rt_ctx_download_wide_bvh(rt_ctx* ctx, rt_node4* nodes4_out, uint32_t capacity, uint32_t* n_nodes4_out) {
    if (int rc = need_scene(ctx)) return rc;
    const uint32_t n = ctx->d_nodes4 ? ctx->n_nodes4_dev : 0u;
    if (n_nodes4_out) *n_nodes4_out = n;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    if (nodes4_out && n) {
        if (capacity < n) return fail(RT_ERR_LIMIT, "wide node buffer too small");
        RT_HIP(hipMemcpy(nodes4_out, ctx->d_nodes4, static_cast<size_t>(n) * sizeof(rt_node4), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

int rt_scene_upload(rt_ctx* ctx, const rt_scene* scene) {
    if (!scene) return fail(RT_ERR_INVALID, "null scene");
    if (int rc = use(ctx)) return rc;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->free_scene();
    ctx->build_ms = 0.0;
    ctx->n_nodes_dev = static_cast<uint32_t>(scene->nodes.size());
    int rc = RT_OK;
    if (scene->device_build && scene->n_slots > 0) {
        rc = build_on_device(ctx, scene);
    } else {
        rc = upload(scene->nodes, &ctx->d_nodes);
        if (rc == RT_OK) rc = upload(scene->recs, &ctx->d_recs);
        if (rc == RT_OK) rc = upload(scene->slot_prim, &ctx->d_slot_prim);
    }
    if (rc == RT_OK) rc = upload(scene->mats, &ctx->d_mats);
    {
        // materials whose Blinn-Phong exponent (2 / roughness^2 - 2, raytracer.cpp:247-249) is above 64: the mixed-precision
        // light term is not used for them.  With the scene's highlights mostly sharp (a sixty-fourth of the materials or
        // more) the single-pass resolve, which holds both forms, is the faster one (0.38 against 0.79 ms on the 4K frame
        // with every roughness at 0.1).
        size_t sharp = 0;
        for (const rt_mat_rec& m : scene->mats) {
            const float r = std::fmin(std::fmax(static_cast<float>(m.roughness), 0.001f), 1.0f);
            sharp += (2.0f / (r * r) - 2.0f > 64.0f) ? 1u : 0u;
        }
        ctx->two_pass_resolve = sharp * 64u < scene->mats.size();
    }
    if (rc == RT_OK) rc = upload(scene->lights, &ctx->d_lights);
    if (rc == RT_OK) rc = upload(scene->texels, &ctx->d_texels);
    if (rc == RT_OK) rc = upload(scene->tex_table, &ctx->d_tex_table);
    if (rc == RT_OK) rc = upload(scene->inst_recs, &ctx->d_inst);
    if (rc == RT_OK && !scene->device_build) rc = upload(scene->nodes4, &ctx->d_nodes4);
    if (rc == RT_OK) rc = upload(scene->loose_boxes, &ctx->d_loose_boxes);
    if (rc != RT_OK) { ctx->free_scene(); return rc; }
    DevScene& d = ctx->sc;
    d.nodes = ctx->d_nodes; d.recs = ctx->d_recs; d.mats = ctx->d_mats; d.lights = ctx->d_lights; d.gamma = ctx->d_gamma;
    d.n_nodes = ctx->n_nodes_dev;
    ctx->n_slots_dev = scene->n_slots + scene->n_loose;
    d.n_slots = static_cast<uint32_t>(scene->n_slots);
    d.n_loose = static_cast<uint32_t>(scene->n_loose);
    d.n_lights = static_cast<uint32_t>(scene->lights.size());
    d.n_prims = static_cast<uint32_t>(scene->prims.size());
    d.texels = ctx->d_texels; d.tex_table = ctx->d_tex_table;
    d.n_textures = static_cast<uint32_t>(scene->tex_table.size());
    ctx->n_nodes4_dev = !ctx->d_nodes4 ? 0u : (scene->device_build ? 2u * static_cast<uint32_t>(scene->n_slots) : static_cast<uint32_t>(scene->nodes4.size()));
    d.nodes4 = reinterpret_cast<const float4*>(ctx->d_nodes4);
    d.loose_boxes = reinterpret_cast<const float4*>(ctx->d_loose_boxes);
    d.root_ref = scene->wide_root_ref;
    for (int k = 0; k < 3; ++k) { d.root_lo[k] = 0.0f; d.root_hi[k] = 0.0f; }
    if (ctx->d_nodes4 && !scene->device_build) for (int k = 0; k < 3; ++k) { d.root_lo[k] = scene->nodes[0].lo[k]; d.root_hi[k] = scene->nodes[0].hi[k]; }
    if (ctx->d_nodes4 && scene->device_build) { for (int k = 0; k < 3; ++k) { d.root_lo[k] = ctx->dev_root_lo[k]; d.root_hi[k] = ctx->dev_root_hi[k]; } d.root_ref = 0; }
    d.inst = ctx->d_inst; d.n_inst = static_cast<uint32_t>(scene->inst_recs.size());
    // round the radius up: the padding must not shrink through the conversion
    d.radius = std::nextafterf(static_cast<float>(scene->info.radius), INFINITY);
    if (!std::isfinite(d.radius)) d.radius = 3.0e38f;
    ctx->have_scene = true;
    return RT_OK;
}

// ---- launches ------------------------------------------------------------------------------------

namespace {

// Work-queue heads of one pipeline call: kQueueSlots counters, RT_QUEUE_PARTS per persistent launch.  A call starts
// with `used` at the end, so its first launch zeroes the block: a captured or replayed sequence must not depend on
// what earlier calls left there.
struct QueueBlock {
    unsigned int* base;
    uint32_t used = kQueueSlots;
    explicit QueueBlock(unsigned int* b) : base(b) {}
};

// A zeroed work-queue head for the next persistent launch (the block of heads is re-zeroed when it runs out).
int next_queue(rt_ctx* ctx, QueueBlock& qb, unsigned int** q) {
    if (qb.used >= kQueueSlots) {
        RT_HIP(hipMemsetAsync(qb.base, 0, kQueueSlots * sizeof(unsigned int), ctx->stream));
        qb.used = 0;
    }
    *q = qb.base + qb.used;
    qb.used += RT_QUEUE_PARTS;
    return RT_OK;
}

// RT_RESOLVE_PARTS zeroed words out of the same block, for the counts of the two-pass resolve.
int take_list_heads(rt_ctx* ctx, QueueBlock& qb, unsigned int** heads) {
    static_assert(RT_RESOLVE_PARTS + 4u * RT_QUEUE_PARTS <= kQueueSlots, "a plain frame's launches and the list counts fit one block");
    if (qb.used + RT_RESOLVE_PARTS > kQueueSlots) {
        RT_HIP(hipMemsetAsync(qb.base, 0, kQueueSlots * sizeof(unsigned int), ctx->stream));
        qb.used = 0;
    }
    *heads = qb.base + qb.used;
    qb.used += RT_RESOLVE_PARTS;
    return RT_OK;
}

// Persistent waves for a launch over n_tiles tiles of 64 rays: as many as stay resident, but no more than leave every
// lane about eight rays to walk - a small launch (one rank's share of a tiled frame) otherwise spends most of its time
// in the ramp and the tail of waves that carry three or four rays per lane.
uint32_t wave_count(const rt_ctx* ctx, uint64_t n_tiles) {
    uint64_t want = n_tiles / ctx->wave_div;
    if (want < 256u) want = n_tiles < 256u ? (n_tiles ? n_tiles : 1u) : 256u;
    return static_cast<uint32_t>(want < ctx->max_waves ? want : ctx->max_waves);
}

// The frame working set of n output slots laid over a block of memory: 7 planes of doubles, four of 32-bit words.
constexpr size_t kFramePlanes = 7;
static_assert(RT_FRAME_BYTES_PER_SLOT == kFramePlanes * sizeof(double) + sizeof(int32_t) + 3 * sizeof(uint32_t), "frame planes");
size_t frame_bytes(size_t n) { return n * RT_FRAME_BYTES_PER_SLOT + 256; }
// Shadow verdicts of one batch of lights, in ray order: 64 bytes per 8x8 tile of the pixel map and light.
size_t verdict_bytes(size_t n_tiles, uint32_t n_batch) { return n_tiles * 64u * n_batch + 256; }
// Per-tile planes of the entry search: the box of the tile's hits, one entry node for its camera rays and one per light of
// a batch for its shadow rays.
size_t tile_bytes(size_t n_tiles, uint32_t n_batch) { return n_tiles * RT_TILE_BYTES(n_batch) + 256; }
void carve_tiles(void* base, size_t n_tiles, FrameBufs& fb) {
    fb.tile_box = static_cast<float*>(base);
    fb.entry_p = reinterpret_cast<float4*>(fb.tile_box + n_tiles * 8u);
    fb.entry_s = fb.entry_p + n_tiles * 8u;
}
void carve_frame(void* base, size_t n, bool textures, FrameBufs& fb) {
    fb.entry_p = nullptr; fb.entry_s = nullptr; fb.tile_box = nullptr;
    double* d = static_cast<double*>(base);
    fb.px = d; fb.py = d + n; fb.pz = d + 2*n; fb.nx = d + 3*n; fb.ny = d + 4*n; fb.nz = d + 5*n;
    fb.lighting = d + 6*n;
    fb.id = reinterpret_cast<int32_t*>(d + kFramePlanes * n);
    fb.n = n;
    fb.ray = nullptr; fb.rawn = nullptr; fb.alive = nullptr; fb.level = 0;
    fb.albedo = textures ? reinterpret_cast<uint32_t*>(fb.id + n) : nullptr;
    fb.slot = reinterpret_cast<uint32_t*>(fb.id + 2 * n); fb.inst = fb.slot + n;
    fb.lit = nullptr;
}

// The same out of the context's grow-only scratch (allocated on first use of a size).
bool wants_entry_search(const rt_ctx* ctx) { return ctx->entry_search && ctx->sc.nodes4; }
int frame_bufs(rt_ctx* ctx, size_t n, uint32_t n_batch, size_t n_tiles, FrameBufs& fb) {
    void* base = nullptr;
    const size_t fbytes = (frame_bytes(n) + 255u) & ~static_cast<size_t>(255);
    const size_t vbytes = (verdict_bytes(n_tiles, n_batch) + 255u) & ~static_cast<size_t>(255);
    // entry nodes are addressed by 32-bit byte offsets (128 bytes each): a frame with more beams goes without the search
    const bool entry = wants_entry_search(ctx) && static_cast<uint64_t>(n_tiles) * n_batch < RT_WIDE_NODE_LIMIT;
    if (int rc = ctx->need(3, fbytes + vbytes + (entry ? tile_bytes(n_tiles, n_batch) : 0), &base)) return rc;
    carve_frame(base, n, ctx->sc.n_textures != 0, fb);
    fb.lit = static_cast<uint8_t*>(base) + fbytes;
    if (entry) carve_tiles(static_cast<uint8_t*>(base) + fbytes + vbytes, n_tiles, fb);
    return RT_OK;
}

// The same frame as a chain of levels: level 0 from the camera, level k + 1 along the mirror directions of level k
// (RayTracer::trace_ray_recursive, raytracer.cpp:281-371).  Every level runs the primary / shadow kernels of the plain
// frame; a resolve kernel closes the level and leaves the next level's rays, a fold kernel blends the levels at the end.
int run_bounce_pipeline(rt_ctx* ctx, const DevCamera& dc, const PixelMap& pm, uint32_t n_waves, size_t n_out, int max_scatters,
                        double min_scatter_intensity, uint8_t* rgb_dev, int32_t* hit_id_dev) {
    const uint32_t n_lights = ctx->sc.n_lights;
    const uint32_t batch = n_lights < RT_LIGHT_BATCH ? n_lights : RT_LIGHT_BATCH;
    FrameBufs fb;
    if (int rc = frame_bufs(ctx, n_out, batch, n_waves, fb)) return rc;
    const size_t levels = static_cast<size_t>(max_scatters) + 1;
    // 9 more double planes (ray, raw normal), Fresnel weights, then the 32-bit and 8-bit planes
    const size_t bytes = n_out * (9 * sizeof(double) + max_scatters * sizeof(double) + levels * sizeof(uint32_t)
                                  + max_scatters * sizeof(uint32_t) + 2) + 256;
    void* base = nullptr;
    if (int rc = ctx->need(6, bytes, &base)) return rc;
    double* d = static_cast<double*>(base);
    fb.ray = d; fb.rawn = d + 6 * n_out;
    LevelBufs lv;
    lv.fr = d + 9 * n_out;
    lv.local = reinterpret_cast<uint32_t*>(d + (9 + static_cast<size_t>(max_scatters)) * n_out);
    lv.tint = lv.local + levels * n_out;
    lv.last = reinterpret_cast<uint8_t*>(lv.tint + static_cast<size_t>(max_scatters) * n_out);
    fb.alive = lv.last + n_out;
    if (pm.mode == 1) RT_HIP(hipMemsetAsync(fb.id, 0xFF, n_out * sizeof(int32_t), ctx->stream));
    QueueBlock qb(ctx->d_queue);
    float4* const entry_s = fb.entry_s;
    float* const tile_box = fb.tile_box;
    for (int level = 0; level <= max_scatters; ++level) {
        fb.level = level;
        // deeper levels: a tile's hits lie all over the scene, its shadow rays are no beam; their walks start at the root
        fb.entry_s = level == 0 ? entry_s : nullptr;
        fb.tile_box = level == 0 ? tile_box : nullptr;
        unsigned int* q = nullptr;
        if (int rc = next_queue(ctx, qb, &q)) return rc;
        if (fb.entry_p && level == 0) ctx->mark_begin(4);
        if (fb.entry_p && level == 0) RT_HIP(rt_launch_entry_search(ctx->stream, &ctx->sc, &dc, &pm, n_waves, 0, 0, ctx->counting, &fb, ctx->d_ctr));
        if (fb.entry_p && level == 0) ctx->mark_end();
        ctx->mark_begin(0);
        RT_HIP(rt_launch_primary(ctx->stream, &ctx->sc, &dc, &pm, n_waves, wave_count(ctx, n_waves), ctx->counting,
                                 level == 0 ? hit_id_dev : nullptr, nullptr, &fb, ctx->d_ctr, q, ctx->refill_primary()));
        ctx->mark_end();
        for (uint32_t l0 = 0; l0 < n_lights; l0 += batch) {
            const uint32_t nb = (n_lights - l0 < batch) ? n_lights - l0 : batch;
            if (int rc = next_queue(ctx, qb, &q)) return rc;
            if (fb.entry_s) ctx->mark_begin(4);
            if (fb.entry_s) RT_HIP(rt_launch_entry_search(ctx->stream, &ctx->sc, &dc, &pm, n_waves, l0, nb, ctx->counting, &fb, ctx->d_ctr));
            if (fb.entry_s) ctx->mark_end();
            ctx->mark_begin(1);
            RT_HIP(rt_launch_shadow(ctx->stream, &ctx->sc, &dc, &pm, n_waves, wave_count(ctx, static_cast<uint64_t>(n_waves) * nb), l0, nb,
                                    ctx->counting, &fb, ctx->d_ctr, q, ctx->refill()));
            ctx->mark_end();
            ctx->mark_begin(2);
            RT_HIP(rt_launch_bounce_resolve(ctx->stream, &ctx->sc, &dc, &pm, &fb, &lv, l0, nb, l0 == 0, l0 + nb >= n_lights, max_scatters,
                                            min_scatter_intensity));
            ctx->mark_end();
        }
    }
    ctx->mark_begin(3);
    RT_HIP(rt_launch_bounce_fold(ctx->stream, &ctx->sc, &lv, n_out, rgb_dev));
    ctx->mark_end();
    return RT_OK;
}

// Primary pass, shadow pass per light batch, resolve: the max_scatters = 0 frame, on a working set and a block of
// queue heads the caller provides.  Allocates nothing unless launch timing is on (events).
int run_pipeline_on(rt_ctx* ctx, const DevCamera& dc, const PixelMap& pm, uint32_t n_waves, size_t n_out,
                    uint8_t* rgb_dev, int32_t* hit_id_dev, FrameBufs& fb, unsigned int* queue_heads) {
    const uint32_t n_lights = ctx->sc.n_lights;
    const uint32_t batch = n_lights < RT_LIGHT_BATCH ? n_lights : RT_LIGHT_BATCH;
    if (pm.mode == 1)   // padded slots of edge tiles and of a short last round: no primitive
        RT_HIP(hipMemsetAsync(fb.id, 0xFF, n_out * sizeof(int32_t), ctx->stream));
    QueueBlock qb(queue_heads);
    unsigned int* q = nullptr;
    if (int rc = next_queue(ctx, qb, &q)) return rc;
    if (fb.entry_p) ctx->mark_begin(4);
    if (fb.entry_p) RT_HIP(rt_launch_entry_search(ctx->stream, &ctx->sc, &dc, &pm, n_waves, 0, 0, ctx->counting, &fb, ctx->d_ctr));
    if (fb.entry_p) ctx->mark_end();
    ctx->mark_begin(0);
    RT_HIP(rt_launch_primary(ctx->stream, &ctx->sc, &dc, &pm, n_waves, wave_count(ctx, n_waves), ctx->counting, hit_id_dev, nullptr,
                             &fb, ctx->d_ctr, q, ctx->refill_primary()));
    ctx->mark_end();
    for (uint32_t l0 = 0; l0 < n_lights; l0 += batch) {
        const uint32_t nb = (n_lights - l0 < batch) ? n_lights - l0 : batch;
        if (int rc = next_queue(ctx, qb, &q)) return rc;
        if (fb.entry_s) ctx->mark_begin(4);
        if (fb.entry_s) RT_HIP(rt_launch_entry_search(ctx->stream, &ctx->sc, &dc, &pm, n_waves, l0, nb, ctx->counting, &fb, ctx->d_ctr));
        if (fb.entry_s) ctx->mark_end();
        ctx->mark_begin(1);
        RT_HIP(rt_launch_shadow(ctx->stream, &ctx->sc, &dc, &pm, n_waves, wave_count(ctx, static_cast<uint64_t>(n_waves) * nb), l0, nb,
                                ctx->counting, &fb, ctx->d_ctr, q, ctx->refill()));
        ctx->mark_end();
        // all lights in one batch: the resolve runs in two passes and counts the slots of the second in zeroed words
        unsigned int* list_head = nullptr;
        if (l0 == 0 && nb >= n_lights && ctx->two_pass_resolve) { if (int rc = take_list_heads(ctx, qb, &list_head)) return rc; }
        ctx->mark_begin(2);
        RT_HIP(rt_launch_resolve(ctx->stream, &ctx->sc, &dc, &pm, &fb, l0, nb, l0 == 0, l0 + nb >= n_lights, rgb_dev, list_head));
        ctx->mark_end();
    }
    return RT_OK;
}

// The same up to the verdicts: which lights reach each pixel's hit (first batch of lights only).
int run_shadow_mask(rt_ctx* ctx, const DevCamera& dc, const PixelMap& pm, uint32_t n_waves, size_t n_out, uint8_t* mask_dev) {
    const uint32_t n_lights = ctx->sc.n_lights;
    FrameBufs fb;
    if (int rc = frame_bufs(ctx, n_out, n_lights, n_waves, fb)) return rc;
    QueueBlock qb(ctx->d_queue);
    unsigned int* q = nullptr;
    if (int rc = next_queue(ctx, qb, &q)) return rc;
    if (fb.entry_p) RT_HIP(rt_launch_entry_search(ctx->stream, &ctx->sc, &dc, &pm, n_waves, 0, 0, ctx->counting, &fb, ctx->d_ctr));
    RT_HIP(rt_launch_primary(ctx->stream, &ctx->sc, &dc, &pm, n_waves, wave_count(ctx, n_waves), ctx->counting, nullptr, nullptr,
                             &fb, ctx->d_ctr, q, ctx->refill_primary()));
    if (int rc = next_queue(ctx, qb, &q)) return rc;
    if (fb.entry_s) RT_HIP(rt_launch_entry_search(ctx->stream, &ctx->sc, &dc, &pm, n_waves, 0, n_lights, ctx->counting, &fb, ctx->d_ctr));
    RT_HIP(rt_launch_shadow(ctx->stream, &ctx->sc, &dc, &pm, n_waves, wave_count(ctx, static_cast<uint64_t>(n_waves) * n_lights), 0, n_lights,
                            ctx->counting, &fb, ctx->d_ctr, q, ctx->refill()));
    RT_HIP(rt_launch_verdict_mask(ctx->stream, &pm, &fb, n_lights, mask_dev));
    return RT_OK;
}

// The reference's own closest-hit loop (raytracer.cpp:143-157) on the device: no hierarchy involved.
int brute_ok(rt_ctx* ctx) {
    if (ctx->sc.inst) return fail(RT_ERR_INVALID, "brute force is not available for instanced scenes");
    return RT_OK;
}
int brute_rays(rt_ctx* ctx, const double* rays6_dev, uint64_t n, int32_t* hit_id_dev, double* dist_dev, double* records16_dev) {
    if (int rc = brute_ok(ctx)) return rc;
    DevCamera dc{}; PixelMap pm{};
    RT_HIP(rt_launch_brute(ctx->stream, &ctx->sc, &dc, &pm, rays6_dev, n, ctx->counting, hit_id_dev, dist_dev, records16_dev, ctx->d_ctr));
    return RT_OK;
}

int run_pipeline(rt_ctx* ctx, const DevCamera& dc, const PixelMap& pm, uint32_t n_waves, size_t n_out,
                 uint8_t* rgb_dev, int32_t* hit_id_dev) {
    const uint32_t n_lights = ctx->sc.n_lights;
    FrameBufs fb;
    if (int rc = frame_bufs(ctx, n_out, n_lights < RT_LIGHT_BATCH ? n_lights : RT_LIGHT_BATCH, n_waves, fb)) return rc;
    return run_pipeline_on(ctx, dc, pm, n_waves, n_out, rgb_dev, hit_id_dev, fb, ctx->d_queue);
}

} // namespace

int rt_trace_primary(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int32_t* hit_id_dev, double* dist_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam)) return fail(RT_ERR_INVALID, "bad camera (resolution must be positive)");
    PixelMap pm; uint32_t nw = 0;
    if (int rc = rect_map(cam, rect, pm, nw)) return rc;
    DevCamera dc; to_dev_camera(*cam, dc);
    if (ctx->brute_force) {
        if (int rc = brute_ok(ctx)) return rc;
        RT_HIP(rt_launch_brute(ctx->stream, &ctx->sc, &dc, &pm, nullptr, static_cast<uint64_t>(rect.w) * rect.h, ctx->counting,
                               hit_id_dev, dist_dev, nullptr, ctx->d_ctr));
        return RT_OK;
    }
    QueueBlock qb(ctx->d_queue);
    unsigned int* q = nullptr;
    if (int rc = next_queue(ctx, qb, &q)) return rc;
    RT_HIP(rt_launch_primary(ctx->stream, &ctx->sc, &dc, &pm, nw, wave_count(ctx, nw), ctx->counting, hit_id_dev, dist_dev, nullptr,
                             ctx->d_ctr, q, ctx->refill_primary()));
    return RT_OK;
}

int rt_trace_rays(rt_ctx* ctx, const double* rays6_dev, uint64_t n, int32_t* hit_id_dev, double* dist_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (n && !rays6_dev) return fail(RT_ERR_INVALID, "null rays");
    if (n > 0x0FFFFFFFull * 64) return fail(RT_ERR_LIMIT, "too many rays for one launch");
    if (ctx->brute_force) return brute_rays(ctx, rays6_dev, n, hit_id_dev, dist_dev, nullptr);
    RT_HIP(rt_launch_trace_rays(ctx->stream, &ctx->sc, rays6_dev, n, ctx->counting, 0, hit_id_dev, dist_dev, nullptr, nullptr, ctx->d_ctr));
    return RT_OK;
}

int rt_shade_rays(rt_ctx* ctx, const double* rays6_dev, uint64_t n, uint8_t* rgb_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (n && (!rays6_dev || !rgb_dev)) return fail(RT_ERR_INVALID, "null argument");
    if (n > 0x0FFFFFFFull * 64) return fail(RT_ERR_LIMIT, "too many rays for one launch");
    RT_HIP(rt_launch_trace_rays(ctx->stream, &ctx->sc, rays6_dev, n, ctx->counting, 1, nullptr, nullptr, rgb_dev, nullptr, ctx->d_ctr));
    return RT_OK;
}

int rt_trace_rays_full(rt_ctx* ctx, const double* rays6_dev, uint64_t n, double* records16_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (n && (!rays6_dev || !records16_dev)) return fail(RT_ERR_INVALID, "null argument");
    if (n > 0x0FFFFFFFull * 64) return fail(RT_ERR_LIMIT, "too many rays for one launch");
    if (ctx->brute_force) return brute_rays(ctx, rays6_dev, n, nullptr, nullptr, records16_dev);
    RT_HIP(rt_launch_trace_rays(ctx->stream, &ctx->sc, rays6_dev, n, ctx->counting, 0, nullptr, nullptr, nullptr, records16_dev, ctx->d_ctr));
    return RT_OK;
}

int rt_shade_hits(rt_ctx* ctx, const double* hits10_dev, const double* view3_dev, uint64_t n, uint8_t* rgb_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (n && (!hits10_dev || !view3_dev || !rgb_dev)) return fail(RT_ERR_INVALID, "null argument");
    if (n > 0x0FFFFFFFull * 64) return fail(RT_ERR_LIMIT, "too many hits for one launch");
    RT_HIP(rt_launch_shade_hits(ctx->stream, &ctx->sc, hits10_dev, view3_dev, n, rgb_dev, ctx->d_ctr));
    return RT_OK;
}

namespace {

// Display transfer on or off for the launches that follow (kernel arguments are copied at launch).
struct LinearOutput {
    rt_ctx* ctx;
    LinearOutput(rt_ctx* c, bool linear) : ctx(c) { if (linear) ctx->sc.gamma = ctx->d_gamma + 256; }
    ~LinearOutput() { ctx->sc.gamma = ctx->d_gamma; }
};

int render_rect(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int max_scatters, double min_scatter_intensity,
                uint8_t* rgb_dev, int32_t* hit_id_dev, uint32_t jitter_seed, bool linear);

} // namespace

int rt_render_rect(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int max_scatters, double min_scatter_intensity,
                   uint8_t* rgb_dev, int32_t* hit_id_dev) {
    return render_rect(ctx, cam, rect, max_scatters, min_scatter_intensity, rgb_dev, hit_id_dev, 0, false);
}

int rt_render_samples(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int samples_per_pixel, uint32_t seed, int max_scatters,
                      double min_scatter_intensity, uint8_t* rgb_dev) {
    if (samples_per_pixel <= 1) return render_rect(ctx, cam, rect, max_scatters, min_scatter_intensity, rgb_dev, nullptr, 0, false);
    if (samples_per_pixel > 65536) return fail(RT_ERR_LIMIT, "samples_per_pixel above 65536");
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam) || !rgb_dev || rect.w <= 0 || rect.h <= 0) return fail(RT_ERR_INVALID, "bad camera, rectangle or null output");
    const size_t n3 = static_cast<size_t>(rect.w) * rect.h * 3;
    void *d_one = nullptr, *d_sum = nullptr;
    if (int rc = ctx->need(4, n3, &d_one)) return rc;
    if (int rc = ctx->need(5, n3 * sizeof(uint32_t), &d_sum)) return rc;
    for (int s = 0; s < samples_per_pixel; ++s) {
        // sample 0 goes through the pixel centre (raytracer.cpp:408-409); later ones get a non-zero jitter seed
        uint32_t js = 0;
        if (s > 0) { js = (seed ^ 0xA511E9B3u) * 0x9E3779B1u + static_cast<uint32_t>(s) * 0x85EBCA6Bu; if (js == 0) js = 1; }
        if (int rc = render_rect(ctx, cam, rect, max_scatters, min_scatter_intensity, static_cast<uint8_t*>(d_one), nullptr, js, true)) return rc;
        RT_HIP(rt_launch_accumulate(ctx->stream, static_cast<const uint8_t*>(d_one), static_cast<uint32_t*>(d_sum), n3, s == 0));
    }
    RT_HIP(rt_launch_finish_samples(ctx->stream, static_cast<const uint32_t*>(d_sum), n3, samples_per_pixel, ctx->d_gamma, rgb_dev));
    return RT_OK;
}

namespace {

int render_rect(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int max_scatters, double min_scatter_intensity,
                uint8_t* rgb_dev, int32_t* hit_id_dev, uint32_t jitter_seed, bool linear) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam) || !rgb_dev) return fail(RT_ERR_INVALID, "bad camera or null output");
    if (max_scatters < 0) max_scatters = 0;
    if (max_scatters > RT_MAX_SCATTERS) return fail(RT_ERR_LIMIT, "max_scatters above 8");
    PixelMap pm; uint32_t nw = 0;
    if (int rc = rect_map(cam, rect, pm, nw)) return rc;
    DevCamera dc; to_dev_camera(*cam, dc);
    dc.jitter_seed = jitter_seed;
    LinearOutput transfer(ctx, linear);
    const size_t n_out = static_cast<size_t>(rect.w) * rect.h;
    if (ctx->sc.n_lights == 0) {
        // no lights: the reference writes an all-black frame without tracing (raytracer.cpp:387-395)
        RT_HIP(hipMemsetAsync(rgb_dev, 0, n_out * 3, ctx->stream));
        if (hit_id_dev) {
            QueueBlock qb(ctx->d_queue);
            unsigned int* q = nullptr;
            if (int rc = next_queue(ctx, qb, &q)) return rc;
            RT_HIP(rt_launch_primary(ctx->stream, &ctx->sc, &dc, &pm, nw, wave_count(ctx, nw), ctx->counting, hit_id_dev, nullptr, nullptr,
                                     ctx->d_ctr, q, ctx->refill_primary()));
        }
        return RT_OK;
    }
    if (max_scatters > 0) return run_bounce_pipeline(ctx, dc, pm, nw, n_out, max_scatters, min_scatter_intensity, rgb_dev, hit_id_dev);
    return run_pipeline(ctx, dc, pm, nw, n_out, rgb_dev, hit_id_dev);
}

} // namespace

int rt_trace_shadows(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, uint8_t* lit_mask_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam) || !lit_mask_dev) return fail(RT_ERR_INVALID, "bad camera or null output");
    if (ctx->sc.n_lights > RT_LIGHT_BATCH) return fail(RT_ERR_LIMIT, "rt_trace_shadows: more than 8 lights");
    PixelMap pm; uint32_t nw = 0;
    if (int rc = rect_map(cam, rect, pm, nw)) return rc;
    const size_t n_out = static_cast<size_t>(rect.w) * rect.h;
    if (ctx->sc.n_lights == 0) { RT_HIP(hipMemsetAsync(lit_mask_dev, 0, n_out, ctx->stream)); return RT_OK; }
    DevCamera dc; to_dev_camera(*cam, dc);
    return run_shadow_mask(ctx, dc, pm, nw, n_out, lit_mask_dev);
}

int rt_render_rect_static(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, uint8_t* rgb_dev, int32_t* hit_id_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam) || !rgb_dev) return fail(RT_ERR_INVALID, "bad camera or null output");
    PixelMap pm; uint32_t nw = 0;
    if (int rc = rect_map(cam, rect, pm, nw)) return rc;
    const size_t n_out = static_cast<size_t>(rect.w) * rect.h;
    const uint32_t n_lights = ctx->sc.n_lights;
    const uint32_t batch = n_lights < RT_LIGHT_BATCH ? n_lights : RT_LIGHT_BATCH;
    const size_t queue_bytes = (kQueueSlots * sizeof(unsigned int) + 255u) & ~static_cast<size_t>(255);
    const size_t fbytes = (frame_bytes(n_out) + 255u) & ~static_cast<size_t>(255);
    const size_t vbytes = (verdict_bytes(nw, batch) + 255u) & ~static_cast<size_t>(255);
    if (n_out > RT_STATIC_MAX_PIXELS)
        return fail(RT_ERR_LIMIT, "rt_render_rect_static: rectangle of " + std::to_string(n_out) + " pixels exceeds the fixed workspace (at most "
                                  + std::to_string(RT_STATIC_MAX_PIXELS) + " pixels per call); render it in pieces or use rt_render_rect");
    // the pixel map covers the rectangle with blocks of 128 x 128 pixels (256 tiles each); the verdict plane has room for
    // RT_STATIC_MAX_TILES tiles, i.e. 32 such blocks - every rectangle of the pixel limit that is not a thin strip
    if (nw > RT_STATIC_MAX_TILES || queue_bytes + fbytes + vbytes > ctx->static_scratch_bytes)
        return fail(RT_ERR_LIMIT, "rt_render_rect_static: a rectangle of " + std::to_string(rect.w) + " x " + std::to_string(rect.h)
                                  + " pixels spans " + std::to_string(nw / 256u) + " blocks of 128 x 128 pixels; the fixed workspace takes "
                                  + std::to_string(RT_STATIC_MAX_TILES / 256u) + " (render a thin strip in shorter pieces, or use rt_render_rect)");
    if (ctx->timing) return fail(RT_ERR_INVALID, "rt_render_rect_static: launch timing creates events; switch it off (rt_ctx_set_timing)");
    DevCamera dc; to_dev_camera(*cam, dc);
    if (n_lights == 0) {        // the reference's all-black frame (raytracer.cpp:387-395); ids on request
        RT_HIP(hipMemsetAsync(rgb_dev, 0, n_out * 3, ctx->stream));
        if (hit_id_dev) {
            QueueBlock qb(static_cast<unsigned int*>(ctx->static_scratch));
            unsigned int* q = nullptr;
            if (int rc = next_queue(ctx, qb, &q)) return rc;
            RT_HIP(rt_launch_primary(ctx->stream, &ctx->sc, &dc, &pm, nw, wave_count(ctx, nw), ctx->counting, hit_id_dev, nullptr, nullptr,
                                     ctx->d_ctr, q, ctx->refill_primary()));
        }
        return RT_OK;
    }
    // queue heads first (zeroed by the pipeline's first launch), then the frame planes, the verdicts, the tile planes; every plane a launch reads
    // is written earlier in the same call, so what the previous call left in the workspace is never seen
    FrameBufs fb;
    carve_frame(static_cast<uint8_t*>(ctx->static_scratch) + queue_bytes, n_out, ctx->sc.n_textures != 0, fb);
    fb.lit = static_cast<uint8_t*>(ctx->static_scratch) + queue_bytes + fbytes;
    if (wants_entry_search(ctx) && queue_bytes + fbytes + vbytes + tile_bytes(nw, batch) <= ctx->static_scratch_bytes)
        carve_tiles(static_cast<uint8_t*>(ctx->static_scratch) + queue_bytes + fbytes + vbytes, nw, fb);
    return run_pipeline_on(ctx, dc, pm, nw, n_out, rgb_dev, hit_id_dev, fb, static_cast<unsigned int*>(ctx->static_scratch));
}

int32_t rt_tile_count(rt_tiling t) {
    if (t.tile <= 0 || t.width <= 0 || t.height <= 0) return 0;
    return ((t.width + t.tile - 1) / t.tile) * ((t.height + t.tile - 1) / t.tile);
}

int32_t rt_tiles_per_rank(rt_tiling t) {
    if (t.n_ranks <= 0) return 0;
    return (rt_tile_count(t) + t.n_ranks - 1) / t.n_ranks;
}

int rt_render_tiles(rt_ctx* ctx, const rt_camera* cam, rt_tiling tiling, int rank, int max_scatters,
                    double min_scatter_intensity, uint8_t* rgb_dev) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam) || !rgb_dev) return fail(RT_ERR_INVALID, "bad camera or null output");
    if (int rc = tiling_ok(tiling)) return rc;
    if (tiling.width != cam->width || tiling.height != cam->height) return fail(RT_ERR_INVALID, "tiling size differs from the camera's film resolution");
    if (rank < 0 || rank >= tiling.n_ranks) return fail(RT_ERR_INVALID, "rank outside the tiling");
    if (max_scatters < 0) max_scatters = 0;
    if (max_scatters > RT_MAX_SCATTERS) return fail(RT_ERR_LIMIT, "max_scatters above 8");
    const int32_t per_rank = rt_tiles_per_rank(tiling);
    const size_t n_out = static_cast<size_t>(per_rank) * tiling.tile * tiling.tile;
    PixelMap pm; std::memset(&pm, 0, sizeof(pm));
    pm.mode = 1; pm.w = tiling.width; pm.h = tiling.height; pm.tile = tiling.tile;
    pm.tiles_x = (tiling.width + tiling.tile - 1) / tiling.tile;
    pm.n_tiles = rt_tile_count(tiling); pm.rank = rank; pm.n_ranks = tiling.n_ranks;
    pixel_map_divisors(pm);
    const uint32_t wpt = static_cast<uint32_t>(tiling.tile / 8) * static_cast<uint32_t>(tiling.tile / 8);
    const uint64_t nw64 = static_cast<uint64_t>(per_rank) * wpt;
    if (nw64 * 64u * RT_LIGHT_BATCH >= 0xFFFFFFFFull) return fail(RT_ERR_LIMIT, "too many tiles for one launch");
    const uint32_t nw = static_cast<uint32_t>(nw64);
    DevCamera dc; to_dev_camera(*cam, dc);
    if (ctx->sc.n_lights == 0) {
        RT_HIP(hipMemsetAsync(rgb_dev, 0, n_out * 3, ctx->stream));
        return RT_OK;
    }
    if (max_scatters > 0) return run_bounce_pipeline(ctx, dc, pm, nw, n_out, max_scatters, min_scatter_intensity, rgb_dev, nullptr);
    return run_pipeline(ctx, dc, pm, nw, n_out, rgb_dev, nullptr);
}

int rt_assemble_tiles(rt_ctx* ctx, rt_tiling tiling, const uint8_t* gathered_dev, uint8_t* frame_rgb_dev) {
    if (int rc = use(ctx)) return rc;
    if (!gathered_dev || !frame_rgb_dev) return fail(RT_ERR_INVALID, "null argument");
    if (int rc = tiling_ok(tiling)) return rc;
    RT_HIP(rt_launch_assemble(ctx->stream, tiling.width, tiling.height, tiling.tile, (tiling.width + tiling.tile - 1) / tiling.tile,
                              tiling.n_ranks, rt_tiles_per_rank(tiling), gathered_dev, frame_rgb_dev));
    return RT_OK;
}

// ---- host-memory conveniences ---------------------------------------------------------------------

int rt_render_frame_host(rt_ctx* ctx, const rt_camera* cam, int max_scatters, double min_scatter_intensity,
                         uint8_t* rgb_host, int32_t* hit_id_host) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam) || !rgb_host) return fail(RT_ERR_INVALID, "bad camera or null output");
    const size_t px = static_cast<size_t>(cam->width) * cam->height;
    void *d_rgb = nullptr, *d_ids = nullptr;
    if (int rc = ctx->need(0, px * 3, &d_rgb)) return rc;
    if (hit_id_host) if (int rc = ctx->need(1, px * sizeof(int32_t), &d_ids)) return rc;
    const rt_rect full = { 0, 0, cam->width, cam->height };
    if (int rc = rt_render_rect(ctx, cam, full, max_scatters, min_scatter_intensity, static_cast<uint8_t*>(d_rgb), static_cast<int32_t*>(d_ids))) return rc;
    RT_HIP(hipMemcpyAsync(rgb_host, d_rgb, px * 3, hipMemcpyDeviceToHost, ctx->stream));
    if (hit_id_host) RT_HIP(hipMemcpyAsync(hit_id_host, d_ids, px * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_trace_primary_host(rt_ctx* ctx, const rt_camera* cam, int32_t* hit_id_host, double* dist_host) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam)) return fail(RT_ERR_INVALID, "bad camera");
    const size_t px = static_cast<size_t>(cam->width) * cam->height;
    void *d_ids = nullptr, *d_dist = nullptr;
    if (int rc = ctx->need(1, px * sizeof(int32_t), &d_ids)) return rc;
    if (int rc = ctx->need(2, px * sizeof(double), &d_dist)) return rc;
    const rt_rect full = { 0, 0, cam->width, cam->height };
    if (int rc = rt_trace_primary(ctx, cam, full, static_cast<int32_t*>(d_ids), static_cast<double*>(d_dist))) return rc;
    if (hit_id_host) RT_HIP(hipMemcpyAsync(hit_id_host, d_ids, px * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (dist_host) RT_HIP(hipMemcpyAsync(dist_host, d_dist, px * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_trace_rays_host(rt_ctx* ctx, const double* rays6_host, uint64_t n, int32_t* hit_id_host, double* dist_host) {
    if (int rc = need_scene(ctx)) return rc;
    if (n == 0) return RT_OK;
    if (!rays6_host) return fail(RT_ERR_INVALID, "null rays");
    void *d_rays = nullptr, *d_ids = nullptr, *d_dist = nullptr;
    if (int rc = ctx->need(0, n * 6 * sizeof(double), &d_rays)) return rc;
    if (int rc = ctx->need(1, n * sizeof(int32_t), &d_ids)) return rc;
    if (int rc = ctx->need(2, n * sizeof(double), &d_dist)) return rc;
    RT_HIP(hipMemcpyAsync(d_rays, rays6_host, n * 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = rt_trace_rays(ctx, static_cast<const double*>(d_rays), n, static_cast<int32_t*>(d_ids), static_cast<double*>(d_dist))) return rc;
    if (hit_id_host) RT_HIP(hipMemcpyAsync(hit_id_host, d_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (dist_host) RT_HIP(hipMemcpyAsync(dist_host, d_dist, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_render_samples_host(rt_ctx* ctx, const rt_camera* cam, int samples_per_pixel, uint32_t seed, int max_scatters,
                           double min_scatter_intensity, uint8_t* rgb_host) {
    if (int rc = need_scene(ctx)) return rc;
    if (!camera_ok(cam) || !rgb_host) return fail(RT_ERR_INVALID, "bad camera or null output");
    const size_t px = static_cast<size_t>(cam->width) * cam->height;
    void* d_rgb = nullptr;
    if (int rc = ctx->need(0, px * 3, &d_rgb)) return rc;
    const rt_rect full = { 0, 0, cam->width, cam->height };
    if (int rc = rt_render_samples(ctx, cam, full, samples_per_pixel, seed, max_scatters, min_scatter_intensity, static_cast<uint8_t*>(d_rgb))) return rc;
    RT_HIP(hipMemcpyAsync(rgb_host, d_rgb, px * 3, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_trace_rays_full_host(rt_ctx* ctx, const double* rays6_host, uint64_t n, double* records16_host) {
    if (int rc = need_scene(ctx)) return rc;
    if (n == 0) return RT_OK;
    if (!rays6_host || !records16_host) return fail(RT_ERR_INVALID, "null argument");
    void *d_rays = nullptr, *d_rec = nullptr;
    if (int rc = ctx->need(0, n * 6 * sizeof(double), &d_rays)) return rc;
    if (int rc = ctx->need(2, n * 16 * sizeof(double), &d_rec)) return rc;
    RT_HIP(hipMemcpyAsync(d_rays, rays6_host, n * 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = rt_trace_rays_full(ctx, static_cast<const double*>(d_rays), n, static_cast<double*>(d_rec))) return rc;
    RT_HIP(hipMemcpyAsync(records16_host, d_rec, n * 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_shade_hits_host(rt_ctx* ctx, const double* hits10_host, const double* view3_host, uint64_t n, uint8_t* rgb_host) {
    if (int rc = need_scene(ctx)) return rc;
    if (n == 0) return RT_OK;
    if (!hits10_host || !view3_host || !rgb_host) return fail(RT_ERR_INVALID, "null argument");
    void *d_hits = nullptr, *d_view = nullptr, *d_rgb = nullptr;
    if (int rc = ctx->need(0, n * 10 * sizeof(double), &d_hits)) return rc;
    if (int rc = ctx->need(1, n * 3 * sizeof(double), &d_view)) return rc;
    if (int rc = ctx->need(2, n * 3, &d_rgb)) return rc;
    RT_HIP(hipMemcpyAsync(d_hits, hits10_host, n * 10 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(hipMemcpyAsync(d_view, view3_host, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = rt_shade_hits(ctx, static_cast<const double*>(d_hits), static_cast<const double*>(d_view), n, static_cast<uint8_t*>(d_rgb))) return rc;
    RT_HIP(hipMemcpyAsync(rgb_host, d_rgb, n * 3, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_intersect_prim_host(rt_ctx* ctx, const rt_prim_desc* prim, const double* rays6_host, uint64_t n, double* records16_host) {
    if (int rc = use(ctx)) return rc;
    if (!prim) return fail(RT_ERR_INVALID, "null primitive");
    if (n == 0) return RT_OK;
    if (!rays6_host || !records16_host) return fail(RT_ERR_INVALID, "null argument");
    if (n > 0x0FFFFFFFull * 64) return fail(RT_ERR_LIMIT, "too many rays for one launch");
    rt_prim_rec rec; rt_box64 box;
    if (rt_make_record(*prim, 0, rec, box) == RT_CLASS_NEVER) {
        // a primitive no ray can hit (a plane with a degenerate frame or fewer than four corners, mesh.cpp:796-838)
        for (uint64_t i = 0; i < n; ++i) {
            double* o = records16_host + i * 16;
            for (int k = 0; k < 16; ++k) o[k] = 0.0;
            o[1] = -1.0; o[2] = INFINITY;
        }
        return RT_OK;
    }
    void *d_rays = nullptr, *d_rec = nullptr, *d_out = nullptr;
    if (int rc = ctx->need(0, n * 6 * sizeof(double), &d_rays)) return rc;
    if (int rc = ctx->need(1, sizeof(rec), &d_rec)) return rc;
    if (int rc = ctx->need(2, n * 16 * sizeof(double), &d_out)) return rc;
    RT_HIP(hipMemcpyAsync(d_rays, rays6_host, n * 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(hipMemcpyAsync(d_rec, &rec, sizeof(rec), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));        // rec lives on this frame
    RT_HIP(rt_launch_intersect_one(ctx->stream, static_cast<const rt_prim_rec*>(d_rec), static_cast<const double*>(d_rays), n,
                                   static_cast<double*>(d_out)));
    RT_HIP(hipMemcpyAsync(records16_host, d_out, n * 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// ---- misc ---------------------------------------------------------------------------------------

const char* rt_strerror(int status) {
    switch (status) {
        case RT_OK: return "ok";
        case RT_ERR_INVALID: return "invalid argument";
        case RT_ERR_IO: return "cannot read file";
        case RT_ERR_NOT_FOUND: return "not found";
        case RT_ERR_NO_DEVICE: return "no usable GPU";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_NO_SCENE: return "no scene uploaded";
        case RT_ERR_LIMIT: return "limit exceeded";
        case RT_ERR_PARSE: return "malformed number in scene file";
        case RT_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}

const char* rt_last_error(void) { return g_last_error.c_str(); }
int rt_abi_version(void) { return RT_ABI_VERSION; }

} // extern "C"
