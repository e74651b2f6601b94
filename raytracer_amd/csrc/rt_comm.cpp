// Multi-GPU tile gather of the C ABI (include/rt_hip.h): RCCL point-to-point over xGMI.
//
// The reference is a single-threaded CPU program with no communication of any kind; this is the one exchange the tiled
// frame needs.  Every peer sends its compact tile buffer straight to rank 0 (each GPU has its own xGMI link to GPU 0, so
// the transfers do not share a link: no ring, no reduction), rank 0 posts the matching receives in one group and then
// scatters the gathered tiles into the frame with the assemble kernel, all on the context's stream.
//
// RCCL is bound at first use with dlopen (the name every ROCm install and the torch wheels carry, "librccl.so.1"), so the
// library loads - and everything single-GPU works - where RCCL is absent, and a process that already holds a copy of
// RCCL (torch.distributed) shares it instead of loading a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_scene.h"

int rt_fail(int code, const std::string& msg);          // rt_capi.cpp: sets rt_last_error()
hipStream_t rt_ctx_stream(rt_ctx* ctx);
int rt_ctx_device(rt_ctx* ctx);
int rt_ctx_scratch(rt_ctx* ctx, int slot, size_t bytes, void** out);

namespace {

// the slice of rccl.h this file needs (types are ABI-stable across NCCL 2.x / RCCL)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                // ncclSuccess = 0
constexpr int kNcclUint8 = 1;                            // ncclUint8 / ncclChar family: ncclInt8 = 0, ncclUint8 = 1

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RT_RCCL_LIBRARY names the one file to load instead (a differently named build; tests use it to force a miss)
        const char* forced = std::getenv("RT_RCCL_LIBRARY");
        std::string last;                                // dlerror() hands a message out once and then forgets it: keep a copy
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            r.handle = dlopen(forced ? forced : name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            if (const char* m = dlerror()) last = m;
            if (forced) break;
        }
        if (!r.handle) { r.error = std::string("RCCL not found: ") + (last.empty() ? std::string("dlopen failed") : last); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int need_rccl(Rccl** out) {
    Rccl* r = rccl();
    if (!r->error.empty()) return rt_fail(RT_ERR_COMM, r->error);
    *out = r;
    return RT_OK;
}

#define RT_NCCL(r, call)                                                                                       \
    do {                                                                                                       \
        const ncclResult_t e_ = (call);                                                                        \
        if (e_ != 0) return rt_fail(RT_ERR_COMM, std::string(#call) + ": " + (r)->GetErrorString(e_));         \
    } while (0)

size_t rank_bytes(rt_tiling t) {
    return static_cast<size_t>(rt_tiles_per_rank(t)) * static_cast<size_t>(t.tile) * static_cast<size_t>(t.tile) * 3u;
}

} // namespace

struct rt_comm {
    rt_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;      // null when n_ranks == 1 was asked for without RCCL (never: RCCL is used for n = 1 too)
    int n_ranks = 1, rank = 0;
};

extern "C" {

int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) {
    if (!id) return rt_fail(RT_ERR_INVALID, "null argument");
    Rccl* r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    ncclUniqueId u;
    RT_NCCL(r, r->GetUniqueId(&u));
    static_assert(sizeof(u) == RT_COMM_ID_BYTES, "unique id size");
    std::memcpy(id, &u, sizeof(u));
    return RT_OK;
}

int rt_comm_create(rt_ctx* ctx, int n_ranks, int rank, const uint8_t id[RT_COMM_ID_BYTES], rt_comm** out) {
    if (!ctx || !id || !out) return rt_fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return rt_fail(RT_ERR_INVALID, "rank outside [0, n_ranks)");
    Rccl* r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    if (hipSetDevice(rt_ctx_device(ctx)) != hipSuccess) return rt_fail(RT_ERR_HIP, "hipSetDevice failed");
    rt_comm* c = new (std::nothrow) rt_comm();
    if (!c) return rt_fail(RT_ERR_LIMIT, "out of memory");
    c->ctx = ctx; c->n_ranks = n_ranks; c->rank = rank;
    ncclUniqueId u; std::memcpy(&u, id, sizeof(u));
    const ncclResult_t e = r->CommInitRank(&c->comm, n_ranks, u, rank);
    if (e != 0) { delete c; return rt_fail(RT_ERR_COMM, std::string("ncclCommInitRank: ") + r->GetErrorString(e)); }
    *out = c;
    return RT_OK;
}

int rt_comm_create_all(rt_ctx** ctxs, int n, rt_comm** comms_out) {
    if (!ctxs || !comms_out || n < 1) return rt_fail(RT_ERR_INVALID, "null argument or no contexts");
    for (int k = 0; k < n; ++k) { comms_out[k] = nullptr; if (!ctxs[k]) return rt_fail(RT_ERR_INVALID, "null context"); }
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (rt_ctx_device(ctxs[a]) == rt_ctx_device(ctxs[b])) return rt_fail(RT_ERR_INVALID, "two contexts on one GPU: a communicator needs one GPU per rank");
    Rccl* r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n, nullptr);
    for (int k = 0; k < n; ++k) devs[k] = rt_ctx_device(ctxs[k]);
    RT_NCCL(r, r->CommInitAll(comms.data(), n, devs.data()));
    for (int k = 0; k < n; ++k) {
        rt_comm* c = new (std::nothrow) rt_comm();
        if (!c) {
            for (int j = 0; j < n; ++j) { if (comms[j]) (void)r->CommDestroy(comms[j]); delete comms_out[j]; comms_out[j] = nullptr; }
            return rt_fail(RT_ERR_LIMIT, "out of memory");
        }
        c->ctx = ctxs[k]; c->comm = comms[k]; c->n_ranks = n; c->rank = k;
        comms_out[k] = c;
    }
    return RT_OK;
}

int rt_comm_destroy(rt_comm* comm) {
    if (!comm) return RT_OK;
    if (comm->comm) {
        Rccl* r = rccl();
        (void)hipSetDevice(rt_ctx_device(comm->ctx));
        (void)hipStreamSynchronize(rt_ctx_stream(comm->ctx));
        if (r->CommDestroy) (void)r->CommDestroy(comm->comm);
    }
    delete comm;
    return RT_OK;
}

// One rank's part of the gather; the caller brackets several of these with a group when it drives several ranks.
static int gather_part(Rccl* r, rt_comm* c, rt_tiling tiling, const uint8_t* tiles_dev, uint8_t* frame_rgb_dev, uint8_t** gathered_out) {
    const size_t per = rank_bytes(tiling);
    hipStream_t st = rt_ctx_stream(c->ctx);
    if (hipSetDevice(rt_ctx_device(c->ctx)) != hipSuccess) return rt_fail(RT_ERR_HIP, "hipSetDevice failed");
    if (c->rank != 0) {
        RT_NCCL(r, r->Send(tiles_dev, per, kNcclUint8, 0, c->comm, st));
        return RT_OK;
    }
    if (!frame_rgb_dev) return rt_fail(RT_ERR_INVALID, "rank 0 needs the frame buffer");
    void* gathered = nullptr;
    if (int rc = rt_ctx_scratch(c->ctx, 7, per * static_cast<size_t>(c->n_ranks), &gathered)) return rc;
    *gathered_out = static_cast<uint8_t*>(gathered);
    for (int peer = 1; peer < c->n_ranks; ++peer)
        RT_NCCL(r, r->Recv(*gathered_out + per * static_cast<size_t>(peer), per, kNcclUint8, peer, c->comm, st));
    return RT_OK;
}

static int finish_rank0(rt_comm* c, rt_tiling tiling, const uint8_t* tiles_dev, uint8_t* gathered, uint8_t* frame_rgb_dev) {
    const size_t per = rank_bytes(tiling);
    hipStream_t st = rt_ctx_stream(c->ctx);
    if (hipSetDevice(rt_ctx_device(c->ctx)) != hipSuccess) return rt_fail(RT_ERR_HIP, "hipSetDevice failed");
    if (hipMemcpyAsync(gathered, tiles_dev, per, hipMemcpyDeviceToDevice, st) != hipSuccess) return rt_fail(RT_ERR_HIP, "hipMemcpyAsync failed");
    return rt_assemble_tiles(c->ctx, tiling, gathered, frame_rgb_dev);
}

static int gather_args_ok(const rt_comm* c, rt_tiling tiling, const uint8_t* tiles_dev) {
    if (!c || !tiles_dev) return rt_fail(RT_ERR_INVALID, "null argument");
    if (tiling.n_ranks != c->n_ranks) return rt_fail(RT_ERR_INVALID, "tiling.n_ranks differs from the communicator's size");
    if (tiling.width <= 0 || tiling.height <= 0 || tiling.tile < 16 || (tiling.tile % 16) != 0 || tiling.tile > 1024)
        return rt_fail(RT_ERR_INVALID, "tiling: tile must be a multiple of 16 in [16, 1024], sizes positive");
    return RT_OK;
}

int rt_gather_tiles(rt_comm* comm, rt_tiling tiling, const uint8_t* tiles_dev, uint8_t* frame_rgb_dev) {
    if (int rc = gather_args_ok(comm, tiling, tiles_dev)) return rc;
    Rccl* r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    uint8_t* gathered = nullptr;
    RT_NCCL(r, r->GroupStart());
    const int rc = gather_part(r, comm, tiling, tiles_dev, frame_rgb_dev, &gathered);
    const ncclResult_t e = r->GroupEnd();          // always closed, also after a failure inside the group
    if (rc != RT_OK) return rc;
    if (e != 0) return rt_fail(RT_ERR_COMM, std::string("ncclGroupEnd: ") + r->GetErrorString(e));
    return comm->rank == 0 ? finish_rank0(comm, tiling, tiles_dev, gathered, frame_rgb_dev) : RT_OK;
}

int rt_gather_tiles_all(rt_comm** comms, int n, rt_tiling tiling, const uint8_t* const* tiles_dev, uint8_t* frame_rgb_dev) {
    if (!comms || !tiles_dev || n < 1) return rt_fail(RT_ERR_INVALID, "null argument");
    for (int k = 0; k < n; ++k) {
        if (int rc = gather_args_ok(comms[k], tiling, tiles_dev[k])) return rc;
        if (comms[k]->rank != k || comms[k]->n_ranks != n) return rt_fail(RT_ERR_INVALID, "communicators must be passed in rank order, all of them");
    }
    Rccl* r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    uint8_t* gathered = nullptr;
    int rc = RT_OK;
    RT_NCCL(r, r->GroupStart());
    for (int k = 0; k < n && rc == RT_OK; ++k) rc = gather_part(r, comms[k], tiling, tiles_dev[k], frame_rgb_dev, &gathered);
    const ncclResult_t e = r->GroupEnd();
    if (rc != RT_OK) return rc;
    if (e != 0) return rt_fail(RT_ERR_COMM, std::string("ncclGroupEnd: ") + r->GetErrorString(e));
    return finish_rank0(comms[0], tiling, tiles_dev[0], gathered, frame_rgb_dev);
}

int rt_render_frame_multi_host(rt_comm** comms, int n, const rt_camera* cam, int tile, int max_scatters, double min_scatter_intensity,
                               uint8_t* rgb_host) {
    if (!comms || n < 1 || !cam || !rgb_host) return rt_fail(RT_ERR_INVALID, "null argument");
    for (int k = 0; k < n; ++k) if (!comms[k] || comms[k]->rank != k || comms[k]->n_ranks != n) return rt_fail(RT_ERR_INVALID, "communicators must be passed in rank order, all of them");
    const rt_tiling tiling = { cam->width, cam->height, tile, n };
    const size_t per = rank_bytes(tiling);
    std::vector<const uint8_t*> bufs(n, nullptr);
    // every GPU renders its share (the launches are asynchronous: the GPUs work side by side) ...
    for (int k = 0; k < n; ++k) {
        void* local = nullptr;
        if (int rc = rt_ctx_scratch(comms[k]->ctx, 0, per, &local)) return rc;
        if (int rc = rt_render_tiles(comms[k]->ctx, cam, tiling, k, max_scatters, min_scatter_intensity, static_cast<uint8_t*>(local))) return rc;
        bufs[k] = static_cast<const uint8_t*>(local);
    }
    // ... the tiles travel to rank 0, which scatters them into the frame and hands it to the host
    void* frame = nullptr;
    const size_t frame_bytes = static_cast<size_t>(cam->width) * static_cast<size_t>(cam->height) * 3u;
    if (int rc = rt_ctx_scratch(comms[0]->ctx, 2, frame_bytes, &frame)) return rc;
    if (int rc = rt_gather_tiles_all(comms, n, tiling, bufs.data(), static_cast<uint8_t*>(frame))) return rc;
    if (hipSetDevice(rt_ctx_device(comms[0]->ctx)) != hipSuccess) return rt_fail(RT_ERR_HIP, "hipSetDevice failed");
    if (hipMemcpyAsync(rgb_host, frame, frame_bytes, hipMemcpyDeviceToHost, rt_ctx_stream(comms[0]->ctx)) != hipSuccess)
        return rt_fail(RT_ERR_HIP, "hipMemcpyAsync failed");
    for (int k = n - 1; k >= 0; --k) if (int rc = rt_ctx_synchronize(comms[k]->ctx)) return rc;
    return RT_OK;
}

} // extern "C"
