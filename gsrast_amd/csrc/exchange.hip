// Multi-GPU row-band exchange over RCCL (the single-GPU reference has nothing of the kind): one
// process per GPU, every rank renders a band of tile rows of the planar frame, one group of exact-size point-to-point
// transfers moves the bands — each rank's three contiguous pieces (rows [y0, y1) of the three colour planes) straight
// into place in its peers' frames, no staging buffer, no padding to the tallest band, no copy back. On a fully
// connected xGMI node each of the seven links of a GPU carries one peer's band.
//
// The host side is the point of this file: the same exchange through torch.distributed (batch_isend_irecv of 42
// P2POps at 8 ranks) costs hundreds of microseconds of Python and dispatcher time per frame, on frames of 0.6 ms;
// here it is one C call that issues ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd (~1 us per transfer).
//
// RCCL is loaded at run time (dlopen of the path the caller names — the copy PyTorch has already mapped — or
// librccl.so.1): libgsrast_amd.so itself does not depend on it, single-GPU users never load it.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "gsr_common.hpp"

namespace gsr {
namespace {

// The handful of RCCL entry points used, with the types of rccl.h (7.2: ncclResult_t is an enum = int; ncclComm_t an
// opaque pointer; ncclUniqueId 128 bytes passed BY VALUE; ncclFloat32 = 7).
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*GroupFn)();
typedef int (*SendFn)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*RecvFn)(void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*ErrorStringFn)(int);
constexpr int kNcclFloat32 = 7;

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GroupFn group_start = nullptr, group_end = nullptr;
    SendFn send = nullptr;
    RecvFn recv = nullptr;
    ErrorStringFn error_string = nullptr;
    char path[512] = "";
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
char g_exchange_error[256] = "";

int load_rccl(const char* path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return GSR_OK;
    const char* candidates[3] = {path && path[0] ? path : nullptr, "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* c : candidates) {
        if (!c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) { snprintf(g_rccl.path, sizeof(g_rccl.path), "%s", c); break; }
    }
    if (!h) {
        snprintf(g_exchange_error, sizeof(g_exchange_error), "dlopen(librccl): %s", dlerror());
        return GSR_ERR_HIP;
    }
    Rccl r;
    r.handle = h;
    r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
    r.group_start = reinterpret_cast<GroupFn>(dlsym(h, "ncclGroupStart"));
    r.group_end = reinterpret_cast<GroupFn>(dlsym(h, "ncclGroupEnd"));
    r.send = reinterpret_cast<SendFn>(dlsym(h, "ncclSend"));
    r.recv = reinterpret_cast<RecvFn>(dlsym(h, "ncclRecv"));
    r.error_string = reinterpret_cast<ErrorStringFn>(dlsym(h, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.group_start || !r.group_end || !r.send || !r.recv) {
        snprintf(g_exchange_error, sizeof(g_exchange_error), "librccl lacks an entry point (ncclSend / ncclRecv / ...)");
        dlclose(h);
        return GSR_ERR_HIP;
    }
    memcpy(r.path, g_rccl.path, sizeof(r.path));
    g_rccl = r;
    return GSR_OK;
}

int rccl_failed(int code, const char* what) {
    snprintf(g_exchange_error, sizeof(g_exchange_error), "%s: %s", what, g_rccl.error_string ? g_rccl.error_string(code) : "RCCL error");
    return GSR_ERR_HIP;
}
#define GSR_RCCL_TRY(call, what) do { const int c_ = (call); if (c_ != 0) return rccl_failed(c_, what); } while (0)

}  // namespace
}  // namespace gsr

using namespace gsr;

struct gsr_exchange {
    void* comm;
    int rank, world;
};

// The transfers rank `rank` of `world` issues for one frame, in issue order: fn(is_send, peer, float offset, float count).
// (Between a pair of ranks RCCL matches sends and receives of a group in order: both sides walk the colour planes in the
// same order.) root < 0: every rank ends up with the whole frame; root = r: only rank r receives.
template <typename F>
static int enumerate_transfers(int rank, int world, int width, int height, const int32_t* bounds, int root, F fn) {
    if (!bounds || width <= 0 || height <= 0 || world < 1 || rank < 0 || rank >= world || root >= world) return GSR_ERR_INVALID_ARG;
    const int grid_y = (height + kTile - 1) / kTile;
    if (bounds[0] != 0 || bounds[world] != grid_y) return GSR_ERR_INVALID_ARG;
    for (int g = 0; g < world; ++g)
        if (bounds[g + 1] < bounds[g]) return GSR_ERR_INVALID_ARG;
    auto rows = [&](int g, int& a, int& b) { a = std::min(bounds[g] * kTile, height); b = std::min(bounds[g + 1] * kTile, height); };
    const size_t plane = (size_t)width * (size_t)height;
    int y0, y1;
    rows(rank, y0, y1);
    for (int step = 1; step < world; ++step) {
        // staggered peers: in round `step` rank r sends to r + step and receives from r - step
        const int dst = (rank + step) % world, src = (rank - step + world) % world;
        int a, b;
        rows(src, a, b);
        const bool i_send = y1 > y0 && (root < 0 || dst == root);
        const bool i_recv = b > a && (root < 0 || rank == root);
        for (int c = 0; c < 3; ++c) {
            if (i_send) { const int rc = fn(true, dst, c * plane + (size_t)y0 * width, (size_t)(y1 - y0) * width); if (rc != GSR_OK) return rc; }
            if (i_recv) { const int rc = fn(false, src, c * plane + (size_t)a * width, (size_t)(b - a) * width); if (rc != GSR_OK) return rc; }
        }
    }
    return GSR_OK;
}

static int exchange_impl(gsr_exchange* x, float* frame, int width, int height, const int32_t* bounds, int root, hipStream_t stream) {
    if (!x || !frame) return GSR_ERR_INVALID_ARG;
    if (x->world == 1) return enumerate_transfers(0, 1, width, height, bounds, root, [](bool, int, size_t, size_t) -> int { return GSR_OK; });
    GSR_RCCL_TRY(g_rccl.group_start(), "ncclGroupStart");
    const int rc = enumerate_transfers(x->rank, x->world, width, height, bounds, root, [&](bool is_send, int peer, size_t off, size_t count) -> int {
        if (is_send) GSR_RCCL_TRY(g_rccl.send(frame + off, count, kNcclFloat32, peer, x->comm, stream), "ncclSend");
        else GSR_RCCL_TRY(g_rccl.recv(frame + off, count, kNcclFloat32, peer, x->comm, stream), "ncclRecv");
        return GSR_OK;
    });
    const int ce = g_rccl.group_end();                                   // (closed whatever happened inside)
    if (rc != GSR_OK) return rc;
    if (ce != 0) return rccl_failed(ce, "ncclGroupEnd");
    return GSR_OK;
}

extern "C" {

const char* gsr_exchange_last_error(void) { return g_exchange_error; }

int gsr_exchange_unique_id(const char* rccl_path, char* id128) {
    g_exchange_error[0] = 0;
    if (!id128) return record_error(GSR_ERR_INVALID_ARG);
    int rc = load_rccl(rccl_path);
    if (rc != GSR_OK) return record_error(rc);
    UniqueId id;
    const int c = g_rccl.get_unique_id(&id);
    if (c != 0) return record_error(rccl_failed(c, "ncclGetUniqueId"));
    memcpy(id128, id.internal, sizeof(id.internal));
    return record_error(GSR_OK);
}

int gsr_exchange_create(const char* rccl_path, const char* id128, int rank, int world, gsr_exchange** out) {
    g_exchange_error[0] = 0;
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return record_error(GSR_ERR_INVALID_ARG);
    *out = nullptr;
    int rc = load_rccl(rccl_path);
    if (rc != GSR_OK) return record_error(rc);
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    const int c = g_rccl.comm_init_rank(&comm, world, id, rank);       // collective: every rank calls, on its own device
    if (c != 0) return record_error(rccl_failed(c, "ncclCommInitRank"));
    gsr_exchange* x = new gsr_exchange;
    x->comm = comm; x->rank = rank; x->world = world;
    *out = x;
    return record_error(GSR_OK);
}

int gsr_exchange_destroy(gsr_exchange* x) {
    if (!x) return GSR_OK;
    int rc = GSR_OK;
    if (x->comm && g_rccl.comm_destroy) {
        const int c = g_rccl.comm_destroy(x->comm);
        if (c != 0) rc = rccl_failed(c, "ncclCommDestroy");
    }
    delete x;
    return record_error(rc);
}

// The plan of gsr_exchange_bands without a communicator (host only): what rank `rank` would issue. Returns the number of
// transfers (<= 6 (world - 1)), or -GSR_ERR_INVALID_ARG; fills up to max_ops entries of each array that is not NULL.
int gsr_exchange_plan(int rank, int world, int width, int height, const int32_t* bounds, int root, int max_ops,
                      int32_t* is_send, int32_t* peer, uint64_t* offset, uint64_t* count) {
    int n = 0;
    const int rc = enumerate_transfers(rank, world, width, height, bounds, root, [&](bool s_, int p_, size_t off, size_t cnt) -> int {
        if (n < max_ops) {
            if (is_send) is_send[n] = s_ ? 1 : 0;
            if (peer) peer[n] = p_;
            if (offset) offset[n] = off;
            if (count) count[n] = cnt;
        }
        ++n;
        return GSR_OK;
    });
    return rc == GSR_OK ? n : -rc;
}

// A rank's transfer to itself: `planes` pieces of `count` floats from src to dst through ncclSend / ncclRecv in one group,
// the shape of a peer's band. What a one-GPU box can run of the transfer path (RCCL matches a send to the own rank with
// the receive from it inside a group).
int gsr_exchange_loopback(gsr_exchange* x, const float* src, float* dst, uint64_t count, int planes, void* stream) {
    g_exchange_error[0] = 0;
    if (!x || !src || !dst || planes < 1) return record_error(GSR_ERR_INVALID_ARG);
    auto run = [&]() -> int {
        GSR_RCCL_TRY(g_rccl.group_start(), "ncclGroupStart");
        int rc = GSR_OK;
        for (int c = 0; c < planes && rc == GSR_OK; ++c) {
            int e = g_rccl.send(src + (size_t)c * count, (size_t)count, kNcclFloat32, x->rank, x->comm, (hipStream_t)stream);
            if (e == 0) e = g_rccl.recv(dst + (size_t)c * count, (size_t)count, kNcclFloat32, x->rank, x->comm, (hipStream_t)stream);
            if (e != 0) rc = rccl_failed(e, "ncclSend / ncclRecv to the own rank");
        }
        const int ce = g_rccl.group_end();
        if (rc != GSR_OK) return rc;
        if (ce != 0) return rccl_failed(ce, "ncclGroupEnd");
        return GSR_OK;
    };
    return record_error(run());
}

int gsr_exchange_bands(gsr_exchange* x, float* frame, int width, int height, const int32_t* bounds, int root, void* stream) {
    g_exchange_error[0] = 0;
    return record_error(exchange_impl(x, frame, width, height, bounds, root, (hipStream_t)stream));
}

}  // extern "C"
