// group_api.cpp — x-slab tiles of one cube over the GPUs of a node (include/thzgpu.h, "Multi-GPU"
// section): member contexts, the two exchange steps of the path as RCCL calls on the members' own
// streams, and a session whose slabs are recomputed side by side.
//
// librccl is opened with dlopen when a group with more than one device is created, so the library
// has no load-time dependency on it (a single-GPU user never maps its 570 MB).  The function
// prototypes come from <rccl/rccl.h>; only the symbols are looked up at run time.
#include "session.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

using namespace thz;

namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string err;
};

Rccl &rccl()
{
    static Rccl r;
    return r;
}

bool rccl_load()
{
    Rccl &r = rccl();
    if (r.h) return true;
    // THZ_RCCL_LIB: developer knob — the library to open in RCCL's place (tests/mock_rccl: several ranks on ONE GPU)
    if (const char *override_path = getenv("THZ_RCCL_LIB")) {
        // never silent: a release process whose collectives go through something else than librccl says so
        fprintf(stderr, "[thzgpu] THZ_RCCL_LIB is set: opening %s in place of librccl (test infrastructure)\n", override_path);
        r.h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
    } else
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
        }
    if (!r.h) {
        r.err = std::string("cannot open librccl: ") + dlerror();
        return false;
    }
#define THZ_SYM(field, sym)                                                  \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, #sym));         \
    if (!r.field) {                                                          \
        r.err = "librccl lacks " #sym;                                       \
        dlclose(r.h);                                                        \
        r.h = nullptr;                                                       \
        return false;                                                        \
    }
    THZ_SYM(GetUniqueId, ncclGetUniqueId)
    THZ_SYM(CommInitRank, ncclCommInitRank)
    THZ_SYM(CommInitAll, ncclCommInitAll)
    THZ_SYM(CommDestroy, ncclCommDestroy)
    THZ_SYM(GetErrorString, ncclGetErrorString)
    THZ_SYM(AllReduce, ncclAllReduce)
    THZ_SYM(Broadcast, ncclBroadcast)
    THZ_SYM(Send, ncclSend)
    THZ_SYM(Recv, ncclRecv)
    THZ_SYM(GroupStart, ncclGroupStart)
    THZ_SYM(GroupEnd, ncclGroupEnd)
#undef THZ_SYM
    return true;
}

}  // namespace

struct thz_group {
    struct Member {
        thz_ctx *ctx = nullptr;
        int rank = 0;
        ncclComm_t comm = nullptr;
        hipEvent_t ev = nullptr;  // same-device groups: orders the members' streams around a collective
    };
    std::vector<Member> m;
    int world = 0;
    bool same_device = false;  // one process, every member on one device: no fabric, device-local copies
    std::string err;
};

struct thz_group_session {
    thz_group *g = nullptr;
    size_t nx = 0, ny = 0, nt = 0;
    std::vector<thz_session *> sess;   // one per local member
    std::vector<size_t> x0, rows;      // one per rank
    int root_local = -1;               // index of rank 0 among the local members, -1: another process has it
    // gathered copies on rank 0's device, allocated when first asked for
    float *d_img = nullptr, *d_data = nullptr, *d_fft = nullptr, *d_amp = nullptr, *d_ph = nullptr;
    size_t cap_img = 0, cap_data = 0, cap_fft = 0, cap_amp = 0, cap_ph = 0;  // floats allocated
    std::vector<size_t> cur_rows;      // rows of the outputs' grid per rank (the block grid behind a scaling stage)
    size_t cur_ny = 0;
    size_t cur_pix() const
    {
        size_t r = 0;
        for (size_t v : cur_rows) r += v;
        return r * cur_ny;
    }
    size_t nt_out = 0;
    int gathered = -1;  // thz_gather level of the last recompute
};

namespace {

int gfail(thz_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    return code;
}

#define NCCL_TRY(g, expr)                                                                            \
    do {                                                                                             \
        ncclResult_t r_ = (expr);                                                                    \
        if (r_ != ncclSuccess) return gfail(g, THZ_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)
#define GHIP_TRY(g, expr)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return gfail(g, THZ_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// same-device groups: stream 0 waits for everything the other members have enqueued ...
int join_on_first(thz_group *g)
{
    for (size_t i = 1; i < g->m.size(); ++i) {
        GHIP_TRY(g, hipEventRecord(g->m[i].ev, g->m[i].ctx->stream));
        GHIP_TRY(g, hipStreamWaitEvent(g->m[0].ctx->stream, g->m[i].ev, 0));
    }
    return THZ_OK;
}
// ... and the others wait for what stream 0 did meanwhile
int fan_out_from_first(thz_group *g)
{
    GHIP_TRY(g, hipEventRecord(g->m[0].ev, g->m[0].ctx->stream));
    for (size_t i = 1; i < g->m.size(); ++i) GHIP_TRY(g, hipStreamWaitEvent(g->m[i].ctx->stream, g->m[0].ev, 0));
    return THZ_OK;
}

template <class T>
int all_reduce(thz_group *g, T *const *d_bufs, size_t count, ncclDataType_t type)
{
    if (!g || !d_bufs) return THZ_ERR_INVALID;
    if (count == 0 || (g->world == 1 && !g->m[0].comm)) return THZ_OK;
    for (size_t i = 0; i < g->m.size(); ++i)
        if (!d_bufs[i]) return gfail(g, THZ_ERR_INVALID, "all-reduce: null buffer");
    if (g->same_device) {
        GHIP_TRY(g, hipSetDevice(g->m[0].ctx->device));
        if (int rc = join_on_first(g)) return rc;
        hipStream_t st = g->m[0].ctx->stream;
        for (size_t i = 1; i < g->m.size(); ++i) {
            if constexpr (sizeof(T) == 4) launch_add_vec(st, (float *)d_bufs[0], (const float *)d_bufs[i], count);
            else launch_add_u64(st, (unsigned long long *)d_bufs[0], (const unsigned long long *)d_bufs[i], count);
        }
        GHIP_TRY(g, hipGetLastError());
        for (size_t i = 1; i < g->m.size(); ++i)
            GHIP_TRY(g, hipMemcpyAsync(d_bufs[i], d_bufs[0], count * sizeof(T), hipMemcpyDeviceToDevice, st));
        return fan_out_from_first(g);
    }
    Rccl &r = rccl();
    NCCL_TRY(g, r.GroupStart());
    for (size_t i = 0; i < g->m.size(); ++i) {
        (void)hipSetDevice(g->m[i].ctx->device);  // the communicator's device is current while its call is enqueued
        ncclResult_t rc = r.AllReduce(d_bufs[i], d_bufs[i], count, type, ncclSum, g->m[i].comm, g->m[i].ctx->stream);
        if (rc != ncclSuccess) {
            (void)r.GroupEnd();
            return gfail(g, THZ_ERR_HIP, std::string("ncclAllReduce: ") + r.GetErrorString(rc));
        }
    }
    NCCL_TRY(g, r.GroupEnd());
    return THZ_OK;
}

int make_member(thz_group *g, int device, int rank)
{
    thz_group::Member mb;
    if (int rc = thz_create(device, &mb.ctx)) return gfail(g, rc, "thz_create(" + std::to_string(device) + ") failed");
    mb.rank = rank;
    if (hipEventCreateWithFlags(&mb.ev, hipEventDisableTiming) != hipSuccess) {
        thz_destroy(mb.ctx);
        return gfail(g, THZ_ERR_HIP, "hipEventCreate failed");
    }
    g->m.push_back(mb);
    return THZ_OK;
}

}  // namespace

extern "C" {

int thz_host_slab(size_t nx, int world, int rank, size_t *x0, size_t *n)
{
    if (world < 1 || rank < 0 || rank >= world) return THZ_ERR_INVALID;
    const size_t base = nx / (size_t)world, rem = nx % (size_t)world, r = (size_t)rank;
    if (n) *n = base + (r < rem ? 1 : 0);
    if (x0) *x0 = r * base + (r < rem ? r : rem);
    return THZ_OK;
}

// g == NULL: why the last thz_group_create* / thz_group_unique_id could not load RCCL (there is no group to ask then)
const char *thz_group_last_error(const thz_group *g) { return g ? g->err.c_str() : (rccl().err.empty() ? "null group" : rccl().err.c_str()); }
int thz_group_world(const thz_group *g) { return g ? g->world : 0; }
int thz_group_local_count(const thz_group *g) { return g ? (int)g->m.size() : 0; }
int thz_group_rank(const thz_group *g, int i) { return (g && i >= 0 && i < (int)g->m.size()) ? g->m[i].rank : -1; }
thz_ctx *thz_group_ctx(thz_group *g, int i) { return (g && i >= 0 && i < (int)g->m.size()) ? g->m[i].ctx : nullptr; }

void thz_group_destroy(thz_group *g)
{
    if (!g) return;
    for (auto &mb : g->m) {
        if (mb.ctx) {
            (void)hipSetDevice(mb.ctx->device);
            (void)hipStreamSynchronize(mb.ctx->stream);
        }
        if (mb.comm) (void)rccl().CommDestroy(mb.comm);
        if (mb.ev) (void)hipEventDestroy(mb.ev);
        if (mb.ctx) thz_destroy(mb.ctx);
    }
    delete g;
}

int thz_group_create(const int *devices, int n, thz_group **out)
{
    if (!out || !devices || n < 1) return THZ_ERR_INVALID;
    *out = nullptr;
    bool all_same = true, distinct = true;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            if (devices[i] == devices[j]) distinct = false;
            else all_same = false;
        }
    if (n > 1 && !all_same && !distinct) return THZ_ERR_INVALID;
    thz_group *g = new thz_group();
    g->world = n;
    g->same_device = n > 1 && all_same;
    for (int i = 0; i < n; ++i)
        if (int rc = make_member(g, devices[i], i)) {
            thz_group_destroy(g);
            return rc;
        }
    if (n > 1 && distinct) {
        if (!rccl_load()) {
            thz_group_destroy(g);
            return THZ_ERR_HIP;
        }
        std::vector<ncclComm_t> comms((size_t)n);
        const ncclResult_t rc = rccl().CommInitAll(comms.data(), n, devices);
        if (rc != ncclSuccess) {
            thz_group_destroy(g);
            return THZ_ERR_HIP;
        }
        for (int i = 0; i < n; ++i) g->m[(size_t)i].comm = comms[(size_t)i];
    }
    *out = g;
    return THZ_OK;
}

int thz_group_unique_id(void *id)
{
    if (!id) return THZ_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == THZ_GROUP_ID_BYTES, "THZ_GROUP_ID_BYTES");
    if (!rccl_load()) return THZ_ERR_HIP;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return THZ_ERR_HIP;
    std::memcpy(id, &u, sizeof u);
    return THZ_OK;
}

int thz_group_create_rank(int device, int rank, int world, const void *id, thz_group **out)
{
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return THZ_ERR_INVALID;
    *out = nullptr;
    thz_group *g = new thz_group();
    g->world = world;
    if (int rc = make_member(g, device, rank)) {
        thz_group_destroy(g);
        return rc;
    }
    // THZ_GROUP_FORCE_RCCL: a single-rank group still opens librccl, builds its communicator and sends its
    // collectives through it — how the RCCL entry points are exercised on a one-GPU box (tests/test_gpu_group.py)
    if (world > 1 || (id && getenv("THZ_GROUP_FORCE_RCCL"))) {
        if (!rccl_load()) {
            thz_group_destroy(g);
            return THZ_ERR_HIP;
        }
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        if (hipSetDevice(device) != hipSuccess || rccl().CommInitRank(&g->m[0].comm, world, u, rank) != ncclSuccess) {
            g->m[0].comm = nullptr;
            thz_group_destroy(g);
            return THZ_ERR_HIP;
        }
    }
    *out = g;
    return THZ_OK;
}

int thz_group_all_reduce_sum(thz_group *g, float *const *d_bufs, size_t count) { return all_reduce<float>(g, d_bufs, count, ncclFloat); }
int thz_group_all_reduce_u64(thz_group *g, uint64_t *const *d_bufs, size_t count) { return all_reduce<uint64_t>(g, d_bufs, count, ncclUint64); }

int thz_group_gather(thz_group *g, const float *const *d_send, const size_t *counts, float *d_recv_root)
{
    if (!g || !d_send || !counts) return THZ_ERR_INVALID;
    std::vector<size_t> off((size_t)g->world + 1, 0);
    for (int q = 0; q < g->world; ++q) off[(size_t)q + 1] = off[(size_t)q] + counts[q];
    int root = -1;
    for (size_t i = 0; i < g->m.size(); ++i)
        if (g->m[i].rank == 0) root = (int)i;
    if (root >= 0 && !d_recv_root) return gfail(g, THZ_ERR_INVALID, "gather: rank 0 needs a receive buffer");
    if (g->same_device) {
        GHIP_TRY(g, hipSetDevice(g->m[0].ctx->device));
        if (int rc = join_on_first(g)) return rc;
        for (size_t i = 0; i < g->m.size(); ++i) {
            const size_t q = (size_t)g->m[i].rank;
            if (counts[q] && d_recv_root + off[q] != d_send[i])
                GHIP_TRY(g, hipMemcpyAsync(d_recv_root + off[q], d_send[i], counts[q] * sizeof(float), hipMemcpyDeviceToDevice,
                                           g->m[0].ctx->stream));
        }
        return fan_out_from_first(g);
    }
    if (root >= 0 && counts[0] && d_recv_root != d_send[root]) {  // rank 0's own rows: a device-local copy
        GHIP_TRY(g, hipSetDevice(g->m[(size_t)root].ctx->device));
        GHIP_TRY(g, hipMemcpyAsync(d_recv_root, d_send[root], counts[0] * sizeof(float), hipMemcpyDeviceToDevice,
                                   g->m[(size_t)root].ctx->stream));
    }
    if (g->world == 1 && !g->m[0].comm) return THZ_OK;
    Rccl &r = rccl();
    NCCL_TRY(g, r.GroupStart());
    ncclResult_t rc = ncclSuccess;
    for (size_t i = 0; i < g->m.size() && rc == ncclSuccess; ++i) {
        const size_t q = (size_t)g->m[i].rank;
        (void)hipSetDevice(g->m[i].ctx->device);
        if (q != 0 && counts[q]) rc = r.Send(d_send[i], counts[q], ncclFloat, 0, g->m[i].comm, g->m[i].ctx->stream);
    }
    if (root >= 0) (void)hipSetDevice(g->m[(size_t)root].ctx->device);
    if (root >= 0)
        for (int q = 1; q < g->world && rc == ncclSuccess; ++q)
            if (counts[q])
                rc = r.Recv(d_recv_root + off[(size_t)q], counts[q], ncclFloat, q, g->m[(size_t)root].comm, g->m[(size_t)root].ctx->stream);
    if (rc != ncclSuccess) {
        (void)r.GroupEnd();
        return gfail(g, THZ_ERR_HIP, std::string("ncclSend / ncclRecv: ") + r.GetErrorString(rc));
    }
    NCCL_TRY(g, r.GroupEnd());
    return THZ_OK;
}

int thz_group_sync(thz_group *g)
{
    if (!g) return THZ_ERR_INVALID;
    for (auto &mb : g->m) {
        GHIP_TRY(g, hipSetDevice(mb.ctx->device));
        GHIP_TRY(g, hipStreamSynchronize(mb.ctx->stream));
    }
    return THZ_OK;
}

/* ------------------------------------------------------------------ group session */

void thz_group_session_destroy(thz_group_session *gs)
{
    if (!gs) return;
    for (thz_session *s : gs->sess) thz_session_destroy(s);
    if (gs->root_local >= 0) {
        (void)hipSetDevice(gs->g->m[(size_t)gs->root_local].ctx->device);
        for (float *p : {gs->d_img, gs->d_data, gs->d_fft, gs->d_amp, gs->d_ph})
            if (p) (void)hipFree(p);
    }
    delete gs;
}

int thz_group_session_create(thz_group *g, size_t nx, size_t ny, size_t nt, const float *time, float dx, float dy,
                             thz_group_session **out)
{
    if (!g || !out || !time || ny == 0 || nt < 2) return THZ_ERR_INVALID;
    *out = nullptr;
    if (nx < (size_t)g->world) return gfail(g, THZ_ERR_INVALID, "fewer x rows than ranks: every slab needs at least one row");
    thz_group_session *gs = new thz_group_session();
    gs->g = g; gs->nx = nx; gs->ny = ny; gs->nt = nt; gs->nt_out = nt;
    gs->x0.resize((size_t)g->world);
    gs->rows.resize((size_t)g->world);
    for (int q = 0; q < g->world; ++q) (void)thz_host_slab(nx, g->world, q, &gs->x0[(size_t)q], &gs->rows[(size_t)q]);
    for (size_t i = 0; i < g->m.size(); ++i) {
        thz_session *s = nullptr;
        const int rc = thz_session_create(g->m[i].ctx, gs->rows[(size_t)g->m[i].rank], ny, nt, time, dx, dy, &s);
        if (rc) {
            gfail(g, rc, std::string("slab session: ") + thz_last_error(g->m[i].ctx));
            thz_group_session_destroy(gs);
            return rc;
        }
        // where the slab sits in the whole grid: the Tilt plan, block means over slab edges and the regions of
        // interest depend on it (session_enqueue derives the current grid's placement from these)
        s->raw_grid_x0 = gs->x0[(size_t)g->m[i].rank];
        s->raw_grid_rows = nx;
        s->slab_rank = g->m[i].rank;
        s->slab_world = g->world;
        s->grid_x0 = s->raw_grid_x0;
        s->grid_rows = nx;
        gs->sess.push_back(s);
        if (g->m[i].rank == 0) gs->root_local = (int)i;
    }
    gs->cur_rows = gs->rows;
    gs->cur_ny = ny;
    if (gs->root_local >= 0) {
        if (hipSetDevice(g->m[(size_t)gs->root_local].ctx->device) != hipSuccess
            || hipMalloc((void **)&gs->d_img, nx * ny * sizeof(float)) != hipSuccess) {
            gfail(g, THZ_ERR_HIP, "gathered image: allocation failed");
            thz_group_session_destroy(gs);
            return THZ_ERR_HIP;
        }
        gs->cap_img = nx * ny;
    }
    *out = gs;
    return THZ_OK;
}

thz_session *thz_group_session_member(thz_group_session *gs, int i)
{
    return (gs && i >= 0 && i < (int)gs->sess.size()) ? gs->sess[(size_t)i] : nullptr;
}

int thz_group_session_upload(thz_group_session *gs, const float *cube, int subtract_bias)
{
    if (!gs) return THZ_ERR_INVALID;
    thz_group *g = gs->g;
    std::vector<float *> sums;
    for (size_t i = 0; i < gs->sess.size(); ++i) {
        const size_t q = (size_t)g->m[i].rank;
        const int rc = thz_session_upload(gs->sess[i], cube ? cube + gs->x0[q] * gs->ny * gs->nt : nullptr, subtract_bias);
        if (rc) return gfail(g, rc, std::string("slab upload: ") + thz_last_error(g->m[i].ctx));
        sums.push_back(gs->sess[i]->d_rawsum);
    }
    // the slabs' raw pixel sums become the cube's: avg_fft of every later recompute follows from them
    if (int rc = thz_group_all_reduce_sum(g, sums.data(), gs->nt)) return rc;
    gs->gathered = -1;
    return thz_group_sync(g);
}

// rank `from` -> rank `to`: src / dst are indexed by LOCAL member; only the members that hold the two ranks act
static int group_p2p(thz_group *g, int from, int to, const float *const *d_src, float *const *d_dst, size_t count)
{
    if (count == 0 || from == to) return THZ_OK;
    int lf = -1, lt = -1;
    for (size_t i = 0; i < g->m.size(); ++i) {
        if (g->m[i].rank == from) lf = (int)i;
        if (g->m[i].rank == to) lt = (int)i;
    }
    if (lf < 0 && lt < 0) return THZ_OK;
    if (g->same_device || (lf >= 0 && lt >= 0 && !g->m[(size_t)lf].comm)) {
        // one process, no fabric: a device-local copy on the receiver's stream behind the sender's work
        GHIP_TRY(g, hipSetDevice(g->m[(size_t)lt].ctx->device));
        GHIP_TRY(g, hipEventRecord(g->m[(size_t)lf].ev, g->m[(size_t)lf].ctx->stream));
        GHIP_TRY(g, hipStreamWaitEvent(g->m[(size_t)lt].ctx->stream, g->m[(size_t)lf].ev, 0));
        GHIP_TRY(g, hipMemcpyAsync(d_dst[lt], d_src[lf], count * sizeof(float), hipMemcpyDeviceToDevice, g->m[(size_t)lt].ctx->stream));
        // ... and the sender's stream behind the copy, as a send on its own stream would be: the sender may write its
        // buffer again right away (the carried means reuse one running-sum buffer for all three arrays)
        GHIP_TRY(g, hipEventRecord(g->m[(size_t)lt].ev, g->m[(size_t)lt].ctx->stream));
        GHIP_TRY(g, hipStreamWaitEvent(g->m[(size_t)lf].ctx->stream, g->m[(size_t)lt].ev, 0));
        return THZ_OK;
    }
    Rccl &r = rccl();
    NCCL_TRY(g, r.GroupStart());
    ncclResult_t rc = ncclSuccess;
    if (lf >= 0) {
        (void)hipSetDevice(g->m[(size_t)lf].ctx->device);
        rc = r.Send(d_src[lf], count, ncclFloat, to, g->m[(size_t)lf].comm, g->m[(size_t)lf].ctx->stream);
    }
    if (lt >= 0 && rc == ncclSuccess) {
        (void)hipSetDevice(g->m[(size_t)lt].ctx->device);
        rc = r.Recv(d_dst[lt], count, ncclFloat, from, g->m[(size_t)lt].comm, g->m[(size_t)lt].ctx->stream);
    }
    if (rc != ncclSuccess) {
        (void)r.GroupEnd();
        return gfail(g, THZ_ERR_HIP, std::string("ncclSend / ncclRecv: ") + r.GetErrorString(rc));
    }
    NCCL_TRY(g, r.GroupEnd());
    return THZ_OK;
}

// want_means == 2 over several slabs: the reference's means are SEQUENTIAL sums over all x rows (ndarray mean_axis on
// axis 0, then on the next: math_tools.rs:421-440), so the slabs take turns in rank order — each continues the running
// sums of the slabs in front of it (k_sum_axis0's carry) and hands them on; the last one divides by nx, sums over y,
// divides by ny; the result reaches every member through an all-reduce in which all others add zeros.  Bit for bit
// one session's means; serial by construction — this mode is for comparing against the reference, not for speed.
static int group_means_reference_order(thz_group_session *gs)
{
    thz_group *g = gs->g;
    const size_t nl = gs->sess.size(), nt = gs->nt_out, nf = nt / 2 + 1, ny = gs->cur_ny;
    size_t nx_total = 0;
    for (size_t v : gs->cur_rows) nx_total += v;
    std::vector<float *> run(nl, nullptr), avg(nl, nullptr);
    std::vector<const float *> run_c(nl, nullptr);
    auto cleanup = [&]() {
        for (size_t i = 0; i < nl; ++i)
            if (run[i]) {
                (void)hipSetDevice(g->m[i].ctx->device);
                (void)hipStreamSynchronize(g->m[i].ctx->stream);
                (void)hipFree(run[i]);
            }
    };
    for (size_t i = 0; i < nl; ++i) {
        GHIP_TRY(g, hipSetDevice(g->m[i].ctx->device));
        if (hipMalloc((void **)&run[i], ny * 2 * nf * sizeof(float)) != hipSuccess) {
            cleanup();
            return gfail(g, THZ_ERR_HIP, "reference-order means: allocation failed");
        }
        run_c[i] = run[i];
        avg[i] = gs->sess[i]->d_avg;
        if (hipMemsetAsync(avg[i], 0, 4 * nf * sizeof(float), g->m[i].ctx->stream) != hipSuccess) { cleanup(); return gfail(g, THZ_ERR_HIP, "memset"); }
    }
    struct Arr { size_t L, off; int which; };
    const Arr arrs[3] = {{2 * nf, 0, 0}, {nf, 2 * nf, 1}, {nf, 3 * nf, 2}};
    int rc = THZ_OK;
    for (const Arr &a : arrs) {
        for (int q = 0; q < g->world && !rc; ++q) {
            if (q > 0) rc = group_p2p(g, q - 1, q, run_c.data(), run.data(), ny * a.L);
            for (size_t i = 0; i < nl && !rc; ++i) {
                if (g->m[i].rank != q) continue;
                thz_session *s = gs->sess[i];
                const float *arr = a.which == 0 ? s->d_fft : (a.which == 1 ? s->d_amp : s->d_ph);
                if (hipSetDevice(g->m[i].ctx->device) != hipSuccess) { rc = THZ_ERR_HIP; break; }
                const bool last = q == g->world - 1;
                launch_sum_axis0(g->m[i].ctx->stream, arr, gs->cur_rows[(size_t)q], ny * a.L, last ? (float)nx_total : 0.0f, run[i], q > 0 ? run[i] : nullptr);
                if (last) launch_sum_axis0(g->m[i].ctx->stream, run[i], ny, a.L, (float)ny, avg[i] + a.off);
                if (hipGetLastError() != hipSuccess) rc = THZ_ERR_HIP;
            }
        }
        if (rc) break;
    }
    if (!rc) rc = thz_group_all_reduce_sum(g, avg.data(), 4 * nf);  // everybody but the last rank holds zeros
    cleanup();
    if (rc) return gfail(g, rc, "reference-order means over the slabs failed");
    for (thz_session *s : gs->sess) s->have_means = true;
    return THZ_OK;
}

// all-reduce of the members' region sums (session_roi.cpp's block layout: the final traces' block is the last
// R x nt floats — the only one a tail-only recompute or the Deconvolution stage renews)
static int group_roi_reduce(thz_group_session *gs, std::vector<float *> &bufs, bool data_only)
{
    thz_session *s0 = gs->sess[0];
    const size_t total = session_roi_floats(s0), fin = s0->rois.size() * s0->nt_out, spec = 2 * s0->rois.size() * s0->nf_out;
    std::vector<float *> tail(bufs);
    for (float *&b : tail) b += total - fin;
    if (data_only) return thz_group_all_reduce_sum(gs->g, tail.data(), fin);
    if (s0->roi_src_fresh) return thz_group_all_reduce_sum(gs->g, bufs.data(), total);
    // the source traces' block in the middle holds the grid's sums of an earlier recompute: left alone
    if (int rc = thz_group_all_reduce_sum(gs->g, bufs.data(), spec)) return rc;
    return thz_group_all_reduce_sum(gs->g, tail.data(), fin);
}

int thz_group_session_set_rois(thz_group_session *gs, size_t n_rois, const size_t *n_vertices, const uint64_t *poly_xy)
{
    if (!gs) return THZ_ERR_INVALID;
    thz_group *g = gs->g;
    for (size_t i = 0; i < gs->sess.size(); ++i) {
        const int rc = thz_session_set_rois(gs->sess[i], n_rois, n_vertices, poly_xy);
        if (rc) return gfail(g, rc, std::string("slab regions of interest: ") + thz_last_error(g->m[i].ctx));
    }
    return THZ_OK;
}

int thz_group_session_roi(thz_group_session *gs, size_t roi, const thz_roi_out *out)
{
    if (!gs || gs->sess.empty()) return THZ_ERR_INVALID;
    const int rc = thz_session_roi(gs->sess[0], roi, out);
    if (rc) return gfail(gs->g, rc, std::string("thz_group_session_roi: ") + thz_last_error(gs->g->m[0].ctx));
    return rc;
}

int thz_group_session_recompute(thz_group_session *gs, const thz_chain_cfg *cfg, int start_stage, int gather)
{
    if (!gs || !cfg) return THZ_ERR_INVALID;
    thz_group *g = gs->g;
    if (gather < THZ_GATHER_SMALL || gather > THZ_GATHER_ALL || start_stage < 0 || start_stage > 8)
        return gfail(g, THZ_ERR_INVALID, "thz_group_session_recompute: bad gather level or chain position");
    const bool single = g->world == 1;  // one slab = the whole grid
    if (start_stage == 8) return THZ_OK;
    // Scaling over slab edges (round 3): a block's rows may lie in two slabs.  Every slab sums the first rows of the
    // block it cannot finish and hands the partial sums to the next slab, which continues the sequence — the block
    // belongs to the slab that holds its LAST row.  Needed only when the walk re-runs the scaling stage.
    const size_t sf = cfg->scale_factor > 1 ? (size_t)cfg->scale_factor : 1;
    if (!single && sf > 1) {
        std::vector<const float *> out_c;
        std::vector<float *> in;
        for (size_t i = 0; i < gs->sess.size(); ++i) {
            const int rc = session_scale_tail(gs->sess[i], cfg);
            if (rc) return gfail(g, rc, std::string("scaling over slabs: ") + thz_last_error(g->m[i].ctx));
            out_c.push_back(gs->sess[i]->d_carry_out);
            in.push_back(gs->sess[i]->d_carry_in);
        }
        const size_t count = (gs->ny / sf) * gs->nt;
        for (int q = 0; q + 1 < g->world; ++q)
            if (slab_scale(gs->nx, g->world, q, sf).tail)
                if (int rc = group_p2p(g, q, q + 1, out_c.data(), in.data(), count)) return rc;
    }
    // every slab's chain, enqueued side by side on the members' streams
    std::vector<char> tail(gs->sess.size(), 0);
    for (size_t i = 0; i < gs->sess.size(); ++i) {
        bool t = false;
        const int rc = session_enqueue(gs->sess[i], cfg, start_stage, &t);
        if (rc) return gfail(g, rc, std::string("slab recompute: ") + thz_last_error(g->m[i].ctx));
        tail[i] = t ? 1 : 0;
    }
    const size_t nt_out = gs->sess.empty() ? gs->nt : gs->sess[0]->nt_out, nf = nt_out / 2 + 1;
    gs->nt_out = nt_out;
    // the outputs' grid: the raw one, or — one slab, scaled — the session's block grid
    gs->cur_rows = gs->rows;
    gs->cur_ny = gs->ny;
    if (!gs->sess.empty() && gs->sess[0]->scale > 1) {
        const size_t s_eff = gs->sess[0]->scale;
        gs->cur_ny = gs->sess[0]->ny_cur;
        for (int q = 0; q < g->world; ++q) gs->cur_rows[(size_t)q] = single ? gs->sess[0]->nx_cur : slab_scale(gs->nx, g->world, q, s_eff).rows;
    }
    // C2: the slabs' undivided amplitude / phase sums -> the cube's, on every member
    const bool additive = !gs->sess.empty() && (gs->sess[0]->msum_fast || gs->sess[0]->msum_passes);
    if (cfg->want_means && !(tail.size() && tail[0]) && !additive && single) {
        // the whole grid in one slab, means in the reference's order: nothing to exchange
        int rc = session_means(gs->sess[0], cfg, gs->sess[0]->nx_cur * gs->sess[0]->ny_cur);
        if (!rc) rc = session_avg_data(gs->sess[0], cfg);
        if (rc) return gfail(g, rc, std::string("slab means: ") + thz_last_error(g->m[0].ctx));
    } else if (cfg->want_means && !(tail.size() && tail[0]) && !additive) {
        if (int rc = group_means_reference_order(gs)) return rc;
        for (size_t i = 0; i < gs->sess.size(); ++i)
            if (int rc = session_avg_data(gs->sess[i], cfg)) return gfail(g, rc, std::string("slab means: ") + thz_last_error(g->m[i].ctx));
    } else if (cfg->want_means && !(tail.size() && tail[0])) {
        std::vector<float *> bufs;
        // amplitude / phase sums of the fused launch (2 nf), or — a tilted cube — spectrum, amplitude and phase sums
        // (4 nf).  The sum of the SOURCE traces in front of them (avg_fft follows from it by linearity) was all-reduced
        // at upload for the raw cube; a block-averaged source's is the slab's own and goes along.
        thz_session *s0 = gs->sess[0];
        const bool src_too = s0->msum_fast && s0->d_src != s0->d_raw;
        for (thz_session *s : gs->sess) bufs.push_back(src_too ? s->d_msum : s->d_msum + nt_out);
        const size_t count = (s0->msum_passes ? 4 * nf : 2 * nf) + (src_too ? nt_out : 0);
        if (int rc = thz_group_all_reduce_sum(g, bufs.data(), count)) return rc;
        for (size_t i = 0; i < gs->sess.size(); ++i) {
            // Σ of the raw traces was all-reduced at upload; the copy in d_msum[0, nt) is already the cube's
            int rc = session_means(gs->sess[i], cfg, gs->cur_pix());
            if (!rc) rc = session_avg_data(gs->sess[i], cfg);
            if (rc) return gfail(g, rc, std::string("slab means: ") + thz_last_error(g->m[i].ctx));
        }
    }
    // C2, second part: the regions of interest's masked sums (every slab's rows of the whole grid's mask)
    if (!gs->sess.empty() && !gs->sess[0]->rois.empty()) {
        bool data_only = tail.size() && tail[0];  // (the members agree: same regions, same history)
        std::vector<float *> bufs;
        for (size_t i = 0; i < gs->sess.size(); ++i) {
            const int rc = session_roi_sums(gs->sess[i], cfg, &data_only);
            if (rc) return gfail(g, rc, std::string("slab regions of interest: ") + thz_last_error(g->m[i].ctx));
            bufs.push_back(gs->sess[i]->d_roi_sum);
        }
        if (int rc = group_roi_reduce(gs, bufs, data_only)) return rc;
        for (size_t i = 0; i < gs->sess.size(); ++i) {
            const int rc = session_roi_finish(gs->sess[i], cfg, data_only);
            if (rc) return gfail(g, rc, std::string("slab regions of interest: ") + thz_last_error(g->m[i].ctx));
        }
    }
    // C1: per-pixel results to rank 0
    auto gather_buf = [&](int which, size_t per_pix, float **d_dst, size_t *cap) -> int {
        std::vector<const float *> send;
        std::vector<size_t> counts((size_t)g->world);
        for (int q = 0; q < g->world; ++q) counts[(size_t)q] = gs->cur_rows[(size_t)q] * gs->cur_ny * per_pix;
        for (thz_session *s : gs->sess) send.push_back(static_cast<const float *>(thz_session_buffer(s, which)));
        const size_t need = gs->cur_pix() * per_pix;
        if (gs->root_local >= 0 && (!*d_dst || *cap < need)) {  // (a tilted cube's outputs are longer than the raw traces)
            GHIP_TRY(g, hipSetDevice(g->m[(size_t)gs->root_local].ctx->device));
            if (*d_dst) {
                GHIP_TRY(g, hipStreamSynchronize(g->m[(size_t)gs->root_local].ctx->stream));
                GHIP_TRY(g, hipFree(*d_dst));
                *d_dst = nullptr;
                *cap = 0;
            }
            GHIP_TRY(g, hipMalloc((void **)d_dst, need * sizeof(float)));
            *cap = need;
        }
        return thz_group_gather(g, send.data(), counts.data(), *d_dst);
    };
    if (int rc = gather_buf(THZ_BUF_IMG, 1, &gs->d_img, &gs->cap_img)) return rc;
    if (gather >= THZ_GATHER_TIME)
        if (int rc = gather_buf(THZ_BUF_DATA, nt_out, &gs->d_data, &gs->cap_data)) return rc;
    if (gather >= THZ_GATHER_ALL) {
        if (int rc = gather_buf(THZ_BUF_FFT, 2 * nf, &gs->d_fft, &gs->cap_fft)) return rc;
        if (int rc = gather_buf(THZ_BUF_AMPLITUDES, nf, &gs->d_amp, &gs->cap_amp)) return rc;
        if (int rc = gather_buf(THZ_BUF_PHASES, nf, &gs->d_ph, &gs->cap_ph)) return rc;
    }
    gs->gathered = gather;
    return thz_group_sync(g);
}

// every member ends with all ranks' rows: d_send[i] (counts[rank_i] floats) -> d_recv[i] + offset(rank), in rank order
static int group_all_gather(thz_group *g, const float *const *d_send, const size_t *counts, float *const *d_recv)
{
    std::vector<size_t> off((size_t)g->world + 1, 0);
    for (int q = 0; q < g->world; ++q) off[(size_t)q + 1] = off[(size_t)q] + counts[q];
    if (g->same_device || (g->world == 1 && !g->m[0].comm)) {
        GHIP_TRY(g, hipSetDevice(g->m[0].ctx->device));
        if (g->world > 1)
            if (int rc = join_on_first(g)) return rc;
        for (size_t i = 0; i < g->m.size(); ++i)
            for (size_t k = 0; k < g->m.size(); ++k) {
                const size_t q = (size_t)g->m[k].rank;
                if (counts[q])
                    GHIP_TRY(g, hipMemcpyAsync(d_recv[i] + off[q], d_send[k], counts[q] * sizeof(float), hipMemcpyDeviceToDevice,
                                               g->m[0].ctx->stream));
            }
        return g->world > 1 ? fan_out_from_first(g) : THZ_OK;
    }
    Rccl &r = rccl();
    NCCL_TRY(g, r.GroupStart());
    ncclResult_t rc = ncclSuccess;
    for (size_t i = 0; i < g->m.size() && rc == ncclSuccess; ++i)
        for (int q = 0; q < g->world && rc == ncclSuccess; ++q)
            if (counts[q] && hipSetDevice(g->m[i].ctx->device) == hipSuccess)
                rc = r.Broadcast(g->m[i].rank == q ? d_send[i] : d_recv[i] + off[(size_t)q], d_recv[i] + off[(size_t)q], counts[q], ncclFloat, q,
                                 g->m[i].comm, g->m[i].ctx->stream);
    if (rc != ncclSuccess) {
        (void)r.GroupEnd();
        return gfail(g, THZ_ERR_HIP, std::string("ncclBroadcast: ") + r.GetErrorString(rc));
    }
    NCCL_TRY(g, r.GroupEnd());
    return THZ_OK;
}

int thz_group_session_deconvolve(thz_group_session *gs, const thz_psf *psf, const thz_deconv_cfg *cfg,
                                 volatile const int *abort_flag, float *progress)
{
    if (!gs || !psf || !cfg) return THZ_ERR_INVALID;
    thz_group *g = gs->g;
    const size_t nl = gs->sess.size();
    for (thz_session *s : gs->sess)
        if (!s->have_outputs) return gfail(g, THZ_ERR_NOT_READY, "thz_group_session_deconvolve: no recompute has run");
    if (g->world == 1 && !g->m[0].comm) {
        // one slab = the whole grid: the session's own stage (no gather of the cube, no second copy of it), then C1
        const int rc = thz_session_deconvolve(gs->sess[0], psf, cfg, abort_flag, progress);
        if (rc < 0) return gfail(g, rc, std::string("thz_group_session_deconvolve: ") + thz_last_error(g->m[0].ctx));
        thz_session *s = gs->sess[0];
        const size_t npix = s->nx_cur * s->ny_cur;
        GHIP_TRY(g, hipSetDevice(g->m[0].ctx->device));
        GHIP_TRY(g, hipMemcpyAsync(gs->d_img, thz_session_buffer(s, THZ_BUF_IMG), npix * sizeof(float), hipMemcpyDeviceToDevice, g->m[0].ctx->stream));
        if (gs->gathered >= THZ_GATHER_TIME && gs->d_data)
            GHIP_TRY(g, hipMemcpyAsync(gs->d_data, thz_session_buffer(s, THZ_BUF_DATA), npix * gs->nt_out * sizeof(float), hipMemcpyDeviceToDevice,
                                       g->m[0].ctx->stream));
        if (int rc2 = thz_group_sync(g)) return rc2;
        return rc;
    }
    // ---- several slabs (round 3).  The stage has three parts with two different independences: the transform, the band
    // energies and the recombination are per PIXEL (every band), the Richardson-Lucy iterations are per BAND (the
    // whole image).  So every member does the per-pixel parts for its own rows and the iterations for its own bands,
    // and what crosses the fabric is two sets of 2-D images — n_filters x Nx x Ny energies out, as many gains back —
    // instead of round 2's all-gather and all-reduce of the whole cube (2 x Nx Ny Nt floats per member, and two
    // whole-cube buffers on every GPU): SURVEY 8e's alternative.
    //   A  thz_dc_slab_energies      own rows, every band                      -> E_slab [nb][npix_slab]
    //   X1 all-gather of the E_slab blocks; a member keeps its bands' images    -> E_mine [bands][npix]
    //   B  thz_dc_band_gains          own bands (dealt out by cost), whole grid -> G_mine [bands][npix]
    //   X2 all-gather of the G_mine blocks (rank order = band order); a member keeps its rows' columns -> G_slab
    //   C  thz_dc_slab_combine        own rows, every band                      -> the slab of the stage's output
    // A guard (the same on every rank) or an abort / error on any rank makes every slab keep its input.
    std::vector<size_t> cur_x0((size_t)g->world, 0);
    for (int q = 1; q < g->world; ++q) cur_x0[(size_t)q] = cur_x0[(size_t)q - 1] + gs->cur_rows[(size_t)q - 1];
    const size_t grid_ny = gs->cur_ny, npix_all = gs->cur_pix(), grid_nx = npix_all / (grid_ny ? grid_ny : 1);
    const size_t nt = gs->nt_out;
    const size_t nb = cfg->n_filters;
    for (size_t i = 0; i < nl; ++i) {  // the members' engines on the chain's current axis
        thz_ctx *ctx = g->m[i].ctx;
        thz_session *s = gs->sess[i];
        GHIP_TRY(g, hipSetDevice(ctx->device));
        if (ctx->time.size() != nt || std::memcmp(ctx->time.data(), s->time_out.data(), nt * sizeof(float)) != 0)
            if (int rc = thz_set_time_axis(ctx, s->time_out.data(), nt)) return gfail(g, rc, thz_last_error(ctx));
    }
    // The bands' iteration counts and tile counts (host arithmetic, identical on every rank) -> contiguous band ranges.
    // A range's iterations run as chains of dependent launches: its time is about alpha x (its longest band's
    // iterations) + beta x (sum of iterations x tiles) — alpha = 13.2 us per iteration (two launches end to end),
    // beta = 20 ns per tile and iteration, fitted to the widest band alone at 128 x 128 and 512 x 512 pixels
    // (profiles/r03_deconv_group_estimate.txt).  The ranges minimise the slowest rank's time (dynamic programme over
    // the cut points; a rank may stay without a band when there are more ranks than bands).
    std::vector<double> costs;
    int status = THZ_OK;
    bool skipped = false;
    {
        // (the reference's guards — no bands, a grid smaller than the widest PSF ... — are rank-independent too)
        const int rc = thz_dc_band_costs(g->m[0].ctx, psf, cfg, grid_nx, grid_ny, gs->sess[0]->dx_cur, gs->sess[0]->dy_cur, &costs);
        if (rc < 0) return gfail(g, rc, std::string("thz_group_session_deconvolve: ") + thz_last_error(g->m[0].ctx));
        skipped = rc == THZ_SKIPPED || costs.size() != 2 * nb;
    }
    std::vector<size_t> band0((size_t)g->world + 1, 0);
    if (!skipped) {
        const size_t W = (size_t)g->world;
        const double alpha = 13.2, beta = 0.0204;
        auto range_cost = [&](size_t a, size_t b) {  // bands [a, b)
            double it = 0.0, w = 0.0;
            for (size_t k = a; k < b; ++k) {
                it = std::max(it, costs[2 * k]);
                w += costs[2 * k + 1];
            }
            return alpha * it + beta * w;
        };
        // best[q][b]: the smallest possible slowest-rank time when ranks 0 .. q-1 share the bands [0, b)
        std::vector<std::vector<double>> best(W + 1, std::vector<double>(nb + 1, 1e300));
        std::vector<std::vector<size_t>> cut(W + 1, std::vector<size_t>(nb + 1, 0));
        best[0][0] = 0.0;
        for (size_t q = 1; q <= W; ++q)
            for (size_t b = 0; b <= nb; ++b)
                for (size_t a = 0; a <= b; ++a) {
                    if (best[q - 1][a] >= 1e300) continue;
                    const double v = std::max(best[q - 1][a], range_cost(a, b));
                    if (v < best[q][b]) { best[q][b] = v; cut[q][b] = a; }
                }
        size_t b = nb;
        for (size_t q = W; q >= 1; --q) {
            band0[q] = b;
            b = cut[q][b];
        }
        band0[0] = 0;
    }
    std::vector<float *> bufA(nl, nullptr), bufB(nl, nullptr), bufC(nl, nullptr), bufD(nl, nullptr), flag(nl, nullptr);
    auto cleanup = [&]() {
        for (size_t i = 0; i < nl; ++i) {
            (void)hipSetDevice(g->m[i].ctx->device);
            (void)hipStreamSynchronize(g->m[i].ctx->stream);
            for (float *p : {bufA[i], bufB[i], bufC[i], bufD[i], flag[i]})
                if (p) (void)hipFree(p);
        }
    };
    // a HIP error past this point frees the exchange buffers before it returns
#define GHIP_TRY_C(g, expr)                                                                          \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            cleanup();                                                                               \
            return gfail(g, THZ_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
        }                                                                                            \
    } while (0)
    auto rank_pix = [&](size_t q) { return gs->cur_rows[q] * grid_ny; };
    // every member's slab output buffers (the stage's result replaces the final cube / image until the next recompute)
    for (size_t i = 0; i < nl; ++i) {
        thz_session *s = gs->sess[i];
        GHIP_TRY_C(g, hipSetDevice(g->m[i].ctx->device));
        const size_t q = (size_t)g->m[i].rank, n = rank_pix(q) * nt;
        if (s->deconv_floats != n) {
            if (s->d_deconv) { (void)hipFree(s->d_deconv); s->d_deconv = nullptr; }
            if (s->d_deconv_img) { (void)hipFree(s->d_deconv_img); s->d_deconv_img = nullptr; }
            s->deconv_floats = 0;
            if (hipMalloc((void **)&s->d_deconv, n * sizeof(float)) != hipSuccess
                || hipMalloc((void **)&s->d_deconv_img, rank_pix(q) * sizeof(float)) != hipSuccess) {
                cleanup();
                return gfail(g, THZ_ERR_HIP, "thz_group_session_deconvolve: slab allocation failed");
            }
            s->deconv_floats = n;
        }
        if (hipMalloc((void **)&flag[i], sizeof(float)) != hipSuccess) { cleanup(); return gfail(g, THZ_ERR_HIP, "thz_group_session_deconvolve: allocation failed"); }
    }
    // runs fn(member) for every local member side by side (the calls wait for their streams)
    auto for_members = [&](const std::function<int(size_t)> &fn, std::vector<int> &rcs) {
        rcs.assign(nl, THZ_OK);
        if (nl == 1) { rcs[0] = fn(0); return; }
        std::vector<std::thread> th;
        for (size_t i = 0; i < nl; ++i) th.emplace_back([&, i]() { rcs[i] = fn(i); });
        for (auto &t : th) t.join();
    };
    // one float per member, summed over the group: did anybody fail?
    auto any_bad = [&](const std::vector<int> &rcs, bool *bad) -> int {
        for (size_t i = 0; i < nl; ++i) {
            GHIP_TRY_C(g, hipSetDevice(g->m[i].ctx->device));
            const float v = rcs[i] < 0 ? 1.0f : 0.0f;
            GHIP_TRY_C(g, hipMemcpyAsync(flag[i], &v, sizeof v, hipMemcpyHostToDevice, g->m[i].ctx->stream));
            GHIP_TRY_C(g, hipStreamSynchronize(g->m[i].ctx->stream));
        }
        if (int rc = thz_group_all_reduce_sum(g, flag.data(), 1)) { cleanup(); return rc; }
        float v = 0.0f;
        GHIP_TRY_C(g, hipSetDevice(g->m[0].ctx->device));
        GHIP_TRY_C(g, hipMemcpyAsync(&v, flag[0], sizeof v, hipMemcpyDeviceToHost, g->m[0].ctx->stream));
        GHIP_TRY_C(g, hipStreamSynchronize(g->m[0].ctx->stream));
        *bad = v != 0.0f;
        return THZ_OK;
    };
    std::vector<int> rcs;
    bool bad = false;
    if (!skipped) {
        for (size_t i = 0; i < nl; ++i) {
            GHIP_TRY_C(g, hipSetDevice(g->m[i].ctx->device));
            const size_t q = (size_t)g->m[i].rank, nbs = band0[q + 1] - band0[q];
            if (hipMalloc((void **)&bufA[i], std::max<size_t>(nb * rank_pix(q), 4) * sizeof(float)) != hipSuccess
                || hipMalloc((void **)&bufB[i], std::max<size_t>(nb * npix_all, 4) * sizeof(float)) != hipSuccess
                || hipMalloc((void **)&bufC[i], std::max<size_t>(nbs * npix_all, 4) * sizeof(float)) != hipSuccess
                || hipMalloc((void **)&bufD[i], std::max<size_t>(nbs * npix_all, 4) * sizeof(float)) != hipSuccess) {
                cleanup();
                return gfail(g, THZ_ERR_HIP, "thz_group_session_deconvolve: allocation of the band images failed");
            }
        }
        // ---- A: own rows, every band
        for_members([&](size_t i) {
            thz_session *s = gs->sess[i];
            (void)hipSetDevice(g->m[i].ctx->device);
            return thz_dc_slab_energies(g->m[i].ctx, psf, cfg, grid_nx, grid_ny, s->dx_cur, s->dy_cur, s->d_data, rank_pix((size_t)g->m[i].rank), bufA[i]);
        }, rcs);
        if (int rc = any_bad(rcs, &bad)) return rc;
    }
    if (!skipped && !bad) {
        // ---- X1: every slab's [nb][npix_slab] block to everybody; a member re-tiles its bands' rows into whole images
        std::vector<const float *> send(nl);
        std::vector<size_t> counts((size_t)g->world), off((size_t)g->world + 1, 0);
        for (int q = 0; q < g->world; ++q) {
            counts[(size_t)q] = nb * rank_pix((size_t)q);
            off[(size_t)q + 1] = off[(size_t)q] + counts[(size_t)q];
        }
        for (size_t i = 0; i < nl; ++i) send[i] = bufA[i];
        if (int rc = group_all_gather(g, send.data(), counts.data(), bufB.data())) { cleanup(); return rc; }
        for (size_t i = 0; i < nl; ++i) {
            GHIP_TRY_C(g, hipSetDevice(g->m[i].ctx->device));
            const size_t me = (size_t)g->m[i].rank, b_lo = band0[me], nbs = band0[me + 1] - b_lo;
            for (int q = 0; q < g->world && nbs; ++q) {
                const size_t pq = rank_pix((size_t)q);
                if (pq)
                    GHIP_TRY_C(g, hipMemcpy2DAsync(bufC[i] + cur_x0[(size_t)q] * grid_ny, npix_all * sizeof(float), bufB[i] + off[(size_t)q] + b_lo * pq,
                                                   pq * sizeof(float), pq * sizeof(float), nbs, hipMemcpyDeviceToDevice, g->m[i].ctx->stream));
            }
        }
        if (int rc = thz_group_sync(g)) { cleanup(); return rc; }
        // ---- B: own bands, whole grid
        for_members([&](size_t i) {
            thz_session *s = gs->sess[i];
            (void)hipSetDevice(g->m[i].ctx->device);
            const size_t me = (size_t)g->m[i].rank;
            thz_deconv_cfg c = *cfg;
            c.band_begin = (uint32_t)band0[me];
            c.band_end = (uint32_t)band0[me + 1];
            if (c.band_begin == c.band_end) return (int)THZ_OK;  // more ranks than bands
            return thz_dc_band_gains(g->m[i].ctx, psf, &c, grid_nx, grid_ny, s->dx_cur, s->dy_cur, bufC[i], bufD[i], abort_flag, i == 0 ? progress : nullptr);
        }, rcs);
        if (int rc = any_bad(rcs, &bad)) return rc;
        if (bad) {
            status = THZ_ERR_ABORTED;
            for (int rc : rcs)
                if (rc < 0 && rc != THZ_ERR_ABORTED) status = rc;
        }
    } else if (!skipped) {
        status = THZ_ERR_HIP;
        for (int rc : rcs)
            if (rc < 0) status = rc;
    }
    if (!skipped && !bad) {
        // ---- X2: the ranks' [bands][npix] gain blocks, in rank order = band order -> [nb][npix] on everybody; a member
        // keeps the columns of its own rows
        std::vector<const float *> send(nl);
        std::vector<size_t> counts((size_t)g->world);
        for (int q = 0; q < g->world; ++q) counts[(size_t)q] = (band0[(size_t)q + 1] - band0[(size_t)q]) * npix_all;
        for (size_t i = 0; i < nl; ++i) send[i] = bufD[i];
        if (int rc = group_all_gather(g, send.data(), counts.data(), bufB.data())) { cleanup(); return rc; }
        for (size_t i = 0; i < nl; ++i) {
            GHIP_TRY_C(g, hipSetDevice(g->m[i].ctx->device));
            const size_t me = (size_t)g->m[i].rank, pq = rank_pix(me);
            if (pq)
                GHIP_TRY_C(g, hipMemcpy2DAsync(bufA[i], pq * sizeof(float), bufB[i] + cur_x0[me] * grid_ny, npix_all * sizeof(float), pq * sizeof(float), nb,
                                               hipMemcpyDeviceToDevice, g->m[i].ctx->stream));
        }
        if (int rc = thz_group_sync(g)) { cleanup(); return rc; }
        // ---- C: own rows, every band
        for_members([&](size_t i) {
            thz_session *s = gs->sess[i];
            (void)hipSetDevice(g->m[i].ctx->device);
            return thz_dc_slab_combine(g->m[i].ctx, psf, cfg, grid_nx, grid_ny, s->dx_cur, s->dy_cur, rank_pix((size_t)g->m[i].rank), bufA[i], s->d_deconv,
                                       s->d_deconv_img);
        }, rcs);
        if (int rc = any_bad(rcs, &bad)) return rc;
        if (bad) {
            status = THZ_ERR_HIP;
            for (int rc : rcs)
                if (rc < 0) status = rc;
        }
    }
    if (skipped) status = THZ_SKIPPED;
    if (skipped || bad) {
        // the stage passes its input through: every slab keeps its own "Time Band Pass" output
        for (size_t i = 0; i < nl; ++i) {
            thz_session *s = gs->sess[i];
            thz_ctx *ctx = g->m[i].ctx;
            GHIP_TRY_C(g, hipSetDevice(ctx->device));
            const size_t pq = rank_pix((size_t)g->m[i].rank);
            GHIP_TRY_C(g, hipMemcpyAsync(s->d_deconv, s->d_data, pq * nt * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            if (int rc = thz_intensity(ctx, pq, s->d_deconv, s->d_deconv_img)) { cleanup(); return gfail(g, rc, thz_last_error(ctx)); }
        }
    }
    for (thz_session *s : gs->sess) s->deconv_current = status >= 0;
    // C1: the new image (and, if the last recompute gathered it, the new final cube) to rank 0
    {
        std::vector<const float *> im, dat;
        std::vector<size_t> ic((size_t)g->world), dc((size_t)g->world);
        for (int q = 0; q < g->world; ++q) {
            ic[(size_t)q] = rank_pix((size_t)q);
            dc[(size_t)q] = rank_pix((size_t)q) * nt;
        }
        for (thz_session *s : gs->sess) {
            im.push_back(s->deconv_current ? s->d_deconv_img : s->d_img);
            dat.push_back(s->deconv_current ? s->d_deconv : s->d_data);
        }
        if (int rc = thz_group_gather(g, im.data(), ic.data(), gs->d_img)) { cleanup(); return rc; }
        if (gs->gathered >= THZ_GATHER_TIME && (gs->root_local < 0 || gs->d_data))
            if (int rc = thz_group_gather(g, dat.data(), dc.data(), gs->d_data)) { cleanup(); return rc; }
    }
    cleanup();
#undef GHIP_TRY_C
    // the regions of interest's means of the FINAL traces follow the stage's output (data_thread.rs:1445-1451)
    if (!gs->sess[0]->rois.empty() && gs->sess[0]->have_last_cfg) {
        std::vector<float *> bufs;
        bool data_only = true;
        for (size_t i = 0; i < nl; ++i) {
            const int rc = session_roi_sums(gs->sess[i], &gs->sess[i]->last_cfg, &data_only);
            if (rc) return gfail(g, rc, std::string("slab regions of interest: ") + thz_last_error(g->m[i].ctx));
            bufs.push_back(gs->sess[i]->d_roi_sum);
        }
        if (int rc = group_roi_reduce(gs, bufs, data_only)) return rc;
        for (size_t i = 0; i < nl; ++i) {
            const int rc = session_roi_finish(gs->sess[i], &gs->sess[i]->last_cfg, data_only);
            if (rc) return gfail(g, rc, std::string("slab regions of interest: ") + thz_last_error(g->m[i].ctx));
        }
        if (int rc = thz_group_sync(g)) return rc;
    }
    if (status < 0) return gfail(g, status, "thz_group_session_deconvolve: aborted or failed on a rank; the stage passes its input through");
    return status;
}

int thz_group_session_grid(const thz_group_session *gs, size_t *nx, size_t *ny)
{
    if (!gs) return THZ_ERR_INVALID;
    if (nx) *nx = gs->cur_ny ? gs->cur_pix() / gs->cur_ny : 0;
    if (ny) *ny = gs->cur_ny;
    return THZ_OK;
}

void *thz_group_session_result(thz_group_session *gs, int which)
{
    if (!gs || gs->root_local < 0 || gs->gathered < 0) return nullptr;
    switch (which) {
    case THZ_BUF_IMG: return gs->d_img;
    case THZ_BUF_DATA: return gs->gathered >= THZ_GATHER_TIME ? gs->d_data : nullptr;
    case THZ_BUF_FFT: return gs->gathered >= THZ_GATHER_ALL ? gs->d_fft : nullptr;
    case THZ_BUF_AMPLITUDES: return gs->gathered >= THZ_GATHER_ALL ? gs->d_amp : nullptr;
    case THZ_BUF_PHASES: return gs->gathered >= THZ_GATHER_ALL ? gs->d_ph : nullptr;
    case THZ_BUF_AVG_FFT: case THZ_BUF_AVG_AMPLITUDES: case THZ_BUF_AVG_PHASES:
        return thz_session_buffer(gs->sess[(size_t)gs->root_local], which);
    default: return nullptr;
    }
}

int thz_group_session_download(thz_group_session *gs, int which, size_t pix0, size_t npix, void *dst)
{
    if (!gs || !dst) return THZ_ERR_INVALID;
    thz_group *g = gs->g;
    if (gs->root_local < 0) return gfail(g, THZ_ERR_NOT_READY, "this process does not drive rank 0");
    thz_session *rs = gs->sess[(size_t)gs->root_local];
    if (which == THZ_BUF_AVG_FFT || which == THZ_BUF_AVG_AMPLITUDES || which == THZ_BUF_AVG_PHASES)
        return thz_session_download(rs, which, 0, 1, dst);
    const float *base = static_cast<const float *>(thz_group_session_result(gs, which));
    if (!base) return gfail(g, THZ_ERR_NOT_READY, "buffer was not gathered by the last recompute");
    const size_t nf = gs->nt_out / 2 + 1;
    size_t per = 0;
    switch (which) {
    case THZ_BUF_IMG: per = 1; break;
    case THZ_BUF_DATA: per = gs->nt_out; break;
    case THZ_BUF_FFT: per = 2 * nf; break;
    case THZ_BUF_AMPLITUDES: case THZ_BUF_PHASES: per = nf; break;
    default: return THZ_ERR_INVALID;
    }
    if (pix0 > gs->cur_pix() || npix > gs->cur_pix() - pix0) return gfail(g, THZ_ERR_INVALID, "pixel range out of bounds");
    return thz_memcpy_d2h(g->m[(size_t)gs->root_local].ctx, dst, base + pix0 * per, npix * per * sizeof(float));
}

}  // extern "C"
