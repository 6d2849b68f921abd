// mock_rccl.cpp — TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so that lets TWO OR MORE PROCESSES ON ONE GPU run
// the library's process-per-GPU entry points (thz_group_create_rank, thz_group_session_*) end to end.  RCCL itself
// refuses two ranks on one device, so on a one-GPU box the real library can only ever be driven with one rank
// (tests/test_gpu_group.py::test_rank_api_through_real_rccl_single_rank); this mock implements the eleven entry
// points libthzgpu.so resolves (group_api.cpp) with their real signatures over POSIX shared memory and a
// process-shared barrier: a call synchronises its stream, stages its buffers through host memory, and returns with the
// result in place.  It knows nothing about a fabric and says nothing about speed; what it checks is the library's
// SEQUENCE of calls — who sends what to whom, counts and offsets per rank, root and non-root roles — which on real
// hardware would only surface as a hang or wrong rows in the driver's multi-GPU bench.
// libthzgpu.so opens it when THZ_RCCL_LIB points here (developer knob of rccl_load()).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr size_t kMaxRanks = 8;
constexpr size_t kSlotBytes = 48u << 20;     // per rank (collectives) and per ordered pair (send / recv)

struct Shared {
    volatile int ready;
    int nranks;
    pthread_barrier_t barrier;
    size_t mail_used[kMaxRanks][kMaxRanks];  // bytes queued from src to dst in the current group
    // followed by: slots[nranks][kSlotBytes], mail[nranks][nranks][kSlotBytes]
};

struct Op {
    int kind;  // 0 all-reduce, 1 broadcast, 2 send, 3 recv
    const void *send;
    void *recv;
    size_t count;
    ncclDataType_t type;
    int peer;  // root / peer
    hipStream_t stream;
};

}  // namespace

struct ncclComm {
    int rank = 0, nranks = 1;
    Shared *sh = nullptr;
    size_t map_bytes = 0;
    std::string name;
    unsigned char *slot(int r) { return reinterpret_cast<unsigned char *>(sh + 1) + (size_t)r * kSlotBytes; }
    unsigned char *mail(int src, int dst)
    {
        return reinterpret_cast<unsigned char *>(sh + 1) + ((size_t)nranks + (size_t)src * nranks + dst) * kSlotBytes;
    }
};

namespace {

thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm *, Op>> g_queue;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclFloat: return 4;
    case ncclUint64: case ncclInt64: case ncclDouble: return 8;
    case ncclInt32: case ncclUint32: return 4;
    default: return 1;
    }
}

bool run_collective(ncclComm *c, const Op &op)
{
    const size_t bytes = op.count * type_bytes(op.type);
    if (bytes > kSlotBytes) { std::fprintf(stderr, "mock rccl: message of %zu bytes exceeds the slot\n", bytes); return false; }
    if (hipStreamSynchronize(op.stream) != hipSuccess) return false;
    if (op.kind == 0) {
        if (hipMemcpy(c->slot(c->rank), op.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
        pthread_barrier_wait(&c->sh->barrier);
        std::vector<unsigned char> acc(bytes);
        std::memcpy(acc.data(), c->slot(0), bytes);
        for (int r = 1; r < c->nranks; ++r) {
            if (op.type == ncclFloat) {
                float *a = reinterpret_cast<float *>(acc.data());
                const float *b = reinterpret_cast<const float *>(c->slot(r));
                for (size_t i = 0; i < op.count; ++i) a[i] += b[i];
            } else if (op.type == ncclUint64) {
                uint64_t *a = reinterpret_cast<uint64_t *>(acc.data());
                const uint64_t *b = reinterpret_cast<const uint64_t *>(c->slot(r));
                for (size_t i = 0; i < op.count; ++i) a[i] += b[i];
            } else {
                return false;
            }
        }
        if (hipMemcpy(op.recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
        pthread_barrier_wait(&c->sh->barrier);
        return true;
    }
    // broadcast from op.peer
    if (c->rank == op.peer && hipMemcpy(c->slot(op.peer), op.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    pthread_barrier_wait(&c->sh->barrier);
    if (hipMemcpy(op.recv, c->slot(op.peer), bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
    pthread_barrier_wait(&c->sh->barrier);
    return true;
}

// executes what a (possibly implicit) group holds, in order: point-to-point messages first (all sends are staged,
// then all receives read them in the order they were posted), collectives one by one
ncclResult_t flush()
{
    std::vector<std::pair<ncclComm *, Op>> q;
    q.swap(g_queue);
    bool ok = true;
    // Point-to-point traffic is pairwise, like RCCL's: only the two ranks of a message take part (round 3: the
    // library's chained reference-order means and its block sums over slab edges talk rank q -> q + 1 while the other
    // ranks do something else).  One mailbox per ordered pair with a "bytes waiting" word: the sender waits until it
    // is empty, fills it and publishes the size; the receiver waits for a size, drains it and clears the word.  Ops
    // run in posting order (the library posts at most one message per pair and direction in a group).
    for (auto &e : q) {
        const Op &op = e.second;
        if (op.kind < 2) continue;
        ncclComm *c = e.first;
        const size_t bytes = op.count * type_bytes(op.type);
        if (bytes > kSlotBytes) { std::fprintf(stderr, "mock rccl: mailbox overflow\n"); ok = false; continue; }
        if (hipStreamSynchronize(op.stream) != hipSuccess) ok = false;
        if (op.kind == 2) {
            volatile size_t *used = &c->sh->mail_used[c->rank][op.peer];
            long spins = 0;
            while (*used != 0 && ++spins < 120000000L) sched_yield();
            if (*used != 0) { std::fprintf(stderr, "mock rccl: send %d -> %d never drained\n", c->rank, op.peer); ok = false; continue; }
            if (hipMemcpy(c->mail(c->rank, op.peer), op.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
            __sync_synchronize();
            *used = bytes;
        } else {
            volatile size_t *used = &c->sh->mail_used[op.peer][c->rank];
            long spins = 0;
            while (*used == 0 && ++spins < 120000000L) sched_yield();
            if (*used != bytes) { std::fprintf(stderr, "mock rccl: recv %d <- %d: %zu bytes waiting, %zu expected\n", c->rank, op.peer, (size_t)*used, bytes); ok = false; if (*used) *used = 0; continue; }
            __sync_synchronize();
            if (hipMemcpy(op.recv, c->mail(op.peer, c->rank), bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
            __sync_synchronize();
            *used = 0;
        }
    }
    for (auto &e : q)
        if (e.second.kind < 2 && !run_collective(e.first, e.second)) ok = false;
    return ok ? ncclSuccess : ncclInternalError;
}

ncclResult_t post(ncclComm *c, const Op &op)
{
    if (!c) return ncclInvalidArgument;
    g_queue.push_back({c, op});
    return g_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    FILE *f = std::fopen("/dev/urandom", "rb");
    unsigned char r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    if (f) { if (std::fread(r, 1, 8, f) != 8) r[0] ^= 0x55; std::fclose(f); }
    std::snprintf(id->internal, sizeof id->internal, "thzmock_%02x%02x%02x%02x%02x%02x%02x%02x_%d", r[0], r[1], r[2], r[3], r[4], r[5],
                  r[6], r[7], (int)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > (int)kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm *c = new ncclComm();
    c->rank = rank;
    c->nranks = nranks;
    c->name = std::string("/") + id.internal;
    c->map_bytes = sizeof(Shared) + ((size_t)nranks + (size_t)nranks * nranks) * kSlotBytes;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 3000 && fd < 0; ++tries) {  // rank 0 creates it
            fd = shm_open(c->name.c_str(), O_RDWR, 0600);
            if (fd < 0) usleep(10000);
        }
        if (fd < 0) { delete c; return ncclSystemError; }
        for (int tries = 0; tries < 3000; ++tries) {  // ... and sizes it
            off_t sz = lseek(fd, 0, SEEK_END);
            if ((size_t)sz >= c->map_bytes) break;
            usleep(10000);
        }
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared *>(p);
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->sh->barrier, &a, (unsigned)nranks);
        pthread_barrierattr_destroy(&a);
        c->sh->nranks = nranks;
        std::memset((void *)c->sh->mail_used, 0, sizeof c->sh->mail_used);
        __sync_synchronize();
        c->sh->ready = 1;
    } else {
        for (int tries = 0; tries < 3000 && !c->sh->ready; ++tries) usleep(10000);
        if (!c->sh->ready) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
    }
    pthread_barrier_wait(&c->sh->barrier);
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }  // one process, n devices: not mocked

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    pthread_barrier_wait(&comm->sh->barrier);
    munmap(comm->sh, comm->map_bytes);
    if (comm->rank == 0) shm_unlink(comm->name.c_str());
    delete comm;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    if (op != ncclSum) return ncclInvalidArgument;
    return post(comm, Op{0, send, recv, count, type, 0, stream});
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm,
                           hipStream_t stream)
{
    return post(comm, Op{1, send, recv, count, type, root, stream});
}

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, Op{2, send, nullptr, count, type, peer, stream});
}

ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, Op{3, nullptr, recv, count, type, peer, stream});
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth > 0) --g_depth;
    return g_depth == 0 ? flush() : ncclSuccess;
}

}  // extern "C"
