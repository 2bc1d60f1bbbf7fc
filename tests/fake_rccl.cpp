// TEST INFRASTRUCTURE: a stand-in for librccl that lets several PROCESSES on ONE GPU run libscythe_hip's in-library exchange
// (csrc/sx_comm.cpp: sx_comm_init / sx_exchange) with more than one rank.  RCCL itself refuses two ranks on one device and the
// GPU box has one, so without this the n > 1 branches of sx_exchange - rank-dependent offsets, the grouped send / receive
// loops, the halo chain, the in-place all-gather - only ever ran through the single-process loopback transport.
//
// It implements exactly the entry points sx_comm.cpp binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclGetErrorString, ncclSend, ncclRecv, ncclAllGather, ncclGroupStart, ncclGroupEnd) over a POSIX shared-memory segment:
// a group is executed at ncclGroupEnd as  stream sync -> every send copied device-to-host into the mailbox [me][peer] ->
// barrier of all ranks -> every receive copied host-to-device from the mailbox [peer][me] -> barrier.  Synchronous and slow
// on purpose; it says nothing about RCCL's performance or about xGMI.  Selected with SX_RCCL_LIB=<this .so>
// (tests/test_gpu_dist_rehearsal.py::test_in_library_exchange_multi_rank_through_a_stand_in_transport).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr size_t SLOT_BYTES = (size_t)24 << 20;          // per (source, destination) mailbox

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> generation;
};

struct Comm {
    int n = 0, me = 0;
    char name[64];
    size_t bytes = 0;
    unsigned char *base = nullptr;
    Header *hdr = nullptr;
    unsigned char *slot(int src, int dst) { return base + 4096 + ((size_t)src * n + dst) * SLOT_BYTES; }
    // returns false when a peer has not arrived after SX_FAKE_RCCL_TIMEOUT seconds (default 120): a rank that died must not leave
    // the others spinning here, holding the GPU, for the rest of the session
    bool barrier() {
        static const double limit = getenv("SX_FAKE_RCCL_TIMEOUT") ? atof(getenv("SX_FAKE_RCCL_TIMEOUT")) : 120.0;
        const int gen = hdr->generation.load();
        if (hdr->arrived.fetch_add(1) + 1 == n) {
            hdr->arrived.store(0);
            hdr->generation.fetch_add(1);
            return true;
        }
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (long spins = 0; hdr->generation.load() == gen; spins++) {
            sched_yield();
            if ((spins & 1023) == 1023) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) {
                    fprintf(stderr, "fake_rccl: rank %d waited %.0f s at a barrier for its peers - giving up\n", me, limit);
                    return false;
                }
            }
        }
        return true;
    }
};

struct Op { int kind; void *buf; size_t bytes; int peer; Comm *comm; hipStream_t stream; };   // kind 0 = send, 1 = recv
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

int run_group() {
    if (g_ops.empty()) return 0;
    Comm *c = g_ops[0].comm;
    for (auto &o : g_ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    for (auto &o : g_ops)
        if (o.kind == 0) {
            if (o.bytes > SLOT_BYTES) { fprintf(stderr, "fake_rccl: message of %zu bytes exceeds the mailbox\n", o.bytes); return 2; }
            if (o.bytes && hipMemcpy(c->slot(c->me, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        }
    if (!c->barrier()) { g_ops.clear(); return 4; }
    for (auto &o : g_ops)
        if (o.kind == 1 && o.bytes && hipMemcpy(o.buf, c->slot(o.peer, c->me), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (!c->barrier()) { g_ops.clear(); return 4; }
    g_ops.clear();
    return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id) {
    std::memset(id->internal, 0, 128);
    snprintf(id->internal, 64, "/sxfake_%d_%ld", (int)getpid(), (long)random());
    return 0;
}

int ncclCommInitRank(void **comm, int n, ncclUniqueId id, int rank) {
    Comm *c = new Comm();
    c->n = n; c->me = rank;
    std::memset(c->name, 0, sizeof c->name);
    std::strncpy(c->name, id.internal, sizeof c->name - 1);
    c->bytes = 4096 + (size_t)n * n * SLOT_BYTES;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 3;
    c->base = (unsigned char *)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) return 3;
    c->hdr = reinterpret_cast<Header *>(c->base);        // a fresh segment is zero-filled: arrived = generation = 0
    if (!c->barrier()) return 4;                         // collective, like the real call
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    munmap(c->base, c->bytes);
    if (c->me == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

const char *ncclGetErrorString(int rc) { return rc == 0 ? "ok" : rc == 1 ? "fake_rccl: HIP call failed" : rc == 2 ? "fake_rccl: message too large" : rc == 4 ? "fake_rccl: a peer did not reach the barrier in time" : "fake_rccl: shared memory"; }

int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() { return --g_depth == 0 ? run_group() : 0; }

static int queue(int kind, void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) {
    if (dtype != 8) return 2;                            // ncclDouble only
    g_ops.push_back(Op{kind, buf, count * sizeof(double), peer, (Comm *)comm, s});
    return g_depth == 0 ? run_group() : 0;
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) { return queue(0, const_cast<void *>(buf), count, dtype, peer, comm, s); }
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) { return queue(1, buf, count, dtype, peer, comm, s); }

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    if (dtype != 8 || count * sizeof(double) > SLOT_BYTES) return 2;
    const size_t bytes = count * sizeof(double);
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (bytes && hipMemcpy(c->slot(c->me, c->me), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!c->barrier()) return 4;
    for (int r = 0; r < c->n; r++)
        if (bytes && hipMemcpy((char *)recv + (size_t)r * bytes, c->slot(r, r), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (!c->barrier()) return 4;
    return 0;
}

}  // extern "C"
