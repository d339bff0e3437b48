// fake_rccl.cpp — TEST-ONLY stand-in for librccl.so, loaded through GUBER_RCCL_LIB (guber_global_sync.h dlopens it first).
//
// Why: RCCL refuses two ranks on one GPU, and the GPU boxes this repository is tested on have ONE GPU — so the RCCL branch of
// guber_global_sync (grouped ncclSend / ncclRecv pairs, the count exchange by ncclAllGather, guber_comm_create_rank /
// ncclCommInitRank) would otherwise run for the first time on the day an 8-GPU node appears.  This library implements exactly the
// entry points the product resolves (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclCommDestroy, ncclGroupStart,
// ncclGroupEnd, ncclSend, ncclRecv, ncclAllGather, ncclGetErrorString) with RCCL's signatures and group semantics, between ranks
// that are processes (or communicators of one process) SHARING one GPU: a message travels device -> POSIX shared memory -> device.
// It is stricter than RCCL in one way — ncclGroupEnd completes the transfers before it returns — and checks what RCCL would
// deadlock or corrupt on: a Recv without a matching Send of the same size, calls outside a group, ranks out of range.
// Nothing in the product links or names this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

namespace {
constexpr int kMaxWorld = 16;
struct Shared {                                   // one per communicator world, in /dev/shm
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> detached;
    std::atomic<uint64_t> sent[kMaxWorld][kMaxWorld];      // messages src -> dst published so far
    std::atomic<uint64_t> size[kMaxWorld][kMaxWorld][4];   // byte size of message (seq & 3)
    std::atomic<uint64_t> taken[kMaxWorld][kMaxWorld];     // messages src -> dst consumed so far
};
struct Comm { std::string name; Shared* sh = nullptr; int world = 0, rank = 0, device = 0; uint64_t calls = 0; };
struct Op { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; Comm* c; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
std::atomic<uint32_t> g_counter{0};
std::atomic<unsigned long long> g_total{0};

size_t tsize(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
                 case ncclInt64: case ncclUint64: case ncclFloat64: return 8; case ncclFloat16: case ncclBfloat16: return 2; default: return 0; }
}
bool wait_until(const std::function<bool()>& ok, double seconds = 60.0) {
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t spin = 0;; ++spin) {
        if (ok()) return true;
        if ((spin & 1023) == 1023) {
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9 > seconds) return false;
            usleep(50);
        }
    }
}
std::string msg_name(const Comm* c, int src, int dst, uint64_t seq) {
    char b[256]; snprintf(b, sizeof b, "%s_m_%d_%d_%llu", c->name.c_str(), src, dst, (unsigned long long)seq); return b;
}
Shared* map_shared(const std::string& name, bool create) {
    const int fd = shm_open(name.c_str(), create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (create && ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return nullptr; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : (Shared*)p;                 // (a fresh segment is zero-filled: every counter starts at 0)
}
ncclResult_t attach(Comm* c) {
    c->sh = map_shared(c->name, false);
    if (!c->sh) { fprintf(stderr, "[fake_rccl] cannot open %s: %s\n", c->name.c_str(), strerror(errno)); return ncclSystemError; }
    c->sh->attached.fetch_add(1);
    if (!wait_until([&] { return c->sh->attached.load() >= (uint32_t)c->world; })) {
        fprintf(stderr, "[fake_rccl] rank %d: only %u of %d ranks arrived\n", c->rank, c->sh->attached.load(), c->world);
        return ncclInternalError;
    }
    return ncclSuccess;
}
// one message src -> dst: its own shared-memory object of exactly its size, published by the counter
ncclResult_t do_send(const Op& o) {
    Comm* c = o.c;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;   // the data is ready
    const uint64_t seq = c->sh->sent[c->rank][o.peer].load();
    if (!wait_until([&] { return seq - c->sh->taken[c->rank][o.peer].load() < 4; })) return ncclInternalError;                   // (4 size slots)
    const std::string nm = msg_name(c, c->rank, o.peer, seq);
    const int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)o.bytes) != 0) { if (fd >= 0) close(fd); return ncclSystemError; }
    void* p = mmap(nullptr, o.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    const hipError_t he = hipMemcpy(p, o.sbuf, o.bytes, hipMemcpyDeviceToHost);
    munmap(p, o.bytes);
    if (he != hipSuccess) return ncclUnhandledCudaError;
    c->sh->size[c->rank][o.peer][seq & 3].store(o.bytes);
    c->sh->sent[c->rank][o.peer].store(seq + 1);
    return ncclSuccess;
}
ncclResult_t do_recv(const Op& o) {
    Comm* c = o.c;
    const uint64_t seq = c->sh->taken[o.peer][c->rank].load();
    if (!wait_until([&] { return c->sh->sent[o.peer][c->rank].load() > seq; })) {
        fprintf(stderr, "[fake_rccl] rank %d: ncclRecv from %d of %zu bytes has no matching ncclSend (RCCL would hang here)\n", c->rank, o.peer, o.bytes);
        return ncclInternalError;
    }
    const uint64_t got = c->sh->size[o.peer][c->rank][seq & 3].load();
    if (got != o.bytes) {
        fprintf(stderr, "[fake_rccl] rank %d: ncclRecv from %d expects %zu bytes, the matching ncclSend carries %llu\n", c->rank, o.peer, o.bytes, (unsigned long long)got);
        return ncclInvalidArgument;
    }
    const std::string nm = msg_name(c, o.peer, c->rank, seq);
    const int fd = shm_open(nm.c_str(), O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    void* p = mmap(nullptr, o.bytes, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    hipError_t he = hipSetDevice(c->device);
    if (he == hipSuccess) he = hipMemcpyAsync(o.rbuf, p, o.bytes, hipMemcpyHostToDevice, o.stream);      // in stream order, like RCCL
    if (he == hipSuccess) he = hipStreamSynchronize(o.stream);
    munmap(p, o.bytes);
    shm_unlink(nm.c_str());
    c->sh->taken[o.peer][c->rank].store(seq + 1);
    return he == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t flush() {
    // every send of the group first (they never block on a receiver), then the receives: the order RCCL's group gives
    ncclResult_t rc = ncclSuccess;
    for (const Op& o : g_ops) if (o.send && rc == ncclSuccess) rc = do_send(o);
    for (const Op& o : g_ops) if (!o.send && rc == ncclSuccess) rc = do_recv(o);
    g_ops.clear();
    return rc;
}
ncclResult_t enqueue(const Op& o) {
    if (!o.c || !o.c->sh) return ncclInvalidArgument;
    if (o.peer < 0 || o.peer >= o.c->world) { fprintf(stderr, "[fake_rccl] peer %d out of range (world %d)\n", o.peer, o.c->world); return ncclInvalidArgument; }
    if (o.peer == o.c->rank) { fprintf(stderr, "[fake_rccl] rank %d sends to / receives from itself\n", o.c->rank); return ncclInvalidArgument; }
    o.c->calls++; g_total.fetch_add(1);
    g_ops.push_back(o);
    return g_depth ? ncclSuccess : flush();
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/guber_fake_rccl_%d_%u_%ld", (int)getpid(), g_counter.fetch_add(1), (long)time(nullptr));
    Shared* sh = map_shared(id->internal, true);
    if (!sh) return ncclSystemError;
    munmap(sh, sizeof(Shared));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxWorld || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->name = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    c->world = nranks; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    const ncclResult_t rc = attach(c);
    if (rc != ncclSuccess) { delete c; return rc; }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1 || ndev > kMaxWorld) return ncclInvalidArgument;
    ncclUniqueId id;
    ncclResult_t rc = ncclGetUniqueId(&id);
    if (rc != ncclSuccess) return rc;
    std::vector<Comm*> cs;
    for (int i = 0; i < ndev; ++i) {                               // (one process: attach all first, nobody waits for the others)
        Comm* c = new Comm();
        c->name = id.internal; c->world = ndev; c->rank = i; c->device = devlist ? devlist[i] : i;
        c->sh = map_shared(c->name, false);
        if (!c->sh) { delete c; for (Comm* x : cs) delete x; return ncclSystemError; }
        c->sh->attached.fetch_add(1);
        cs.push_back(c);
    }
    for (int i = 0; i < ndev; ++i) comms[i] = (ncclComm_t)cs[i];
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    if (c->sh) {
        if (c->sh->detached.fetch_add(1) + 1 == (uint32_t)c->world) shm_unlink(c->name.c_str());
        munmap(c->sh, sizeof(Shared));
    }
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) { fprintf(stderr, "[fake_rccl] ncclGroupEnd without ncclGroupStart\n"); return ncclInvalidUsage; }
    return --g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(Op{true, sendbuff, nullptr, count * tsize(datatype), peer, (Comm*)comm, stream});
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(Op{false, nullptr, recvbuff, count * tsize(datatype), peer, (Comm*)comm, stream});
}
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    const size_t bytes = sendcount * tsize(datatype);
    ++g_depth;
    ncclResult_t rc = ncclSuccess;
    for (int p = 0; p < c->world && rc == ncclSuccess; ++p) {
        if (p == c->rank) continue;
        rc = enqueue(Op{true, sendbuff, nullptr, bytes, p, c, stream});
        if (rc == ncclSuccess) rc = enqueue(Op{false, nullptr, (char*)recvbuff + (size_t)p * bytes, bytes, p, c, stream});
    }
    --g_depth;
    if (rc != ncclSuccess) { g_ops.clear(); return rc; }
    if (hipSetDevice(c->device) != hipSuccess ||
        hipMemcpyAsync((char*)recvbuff + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return g_depth ? ncclSuccess : flush();
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) { case ncclSuccess: return "fake_rccl: success"; case ncclInvalidArgument: return "fake_rccl: invalid argument"; case ncclInvalidUsage: return "fake_rccl: invalid usage";
                 case ncclSystemError: return "fake_rccl: system error"; case ncclUnhandledCudaError: return "fake_rccl: HIP error"; default: return "fake_rccl: internal error (see stderr)"; }
}
// how many Send / Recv calls went through a communicator: the tests assert that the RCCL branch really ran
unsigned long long fake_rccl_calls(ncclComm_t comm) { return comm ? ((Comm*)comm)->calls : 0; }
unsigned long long fake_rccl_total_calls(void) { return g_total.load(); }
}
