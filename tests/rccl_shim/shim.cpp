// tests/rccl_shim/shim.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl that lets several ranks share ONE GPU.
//
// RCCL refuses two ranks on one device, and the build container / the one-GPU test box have no second GPU, so the native
// collectives of libdenovo_kmer (dk_set_allreduce_or, dk_accum_exchange_finish: grouped ncclSend / ncclRecv + ncclAllGather on
// the engine's stream) could only ever run with one rank as its own peer -- which cannot catch a wrong peer index, a wrong
// staging slot or a missed receive.  This shim implements the handful of entry points the library loads (csrc/dk_comm.h) over
// POSIX shared memory between processes: a send is a device-to-host copy into the (source, destination) mailbox, a receive a
// host-to-device copy out of it, with inter-process barriers between the phases of a group.  Everything is synchronous on the
// host after a stream synchronisation, which preserves stream order.  Loaded through DK_RCCL_LIBRARY=<path> by the tests in
// tests/test_gpu_multirank.py; it is never part of the product.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <vector>

namespace {

constexpr size_t MAILBOX = 48u << 20;          // bytes per (source, destination) pair
constexpr int MAX_RANKS = 8;

struct Header {
    volatile int arrived;                      // barrier: ranks arrived in the current generation
    volatile int generation;
    volatile size_t size[MAX_RANKS][MAX_RANKS];
};

struct Comm {
    int n, rank;
    char name[128];
    Header *hdr;
    char *boxes;
    size_t bytes;
};

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

char *box(Comm *c, int src, int dst) { return c->boxes + ((size_t)src * c->n + dst) * MAILBOX; }

void barrier(Comm *c)
{
    const int gen = c->hdr->generation;
    if (__atomic_add_fetch(&c->hdr->arrived, 1, __ATOMIC_SEQ_CST) == c->n) {
        c->hdr->arrived = 0;
        __atomic_store_n(&c->hdr->generation, gen + 1, __ATOMIC_SEQ_CST);
    } else {
        while (__atomic_load_n(&c->hdr->generation, __ATOMIC_SEQ_CST) == gen) usleep(50);
    }
}

size_t type_bytes(int t) { return t == 0 || t == 1 ? 1 : t == 2 || t == 3 ? 4 : t == 4 || t == 5 ? 8 : t == 6 ? 2 : t == 7 ? 4 : t == 8 ? 8 : 1; }

int run_group()
{
    if (g_ops.empty()) return 0;
    Comm *c = g_ops[0].comm;
    for (const Op &o : g_ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    for (const Op &o : g_ops) {
        if (!o.send) continue;
        if (o.bytes > MAILBOX) { fprintf(stderr, "rccl shim: message of %zu bytes exceeds the mailbox\n", o.bytes); return 5; }
        if (hipMemcpy(box(c, c->rank, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        c->hdr->size[c->rank][o.peer] = o.bytes;
    }
    barrier(c);
    int rc = 0;
    for (const Op &o : g_ops) {
        if (o.send) continue;
        if (c->hdr->size[o.peer][c->rank] != o.bytes) {
            fprintf(stderr, "rccl shim: rank %d expects %zu bytes from %d, which sent %zu\n", c->rank, o.bytes, o.peer,
                    (size_t)c->hdr->size[o.peer][c->rank]);
            rc = 5;
            continue;
        }
        if (hipMemcpy(o.buf, box(c, o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    }
    barrier(c);
    for (const Op &o : g_ops)
        if (o.send) c->hdr->size[c->rank][o.peer] = 0;
    barrier(c);
    g_ops.clear();
    return rc;
}

}  // namespace

extern "C" {

struct ncclUniqueId_t { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId_t *id)
{
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/dk_rccl_shim_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
    return 0;
}

int ncclCommInitRank(Comm **out, int n, ncclUniqueId_t id, int rank)
{
    if (n < 1 || n > MAX_RANKS || rank < 0 || rank >= n) return 4;
    Comm *c = new Comm();
    c->n = n;
    c->rank = rank;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->bytes = sizeof(Header) + (size_t)n * n * MAILBOX;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;
    } else {
        for (int tries = 0; tries < 20000; tries++) {
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;
            if (fd >= 0) close(fd);
            fd = -1;
            usleep(1000);
        }
        if (fd < 0) return 2;
    }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->hdr = (Header *)p;
    c->boxes = (char *)p + sizeof(Header);
    barrier(c);
    *out = c;
    return 0;
}

int ncclCommDestroy(Comm *c)
{
    if (!c) return 0;
    barrier(c);
    munmap((void *)c->hdr, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

int ncclCommAbort(Comm *c)
{
    if (!c) return 0;
    munmap((void *)c->hdr, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

int ncclGroupStart() { g_depth++; return 0; }

int ncclGroupEnd()
{
    if (--g_depth > 0) return 0;
    return run_group();
}

int ncclSend(const void *buf, size_t count, int type, int peer, Comm *c, hipStream_t stream)
{
    g_ops.push_back(Op{true, (void *)buf, count * type_bytes(type), peer, c, stream});
    return g_depth ? 0 : run_group();
}

int ncclRecv(void *buf, size_t count, int type, int peer, Comm *c, hipStream_t stream)
{
    g_ops.push_back(Op{false, buf, count * type_bytes(type), peer, c, stream});
    return g_depth ? 0 : run_group();
}

int ncclAllGather(const void *send, void *recv, size_t count, int type, Comm *c, hipStream_t stream)
{
    const size_t nb = count * type_bytes(type);
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    int rc = 0;
    for (size_t off = 0; off < nb || (nb == 0 && off == 0); off += MAILBOX) {        // contributions larger than a mailbox go in chunks
        const size_t n = nb - off < MAILBOX ? nb - off : MAILBOX;
        if (n && hipMemcpy(box(c, c->rank, c->rank), (const char *)send + off, n, hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
        barrier(c);
        for (int q = 0; q < c->n && n; q++)
            if (hipMemcpy((char *)recv + (size_t)q * nb + off, box(c, q, q), n, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
        barrier(c);
        if (nb == 0) break;
    }
    return rc;
}

const char *ncclGetErrorString(int r)
{
    return r == 0 ? "success" : r == 1 ? "HIP error (shim)" : r == 2 ? "shared memory error (shim)" : r == 4 ? "invalid argument (shim)"
                                                                                                            : "message mismatch (shim)";
}

}  // extern "C"
