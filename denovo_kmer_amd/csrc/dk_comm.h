// dk_comm.h -- RCCL behind the C ABI: dk_comm_* and dk_set_allreduce (include/denovo_kmer.h).
//
// The one exchange step of the hot path: every rank holds a partial parent set of the FULL size,
// built from its parent-read shard; the partials are combined in place.  RCCL has no bitwise-OR reduction
// (ncclRedOp_t = sum/prod/max/min/avg, rccl.h:448-453), so the all-reduce is composed on the engine's stream:
//
//   chunk by chunk:  all-to-all (ncclSend / ncclRecv inside one group): rank r receives piece j of slice r from
//                    every other rank into a staging buffer;  or_slices_kernel (Bloom) / union_slices_kernel (exact)
//                    combines the P - 1 pieces into rank r's own slice
//   once:            ncclAllGather in place: every rank receives every combined slice
//
// On a fully connected xGMI node every phase drives all 7 links of a GPU at once (a ring all-reduce is bound by
// one link).  The staging buffer is bounded (1 GiB by default), so a 64-GiB filter needs no second 64 GiB.
// librccl is loaded on first use (dlopen "librccl.so.1": in a process that already holds PyTorch's copy the loader
// hands back that one), so the library itself carries no link-time dependency on it.
#pragma once
#include <dlfcn.h>

#include "dk_internal.h"
#include "dk_kernels_bucket.h"

namespace dk {

// the subset of rccl.h this file calls (types as declared there: rccl.h:43, 459-470, 187-720)
typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId_t { char internal[128]; };
enum { RCCL_SUCCESS = 0, RCCL_UINT8 = 1 };

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId_t *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

inline RcclApi *rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    // DK_RCCL_LIBRARY names the library outright (a non-standard install; the tests point it at tests/rccl_shim, which
    // runs several ranks on one GPU).  A named library that does not load is an error, not a reason to fall back.
    const char *named = getenv("DK_RCCL_LIBRARY");
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (named && *named) {
        api.lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
    }
    if (!api.lib) { api.err = std::string("cannot load librccl: ") + dlerror(); return &api; }
    struct { const char *name; void **slot; } syms[] = {
        {"ncclGetUniqueId", (void **)&api.GetUniqueId}, {"ncclCommInitRank", (void **)&api.CommInitRank},
        {"ncclCommDestroy", (void **)&api.CommDestroy}, {"ncclGroupStart", (void **)&api.GroupStart},
        {"ncclCommAbort", (void **)&api.CommAbort},
        {"ncclGroupEnd", (void **)&api.GroupEnd},       {"ncclSend", (void **)&api.Send},
        {"ncclRecv", (void **)&api.Recv},               {"ncclAllGather", (void **)&api.AllGather},
        {"ncclGetErrorString", (void **)&api.GetErrorString},
    };
    for (auto &s : syms) {
        *s.slot = dlsym(api.lib, s.name);
        if (!*s.slot) { api.err = std::string("librccl lacks ") + s.name; api.lib = nullptr; return &api; }
    }
    return &api;
}

}  // namespace dk

struct dk_comm {
    dk::ncclComm_t comm;
    uint32_t rank, world;
    void *staging;              // pool block for the received pieces
    uint64_t staging_bytes;
    bool dead;                  // a collective failed half-way: the communicator was aborted, every later call refuses
};

namespace dk {

// A failed RCCL call leaves the peers waiting in their half of the collective: the group (if one is open) is closed, the
// communicator aborted -- which releases the peers with an error of their own -- and marked dead.
inline dk_status comm_fail(dk_engine *e, bool in_group, const char *what, int r, const char *file, int line)
{
    RcclApi *api = rccl();
    dk_comm *c = e->comm;
    const std::string msg = api->GetErrorString ? api->GetErrorString(r) : "?";
    if (in_group) (void)api->GroupEnd();
    if (c && c->comm && !c->dead) {
        (void)api->CommAbort(c->comm);
        c->comm = nullptr;
        c->dead = true;
    }
    return fail(e, DK_ERR_HIP, "%s failed: %s (%s:%d); the communicator was aborted", what, msg.c_str(), file, line);
}

#define DK_RCCL(e, in_group, call)                                                            \
    do {                                                                                      \
        const int _r = (call);                                                                \
        if (_r != RCCL_SUCCESS) return comm_fail((e), (in_group), #call, _r, __FILE__, __LINE__); \
    } while (0)

// Where the pieces of an all-to-all over slices land in the staging buffer: rank r keeps its own slice in place and
// receives the piece of rank q at index (q < r ? q : q - 1) of P - 1 (one rank as its own peer: index 0 of 1).
// Host-only arithmetic, unit-tested on the CPU for P = 1..8 (tests/test_abi.py through dk_comm_layout).
inline uint64_t staging_index(uint64_t q, uint64_t r, uint64_t P) { return P == 1 ? 0 : q < r ? q : q - 1; }
inline uint64_t exchange_piece_bytes(uint64_t staging_bytes, uint64_t slice_bytes, uint64_t P, uint64_t granule)
{
    const uint64_t n_peers = P == 1 ? 1 : P - 1;
    uint64_t piece = staging_bytes / n_peers / granule * granule;
    if (piece > slice_bytes) piece = slice_bytes;
    return piece;
}

// the slices of the set are combined in pieces of `piece` bytes (a multiple of 64 KiB) so that the P - 1 received
// pieces fit the staging buffer
inline dk_status set_allreduce(dk_engine *e, dk_set *s, uint64_t *bytes_sent)
{
    dk_comm *c = e->comm;
    if (bytes_sent) *bytes_sent = 0;
    if (c && c->dead) return fail(e, DK_ERR_HIP, "the communicator was aborted by an earlier failure");
    if (!c || !c->comm) return DK_OK;           // no communicator, or one rank without RCCL
    RcclApi *api = rccl();
    const uint64_t P = c->world, r = c->rank;
    // A communicator of ONE rank (dk_comm_init with an id and world_size 1) runs every call below with the rank as its own
    // peer: its slice goes through ncclSend / ncclRecv into the staging buffer and is combined with itself (x | x = x, a
    // table united with itself), then the in-place all-gather -- the whole path on a single GPU, results unchanged.
    const bool own_peer = P == 1;
    const uint64_t n_peers = own_peer ? 1 : P - 1;
    if (s->n_bytes % (P * SEG_BYTES) != 0)
        return fail(e, DK_ERR_UNSUPPORTED, "the set (%llu bytes) does not split into whole 64-KiB segments over %llu ranks",
                    (unsigned long long)s->n_bytes, (unsigned long long)P);
    const uint64_t sl = s->n_bytes / P;
    const uint64_t piece = exchange_piece_bytes(c->staging_bytes, sl, P, SEG_BYTES);
    if (piece == 0) return fail(e, DK_ERR_INVALID_ARG, "staging buffer below 64 KiB per peer");
    char *base = (char *)s->d_words;
    const int T = (int)e->cfg.filter_log2_bits - 19;
    hipError_t h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
    if (h != hipSuccess) return fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h));
    for (uint64_t off = 0; off < sl; off += piece) {
        const uint64_t nb = std::min(piece, sl - off);
        DK_RCCL(e, false, api->GroupStart());
        for (uint64_t q = 0; q < P; q++) {
            if (q == r && !own_peer) continue;
            // piece of slice q goes to rank q; the same piece of my slice comes from rank q
            DK_RCCL(e, true, api->Send(base + q * sl + off, nb, RCCL_UINT8, (int)q, c->comm, e->stream));
            DK_RCCL(e, true, api->Recv((char *)c->staging + staging_index(q, r, P) * nb, nb, RCCL_UINT8, (int)q, c->comm, e->stream));
        }
        DK_RCCL(e, false, api->GroupEnd());
        char *dst = base + r * sl + off;
        if (s->exact) {
            const uint64_t first_seg = (r * sl + off) / SEG_BYTES, n_seg = nb / SEG_BYTES;
            if (e->cfg.k > 32)
                union_slices_kernel<true><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(
                    (unsigned long long *)dst, (const unsigned long long *)c->staging, n_peers, nb / 8, first_seg, T, e->d_ctr);
            else
                union_slices_kernel<false><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(
                    (unsigned long long *)dst, (const unsigned long long *)c->staging, n_peers, nb / 8, first_seg, T, e->d_ctr);
        } else {
            or_slices_kernel<<<grid_for(e, nb / 16, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                (uint4 *)dst, (const uint4 *)c->staging, n_peers, nb / 16);
        }
        h = hipGetLastError();
        if (h != hipSuccess) return fail(e, DK_ERR_HIP, "slice reduction failed: %s", hipGetErrorString(h));
    }
    DK_RCCL(e, false, api->AllGather(base + r * sl, base, sl, RCCL_UINT8, c->comm, e->stream));
    h = hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream);
    if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
    if (h != hipSuccess) return fail(e, DK_ERR_HIP, "all-reduce of the set failed: %s", hipGetErrorString(h));
    if (bytes_sent) *bytes_sent = 2 * (P - 1) * sl;
    return DK_OK;
}

// In-place all-to-all over the P equal slices of `base` (slice q goes to rank q; what rank q sends lands where slice q
// was): piece by piece through the staging buffer -- grouped ncclSend / ncclRecv, then device-to-device copies of the
// received pieces over the pieces just sent, all in stream order.  Afterwards slice q holds rank q's slice r.
inline dk_status alltoall_in_place(dk_engine *e, char *base, uint64_t slice_bytes, uint64_t granule, uint64_t *bytes_sent)
{
    dk_comm *c = e->comm;
    RcclApi *api = rccl();
    const uint64_t P = c->world, r = c->rank;
    const bool own_peer = P == 1;
    const uint64_t piece = exchange_piece_bytes(c->staging_bytes, slice_bytes, P, granule);
    if (slice_bytes && piece == 0) return fail(e, DK_ERR_INVALID_ARG, "staging buffer too small for one granule per peer");
    for (uint64_t off = 0; off < slice_bytes; off += piece) {
        const uint64_t nb = std::min(piece, slice_bytes - off);
        DK_RCCL(e, false, api->GroupStart());
        for (uint64_t q = 0; q < P; q++) {
            if (q == r && !own_peer) continue;
            DK_RCCL(e, true, api->Send(base + q * slice_bytes + off, nb, RCCL_UINT8, (int)q, c->comm, e->stream));
            DK_RCCL(e, true, api->Recv((char *)c->staging + staging_index(q, r, P) * nb, nb, RCCL_UINT8, (int)q, c->comm, e->stream));
        }
        DK_RCCL(e, false, api->GroupEnd());
        for (uint64_t q = 0; q < P; q++) {
            if (q == r && !own_peer) continue;
            DK_HIP(e, hipMemcpyAsync(base + q * slice_bytes + off, (char *)c->staging + staging_index(q, r, P) * nb, nb,
                                     hipMemcpyDeviceToDevice, e->stream));
        }
        if (bytes_sent) *bytes_sent += (own_peer ? 1 : P - 1) * nb;
    }
    return DK_OK;
}

}  // namespace dk
