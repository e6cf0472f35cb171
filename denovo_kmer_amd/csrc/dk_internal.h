// dk_internal.h -- host-side state behind the opaque handles of include/denovo_kmer.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/denovo_kmer.h"
#include "dk_kernels_direct.h"

struct dk_pool_block {
    void *ptr;
    size_t bytes;
    bool in_use;
};

struct dk_engine {
    dk_config cfg;
    int device;
    int n_cu;
    hipStream_t stream;
    bool own_stream;
    std::string err;
    dk::Counters *d_ctr;          // device counters of the running operation
    dk::Counters *h_ctr;          // pinned host mirror
    std::vector<dk_pool_block> pool;
    // stage timing of the last operation
    hipEvent_t ev[DK_MAX_STAGES + 1];
    int n_ev;
    char ev_name[DK_MAX_STAGES][24];
    dk_timings timings;
};

struct dk_reads {
    dk_engine *e;
    uint64_t *d_bases;
    uint64_t *d_mask;
    uint64_t n_bases, n_reads, n_windows;
    bool owns;
};

struct dk_set {
    dk_engine *e;
    unsigned long long *d_words;
    uint64_t n_bytes;
    bool owns;
    bool exact;                   // DK_SET_EXACT: d_words is read as open-addressing tables (dk_device.h)
};

struct dk_result {
    dk_engine *e;
    uint64_t *d_lo, *d_hi;
    uint32_t *d_cnt;
    uint64_t n;
    bool wide;
    bool owns;                    // false: the arrays belong to the caller (dk_result_attach)
    // Region r of the arrays holds region_n[r] entries starting at r * region_cap.  The direct
    // family writes one dense region; the bucketed count kernel appends through RESULT_REGIONS
    // independent fill counters (one global counter saturates near 10^8 atomics/s) and the regions
    // are stitched on copy-out, or compacted on the first dk_result_device_view.
    uint32_t n_regions;
    uint64_t region_cap;
    uint64_t region_n[dk::RESULT_REGIONS];
};

namespace dk {

dk_status fail(dk_engine *e, dk_status s, const char *fmt, ...);
dk_status pool_alloc(dk_engine *e, size_t bytes, void **out);
void pool_free(dk_engine *e, void *p);
void stage_begin(dk_engine *e);
void stage_mark(dk_engine *e, const char *name);
dk_status stage_end(dk_engine *e);      // synchronises the stream and fills e->timings
int grid_for(const dk_engine *e, uint64_t n_threads, int block);

}  // namespace dk

#define DK_HIP(e, call)                                                                      \
    do {                                                                                     \
        hipError_t _r = (call);                                                              \
        if (_r != hipSuccess)                                                                \
            return dk::fail((e), _r == hipErrorOutOfMemory ? DK_ERR_OOM : DK_ERR_HIP,        \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, \
                            __LINE__);                                                       \
    } while (0)

#define DK_TRY(call)                         \
    do {                                     \
        dk_status _s = (call);               \
        if (_s != DK_OK) return _s;          \
    } while (0)
