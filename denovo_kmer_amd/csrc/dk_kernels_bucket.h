// dk_kernels_bucket.h -- "bucketed" kernel family (k <= 32): every random access of the hot path
// is moved from HBM into LDS.
//
//   scan_part   packed stream -> canonical k-mer -> hash h (a bijection of the k-mer, so h IS the
//               record); LDS multisplit of a tile of 8192 positions by the top b1 hash bits,
//               runs written coalesced to per-bin regions in HBM
//   repart      second multisplit level by the next b2 bits (filters above 2^9 segments)
//   seg_insert  one workgroup per 64-KiB filter segment: segment -> LDS, ds_or per record, back
//   seg_probe   segment -> LDS, test per record, absent records compacted per segment
//   seg_count   absent records of a segment -> LDS hash table -> (k-mer, count) appended
//
// The filter bits produced are identical to the direct family's (same hash, same geometry), so
// both families are checked against the same oracle.  HBM traffic per k-mer:
// 3L/(8(L-k+1)) + 8 (scan_part) + 16 (repart) + 8 + filter/batch (seg_*) instead of one random
// 64-B block per k-mer.
//
// Parts: dk_bucket_common.h (records, overflow list, LDS multisplit), dk_bucket_scan.h (scan_part, kmers_tile),
// dk_bucket_repart.h, dk_bucket_seg.h (seg_insert / seg_probe / seg_exact_*), dk_bucket_count.h (seg_count),
// dk_bucket_rare.h (overflow and accumulator rare paths), dk_bucket_host.h (plans and stages).
#pragma once
#include "dk_bucket_host.h"
