// dk_kernels_bucket.h -- "bucketed" kernel family (placeholder until the LDS-segment pipeline lands)
#pragma once
#include "dk_internal.h"

namespace dk {

inline bool bucketed_pays(const dk_engine *, uint64_t) { return false; }

inline dk_status bucketed_insert(dk_engine *e, dk_set *, const dk_reads *)
{
    return fail(e, DK_ERR_UNSUPPORTED, "bucketed kernels not built");
}

inline dk_status bucketed_probe(dk_engine *e, dk_set *, const dk_reads *, dk_result *)
{
    return fail(e, DK_ERR_UNSUPPORTED, "bucketed kernels not built");
}

}  // namespace dk
