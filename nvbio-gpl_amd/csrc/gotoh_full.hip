// gotoh_full.hip -- full-matrix Gotoh (placeholder until the kernel lands in this round)
#include "common.h"
using namespace nvbio_amd;
extern "C" nvbio_status nvbio_full_gotoh_temp_bytes(const nvbio_alignment_batch*, uint32_t, uint32_t, int, uint64_t* bytes)
{
    if (bytes) *bytes = 0;
    set_error( "nvbio_full_gotoh_temp_bytes: not built yet" );
    return NVBIO_ERR_UNSUPPORTED;
}
extern "C" nvbio_status nvbio_full_gotoh_score(int, nvbio_alignment_type, int, const nvbio_gotoh_scheme*, const nvbio_alignment_batch*,
                                               uint32_t, uint32_t, const int32_t*, int32_t*, nvbio_uint2*, void*, uint64_t, void*)
{
    set_error( "nvbio_full_gotoh_score: not built yet" );
    return NVBIO_ERR_UNSUPPORTED;
}
