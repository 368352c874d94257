// fm_build.hip -- GPU FM-index construction (placeholder until the builder lands in this round)
#include "common.h"
using namespace nvbio_amd;
extern "C" nvbio_status nvbio_fm_index_build(const uint32_t*, uint32_t, int, uint32_t, uint32_t, void*, nvbio_fm_index_t*)
{
    set_error( "nvbio_fm_index_build: not built yet" );
    return NVBIO_ERR_UNSUPPORTED;
}
