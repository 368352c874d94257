// gotoh_common.h -- types shared by the banded and full-matrix Gotoh kernels.
#pragma once
#include "common.h"

namespace nvbio_amd {

// flattened stream of alignment jobs (see nvbio_alignment_batch)
struct BatchDev
{
    const void*     reads;
    const uint32_t* read_offsets;
    const uint8_t*  quals;
    const uint32_t* read_id;
    const uint8_t*  flags;
    const void*     text;
    const uint32_t* win_begin;
    const uint32_t* win_end;
    uint32_t        n;
    uint32_t        max_read_len;
};

struct SchemeDev
{
    int32_t match, mm_min, mm_max, pat_go, pat_ge, txt_go, txt_ge;
};

nvbio_status make_batch(const nvbio_alignment_batch* in, BatchDev* b);      // gotoh_banded.hip

// QualCost (nvBowtie/bowtie2/cuda/scoring.h:84-88) negated (:280-281); IEEE float ops, no contraction
__device__ __forceinline__ int32_t mismatch_score(const SchemeDev& sc, const uint32_t q)
{
    const int   qi   = (int)q < 40 ? (int)q : 40;
    const float frac = (float)qi / 40.0f;
    return -( sc.mm_min + (int)( frac * (float)(sc.mm_max - sc.mm_min) ) );
}

__device__ __forceinline__ int32_t max2(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t max3(int32_t a, int32_t b, int32_t c) { return max2( max2( a, b ), c ); }

} // namespace nvbio_amd
