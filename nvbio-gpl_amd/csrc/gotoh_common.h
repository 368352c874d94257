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
    uint32_t        algo;              // NVBIO_ALN_* (host-side kernel selection only)
};

// Scoring scheme as the kernels see it.  The reference's Gotoh kernels take BOTH gap recurrences from the pattern-gap terms and
// use the text-gap terms only to initialise the boundaries (gotoh_banded_inl.h:435-439,48; gotoh_inl.h:641-642,69-70); its
// linear-gap Smith-Waterman kernels charge `deletion` where the text advances alone and `insertion` where the pattern does, in
// the recurrences and on the boundaries alike (sw/sw_banded_inl.h:369-370,44; sw/sw_inl.h:69-70,628-629,650).  Both fit:
//   pat_go/pat_ge : the gap that consumes text only   (banded: F, from band j+1; full: along the text)
//   ins_go/ins_ge : the gap that consumes pattern only (banded: E, along the band; full: along the pattern)
//   txt_go/txt_ge : boundary that is read first (banded row zero, the full matrix's boundary column)
//   top_go/top_ge : the full matrix's stripe-top boundary
//   wide          : full matrix swept in logical stripes of 16 (sw_bandlen_selector, sw/sw_inl.h:1322-1325) instead of 8: it
//                   decides LOCAL ties between cells of equal score and where the early exit is tested
struct SchemeDev
{
    int32_t match, mm_min, mm_max, pat_go, pat_ge, txt_go, txt_ge;
    int32_t ins_go, ins_ge, top_go, top_ge, wide;
};

inline SchemeDev scheme_dev(const nvbio_gotoh_scheme* g)
{
    return SchemeDev{ g->match, g->mm_min, g->mm_max, g->pat_gap_open, g->pat_gap_ext, g->txt_gap_open, g->txt_gap_ext,
                      g->pat_gap_open, g->pat_gap_ext, g->pat_gap_open, g->pat_gap_ext, 0 };
}
// boundary_over_pattern: the first-read boundary runs along the pattern (full matrix, text blocking) instead of the text
inline SchemeDev scheme_dev(const nvbio_sw_scheme* w, const bool boundary_over_pattern)
{
    const int32_t D = w->deletion, I = w->insertion;
    const int32_t b = boundary_over_pattern ? I : D, t = boundary_over_pattern ? D : I;
    return SchemeDev{ w->match, -w->mismatch, -w->mismatch, D, D, b, b, I, I, t, t, 1 };
}
// true when the scheme is an ordinary reference-Gotoh one (what the packed kernels and the ungapped shortcuts assume)
inline bool plain_gotoh(const SchemeDev& sc)
{
    return !sc.wide && sc.ins_go == sc.pat_go && sc.ins_ge == sc.pat_ge && sc.top_go == sc.pat_go && sc.top_ge == sc.pat_ge;
}

nvbio_status make_batch(const nvbio_alignment_batch* in, BatchDev* b);      // gotoh_banded.hip
nvbio_status banded15_full_ties_traceback(const BatchDev& b, const SchemeDev& sc, const uint32_t rbits, const uint32_t tbits, const uint32_t max_jobs,
                                          const uint32_t* job_list, const uint32_t* job_count, uint32_t* dirs, const uint64_t dirs_bytes,
                                          int32_t* scores, uint2* sources, uint2* sinks, uint16_t* cigars, const uint32_t stride, uint32_t* lens,
                                          hipStream_t s);                   // gotoh_traceback.hip
bool banded31_packed_ok(const SchemeDev& sc, const uint32_t max_read_len);  // gotoh_banded.hip
void banded31_packed_launch(const BatchDev& b, const SchemeDev& sc, const uint32_t read_bits, const uint32_t max_jobs, int32_t* scores, uint2* sinks,
                            const uint32_t* job_list, const uint32_t* job_count, hipStream_t s);

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
