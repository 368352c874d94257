// oracle/ref/nvbio_ref.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A thin extern "C" driver around the *reference's own* host-callable template
// code (headers under /root/reference, compiled where they lie; nothing is
// copied).  It is built only in the development container by oracle/Makefile
// into oracle/_ref/libnvbio_ref.so and is used to
//   (1) pin the C restatement in oracle/nvbio_oracle.c, and
//   (2) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py).
//
// Reference entry points exercised (file:line relative to /root/reference):
//   gen_sa / gen_bwt_from_sa / gen_bwt_count_table    nvbio/fmindex/bwt.h:28-89
//   build_occurrence_table<64>                        nvbio/fmindex/rank_dictionary_inl.h:33-66
//   rank / rank4 / match / match_reverse / locate /
//   locate_ssa_iterator / lookup_ssa_iterator         nvbio/fmindex/fmindex_inl.h:27-460
//   dispatch_rank<2,64,...,uint4,uint4>               nvbio/fmindex/rank_dictionary_inl.h:338-479
//   (production interleaved bwt_occ layout)           nvbio/io/fmindex/fmindex.h:151-176
//   aln::banded_alignment_score<BAND>                 nvbio/alignment/gotoh/gotoh_banded_inl.h:397-688
//   aln::alignment_score (text / pattern blocking)    nvbio/alignment/gotoh/gotoh_inl.h:444-1256
//   aln::BestSink<int32>                              nvbio/alignment/sink_inl.h:31-49
//   priority_deque + interval heap                    nvbio/basic/priority_deque.h, interval_heap.h (the container of nvBowtie's
//                                                     seed-hit deques, seed_hit_deque_array.h:145; the element type SeedHit and its
//                                                     comparator live in nvBowtie/bowtie2/cuda/seed_hit.h, which pulls thrust device
//                                                     vectors and does not compile host-only: the driver passes a user-side element
//                                                     with the same two words and the same ordering, size(f) > size(s))
//
// The sampled suffix array type is a template parameter of nvbio::fm_index
// (interface: nvbio/fmindex/ssa.h:67-74).  nvbio/fmindex/ssa.h itself pulls in
// thrust::device_vector and does not compile host-only, so the driver passes a
// user-defined context with the same fetch()/has() semantics as
// SSA_index_multiple_context<16> (nvbio/fmindex/ssa_inl.h:477-495) backed by the
// SA the reference's own gen_sa() produced, with entry 0 = uint32(-1)
// (nvbio/fmindex/ssa_inl.h:299).
#include <nvbio/basic/types.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/basic/deinterleaved_iterator.h>
#include <nvbio/fmindex/bwt.h>
#include <nvbio/fmindex/rank_dictionary.h>
#include <nvbio/fmindex/fmindex.h>
#include <nvbio/fmindex/backtrack.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/basic/priority_deque.h>
#include <vector>
#include <cstring>
#include <cstdint>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace nvbio;

namespace {

struct SampledSA16
{
    typedef SampledSA16 context_type;
    const uint32* ssa;      // ssa[j] = SA[16*j], ssa[0] = uint32(-1)
    NVBIO_FORCEINLINE bool fetch(const uint32 i, uint32& r) const { if (i & 15u) return false; r = ssa[i >> 4]; return true; }
    NVBIO_FORCEINLINE bool has(const uint32 i) const { return (i & 15u) == 0; }
};

struct RefIndex
{
    uint32                n;
    uint32                primary;
    uint32                L2[5];
    std::vector<uint32>   text;        // packed 2-bit big-endian
    std::vector<uint32>   bwt;         // packed 2-bit big-endian
    std::vector<uint32>   occ;         // 4 counters per 64 symbols
    std::vector<uint32>   bwt_occ;     // interleaved production layout
    std::vector<int32>    sa;          // sa[0] = n (then -1), sa[1..n]
    std::vector<uint32>   ssa;         // every 16-th row
    std::vector<uint32>   count_table; // 256 entries

    typedef const uint4*                                         bwt_occ_type;
    typedef deinterleaved_iterator<2,0,bwt_occ_type>             bwt_type;
    typedef deinterleaved_iterator<2,1,bwt_occ_type>             occ_type;
    typedef PackedStream<bwt_type,uint8,2u,true>                 bwt_stream_type;
    typedef rank_dictionary<2u,64u,bwt_stream_type,occ_type,const uint32*> rank_dict_type;
    typedef fm_index<rank_dict_type,SampledSA16>                 fm_index_type;

    fm_index_type fmi() const
    {
        const bwt_occ_type base = (bwt_occ_type)&bwt_occ[0];
        SampledSA16 s; s.ssa = &ssa[0];
        return fm_index_type(
            n, primary, &L2[0],
            rank_dict_type( bwt_stream_type( bwt_type( base ) ), occ_type( base ), &count_table[0] ),
            s );
    }
};

// A user-side model of the reference's Gotoh scoring-scheme concept (the aligners are
// templated on it, nvbio/alignment/alignment.h:437-449) with the quality-ramp mismatch
// cost and the separate read/reference gap costs of nvBowtie's
// SmithWatermanScoringScheme<QualCost,ConstantCost> (nvBowtie/bowtie2/cuda/scoring.h:73-92,278-285).
// nvBowtie's own header cannot be compiled host-only (it includes thrust::device_vector
// through defs.h -> nvbio/basic/cuda/arch.h), so the float->int formula is restated here;
// what this pins is the DP's use of quals[i] and of the four gap costs.
struct QualRampScheme
{
    int32 m_match, m_mm_min, m_mm_max, m_pat_go, m_pat_ge, m_txt_go, m_txt_ge;

    NVBIO_FORCEINLINE int32 mmp(const int q) const
    {
        const float frac = (float)(nvbio::min( q, 40 ) / 40.0f);
        return m_mm_min + int( frac * (m_mm_max - m_mm_min) );
    }
    NVBIO_FORCEINLINE int32 match(const uint8 q = 0)      const { return m_match; }
    NVBIO_FORCEINLINE int32 mismatch(const uint8 q = 0)   const { return -mmp(q); }
    NVBIO_FORCEINLINE int32 mismatch(const uint8 a, const uint8 b, const uint8 q = 0) const { return -mmp(q); }
    NVBIO_FORCEINLINE int32 pattern_gap_open()            const { return m_pat_go; }
    NVBIO_FORCEINLINE int32 pattern_gap_extension()       const { return m_pat_ge; }
    NVBIO_FORCEINLINE int32 text_gap_open()               const { return m_txt_go; }
    NVBIO_FORCEINLINE int32 text_gap_extension()          const { return m_txt_ge; }
};

template <uint32 BAND, aln::AlignmentType TYPE, typename scheme_type, typename qual_type>
int banded_run(const scheme_type& scheme,
               const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
               int32* score, uint32* sink)
{
    typedef vector_view<const uint8*> string_type;
    aln::BestSink<int32> best;
    const bool ok = aln::banded_alignment_score<BAND>(
        aln::make_gotoh_aligner<TYPE>( scheme ),
        string_type( M, pat ),
        quals,
        string_type( N, txt ),
        min_score,
        best );
    *score = best.score; sink[0] = best.sink.x; sink[1] = best.sink.y;
    return ok ? 1 : 0;
}

template <uint32 BAND, typename scheme_type, typename qual_type>
int banded_type(int type, const scheme_type& scheme,
                const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                int32* score, uint32* sink)
{
    switch (type)
    {
    case 0: return banded_run<BAND,aln::GLOBAL>     ( scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 1: return banded_run<BAND,aln::LOCAL>      ( scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 2: return banded_run<BAND,aln::SEMI_GLOBAL>( scheme, pat, quals, M, txt, N, min_score, score, sink );
    }
    return -1;
}

template <typename scheme_type, typename qual_type>
int banded_band(uint32 band, int type, const scheme_type& scheme,
                const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                int32* score, uint32* sink)
{
    switch (band)
    {
    case 3:  return banded_type<3> ( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 7:  return banded_type<7> ( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 15: return banded_type<15>( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 31: return banded_type<31>( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    }
    return -1;
}

// The staged scheduler's scoring of one job: StagedAlignmentUnitBase::run (nvbio/alignment/batched_stream.h:145-180)
// calls BandedScoreUnit::execute (:259-284) = the windowed aln::banded_alignment_score<BAND> (banded_inl.h:179-208)
// over 32-row windows (WINDOW_SIZE, batched_stream.h:119) through one band of short2 checkpoints
// (batched_banded_inl.h:176-183), until a window returns false or the pattern is consumed.
template <uint32 BAND, aln::AlignmentType TYPE, typename scheme_type, typename qual_type>
int banded_staged_run(const scheme_type& scheme,
                      const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                      int32* score, uint32* sink)
{
    typedef vector_view<const uint8*> string_type;
    aln::BestSink<int32> best;
    short2 column[BAND];
    bool valid = true; uint32 windows = 0;
    for (uint32 window_begin = 0; valid; )
    {
        const uint32 window_end = nvbio::min( window_begin + 32u, M );
        valid = aln::banded_alignment_score<BAND>(
            aln::make_gotoh_aligner<TYPE>( scheme ),
            string_type( M, pat ),
            quals,
            string_type( N, txt ),
            min_score,
            window_begin,
            window_end,
            best,
            &column[0] );
        ++windows;
        if (window_end >= M) break;
        window_begin = window_end;
    }
    *score = best.score; sink[0] = best.sink.x; sink[1] = best.sink.y;
    return (int)windows * 2 + (valid ? 1 : 0);
}
template <uint32 BAND, typename scheme_type, typename qual_type>
int banded_staged_type(int type, const scheme_type& scheme,
                       const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                       int32* score, uint32* sink)
{
    switch (type)
    {
    case 0: return banded_staged_run<BAND,aln::GLOBAL>     ( scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 1: return banded_staged_run<BAND,aln::LOCAL>      ( scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 2: return banded_staged_run<BAND,aln::SEMI_GLOBAL>( scheme, pat, quals, M, txt, N, min_score, score, sink );
    }
    return -1;
}
template <typename scheme_type, typename qual_type>
int banded_staged_band(uint32 band, int type, const scheme_type& scheme,
                       const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                       int32* score, uint32* sink)
{
    switch (band)
    {
    case 3:  return banded_staged_type<3> ( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 7:  return banded_staged_type<7> ( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 15: return banded_staged_type<15>( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 31: return banded_staged_type<31>( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    }
    return -1;
}

// A Backtracer model (concept: nvbio/alignment/alignment.h:130-140) that records exactly what the
// reference's traceback hands it: the ops in backtracking order and the two clip() calls.
struct RecordingBacktracer
{
    uint8* ops; uint32 cap; uint32 n; uint32 clips[2]; uint32 n_clips;
    void clip(const uint32 l) { if (n_clips < 2) clips[n_clips] = l; ++n_clips; }
    void push(const uint8 op) { if (n < cap) ops[n] = op; ++n; }
};

// aln::banded_alignment_traceback<BAND,MAX_PATTERN_LEN,CHECKPOINTS> (nvbio/alignment/banded_inl.h:440-483 ->
// :354-417 -> gotoh/gotoh_banded_inl.h:730-950); CHECKPOINTS = 16 as nvBowtie (BANDED_DP_CHECKPOINTS, defs.h:95)
template <uint32 BAND, aln::AlignmentType TYPE, typename scheme_type, typename qual_type>
int banded_tb_run(const scheme_type& scheme,
                  const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                  int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    typedef vector_view<const uint8*> string_type;
    const aln::Alignment<int32> a = aln::banded_alignment_traceback<BAND,1024u,16u>(
        aln::make_gotoh_aligner<TYPE>( scheme ),
        string_type( M, pat ),
        quals,
        string_type( N, txt ),
        min_score,
        bt );
    *score = a.score; source[0] = a.source.x; source[1] = a.source.y; sink[0] = a.sink.x; sink[1] = a.sink.y;
    return (int)bt.n_clips;
}

template <uint32 BAND, typename scheme_type, typename qual_type>
int banded_tb_type(int type, const scheme_type& scheme,
                   const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                   int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    switch (type)
    {
    case 0: return banded_tb_run<BAND,aln::GLOBAL>     ( scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    case 1: return banded_tb_run<BAND,aln::LOCAL>      ( scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    case 2: return banded_tb_run<BAND,aln::SEMI_GLOBAL>( scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    }
    return -1;
}

template <typename scheme_type, typename qual_type>
int banded_tb_band(uint32 band, int type, const scheme_type& scheme,
                   const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                   int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    switch (band)
    {
    case 3:  return banded_tb_type<3> ( type, scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    case 7:  return banded_tb_type<7> ( type, scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    case 15: return banded_tb_type<15>( type, scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    case 31: return banded_tb_type<31>( type, scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    }
    return -1;
}

template <aln::AlignmentType TYPE, typename algorithm_tag, typename scheme_type, typename qual_type>
int full_run(const scheme_type& scheme,
             const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
             int32* score, uint32* sink)
{
    typedef vector_view<const uint8*> string_type;
    typedef aln::GotohAligner<TYPE,scheme_type,algorithm_tag>       aligner_type;
    typedef typename aln::column_storage_type<aligner_type>::type   cell_type;

    // the column must be as large as the text (pattern blocking) or the pattern (text blocking)
    std::vector<cell_type> column( (M > N ? M : N) + 16u );

    aln::BestSink<int32> best;
    const bool ok = aln::alignment_score(
        aligner_type( scheme ),
        string_type( M, pat ),
        quals,
        string_type( N, txt ),
        min_score,
        best,
        &column[0] );
    *score = best.score; sink[0] = best.sink.x; sink[1] = best.sink.y;
    return ok ? 1 : 0;
}

template <typename algorithm_tag, typename scheme_type, typename qual_type>
int full_type(int type, const scheme_type& scheme,
              const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
              int32* score, uint32* sink)
{
    switch (type)
    {
    case 0: return full_run<aln::GLOBAL,algorithm_tag>     ( scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 1: return full_run<aln::LOCAL,algorithm_tag>      ( scheme, pat, quals, M, txt, N, min_score, score, sink );
    case 2: return full_run<aln::SEMI_GLOBAL,algorithm_tag>( scheme, pat, quals, M, txt, N, min_score, score, sink );
    }
    return -1;
}

} // anonymous namespace

// banded edit distance: aln::banded_alignment_score<BAND>( EditDistanceAligner<TYPE>, ... ) (nvbio/alignment/ed/ed_banded_inl.h:37-69 ->
// sw/sw_banded_inl.h:281-520 with EditDistanceSWScheme) -- the aligner of examples/fmmap/fmmap.cu:346-359 and of nvBowtie --scoring ed
template <uint32 BAND, aln::AlignmentType TYPE>
int banded_ed_run(const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32* score, uint32* sink)
{
    typedef vector_view<const uint8*> string_type;
    aln::BestSink<int32> best;
    const bool ok = aln::banded_alignment_score<BAND>(
        aln::make_edit_distance_aligner<TYPE>(),
        string_type( M, pat ),
        aln::trivial_quality_string(),
        string_type( N, txt ),
        Field_traits<int32>::min(),
        best );
    *score = best.score; sink[0] = best.sink.x; sink[1] = best.sink.y;
    return ok ? 1 : 0;
}
template <uint32 BAND>
int banded_ed_type(int type, const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32* score, uint32* sink)
{
    switch (type)
    {
    case 0: return banded_ed_run<BAND,aln::GLOBAL>     ( pat, M, txt, N, score, sink );
    case 1: return banded_ed_run<BAND,aln::LOCAL>      ( pat, M, txt, N, score, sink );
    case 2: return banded_ed_run<BAND,aln::SEMI_GLOBAL>( pat, M, txt, N, score, sink );
    }
    return -1;
}

// the Myers bit-vector aligner, aln::banded_alignment_score<BAND>( EditDistanceAligner<TYPE, MyersTag<5> >, ... ) (nvbio/alignment/myers/
// myers_banded_inl.h:247-342): the aligner examples/fmmap/fmmap.cu:346-359 actually instantiates (GLOBAL and SEMI_GLOBAL only)
template <uint32 BAND, aln::AlignmentType TYPE>
int banded_myers_run(const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32 min_score, int32* score, uint32* sink)
{
    typedef vector_view<const uint8*> string_type;
    aln::BestSink<int32> best;
    const bool ok = aln::banded_alignment_score<BAND>(
        aln::make_edit_distance_aligner<TYPE, aln::MyersTag<5u> >(),
        string_type( M, pat ),
        aln::trivial_quality_string(),
        string_type( N, txt ),
        min_score,
        best );
    *score = best.score; sink[0] = best.sink.x; sink[1] = best.sink.y;
    return ok ? 1 : 0;
}

// Best2Sink<int32> through the same dispatches (banded and full matrix)
template <uint32 BAND, aln::AlignmentType TYPE>
int banded_best2_run(const QualRampScheme& scheme, const uint8* pat, const uint8* quals, uint32 M, const uint8* txt, uint32 N, uint32 dist, int64* out)
{
    typedef vector_view<const uint8*> string_type;
    aln::Best2Sink<int32> best( dist );
    bool ok;
    if (quals) ok = aln::banded_alignment_score<BAND>( aln::make_gotoh_aligner<TYPE>( scheme ), string_type( M, pat ), quals, string_type( N, txt ), Field_traits<int32>::min(), best );
    else       ok = aln::banded_alignment_score<BAND>( aln::make_gotoh_aligner<TYPE>( scheme ), string_type( M, pat ), aln::trivial_quality_string(), string_type( N, txt ), Field_traits<int32>::min(), best );
    out[0] = best.score1; out[1] = best.sink1.x; out[2] = best.sink1.y; out[3] = best.score2; out[4] = best.sink2.x; out[5] = best.sink2.y;
    return ok ? 1 : 0;
}
template <uint32 BAND>
int banded_best2_type(int type, const QualRampScheme& scheme, const uint8* pat, const uint8* quals, uint32 M, const uint8* txt, uint32 N, uint32 dist, int64* out)
{
    switch (type)
    {
    case 0: return banded_best2_run<BAND,aln::GLOBAL>     ( scheme, pat, quals, M, txt, N, dist, out );
    case 1: return banded_best2_run<BAND,aln::LOCAL>      ( scheme, pat, quals, M, txt, N, dist, out );
    case 2: return banded_best2_run<BAND,aln::SEMI_GLOBAL>( scheme, pat, quals, M, txt, N, dist, out );
    }
    return -1;
}
template <aln::AlignmentType TYPE, typename tag>
int full_best2_run(const QualRampScheme& scheme, const uint8* pat, const uint8* quals, uint32 M, const uint8* txt, uint32 N, int32 min_score, uint32 dist, int64* out)
{
    typedef vector_view<const uint8*> string_type;
    typedef aln::GotohAligner<TYPE,QualRampScheme,tag>              aligner_type;
    typedef typename aln::column_storage_type<aligner_type>::type   cell_type;
    std::vector<cell_type> column( (M > N ? M : N) + 16u );
    aln::Best2Sink<int32> best( dist );
    bool ok;
    if (quals) ok = aln::alignment_score( aligner_type( scheme ), string_type( M, pat ), quals, string_type( N, txt ), min_score, best, &column[0] );
    else       ok = aln::alignment_score( aligner_type( scheme ), string_type( M, pat ), aln::trivial_quality_string(), string_type( N, txt ), min_score, best, &column[0] );
    out[0] = best.score1; out[1] = best.sink1.x; out[2] = best.sink1.y; out[3] = best.score2; out[4] = best.sink2.x; out[5] = best.sink2.y;
    return ok ? 1 : 0;
}
template <typename tag>
int full_best2_type(int type, const QualRampScheme& scheme, const uint8* pat, const uint8* quals, uint32 M, const uint8* txt, uint32 N, int32 min_score, uint32 dist, int64* out)
{
    switch (type)
    {
    case 0: return full_best2_run<aln::GLOBAL,tag>     ( scheme, pat, quals, M, txt, N, min_score, dist, out );
    case 1: return full_best2_run<aln::LOCAL,tag>      ( scheme, pat, quals, M, txt, N, min_score, dist, out );
    case 2: return full_best2_run<aln::SEMI_GLOBAL,tag>( scheme, pat, quals, M, txt, N, min_score, dist, out );
    }
    return -1;
}

// the other two aligner families through the same entry points: SmithWatermanAligner<TYPE,SimpleSmithWatermanScheme>
// (linear gaps; sw/sw_banded_inl.h:281-520, sw/sw_inl.h) and the full-matrix EditDistanceAligner (ed/ed_inl.h -> sw/sw_inl.h
// with EditDistanceSWScheme); column cells are int16 (nvbio/alignment/utils.h:55-56)
template <uint32 BAND, aln::AlignmentType TYPE>
int banded_sw_run(const aln::SimpleSmithWatermanScheme& scheme, const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                  int32* score, uint32* sink)
{
    typedef vector_view<const uint8*> string_type;
    aln::BestSink<int32> best;
    const bool ok = aln::banded_alignment_score<BAND>(
        aln::make_smith_waterman_aligner<TYPE>( scheme ),
        string_type( M, pat ),
        aln::trivial_quality_string(),
        string_type( N, txt ),
        min_score,
        best );
    *score = best.score; sink[0] = best.sink.x; sink[1] = best.sink.y;
    return ok ? 1 : 0;
}
template <uint32 BAND>
int banded_sw_type(int type, const aln::SimpleSmithWatermanScheme& scheme, const uint8* pat, uint32 M, const uint8* txt, uint32 N,
                   int32 min_score, int32* score, uint32* sink)
{
    switch (type)
    {
    case 0: return banded_sw_run<BAND,aln::GLOBAL>     ( scheme, pat, M, txt, N, min_score, score, sink );
    case 1: return banded_sw_run<BAND,aln::LOCAL>      ( scheme, pat, M, txt, N, min_score, score, sink );
    case 2: return banded_sw_run<BAND,aln::SEMI_GLOBAL>( scheme, pat, M, txt, N, min_score, score, sink );
    }
    return -1;
}
template <typename aligner_type>
int full_aligner_run(const aligner_type& aligner, const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                     int32* score, uint32* sink)
{
    typedef vector_view<const uint8*> string_type;
    typedef typename aln::column_storage_type<aligner_type>::type cell_type;
    std::vector<cell_type> column( (M > N ? M : N) + 16u );
    aln::BestSink<int32> best;
    const bool ok = aln::alignment_score(
        aligner,
        string_type( M, pat ),
        aln::trivial_quality_string(),
        string_type( N, txt ),
        min_score,
        best,
        &column[0] );
    *score = best.score; sink[0] = best.sink.x; sink[1] = best.sink.y;
    return ok ? 1 : 0;
}
template <typename tag>
int full_sw_type(int type, const aln::SimpleSmithWatermanScheme& scheme, const uint8* pat, uint32 M, const uint8* txt, uint32 N,
                 int32 min_score, int32* score, uint32* sink)
{
    typedef aln::SimpleSmithWatermanScheme S;
    switch (type)
    {
    case 0: return full_aligner_run( aln::SmithWatermanAligner<aln::GLOBAL,S,tag>( scheme ),      pat, M, txt, N, min_score, score, sink );
    case 1: return full_aligner_run( aln::SmithWatermanAligner<aln::LOCAL,S,tag>( scheme ),       pat, M, txt, N, min_score, score, sink );
    case 2: return full_aligner_run( aln::SmithWatermanAligner<aln::SEMI_GLOBAL,S,tag>( scheme ), pat, M, txt, N, min_score, score, sink );
    }
    return -1;
}
template <typename tag>
int full_ed_type(int type, const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32 min_score, int32* score, uint32* sink)
{
    switch (type)
    {
    case 0: return full_aligner_run( aln::EditDistanceAligner<aln::GLOBAL,tag>(),      pat, M, txt, N, min_score, score, sink );
    case 1: return full_aligner_run( aln::EditDistanceAligner<aln::LOCAL,tag>(),       pat, M, txt, N, min_score, score, sink );
    case 2: return full_aligner_run( aln::EditDistanceAligner<aln::SEMI_GLOBAL,tag>(), pat, M, txt, N, min_score, score, sink );
    }
    return -1;
}

// aln::alignment_traceback<MAX_PATTERN_LEN,MAX_TEXT_LEN,CHECKPOINTS> (nvbio/alignment/alignment_inl.h:478-517 -> :355-455 ->
// gotoh/gotoh_inl.h:1573-1640); CHECKPOINTS = 64 as nvBowtie (FULL_DP_CHECKPOINTS, defs.h:96); M <= 256, N <= 1024
template <aln::AlignmentType TYPE, typename scheme_type, typename qual_type>
int full_tb_run(const scheme_type& scheme,
                const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    typedef vector_view<const uint8*> string_type;
    const aln::Alignment<int32> a = aln::alignment_traceback<256u,1024u,64u>(
        aln::make_gotoh_aligner<TYPE>( scheme ),
        string_type( M, pat ),
        quals,
        string_type( N, txt ),
        min_score,
        bt );
    *score = a.score; source[0] = a.source.x; source[1] = a.source.y; sink[0] = a.sink.x; sink[1] = a.sink.y;
    return (int)bt.n_clips;
}
template <typename scheme_type, typename qual_type>
int full_tb_type(int type, const scheme_type& scheme,
                 const uint8* pat, const qual_type quals, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                 int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    switch (type)
    {
    case 0: return full_tb_run<aln::GLOBAL>     ( scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    case 1: return full_tb_run<aln::LOCAL>      ( scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    case 2: return full_tb_run<aln::SEMI_GLOBAL>( scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt );
    }
    return -1;
}

// the banded traceback of the linear-gap Smith-Waterman aligner: aln::banded_alignment_traceback<BAND,1024,16>( SmithWatermanAligner<TYPE>, ... )
// (nvbio/alignment/banded_inl.h:354-417 over sw/sw_banded_inl.h); sw[4] = { match, mismatch, deletion, insertion }; outputs as
// ref_banded_gotoh_traceback_ex
template <uint32 BAND, aln::AlignmentType TYPE>
int banded_sw_tb_run(const aln::SimpleSmithWatermanScheme& scheme, const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                     int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    typedef vector_view<const uint8*> string_type;
    const aln::Alignment<int32> a = aln::banded_alignment_traceback<BAND,1024u,16u>(
        aln::make_smith_waterman_aligner<TYPE>( scheme ),
        string_type( M, pat ),
        aln::trivial_quality_string(),
        string_type( N, txt ),
        min_score,
        bt );
    *score = a.score; source[0] = a.source.x; source[1] = a.source.y; sink[0] = a.sink.x; sink[1] = a.sink.y;
    return (int)bt.n_clips;
}
template <uint32 BAND>
int banded_sw_tb_type(int type, const aln::SimpleSmithWatermanScheme& scheme, const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                      int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    switch (type)
    {
    case 0: return banded_sw_tb_run<BAND,aln::GLOBAL>     ( scheme, pat, M, txt, N, min_score, score, source, sink, bt );
    case 1: return banded_sw_tb_run<BAND,aln::LOCAL>      ( scheme, pat, M, txt, N, min_score, score, source, sink, bt );
    case 2: return banded_sw_tb_run<BAND,aln::SEMI_GLOBAL>( scheme, pat, M, txt, N, min_score, score, source, sink, bt );
    }
    return -1;
}

// the full-matrix traceback of the linear-gap Smith-Waterman aligner: aln::alignment_traceback<256,1024,64>( SmithWatermanAligner<TYPE>, ... )
// (nvbio/alignment/alignment_inl.h:478-517 -> :355-455 over sw/sw_inl.h:1476-1694); outputs as ref_full_gotoh_traceback_ex
template <aln::AlignmentType TYPE>
int full_sw_tb_run(const aln::SimpleSmithWatermanScheme& scheme, const uint8* pat, uint32 M, const uint8* txt, uint32 N, int32 min_score,
                   int32* score, uint32* source, uint32* sink, RecordingBacktracer& bt)
{
    typedef vector_view<const uint8*> string_type;
    const aln::Alignment<int32> a = aln::alignment_traceback<256u,1024u,64u>(
        aln::make_smith_waterman_aligner<TYPE>( scheme ),
        string_type( M, pat ),
        aln::trivial_quality_string(),
        string_type( N, txt ),
        min_score,
        bt );
    *score = a.score; source[0] = a.source.x; source[1] = a.source.y; sink[0] = a.sink.x; sink[1] = a.sink.y;
    return (int)bt.n_clips;
}

extern "C" {

// Build an FM-index over text[0,n) (one 2-bit symbol per byte) exactly as
// nvbio-test/fmindex_test.cu:441-534 does.
void* ref_fm_create(const uint8_t* text_bytes, uint32_t n)
{
    RefIndex* idx = new RefIndex;
    idx->n = n;
    const uint32 words = ((n + 15u) / 16u + 3u) & ~3u;
    // +4 words of slack: gen_bwt_from_sa writes the (n+1)-th symbol before squeezing '$' out
    idx->text.assign( words + 4u, 0u );
    idx->bwt.assign(  words + 4u, 0u );
    idx->occ.assign(  words, 0u );     // occ words == bwt words (4 per 64 symbols)
    idx->count_table.resize( 256 );

    typedef PackedStream<uint32*,uint8,2,true,uint32> stream_type;
    stream_type text( &idx->text[0] );
    for (uint32 i = 0; i < n; ++i)
        text[i] = text_bytes[i] & 3u;

    idx->sa.assign( n + 1u, 0 );
    gen_sa( n, text, &idx->sa[0] );

    stream_type bwt( &idx->bwt[0] );
    idx->primary = gen_bwt_from_sa( n, text, &idx->sa[0], bwt );
    idx->sa[0] = -1;

    build_occurrence_table<64u>( bwt, bwt + n, &idx->occ[0], &idx->L2[1] );
    idx->L2[0] = 0;
    for (uint32 c = 0; c < 4; ++c)
        idx->L2[c+1] += idx->L2[c];

    gen_bwt_count_table( &idx->count_table[0] );

    // interleave as nvbio/io/fmindex/fmindex_impl.cu:300-313 / fmindex_test.cu:512-524
    idx->bwt_occ.assign( size_t(words) * 2u, 0u );
    for (uint32 w = 0; w < words; w += 4)
    {
        for (uint32 k = 0; k < 4; ++k)
        {
            idx->bwt_occ[ w*2 + k     ] = idx->bwt[ w + k ];
            idx->bwt_occ[ w*2 + 4 + k ] = idx->occ[ w + k ];
        }
    }

    idx->ssa.assign( (n + 16u) / 16u, 0u );
    for (uint32 i = 0; i <= n; i += 16)
        idx->ssa[i >> 4] = uint32( idx->sa[i] );
    return idx;
}

// Adopt an index built elsewhere (e.g. on the GPU by nvbio_fm_index_build) in the reference's
// production layout, so that the reference's own host code can be run -- and timed -- on it.
void* ref_fm_adopt(uint32_t n, uint32_t primary, const uint32_t* L2, const uint32_t* bwt_occ, uint64_t bwt_occ_words,
                   const uint32_t* ssa, uint64_t ssa_words)
{
    RefIndex* idx = new RefIndex;
    idx->n = n; idx->primary = primary;
    for (int c = 0; c < 5; ++c) idx->L2[c] = L2[c];
    idx->bwt_occ.assign( bwt_occ, bwt_occ + bwt_occ_words );
    idx->ssa.assign( ssa, ssa + ssa_words );
    idx->occ.resize( bwt_occ_words / 2 );
    idx->count_table.resize( 256 );
    gen_bwt_count_table( &idx->count_table[0] );
    return idx;
}

int ref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ref_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads( n );
#else
    (void)n;
#endif
}

void ref_fm_destroy(void* h) { delete (RefIndex*)h; }

uint32_t ref_fm_primary(void* h) { return ((RefIndex*)h)->primary; }
void     ref_fm_L2(void* h, uint32_t* out) { memcpy( out, ((RefIndex*)h)->L2, 5*sizeof(uint32) ); }
uint32_t ref_fm_words(void* h) { return uint32( ((RefIndex*)h)->occ.size() ); }
uint32_t ref_fm_ssa_words(void* h) { return uint32( ((RefIndex*)h)->ssa.size() ); }

void ref_fm_export(void* h, uint32_t* bwt, uint32_t* occ, uint32_t* bwt_occ, int32_t* sa, uint32_t* ssa)
{
    RefIndex* idx = (RefIndex*)h;
    if (bwt)     memcpy( bwt,     &idx->bwt[0],     idx->occ.size()*4 );
    if (occ)     memcpy( occ,     &idx->occ[0],     idx->occ.size()*4 );
    if (bwt_occ) memcpy( bwt_occ, &idx->bwt_occ[0], idx->bwt_occ.size()*4 );
    if (sa)      memcpy( sa,      &idx->sa[0],      idx->sa.size()*4 );
    if (ssa)     memcpy( ssa,     &idx->ssa[0],     idx->ssa.size()*4 );
}

uint32_t ref_fm_rank(void* h, uint32_t k, uint32_t c)
{
    return rank( ((RefIndex*)h)->fmi(), k, uint8(c) );
}
void ref_fm_rank_range(void* h, uint32_t l, uint32_t r, uint32_t c, uint32_t* out)
{
    const uint2 res = rank( ((RefIndex*)h)->fmi(), make_uint2( l, r ), uint8(c) );
    out[0] = res.x; out[1] = res.y;
}
void ref_fm_rank4(void* h, uint32_t k, uint32_t* out)
{
    const uint4 r = rank4( ((RefIndex*)h)->fmi(), k );
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// queries: one symbol per byte (values > 3 are 'N'), offsets[n_q+1]
// nvbio::hamming_backtrack (nvbio/fmindex/backtrack.h:51-157) exactly as the reference's benchmark calls it (count_core,
// nvbio-test/fmindex_test.cu:744-767): the pattern is an iterator into a 2-bit PackedStream of all reads (32-bit index
// arithmetic), the stack holds 128 uint4, the delegate adds up the range sizes -- here it also records the ranges.
struct RecordingCountDelegate
{
    uint32 count, n; uint32* ranges; uint32 cap;
    void operator() (const uint2 range)
    {
        count += range.y + 1u - range.x;
        if (ranges && n < cap) { ranges[2*n] = range.x; ranges[2*n+1] = range.y; }
        ++n;
    }
};
uint32_t ref_hamming_backtrack(void* h, const uint32_t* stream_words, uint32_t begin, uint32_t len, uint32_t seed, uint32_t mismatches,
                               uint32_t* count, uint32_t* ranges, uint32_t cap)
{
    const RefIndex::fm_index_type fmi = ((RefIndex*)h)->fmi();
    typedef PackedStream<const uint32*,uint8,2u,true> stream_type;
    const stream_type stream( stream_words );
    uint4 stack[32*4 + 1024];                      // the benchmark's 32*4 plus slack: an overrun must not corrupt the driver
    RecordingCountDelegate d; d.count = 0; d.n = 0; d.ranges = ranges; d.cap = cap;
    hamming_backtrack( fmi, stream.begin() + begin, len, seed, mismatches, stack, d );
    *count = d.count;
    return d.n;
}

void ref_fm_match(void* h, const uint8_t* syms, const uint32_t* offsets, uint32_t n_q, uint32_t* ranges, int reverse)
{
    const RefIndex::fm_index_type fmi = ((RefIndex*)h)->fmi();
    #pragma omp parallel for
    for (int64 q = 0; q < int64(n_q); ++q)
    {
        const uint8* p   = syms + offsets[q];
        const uint32 len = offsets[q+1] - offsets[q];
        const uint2 r = reverse ? match_reverse( fmi, p, len ) : match( fmi, p, len );
        ranges[2*q] = r.x; ranges[2*q+1] = r.y;
    }
}

void ref_fm_locate(void* h, const uint32_t* rows, uint32_t n, uint32_t* pos)
{
    const RefIndex::fm_index_type fmi = ((RefIndex*)h)->fmi();
    #pragma omp parallel for
    for (int64 i = 0; i < int64(n); ++i)
        pos[i] = locate( fmi, rows[i] );
}

void ref_fm_locate_ssa(void* h, const uint32_t* rows, uint32_t n, uint32_t* jt)
{
    const RefIndex::fm_index_type fmi = ((RefIndex*)h)->fmi();
    #pragma omp parallel for
    for (int64 i = 0; i < int64(n); ++i)
    {
        const uint2 r = locate_ssa_iterator( fmi, rows[i] );
        jt[2*i] = r.x; jt[2*i+1] = r.y;
    }
}

void ref_fm_lookup_ssa(void* h, const uint32_t* jt, uint32_t n, uint32_t* pos)
{
    const RefIndex::fm_index_type fmi = ((RefIndex*)h)->fmi();
    for (uint32 i = 0; i < n; ++i)
        pos[i] = lookup_ssa_iterator( fmi, make_uint2( jt[2*i], jt[2*i+1] ) );
}

// type: 0 GLOBAL, 1 LOCAL, 2 SEMI_GLOBAL (aln::AlignmentType, nvbio/alignment/alignment.h:242)
int ref_banded_gotoh(uint32_t band, int type, int match, int mm, int gap_open, int gap_ext,
                     const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                     int32_t* score, uint32_t* sink)
{
    const aln::SimpleGotohScheme scheme( match, mm, gap_open, gap_ext );
    return banded_band( band, type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, sink );
}

// general scheme: sc[7] = { match, mm_min, mm_max, pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext }
// (mm_min/mm_max are positive penalties as in nvBowtie; gap costs are signed scores); quals may be NULL.
int ref_banded_gotoh_ex(uint32_t band, int type, const int32_t* sc,
                        const uint8_t* pat, const uint8_t* quals, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                        int32_t* score, uint32_t* sink)
{
    QualRampScheme scheme; scheme.m_match = sc[0]; scheme.m_mm_min = sc[1]; scheme.m_mm_max = sc[2];
    scheme.m_pat_go = sc[3]; scheme.m_pat_ge = sc[4]; scheme.m_txt_go = sc[5]; scheme.m_txt_ge = sc[6];
    return quals ?
        banded_band( band, type, scheme, pat, quals, M, txt, N, min_score, score, sink ) :
        banded_band( band, type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, sink );
}

// the same through the staged scheduler's 32-row windows (see banded_staged_run); returns 2*windows run + (1 if the last
// window returned true)
int ref_banded_gotoh_staged_ex(uint32_t band, int type, const int32_t* sc,
                               const uint8_t* pat, const uint8_t* quals, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                               int32_t* score, uint32_t* sink)
{
    QualRampScheme scheme; scheme.m_match = sc[0]; scheme.m_mm_min = sc[1]; scheme.m_mm_max = sc[2];
    scheme.m_pat_go = sc[3]; scheme.m_pat_ge = sc[4]; scheme.m_txt_go = sc[5]; scheme.m_txt_ge = sc[6];
    return quals ?
        banded_staged_band( band, type, scheme, pat, quals, M, txt, N, min_score, score, sink ) :
        banded_staged_band( band, type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, sink );
}

int ref_banded_ed(uint32_t band, int type, const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t* score, uint32_t* sink)
{
    switch (band)
    {
    case 3:  return banded_ed_type<3> ( type, pat, M, txt, N, score, sink );
    case 7:  return banded_ed_type<7> ( type, pat, M, txt, N, score, sink );
    case 15: return banded_ed_type<15>( type, pat, M, txt, N, score, sink );
    case 31: return banded_ed_type<31>( type, pat, M, txt, N, score, sink );
    }
    return -1;
}

int ref_banded_gotoh_best2(uint32_t band, int type, const int32_t* sc, const uint8_t* pat, const uint8_t* quals, uint32_t M,
                           const uint8_t* txt, uint32_t N, uint32_t dist, int64_t* out)
{
    QualRampScheme scheme; scheme.m_match = sc[0]; scheme.m_mm_min = sc[1]; scheme.m_mm_max = sc[2];
    scheme.m_pat_go = sc[3]; scheme.m_pat_ge = sc[4]; scheme.m_txt_go = sc[5]; scheme.m_txt_ge = sc[6];
    switch (band)
    {
    case 3:  return banded_best2_type<3> ( type, scheme, pat, quals, M, txt, N, dist, (int64*)out );
    case 7:  return banded_best2_type<7> ( type, scheme, pat, quals, M, txt, N, dist, (int64*)out );
    case 15: return banded_best2_type<15>( type, scheme, pat, quals, M, txt, N, dist, (int64*)out );
    case 31: return banded_best2_type<31>( type, scheme, pat, quals, M, txt, N, dist, (int64*)out );
    }
    return -1;
}
int ref_full_gotoh_best2(int type, int blocking, const int32_t* sc, const uint8_t* pat, const uint8_t* quals, uint32_t M,
                         const uint8_t* txt, uint32_t N, int32_t min_score, uint32_t dist, int64_t* out)
{
    QualRampScheme scheme; scheme.m_match = sc[0]; scheme.m_mm_min = sc[1]; scheme.m_mm_max = sc[2];
    scheme.m_pat_go = sc[3]; scheme.m_pat_ge = sc[4]; scheme.m_txt_go = sc[5]; scheme.m_txt_ge = sc[6];
    return blocking ? full_best2_type<aln::TextBlockingTag>   ( type, scheme, pat, quals, M, txt, N, min_score, dist, (int64*)out )
                    : full_best2_type<aln::PatternBlockingTag>( type, scheme, pat, quals, M, txt, N, min_score, dist, (int64*)out );
}

int ref_banded_sw(uint32_t band, int type, int match, int mm, int del, int ins,
                  const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score, int32_t* score, uint32_t* sink)
{
    const aln::SimpleSmithWatermanScheme scheme( match, mm, del, ins );
    switch (band)
    {
    case 3:  return banded_sw_type<3> ( type, scheme, pat, M, txt, N, min_score, score, sink );
    case 7:  return banded_sw_type<7> ( type, scheme, pat, M, txt, N, min_score, score, sink );
    case 15: return banded_sw_type<15>( type, scheme, pat, M, txt, N, min_score, score, sink );
    case 31: return banded_sw_type<31>( type, scheme, pat, M, txt, N, min_score, score, sink );
    }
    return -1;
}

// blocking: 0 = PatternBlockingTag, 1 = TextBlockingTag
int ref_full_sw(int type, int blocking, int match, int mm, int del, int ins,
                const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score, int32_t* score, uint32_t* sink)
{
    const aln::SimpleSmithWatermanScheme scheme( match, mm, del, ins );
    return blocking ? full_sw_type<aln::TextBlockingTag>   ( type, scheme, pat, M, txt, N, min_score, score, sink )
                    : full_sw_type<aln::PatternBlockingTag>( type, scheme, pat, M, txt, N, min_score, score, sink );
}

int ref_full_ed(int type, int blocking, const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                int32_t* score, uint32_t* sink)
{
    return blocking ? full_ed_type<aln::TextBlockingTag>   ( type, pat, M, txt, N, min_score, score, sink )
                    : full_ed_type<aln::PatternBlockingTag>( type, pat, M, txt, N, min_score, score, sink );
}

// banded traceback through the reference (M <= 1024).  ops: one byte per op in BACKTRACKING order
// (0 SUBSTITUTION, 1 INSERTION, 2 DELETION -- aln::DirectionVector, alignment.h:326-330); *n_ops is the
// number of ops produced (ops beyond cap are dropped); clips[0] = clip before the ops (pattern_len - sink.y),
// clips[1] = clip after them (source.y).  Returns the number of clip() calls (0: not aligned, nothing reported).
int ref_banded_gotoh_traceback_ex(uint32_t band, int type, const int32_t* sc,
                                  const uint8_t* pat, const uint8_t* quals, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                                  int32_t* score, uint32_t* source, uint32_t* sink,
                                  uint8_t* ops, uint32_t cap, uint32_t* n_ops, uint32_t* clips)
{
    if (M > 1024u) return -1;
    QualRampScheme scheme; scheme.m_match = sc[0]; scheme.m_mm_min = sc[1]; scheme.m_mm_max = sc[2];
    scheme.m_pat_go = sc[3]; scheme.m_pat_ge = sc[4]; scheme.m_txt_go = sc[5]; scheme.m_txt_ge = sc[6];
    RecordingBacktracer bt; bt.ops = ops; bt.cap = cap; bt.n = 0; bt.n_clips = 0; bt.clips[0] = bt.clips[1] = 0;
    const int r = quals ?
        banded_tb_band( band, type, scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt ) :
        banded_tb_band( band, type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, source, sink, bt );
    *n_ops = bt.n; clips[0] = bt.clips[0]; clips[1] = bt.clips[1];
    return r;
}

// the banded traceback of the linear-gap Smith-Waterman aligner (banded_sw_tb_type above); sw[4] = { match, mismatch, deletion, insertion }
int ref_banded_sw_traceback(uint32_t band, int type, const int32_t* sw,
                            const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                            int32_t* score, uint32_t* source, uint32_t* sink,
                            uint8_t* ops, uint32_t cap, uint32_t* n_ops, uint32_t* clips)
{
    if (M > 1024u) return -1;
    const aln::SimpleSmithWatermanScheme scheme( sw[0], sw[1], sw[2], sw[3] );
    RecordingBacktracer bt; bt.ops = ops; bt.cap = cap; bt.n = 0; bt.n_clips = 0; bt.clips[0] = bt.clips[1] = 0;
    int r = -1;
    switch (band)
    {
    case 3:  r = banded_sw_tb_type<3> ( type, scheme, pat, M, txt, N, min_score, score, source, sink, bt ); break;
    case 7:  r = banded_sw_tb_type<7> ( type, scheme, pat, M, txt, N, min_score, score, source, sink, bt ); break;
    case 15: r = banded_sw_tb_type<15>( type, scheme, pat, M, txt, N, min_score, score, source, sink, bt ); break;
    case 31: r = banded_sw_tb_type<31>( type, scheme, pat, M, txt, N, min_score, score, source, sink, bt ); break;
    }
    *n_ops = bt.n; clips[0] = bt.clips[0]; clips[1] = bt.clips[1];
    return r;
}

// the full-matrix traceback of the linear-gap Smith-Waterman aligner (full_sw_tb_run above; M <= 256, N <= 1024);
// sw[4] = { match, mismatch, deletion, insertion }
int ref_full_sw_traceback(int type, const int32_t* sw, const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                          int32_t* score, uint32_t* source, uint32_t* sink, uint8_t* ops, uint32_t cap, uint32_t* n_ops, uint32_t* clips)
{
    if (M > 256u || N > 1024u) return -1;
    const aln::SimpleSmithWatermanScheme scheme( sw[0], sw[1], sw[2], sw[3] );
    RecordingBacktracer bt; bt.ops = ops; bt.cap = cap; bt.n = 0; bt.n_clips = 0; bt.clips[0] = bt.clips[1] = 0;
    int r = -1;
    switch (type)
    {
    case 0: r = full_sw_tb_run<aln::GLOBAL>     ( scheme, pat, M, txt, N, min_score, score, source, sink, bt ); break;
    case 1: r = full_sw_tb_run<aln::LOCAL>      ( scheme, pat, M, txt, N, min_score, score, source, sink, bt ); break;
    case 2: r = full_sw_tb_run<aln::SEMI_GLOBAL>( scheme, pat, M, txt, N, min_score, score, source, sink, bt ); break;
    }
    *n_ops = bt.n; clips[0] = bt.clips[0]; clips[1] = bt.clips[1];
    return r;
}

// full-matrix traceback through the reference (M <= 256, N <= 1024); outputs as ref_banded_gotoh_traceback_ex
int ref_full_gotoh_traceback_ex(int type, const int32_t* sc,
                                const uint8_t* pat, const uint8_t* quals, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                                int32_t* score, uint32_t* source, uint32_t* sink,
                                uint8_t* ops, uint32_t cap, uint32_t* n_ops, uint32_t* clips)
{
    if (M > 256u || N > 1024u) return -1;
    QualRampScheme scheme; scheme.m_match = sc[0]; scheme.m_mm_min = sc[1]; scheme.m_mm_max = sc[2];
    scheme.m_pat_go = sc[3]; scheme.m_pat_ge = sc[4]; scheme.m_txt_go = sc[5]; scheme.m_txt_ge = sc[6];
    RecordingBacktracer bt; bt.ops = ops; bt.cap = cap; bt.n = 0; bt.n_clips = 0; bt.clips[0] = bt.clips[1] = 0;
    const int r = quals ?
        full_tb_type( type, scheme, pat, quals, M, txt, N, min_score, score, source, sink, bt ) :
        full_tb_type( type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, source, sink, bt );
    *n_ops = bt.n; clips[0] = bt.clips[0]; clips[1] = bt.clips[1];
    return r;
}

int ref_full_gotoh_ex(int type, int blocking, const int32_t* sc,
                      const uint8_t* pat, const uint8_t* quals, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                      int32_t* score, uint32_t* sink)
{
    QualRampScheme scheme; scheme.m_match = sc[0]; scheme.m_mm_min = sc[1]; scheme.m_mm_max = sc[2];
    scheme.m_pat_go = sc[3]; scheme.m_pat_ge = sc[4]; scheme.m_txt_go = sc[5]; scheme.m_txt_ge = sc[6];
    if (quals)
        return blocking ?
            full_type<aln::TextBlockingTag>   ( type, scheme, pat, quals, M, txt, N, min_score, score, sink ) :
            full_type<aln::PatternBlockingTag>( type, scheme, pat, quals, M, txt, N, min_score, score, sink );
    return blocking ?
        full_type<aln::TextBlockingTag>   ( type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, sink ) :
        full_type<aln::PatternBlockingTag>( type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, sink );
}

// blocking: 0 = PatternBlockingTag (the alignment_score default), 1 = TextBlockingTag (sw-benchmark)
int ref_full_gotoh(int type, int blocking, int match, int mm, int gap_open, int gap_ext,
                   const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                   int32_t* score, uint32_t* sink)
{
    const aln::SimpleGotohScheme scheme( match, mm, gap_open, gap_ext );
    return blocking ?
        full_type<aln::TextBlockingTag>   ( type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, sink ) :
        full_type<aln::PatternBlockingTag>( type, scheme, pat, aln::trivial_quality_string(), M, txt, N, min_score, score, sink );
}

// batched forms used for timing the reference's host path (OpenMP over work items, as
// BatchedBandedAlignmentScore<..,HostThreadScheduler>::enact, nvbio/alignment/batched_banded_inl.h:113-121)
void ref_banded_gotoh_batch(uint32_t band, int type, int match, int mm, int gap_open, int gap_ext,
                            const uint8_t* pats, const uint32_t* pat_off,
                            const uint8_t* txts, const uint32_t* txt_off, uint32_t n,
                            int32_t min_score, int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel for
    for (int64 i = 0; i < int64(n); ++i)
    {
        ref_banded_gotoh( band, type, match, mm, gap_open, gap_ext,
            pats + pat_off[i], pat_off[i+1] - pat_off[i],
            txts + txt_off[i], txt_off[i+1] - txt_off[i],
            min_score, scores + i, sinks + 2*i );
    }
}

// general-scheme batch: the host path of BatchedBandedAlignmentScore<BAND,stream,HostThreadScheduler>
// (an OpenMP parallel for over banded_alignment_score, nvbio/alignment/batched_banded_inl.h:113-121)
void ref_banded_gotoh_ex_batch(uint32_t band, int type, const int32_t* sc,
                               const uint8_t* pats, const uint8_t* quals, const uint32_t* pat_off,
                               const uint8_t* txts, const uint32_t* txt_off, uint32_t n,
                               int32_t min_score, int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel for
    for (int64 i = 0; i < int64(n); ++i)
    {
        ref_banded_gotoh_ex( band, type, sc,
            pats + pat_off[i], quals ? quals + pat_off[i] : NULL, pat_off[i+1] - pat_off[i],
            txts + txt_off[i], txt_off[i+1] - txt_off[i],
            min_score, scores + i, sinks + 2*i );
    }
}

void ref_full_gotoh_batch(int type, int blocking, int match, int mm, int gap_open, int gap_ext,
                          const uint8_t* pats, const uint32_t* pat_off,
                          const uint8_t* txts, const uint32_t* txt_off, uint32_t n,
                          int32_t min_score, int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel for
    for (int64 i = 0; i < int64(n); ++i)
    {
        ref_full_gotoh( type, blocking, match, mm, gap_open, gap_ext,
            pats + pat_off[i], pat_off[i+1] - pat_off[i],
            txts + txt_off[i], txt_off[i+1] - txt_off[i],
            min_score, scores + i, sinks + 2*i );
    }
}


// A sequence of operations on nvBowtie's per-read hit deque, run on the reference's own priority_deque:
//   op 0: push (begin, bits) the way seed_mapper<EXACT_MAPPING>::enact does (mapping_inl.h:242-244: pop_bottom first when the
//         deque holds max_hits elements)
//   op 1: pop_top     op 2: pop_bottom
//   op 3: what select_kernel does to the top hit (select_inl.h:96-118): pop an exhausted top first, then take a row off its
//         front in place -- begin + 1, size - 1, no re-heapify; out_rows receives the row (0xFFFFFFFF if the deque was empty)
// heap_out / *size_out: the underlying array afterwards.
struct RefSeedHit { uint32 begin; uint32 bits; uint32 get_range_size() const { return bits & 0xFFFFFu; } };
struct ref_hit_compare { bool operator() (const RefSeedHit& f, const RefSeedHit& s) { return f.get_range_size() > s.get_range_size(); } };

void ref_hit_deque_run(const uint32_t* ops, const uint32_t* begins, const uint32_t* bits, uint32_t n_ops, uint32_t max_hits,
                       uint32_t* heap_out, uint32_t* size_out, uint32_t* out_rows)
{
    std::vector<RefSeedHit> storage( n_ops + 1u );
    typedef vector_view<RefSeedHit*> vec_t;
    priority_deque<RefSeedHit, vec_t, ref_hit_compare> heap( vec_t( 0, &storage[0] ), true );
    for (uint32 k = 0; k < n_ops; ++k)
    {
        out_rows[k] = 0xFFFFFFFFu;
        if (ops[k] == 0u)
        {
            RefSeedHit h = { begins[k], bits[k] };
            if (heap.size() == max_hits) heap.pop_bottom();
            heap.push( h );
        }
        else if (ops[k] == 1u) { if (heap.size()) heap.pop_top(); }
        else if (ops[k] == 2u) { if (heap.size()) heap.pop_bottom(); }
        else
        {
            if (heap.size() == 0) continue;
            RefSeedHit* hit = const_cast<RefSeedHit*>( &heap.top() );
            if (hit->get_range_size() == 0u)
            {
                heap.pop_top();
                if (heap.size() == 0) continue;
                hit = const_cast<RefSeedHit*>( &heap.top() );
            }
            out_rows[k] = hit->begin;
            hit->begin += 1u; hit->bits = (hit->bits & ~0xFFFFFu) | ((hit->get_range_size() - 1u) & 0xFFFFFu);
        }
    }
    *size_out = uint32( heap.size() );
    for (uint32 k = 0; k < heap.size(); ++k) { heap_out[2*k] = storage[k].begin; heap_out[2*k+1] = storage[k].bits; }
}


// The GENERIC rank dictionary (nvbio/fmindex/rank_dictionary_inl.h:206-336: plain word storage, separate occurrence table, any K, 32- or
// 64-bit indices), in the two configurations the reference's own test runs besides the production one (nvbio-test/rank_test.cu:83-227):
//   word_bits 32: PackedStream<const uint32*,uint8,2,true>,        K = 64,  uint32 indices
//   word_bits 64: PackedStream<const uint64*,uint8,2,true,uint64>, K = 128, uint64 indices
// occ_out = build_occurrence_table<K> of the text (4 entries per block, as index-width words); rank_out[q] = rank( dict, idx[q], sym[q] ),
// rank4_out[4 q ..] = rank4( dict, idx[q] ).  idx 0xFFFF...F is passed through to rank() (-> 0) and skipped for rank4 (undefined there).
void ref_rank_generic(uint32_t word_bits, const void* text_words, uint64_t length, const uint64_t* idx, const uint8_t* sym, uint32_t n,
                      void* occ_out, uint64_t* counts_out, uint64_t* rank_out, uint64_t* rank4_out)
{
    std::vector<uint32> count_table( 256 );
    gen_bwt_count_table( &count_table[0] );
    if (word_bits == 32)
    {
        typedef PackedStream<const uint32*,uint8,2,true> stream_type;
        stream_type text( (const uint32*)text_words );
        uint32* occ = (uint32*)occ_out; uint32 cnt[4];
        build_occurrence_table<64>( text.begin(), text.begin() + uint32(length), occ, cnt );
        for (int c = 0; c < 4; ++c) counts_out[c] = cnt[c];
        typedef rank_dictionary<2u, 64, stream_type, const uint32*, const uint32*> dict_type;
        dict_type dict( text, occ, &count_table[0] );
        for (uint32 q = 0; q < n; ++q)
        {
            rank_out[q] = rank( dict, uint32( idx[q] ), uint32( sym[q] ) );
            if (uint32( idx[q] ) != uint32(-1)) { const uint4 r = rank4( dict, uint32( idx[q] ) ); rank4_out[4*q] = r.x; rank4_out[4*q+1] = r.y; rank4_out[4*q+2] = r.z; rank4_out[4*q+3] = r.w; }
        }
    }
    else
    {
        typedef PackedStream<const uint64*,uint8,2,true,uint64> stream_type;
        stream_type text( (const uint64*)text_words );
        uint64* occ = (uint64*)occ_out; uint64 cnt[4];
        build_occurrence_table<128>( text.begin(), text.begin() + length, occ, cnt );
        for (int c = 0; c < 4; ++c) counts_out[c] = cnt[c];
        typedef rank_dictionary<2u, 128, stream_type, const uint64*, const uint32*> dict_type;
        dict_type dict( text, occ, &count_table[0] );
        for (uint32 q = 0; q < n; ++q)
        {
            rank_out[q] = rank( dict, uint64( idx[q] ), uint32( sym[q] ) );
            if (idx[q] != uint64(-1)) { const uint64_4 r = rank4( dict, uint64( idx[q] ) ); rank4_out[4*q] = r.x; rank4_out[4*q+1] = r.y; rank4_out[4*q+2] = r.z; rank4_out[4*q+3] = r.w; }
        }
    }
}


// sw-benchmark's shape (sw-benchmark/sw-benchmark.cu:152,362-369): every pattern against the WHOLE of one text, the host scheduler's
// OpenMP parallel-for over work items (batched_inl.h:283-306)
void ref_full_gotoh_many_to_one(int type, int blocking, int match, int mm, int gap_open, int gap_ext,
                                const uint8_t* pats, const uint32_t* pat_off, const uint8_t* text, uint32_t text_len, uint32_t n,
                                int32_t min_score, int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel for schedule(dynamic,64)
    for (int64 i = 0; i < int64(n); ++i)
        ref_full_gotoh( type, blocking, match, mm, gap_open, gap_ext, pats + pat_off[i], pat_off[i+1] - pat_off[i], text, text_len,
                        min_score, scores + i, sinks + 2*i );
}


int ref_banded_myers(uint32_t band, int type, const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score, int32_t* score, uint32_t* sink)
{
#define REF_MY(B) (type == 0 ? banded_myers_run<B,aln::GLOBAL>( pat, M, txt, N, min_score, score, sink ) : banded_myers_run<B,aln::SEMI_GLOBAL>( pat, M, txt, N, min_score, score, sink ))
    switch (band)
    {
    case 3:  return REF_MY( 3 );
    case 7:  return REF_MY( 7 );
    case 15: return REF_MY( 15 );
    case 31: return REF_MY( 31 );
    }
#undef REF_MY
    return -1;
}

} // extern "C"
