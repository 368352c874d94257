// reference_streams.hpp -- enacting the reference's OWN alignment streams through the library.
//
// aln::BatchedBandedAlignmentScore / aln::BatchedAlignmentScore call a stream's device functors per work item
// (init_context / load_strings / output); functors cannot cross a C ABI, but the two streams the north star names keep
// everything those functors read in plain device arrays, reachable from the stream object on the host.  The functions here
// take such a stream object AS THE REFERENCE DEFINES IT and use only members it has:
//
//   bowtie2::cuda::BestScoreStream<AlignerType,PipelineType>          nvBowtie/bowtie2/cuda/score_inl.h:44-136
//     stream.size()                       :79     stream.m_band_len        :135
//     stream.aligner()                    alignment_utils.h:245            stream.m_pipeline   alignment_utils.h:302 (public)
//     m_pipeline.idx_queue                pipeline_states.h:104 (uint32*)  m_pipeline.genome_length :94
//     m_pipeline.genome                   :95  (PackedStream over the 2-bit genome: .stream() is its storage iterator)
//     m_pipeline.reads                    :91  (io::SequenceDataAccess<DNA_N>: sequence_index() / sequence_storage() / qual_stream(),
//                                               nvbio/io/sequence/sequence_access.h:115-135)
//     m_pipeline.scoring_queues.hits.{read_id, seed, loc, score, sink}     scoring_queues.h:280-285 (vector_view<T*>: &v[0] is the array)
//   sw-benchmark's AlignmentStream<aligner_type>                      sw-benchmark/sw-benchmark.cu:70-209
//     m_aligner, m_count, m_max_pattern_len, m_text_len, m_offsets, m_patterns, m_text, m_scores   :199-208 (public)
//
// A maintainer's specialisation of the batched classes for a new scheduler tag is then three lines.
// Iterator types that wrap a pointer (cuda::ldg_pointer<T>, nvbio/basic/cuda/ldg.h) reach it through raw_pointer(): add an
// overload next to the two below where the build uses one.
#pragma once
#include "nvbio_amd.hpp"

namespace nvbio_amd {

template <typename T> inline const T* raw_pointer(const T* p) { return p; }
template <typename T> inline T*       raw_pointer(T* p)       { return p; }

namespace aln {

// how a reference scoring scheme becomes an nvbio_gotoh_scheme: the reference's own accessors
// (aln::SimpleGotohScheme, nvbio/alignment/utils.h:103-123: match(q), mismatch(q), pattern_gap_open/extension, text_gap_open/extension;
// nvBowtie's SmithWatermanScoringScheme has the same interface, scoring.h:278-285, with mismatch(q) a quality ramp between q = 0 and q >= 40)
template <typename scheme_type>
inline nvbio_gotoh_scheme flat_scheme(const scheme_type& s)
{
    nvbio_gotoh_scheme f;
    f.match        = s.match( 0 );
    f.mm_min       = -s.mismatch( 0 );
    f.mm_max       = -s.mismatch( 40 );
    f.pat_gap_open = s.pattern_gap_open();
    f.pat_gap_ext  = s.pattern_gap_extension();
    f.txt_gap_open = s.text_gap_open();
    f.txt_gap_ext  = s.text_gap_extension();
    return f;
}

// scratch of enact_best_score_stream: read_id (4) + flags (1) + window begin / end (8) + score (4) + sink (8) per work item
inline uint64_t best_score_stream_temp_storage(uint32_t stream_size) { return ((uint64_t)stream_size + 64u) * 25u + 1024u; }

// aln::BatchedBandedAlignmentScore<BAND_LEN, bowtie2::cuda::BestScoreStream<...>, scheduler>::enact( stream, temp_size, temp )
// (nvBowtie/bowtie2/cuda/score_inl.h:461-509 instantiates it for BAND_LEN 3/7/15/31).  TYPE = the aligner's AlignmentType.
// temp: device scratch of best_score_stream_temp_storage( stream.size() ) bytes (nvBowtie hands over pipeline.dp_buffer).
template <uint32_t BAND_LEN, typename stream_type>
void enact_best_score_stream(const stream_type& stream, nvbio_alignment_type type, uint8_t* temp, uint64_t temp_size,
                             int32_t worst_score, int device = 0, hipStream_t s = 0)
{
    const uint32_t n = stream.size();
    if (n == 0) return;
    if (temp == nullptr || temp_size < best_score_stream_temp_storage( n )) throw error( NVBIO_ERR_INVALID, "enact_best_score_stream: temp storage too small" );
    const auto& p = stream.m_pipeline;
    nvbio_hit_queues hq;
    hq.idx_queue_dev   = p.idx_queue;
    hq.hit_read_id_dev = &p.scoring_queues.hits.read_id[0];
    hq.hit_seed_dev    = (uint32_t*)&p.scoring_queues.hits.seed[0];             // packed_seed is one 32-bit word (defs.h:162-172)
    hq.hit_loc_dev     = &p.scoring_queues.hits.loc[0];
    hq.hit_score_dev   = (int32_t*)&p.scoring_queues.hits.score[0];
    hq.hit_sink_dev    = (uint32_t*)&p.scoring_queues.hits.sink[0];
    hq.n               = n;
    // carve the scratch
    const uint64_t n4 = ((uint64_t)n + 63u) & ~63ull;
    uint32_t*    read_id = (uint32_t*)(((uintptr_t)temp + 255u) & ~(uintptr_t)255u);
    uint32_t*    wb      = read_id + n4;
    uint32_t*    we      = wb + n4;
    int32_t*     scores  = (int32_t*)(we + n4);
    nvbio_uint2* sinks   = (nvbio_uint2*)(scores + n4);
    uint8_t*     flags   = (uint8_t*)(sinks + n4);
    const uint32_t* read_index = raw_pointer( p.reads.sequence_index() );
    check( nvbio_score_stream_flatten( device, &hq, read_index, stream.m_band_len, p.genome_length, /*reads_reversed*/1u,
                                       read_id, flags, wb, we, s ) );
    nvbio_alignment_batch b = {};
    b.reads_dev        = raw_pointer( p.reads.sequence_storage() );
    b.read_bits        = 4;                                                      // io::SequenceDataTraits<DNA_N>::SEQUENCE_BITS
    b.read_offsets_dev = read_index;
    b.quals_dev        = (const uint8_t*)raw_pointer( p.reads.qual_stream() );
    b.read_id_dev      = read_id;
    b.flags_dev        = flags;
    b.text_dev         = raw_pointer( p.genome.stream() );
    b.text_bits        = 2;
    b.win_begin_dev    = wb;
    b.win_end_dev      = we;
    b.n                = n;
    b.max_read_len     = stream.max_pattern_length();
    b.algo_flags       = 0;
    const nvbio_gotoh_scheme sc = flat_scheme( stream.aligner().scheme );
    check( nvbio_banded_gotoh_score( device, BAND_LEN, type, &sc, &b, scores, sinks, s ) );
    check( nvbio_score_stream_output( device, &hq, scores, sinks, wb, worst_score, s ) );
}

// scratch of enact_sw_benchmark_stream: big-endian copy of the text + window arrays + int32 scores + sinks + the DP's boundary columns
inline uint64_t sw_benchmark_stream_temp_storage(uint32_t stream_size, uint32_t max_pattern_len, uint32_t text_len, bool text_blocking = true)
{
    uint64_t dp = 0; nvbio_alignment_batch b = {}; b.n = stream_size;
    check( nvbio_full_gotoh_temp_bytes( &b, max_pattern_len, text_len, text_blocking ? 1 : 0, &dp ) );
    return dp + ((uint64_t)stream_size + 64u) * 20u + ((uint64_t)text_len / 16u + 8u) * 4u + 2048u;
}

// aln::BatchedAlignmentScore<AlignmentStream<aligner>, scheduler>::enact( stream, temp_size, temp ) for sw-benchmark's stream
// (sw-benchmark/sw-benchmark.cu:362-369): every pattern against the whole reference text, int16 scores out.
template <typename stream_type>
void enact_sw_benchmark_stream(const stream_type& stream, nvbio_alignment_type type, bool text_blocking, uint8_t* temp, uint64_t temp_size,
                               int device = 0, hipStream_t s = 0)
{
    const uint32_t n = stream.m_count;
    if (n == 0) return;
    const uint32_t text_len = stream.m_text_len, text_words = (text_len + 15u) / 16u;
    if (temp == nullptr || temp_size < sw_benchmark_stream_temp_storage( n, stream.m_max_pattern_len, text_len, text_blocking ))
        throw error( NVBIO_ERR_INVALID, "enact_sw_benchmark_stream: temp storage too small" );
    const uint64_t n4 = ((uint64_t)n + 63u) & ~63ull;
    uint32_t*    wb     = (uint32_t*)(((uintptr_t)temp + 255u) & ~(uintptr_t)255u);
    uint32_t*    we     = wb + n4;
    int32_t*     scores = (int32_t*)(we + n4);
    nvbio_uint2* sinks  = (nvbio_uint2*)(scores + n4);
    uint32_t*    text_be = (uint32_t*)(sinks + n4);
    uint8_t*     dp     = (uint8_t*)(((uintptr_t)(text_be + text_words + 8u) + 255u) & ~(uintptr_t)255u);
    const uint64_t dp_size = temp_size - (uint64_t)(dp - temp);
    check( nvbio_text_2bit_le_to_be( device, raw_pointer( stream.m_text ), text_words, text_be, s ) );   // REF_BIG_ENDIAN = false (:65)
    check_hip( hipMemsetAsync( wb, 0, n * sizeof(uint32_t), s ), "hipMemsetAsync" );                       // text_length(i) = m_text_len: the whole text
    check_hip( hipMemsetD32Async( (hipDeviceptr_t)we, (int)text_len, n, s ), "hipMemsetD32Async" );
    nvbio_alignment_batch b = {};
    b.reads_dev        = raw_pointer( stream.m_patterns );                       // 4-bit big-endian (SequenceDataTraits<DNA_N>)
    b.read_bits        = 4;
    b.read_offsets_dev = stream.m_offsets;                                       // pattern i = [m_offsets[i], m_offsets[i+1])
    b.quals_dev        = nullptr;                                                // aln::trivial_quality_string
    b.read_id_dev      = nullptr;
    b.flags_dev        = nullptr;
    b.text_dev         = text_be;
    b.text_bits        = 2;
    b.win_begin_dev    = wb;
    b.win_end_dev      = we;
    b.n                = n;
    b.max_read_len     = stream.m_max_pattern_len;
    b.algo_flags       = 0;
    const nvbio_gotoh_scheme sc = flat_scheme( stream.m_aligner.scheme );
    // init_context sets min_score = Field_traits<int32>::min() (:163): no early exit -> no min_scores array
    check( nvbio_full_gotoh_score( device, type, text_blocking ? 1 : 0, &sc, &b, stream.m_max_pattern_len, text_len, nullptr,
                                   scores, sinks, dp, dp_size, s ) );
    check( nvbio_scores_to_int16( device, scores, n, stream.m_scores, s ) );    // output(): m_scores[i] = sink.score (:197)
}

} // namespace aln
} // namespace nvbio_amd
