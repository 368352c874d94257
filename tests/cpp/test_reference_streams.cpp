// tests/cpp/test_reference_streams.cpp -- the binding of the reference's OWN alignment streams (reference_streams.hpp) compiled
// and run.  nvBowtie's headers need nvcc + thrust device vectors, so the two stream classes are stood in for by test doubles
// that declare exactly the members the reference's classes have -- same names, same types (raw device pointers where the
// reference holds vector_view<T*> / uint32*) -- and nothing else:
//   bowtie2::cuda::BestScoreStream + AlignmentStreamBase + BaseScoringPipelineState + HitQueuesDeviceView + packed_seed
//       nvBowtie/bowtie2/cuda/score_inl.h:44-136, alignment_utils.h:194-308, pipeline_states.h:49-115, scoring_queues.h:244-289, defs.h:162-172
//   sw-benchmark's AlignmentStream                                                   sw-benchmark/sw-benchmark.cu:70-209
// The doubles are test infrastructure; what is under test is that enact_best_score_stream / enact_sw_benchmark_stream touch only
// those members and produce, hit for hit, what the reference's per-item functors would: checked with the oracle's DP on reads
// oriented the way load_strings orients them.
#include <nvbio_amd/reference_streams.hpp>
#include "../../oracle/nvbio_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <algorithm>

using namespace nvbio_amd;

#define REQUIRE(cond) do { if (!(cond)) { fprintf( stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond ); exit( 1 ); } } while (0)

namespace refdouble {

// nvbio::vector_view<T*,uint64> (nvbio/basic/vector_view.h): size + pointer, operator[] returns a reference
template <typename T> struct vector_view { uint64_t m_size; T* m_vec; T& operator[](const uint64_t i) const { return m_vec[i]; } uint64_t size() const { return m_size; } };

struct packed_seed { uint32_t pos_in_read:12, index_dir:1, rc:1, top_flag:1; };              // defs.h:162-172

struct HitQueuesDeviceView                                                                       // scoring_queues.h:244-289
{
    vector_view<uint32_t>    read_id;
    vector_view<packed_seed> seed;
    vector_view<uint32_t>    ssa;
    vector_view<uint32_t>    loc;
    vector_view<int32_t>     score;
    vector_view<uint32_t>    sink;
};
struct ScoringQueuesDeviceView { HitQueuesDeviceView hits; };                                    // scoring_queues.h:303-...

struct ReadBatch                                                                                 // io::SequenceDataAccess<DNA_N> (sequence_access.h:115-135)
{
    const uint32_t* m_index; const uint32_t* m_storage; const uint8_t* m_quals; uint32_t m_max_len;
    const uint32_t* sequence_index()   const { return m_index; }
    const uint32_t* sequence_storage() const { return m_storage; }
    const uint8_t*  qual_stream()      const { return m_quals; }
    uint32_t        max_read_len()     const { return m_max_len; }
};
struct GenomeStream { const uint32_t* m_stream; const uint32_t* stream() const { return m_stream; } };   // PackedStream<const uint32*,uint8,2,true>

struct Pipeline                                                                                  // BaseScoringPipelineState (pipeline_states.h:49-115)
{
    ReadBatch               reads;
    uint32_t                genome_length;
    GenomeStream            genome;
    ScoringQueuesDeviceView scoring_queues;
    uint32_t                hits_queue_size;
    uint32_t*               idx_queue;
    int32_t                 score_limit;
};

template <typename scheme_t> struct Aligner { scheme_t scheme; };                                // aln::GotohAligner<TYPE,scheme>: public `scheme`

template <typename AlignerType, typename PipelineType>
struct BestScoreStream                                                                           // score_inl.h:44-136 + AlignmentStreamBase
{
    uint32_t max_pattern_length() const { return m_pipeline.reads.max_read_len(); }
    uint32_t size() const { return m_pipeline.hits_queue_size; }
    const AlignerType& aligner() const { return m_aligner; }
    PipelineType m_pipeline; AlignerType m_aligner; uint32_t m_band_len;
};

struct SimpleGotohScheme                                                                         // nvbio/alignment/utils.h:103-123
{
    int32_t match(const uint8_t q = 0) const { return m_match; }
    int32_t mismatch(const uint8_t q = 0) const { return m_mismatch; }
    int32_t pattern_gap_open() const { return m_gap_open; }   int32_t pattern_gap_extension() const { return m_gap_ext; }
    int32_t text_gap_open() const { return m_gap_open; }      int32_t text_gap_extension() const { return m_gap_ext; }
    int32_t m_match, m_mismatch, m_gap_open, m_gap_ext;
};
struct QualityScheme                                                                             // nvBowtie SmithWatermanScoringScheme's aligner interface (scoring.h:278-285)
{
    int32_t match(const uint8_t q = 0) const { return 0; }
    int32_t mismatch(const uint8_t q = 0) const { return -(2 + int32_t( float( q < 40 ? q : 40 ) / 40.0f * 4.0f )); }   // QualCost, scoring.h:84-88
    int32_t pattern_gap_open() const { return -8; }   int32_t pattern_gap_extension() const { return -3; }
    int32_t text_gap_open() const { return -8; }      int32_t text_gap_extension() const { return -3; }
};

template <typename aligner_t>
struct SwBenchmarkStream                                                                         // sw-benchmark.cu:70-209 (public members :199-208)
{
    aligner_t       m_aligner;
    uint32_t        m_count, m_max_pattern_len, m_total_pattern_len, m_text_len;
    const uint32_t* m_offsets;
    const uint32_t* m_patterns;
    const uint32_t* m_text;
    int16_t*        m_scores;
};

} // namespace refdouble

static void pack4(const std::vector<uint8_t>& sym, std::vector<uint32_t>& out)
{
    out.assign( sym.size() / 8 + 8, 0u );
    for (size_t i = 0; i < sym.size(); ++i) out[i >> 3] |= (uint32_t)sym[i] << (28 - 4 * (i & 7));
}

int main()
{
    std::mt19937 rng( 11 );
    const uint32_t G = 200000;
    std::vector<uint8_t> text( G );
    for (auto& c : text) c = rng() & 3;
    std::vector<uint32_t> text2( (G + 15) / 16 + 8, 0u );
    orc_pack2( text.data(), G, text2.data() );
    device_vector<uint32_t> d_genome( text2 );

    // ---- nvBowtie: a scoring queue over reads stored REVERSED, hits on both strands, scored through BestScoreStream ----
    {
        const uint32_t R = 500, H = 4000, M = 100, BAND = 31;
        std::vector<uint8_t> reads( R * M ), quals( R * M );
        for (auto& c : reads) c = rng() & 3;
        for (auto& q : quals) q = rng() % 50;
        std::vector<uint32_t> hit_read( H ), hit_loc( H ), idxq;
        std::vector<refdouble::packed_seed> hit_seed( H );
        for (uint32_t h = 0; h < H; ++h)
        {
            hit_read[h] = rng() % R; hit_loc[h] = (h < 30) ? rng() % 12 : (h < 60 ? G - 1 - rng() % 50 : 20 + rng() % (G - M - 60));
            hit_seed[h].pos_in_read = rng() % 100; hit_seed[h].index_dir = rng() & 1; hit_seed[h].rc = rng() & 1; hit_seed[h].top_flag = rng() & 1;
            if (h % 5) idxq.push_back( h );
        }
        std::shuffle( idxq.begin(), idxq.end(), rng );
        // make most reads resemble the text at one of their hits
        for (uint32_t h = 60; h < H; h += 2)
        {
            const uint32_t g = hit_loc[h], r = hit_read[h];
            for (uint32_t k = 0; k < M; ++k)
            {
                const uint8_t c = (rng() % 40) ? text[g + k] : (uint8_t)(rng() & 3);
                if (hit_seed[h].rc) reads[r * M + (M - 1 - k)] = 3 - c; else reads[r * M + k] = c;
            }
        }
        // io::REVERSE: the stored stream holds every read backwards (nvBowtie.cpp:322), qualities alongside
        std::vector<uint8_t> stored( R * M ), stored_q( R * M );
        for (uint32_t r = 0; r < R; ++r) for (uint32_t k = 0; k < M; ++k) { stored[r*M + k] = reads[r*M + M-1-k]; stored_q[r*M + k] = quals[r*M + M-1-k]; }
        std::vector<uint32_t> stored4; pack4( stored, stored4 );
        std::vector<uint32_t> index( R + 1 ); for (uint32_t r = 0; r <= R; ++r) index[r] = r * M;

        device_vector<uint32_t> d_reads( stored4 ), d_index( index ), d_hit_read( hit_read ), d_hit_loc( hit_loc ), d_idxq( idxq ), d_sink( H );
        device_vector<uint8_t>  d_quals( stored_q );
        device_vector<refdouble::packed_seed> d_seed( hit_seed );
        device_vector<int32_t>  d_score( H );
        check_hip( hipMemset( d_score.data(), 0, H * 4 ), "memset" );

        typedef refdouble::Aligner<refdouble::QualityScheme> aligner_t;
        refdouble::BestScoreStream<aligner_t, refdouble::Pipeline> stream;
        stream.m_band_len = BAND;
        stream.m_pipeline.reads = { d_index.data(), d_reads.data(), d_quals.data(), M };
        stream.m_pipeline.genome_length = G;
        stream.m_pipeline.genome = { d_genome.data() };
        stream.m_pipeline.scoring_queues.hits = { { H, d_hit_read.data() }, { H, d_seed.data() }, { 0, nullptr }, { H, d_hit_loc.data() }, { H, d_score.data() }, { H, d_sink.data() } };
        stream.m_pipeline.hits_queue_size = (uint32_t)idxq.size();
        stream.m_pipeline.idx_queue = d_idxq.data();
        stream.m_pipeline.score_limit = -1000;

        device_vector<uint8_t> d_temp( aln::best_score_stream_temp_storage( stream.size() ) );
        aln::enact_best_score_stream<BAND>( stream, NVBIO_SEMI_GLOBAL, d_temp.data(), d_temp.size(), /*worst_score*/ -65536 );
        check_hip( hipDeviceSynchronize(), "sync" );
        const std::vector<int32_t>  got_score = d_score.to_host();
        const std::vector<uint32_t> got_sink  = d_sink.to_host();

        const orc_gotoh_scheme os = { 0, 2, 6, -8, -3, -8, -3 };
        std::vector<bool> touched( H, false );
        for (size_t i = 0; i < idxq.size(); ++i)
        {
            const uint32_t h = idxq[i], r = hit_read[h], g = hit_loc[h];
            touched[h] = true;
            const uint32_t begin = g > BAND / 2 ? g - BAND / 2 : 0u, end = std::min( begin + BAND + M, G );
            std::vector<uint8_t> pat( M ), pq( M );
            for (uint32_t k = 0; k < M; ++k)
            {
                // what load_strings gives the aligner: the read forwards, or its reverse complement (qualities follow the read)
                pat[k] = hit_seed[h].rc ? 3 - reads[r*M + M-1-k] : reads[r*M + k];
                pq[k]  = hit_seed[h].rc ? quals[r*M + M-1-k]     : quals[r*M + k];
            }
            int32_t score; uint32_t sink[2];
            orc_banded_gotoh( BAND, 2 /*SEMI_GLOBAL*/, &os, pat.data(), pq.data(), M, text.data() + begin, end - begin, &score, sink );
            REQUIRE( got_score[h] == std::max( score, -65536 ) );
            REQUIRE( got_sink[h] == begin + sink[0] );
        }
        for (uint32_t h = 0; h < H; ++h) if (!touched[h]) REQUIRE( got_score[h] == 0 );      // hits outside idx_queue are not written
        printf( "BestScoreStream binding ok: %zu hits\n", idxq.size() );
    }

    // ---- sw-benchmark: every pattern against one little-endian reference text, int16 scores ----
    {
        const uint32_t N = 3000, T = 1500;
        std::vector<uint32_t> offs( N + 1 ); offs[0] = 0;
        for (uint32_t i = 0; i < N; ++i) offs[i + 1] = offs[i] + 60 + rng() % 41;
        std::vector<uint8_t> pats( offs[N] );
        uint32_t max_len = 0;
        for (uint32_t i = 0; i < N; ++i)
        {
            const uint32_t len = offs[i+1] - offs[i], p = rng() % (T - len);
            max_len = std::max( max_len, len );
            for (uint32_t k = 0; k < len; ++k) pats[offs[i] + k] = (rng() % 20) ? text[p + k] : (uint8_t)(rng() & 3);
        }
        std::vector<uint32_t> pats4; pack4( pats, pats4 );
        std::vector<uint32_t> text_le( (T + 15) / 16 + 8, 0u );                                  // REF_BIG_ENDIAN = false
        for (uint32_t i = 0; i < T; ++i) text_le[i >> 4] |= (uint32_t)text[i] << (2 * (i & 15));
        device_vector<uint32_t> d_offs( offs ), d_pats( pats4 ), d_text( text_le );
        device_vector<int16_t>  d_scores( N );
        typedef refdouble::Aligner<refdouble::SimpleGotohScheme> aligner_t;
        const orc_gotoh_scheme os = { 2, 1, 1, -2, -1, -2, -1 };
        for (int type = 0; type < 3; ++type)
        {
            refdouble::SwBenchmarkStream<aligner_t> stream = { { { 2, -1, -2, -1 } }, N, max_len, offs[N], T, d_offs.data(), d_pats.data(), d_text.data(), d_scores.data() };
            device_vector<uint8_t> d_temp( aln::sw_benchmark_stream_temp_storage( N, max_len, T ) );
            aln::enact_sw_benchmark_stream( stream, (nvbio_alignment_type)type, /*text_blocking*/ true, d_temp.data(), d_temp.size() );
            check_hip( hipDeviceSynchronize(), "sync" );
            const std::vector<int16_t> got = d_scores.to_host();
            for (uint32_t i = 0; i < N; i += 3)
            {
                int32_t score; uint32_t sink[2];
                orc_full_gotoh( type, 1 /*text blocking*/, &os, pats.data() + offs[i], nullptr, offs[i+1] - offs[i], text.data(), T, -(1 << 30), &score, sink );
                REQUIRE( got[i] == (int16_t)score );
            }
        }
        printf( "sw-benchmark stream binding ok: %u patterns x 3 types\n", N );
    }
    printf( "reference streams ok\n" );
    return 0;
}
