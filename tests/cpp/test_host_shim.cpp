// tests/cpp/test_host_shim.cpp -- GPU parity test of the C++ host mirror (nvbio_amd.hpp), written
// the way the reference's own tests are (self-checking, exit(1) on mismatch:
// nvbio-test/fmindex_test.cu:575-657, alignment_test.cu:709-786), with the oracle as the checker.
#include <nvbio_amd/nvbio_amd.hpp>
#include "../../oracle/nvbio_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

using namespace nvbio_amd;

#define REQUIRE(cond) do { if (!(cond)) { fprintf( stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond ); exit( 1 ); } } while (0)

int main()
{
    std::mt19937 rng( 5 );
    const uint32_t n = 300000;
    std::vector<uint8_t> text( n );
    for (auto& c : text) c = rng() & 3;

    // oracle index (the checker)
    std::vector<uint32_t> sa( n + 1 ), bwt_occ( 2 * orc_bwt_words( n ) ), ssa( (n + 16) / 16 );
    orc_suffix_sort( text.data(), n, sa.data() );
    orc_fm_index oidx; memset( &oidx, 0, sizeof(oidx) );
    oidx.length = n; oidx.primary = orc_fm_build( text.data(), n, sa.data(), bwt_occ.data(), ssa.data(), oidx.L2 );
    oidx.bwt_occ = bwt_occ.data(); oidx.ssa = ssa.data();

    // device index built on the GPU through the shim
    std::vector<uint32_t> text2( (n + 15) / 16 + 8, 0u );
    orc_pack2( text.data(), n, text2.data() );
    device_vector<uint32_t> d_text2( text2 );
    fm_index fmi( d_text2.data(), n, 0, 8 );
    REQUIRE( fmi.length() == n && fmi.primary() == oidx.primary );
    for (int c = 0; c < 5; ++c) REQUIRE( fmi.L2( c ) == oidx.L2[c] );

    // FMIndexFilter: rank + locate, against the oracle's host filter
    const uint32_t Q = 5000, L = 20;
    std::vector<uint8_t> qsyms( Q * L );
    for (uint32_t q = 0; q < Q; ++q)
    {
        const uint32_t p = rng() % (n - L);
        for (uint32_t k = 0; k < L; ++k) qsyms[q*L + k] = (q % 3) ? text[p + k] : (rng() & 3);
    }
    std::vector<uint32_t> offs( Q + 1 ); for (uint32_t q = 0; q <= Q; ++q) offs[q] = q * L;
    std::vector<uint32_t> want_ranges( 2 * Q ); std::vector<uint64_t> want_slots( Q );
    const uint64_t want_total = orc_filter_rank( &oidx, qsyms.data(), offs.data(), Q, want_ranges.data(), want_slots.data() );

    device_vector<uint8_t> d_q( qsyms );
    FMIndexFilter<amd_device_tag> filter;
    const uint64_t total = filter.rank( fmi, string_set::uniform( d_q.data(), 8, L, Q ) );
    REQUIRE( total == want_total && filter.n_hits() == total );
    device_vector<nvbio_uint2> d_hits( total );
    filter.locate( 0, total, d_hits.data() );
    check_hip( hipDeviceSynchronize(), "sync" );
    std::vector<nvbio_uint2> hits = d_hits.to_host();
    std::vector<uint32_t> want_hits( 2 * total );
    orc_filter_locate( &oidx, want_ranges.data(), want_slots.data(), Q, 0, total, want_hits.data() );
    for (uint64_t h = 0; h < total; ++h) REQUIRE( hits[h].x == want_hits[2*h] && hits[h].y == want_hits[2*h+1] );

    // the same filter on an index that holds the full SA and the text, allowed to finish single-row searches on the
    // text (nvbio_fm_match_direct): identical hits
    {
        fm_index fmi1( d_text2.data(), n, 0, 8, 0, 1 );
        FMIndexFilter<amd_device_tag> direct;
        REQUIRE( direct.rank( fmi1, string_set::uniform( d_q.data(), 8, L, Q ), 0, true ) == want_total && direct.direct() != nullptr );
        device_vector<nvbio_uint2> d_hits1( total );
        direct.locate( 0, total, d_hits1.data() );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<nvbio_uint2> hits1 = d_hits1.to_host();
        for (uint64_t h = 0; h < total; ++h) REQUIRE( hits1[h].x == want_hits[2*h] && hits1[h].y == want_hits[2*h+1] );
    }

    // approximate search (hamming_backtrack), default semantics, against the oracle
    {
        std::vector<uint8_t> stream( 64 + Q * L );
        for (uint32_t k = 0; k < 64; ++k) stream[k] = rng() & 3;
        for (uint32_t k = 0; k < Q * L; ++k) stream[64 + k] = qsyms[k];
        std::vector<uint32_t> offs( Q + 1 );
        for (uint32_t k = 0; k <= Q; ++k) offs[k] = 64 + k * L;
        device_vector<uint8_t> d_stream( stream ); device_vector<uint32_t> d_offs( offs ), d_cnt( Q );
        hamming_backtrack( fmi, string_set::concatenated( d_stream.data(), 8, d_offs.data(), Q ), L / 2, 1, d_cnt.data() );
        std::vector<uint32_t> cnt = d_cnt.to_host();
        for (uint32_t k = 0; k < Q; ++k)
        {
            uint32_t want = 0;
            orc_hamming_backtrack( &oidx, stream.data(), offs[k], L, L / 2, 1, 0, &want, nullptr, 0 );
            REQUIRE( cnt[k] == want );
        }
    }

    // banded Gotoh through BatchedBandedAlignmentScore, SimpleGotohScheme, all three types
    const uint32_t J = 3000, M = 100;
    std::vector<uint8_t> pats( J * M ); std::vector<uint32_t> poffs( J + 1 ), wb( J ), we( J );
    for (uint32_t j = 0; j < J; ++j)
    {
        const uint32_t p = 15 + rng() % (n - M - 64);
        for (uint32_t k = 0; k < M; ++k) pats[j*M + k] = (rng() % 50) ? text[p + k] : (rng() & 3);
        poffs[j] = j * M; wb[j] = p - 15; we[j] = p - 15 + M + 31;
    }
    poffs[J] = J * M;
    device_vector<uint8_t> d_p( pats ), d_t( text );
    device_vector<uint32_t> d_po( poffs ), d_wb( wb ), d_we( we );
    device_vector<int32_t> d_scores( J ); device_vector<nvbio_uint2> d_sinks( J );
    nvbio_alignment_batch batch; memset( &batch, 0, sizeof(batch) );
    batch.reads_dev = d_p.data(); batch.read_bits = 8; batch.read_offsets_dev = d_po.data();
    batch.text_dev = d_t.data(); batch.text_bits = 8; batch.win_begin_dev = d_wb.data(); batch.win_end_dev = d_we.data(); batch.n = J;
    const aln::SimpleGotohScheme scheme( 2, -1, -2, -1 );
    const nvbio_gotoh_scheme fs = scheme.flat();
    orc_gotoh_scheme os = { fs.match, fs.mm_min, fs.mm_max, fs.pat_gap_open, fs.pat_gap_ext, fs.txt_gap_open, fs.txt_gap_ext };
    for (int type = 0; type < 3; ++type)
    {
        if (type == 0) aln::batch_banded_alignment_score<31>( aln::make_gotoh_aligner<aln::GLOBAL>( scheme ), batch, d_scores.data(), d_sinks.data() );
        if (type == 1) aln::batch_banded_alignment_score<31>( aln::make_gotoh_aligner<aln::LOCAL>( scheme ), batch, d_scores.data(), d_sinks.data() );
        if (type == 2) aln::batch_banded_alignment_score<31>( aln::make_gotoh_aligner<aln::SEMI_GLOBAL>( scheme ), batch, d_scores.data(), d_sinks.data() );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<int32_t> sc = d_scores.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host();
        for (uint32_t j = 0; j < J; ++j)
        {
            int32_t ws; uint32_t wk[2];
            orc_banded_gotoh( 31, type, &os, &pats[j*M], nullptr, M, &text[wb[j]], we[j] - wb[j], &ws, wk );
            REQUIRE( sc[j] == ws && sk[j].x == wk[0] && sk[j].y == wk[1] );
        }
    }
    // the staged scheduler's specialization (32-row windows, min_score exit): LOCAL with a limit most pairs reach and some do not
    {
        typedef aln::GotohAligner<aln::LOCAL,aln::SimpleGotohScheme> aligner_t;
        typedef aln::FlatStagedAlignmentStream<aligner_t> stream_t;
        aln::BatchedBandedAlignmentScore<31,stream_t,aln::DeviceStagedThreadScheduler> staged;
        const int32_t limit = -80;     // stops a job with 3+ mismatches in its first window (see the sign of the test, gotoh_banded_inl.h:619-621)
        staged.enact( stream_t( aln::make_gotoh_aligner<aln::LOCAL>( scheme ), batch, nullptr, limit, d_scores.data(), d_sinks.data(), M, M + 31 ) );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<int32_t> sc = d_scores.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host();
        uint32_t stopped = 0;
        for (uint32_t j = 0; j < J; ++j)
        {
            int32_t ws; uint32_t wk[2];
            stopped += orc_banded_gotoh_staged( 31, 1, &os, &pats[j*M], nullptr, M, &text[wb[j]], we[j] - wb[j], limit, &ws, wk ) == 0;
            REQUIRE( sc[j] == ws && sk[j].x == wk[0] && sk[j].y == wk[1] );
        }
        REQUIRE( stopped > 0 && stopped < J );
    }
    // the linear-gap Smith-Waterman aligner (unequal deletion / insertion) and the edit-distance aligner through the same class
    {
        const aln::SimpleSmithWatermanScheme sws( 2, -3, -5, -2 );
        const int32_t sw[4] = { 2, -3, -5, -2 }, ed[4] = { 0, -1, -1, -1 };
        for (int pass = 0; pass < 2; ++pass)
        {
            if (pass == 0) aln::batch_banded_alignment_score<31>( aln::make_smith_waterman_aligner<aln::LOCAL>( sws ), batch, d_scores.data(), d_sinks.data() );
            else           aln::batch_banded_alignment_score<31>( aln::make_edit_distance_aligner<aln::SEMI_GLOBAL>(), batch, d_scores.data(), d_sinks.data() );
            check_hip( hipDeviceSynchronize(), "sync" );
            std::vector<int32_t> sc = d_scores.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host();
            for (uint32_t j = 0; j < J; ++j)
            {
                int32_t ws; uint32_t wk[2];
                orc_banded_sw( 31, pass == 0 ? 1 : 2, pass == 0 ? sw : ed, &pats[j*M], M, &text[wb[j]], we[j] - wb[j], &ws, wk );
                REQUIRE( sc[j] == ws && sk[j].x == wk[0] && sk[j].y == wk[1] );
            }
        }
        // full matrix, pattern blocking, local: 16-wide logical stripes
        typedef aln::SmithWatermanAligner<aln::LOCAL,aln::SimpleSmithWatermanScheme> aligner_t;
        typedef aln::FlatAlignmentStream<aligner_t> stream_t;
        aln::BatchedAlignmentScore<stream_t> full;
        full.enact( stream_t( aligner_t( sws ), batch, d_scores.data(), d_sinks.data(), M, M + 31 ), 0u, nullptr, 0, 0, false );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<int32_t> sc = d_scores.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host();
        for (uint32_t j = 0; j < J; ++j)
        {
            int32_t ws; uint32_t wk[2];
            orc_full_sw( 1, 0, sw, &pats[j*M], M, &text[wb[j]], we[j] - wb[j], -(1 << 30), &ws, wk );
            REQUIRE( sc[j] == ws && sk[j].x == wk[0] && sk[j].y == wk[1] );
        }
    }
    // Best2Sink: best and second-best distinct alignment (local), against the oracle
    {
        device_vector<int32_t> d_s2( J ); device_vector<nvbio_uint2> d_k2( J );
        aln::batch_banded_alignment_score_best2<31>( aln::make_gotoh_aligner<aln::LOCAL>( scheme ), batch, 8u, d_scores.data(), d_sinks.data(), d_s2.data(), d_k2.data() );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<int32_t> sc = d_scores.to_host(), s2 = d_s2.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host(), k2 = d_k2.to_host();
        for (uint32_t j = 0; j < J; ++j)
        {
            int64_t out[6];
            orc_banded_gotoh_best2( 31, 1, &os, &pats[j*M], nullptr, M, &text[wb[j]], we[j] - wb[j], 8u, out );
            REQUIRE( sc[j] == out[0] && sk[j].x == (uint32_t)out[1] && sk[j].y == (uint32_t)out[2] );
            REQUIRE( s2[j] == out[3] && k2[j].x == (uint32_t)out[4] && k2[j].y == (uint32_t)out[5] );
        }
    }
    // banded traceback through BatchedBandedAlignmentTraceback (semi-global): Alignment + CIGAR runs equal to the oracle
    {
        const uint32_t STRIDE = 32;
        batch.max_read_len = M;
        device_vector<nvbio_uint2> d_src( J ); device_vector<uint16_t> d_cig( (size_t)J * STRIDE ); device_vector<uint32_t> d_len( J );
        typedef aln::GotohAligner<aln::SEMI_GLOBAL,aln::SimpleGotohScheme> aligner_t;
        typedef aln::FlatTracebackStream<aligner_t> stream_t;
        aln::BatchedBandedAlignmentTraceback<31,16,stream_t> tb;
        REQUIRE( tb.min_temp_storage( M, M + 31, J ) == (uint64_t)J * M * 16 );
        tb.enact( stream_t( aln::make_gotoh_aligner<aln::SEMI_GLOBAL>( scheme ), batch, d_scores.data(), d_src.data(), d_sinks.data(),
                            d_cig.data(), STRIDE, d_len.data() ) );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<int32_t> sc = d_scores.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host(), so = d_src.to_host();
        std::vector<uint16_t> cg = d_cig.to_host(); std::vector<uint32_t> ln = d_len.to_host();
        uint32_t gapped = 0;
        for (uint32_t j = 0; j < J; ++j)
        {
            int32_t ws; uint32_t wsrc[2], wsnk[2], wl; uint16_t wc[STRIDE];
            orc_banded_gotoh_traceback( 31, 2, &os, &pats[j*M], nullptr, M, &text[wb[j]], we[j] - wb[j], &ws, wsrc, wsnk, wc, STRIDE, &wl, nullptr, 0, nullptr );
            REQUIRE( sc[j] == ws && sk[j].x == wsnk[0] && sk[j].y == wsnk[1] && so[j].x == wsrc[0] && so[j].y == wsrc[1] && ln[j] == wl );
            for (uint32_t k = 0; k < wl && k < STRIDE; ++k) REQUIRE( cg[(size_t)j*STRIDE + k] == wc[k] );
            gapped += wl > 1;
        }
        printf( "host shim: %u tracebacks equal to the oracle (%u with more than one CIGAR run)\n", J, gapped );
    }
    // full-matrix traceback through BatchedAlignmentTraceback (local): equal to the oracle
    {
        const uint32_t STRIDE = 48;
        device_vector<nvbio_uint2> d_src( J ); device_vector<uint16_t> d_cig( (size_t)J * STRIDE ); device_vector<uint32_t> d_len( J );
        typedef aln::GotohAligner<aln::LOCAL,aln::SimpleGotohScheme> aligner_t;
        typedef aln::FlatTracebackStream<aligner_t> stream_t;
        aln::BatchedAlignmentTraceback<64,stream_t> tb;
        REQUIRE( tb.min_temp_storage( M, M + 31, J ) == (uint64_t)((J + 63) / 64 * 64) * (M + 31) * 4 * (1 + (M + 7) / 8) );   // whole waves own scratch
        tb.enact( stream_t( aln::make_gotoh_aligner<aln::LOCAL>( scheme ), batch, d_scores.data(), d_src.data(), d_sinks.data(),
                            d_cig.data(), STRIDE, d_len.data() ), M, M + 31 );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<int32_t> sc = d_scores.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host(), so = d_src.to_host();
        std::vector<uint16_t> cg = d_cig.to_host(); std::vector<uint32_t> ln = d_len.to_host();
        for (uint32_t j = 0; j < J; ++j)
        {
            int32_t ws; uint32_t wsrc[2], wsnk[2], wl; uint16_t wc[STRIDE];
            orc_full_gotoh_traceback( 1, &os, &pats[j*M], nullptr, M, &text[wb[j]], we[j] - wb[j], -(1 << 30), &ws, wsrc, wsnk, wc, STRIDE, &wl );
            REQUIRE( sc[j] == ws && sk[j].x == wsnk[0] && sk[j].y == wsnk[1] && so[j].x == wsrc[0] && so[j].y == wsrc[1] && ln[j] == wl );
            for (uint32_t k = 0; k < wl && k < STRIDE; ++k) REQUIRE( cg[(size_t)j*STRIDE + k] == wc[k] );
        }
        printf( "host shim: %u full-matrix tracebacks equal to the oracle\n", J );
    }
    // ... and of the linear-gap Smith-Waterman aligner with deletion != insertion (end-to-end): nvbio_full_sw_traceback behind the same class
    {
        const uint32_t STRIDE = 48;
        const aln::SimpleSmithWatermanScheme sws( 2, -3, -5, -2 );
        const int32_t swv[4] = { 2, -3, -5, -2 };
        device_vector<nvbio_uint2> d_src( J ); device_vector<uint16_t> d_cig( (size_t)J * STRIDE ); device_vector<uint32_t> d_len( J );
        typedef aln::SmithWatermanAligner<aln::SEMI_GLOBAL,aln::SimpleSmithWatermanScheme> aligner_t;
        typedef aln::FlatTracebackStream<aligner_t> stream_t;
        aln::BatchedAlignmentTraceback<64,stream_t> tb;
        tb.enact( stream_t( aln::make_smith_waterman_aligner<aln::SEMI_GLOBAL>( sws ), batch, d_scores.data(), d_src.data(), d_sinks.data(),
                            d_cig.data(), STRIDE, d_len.data() ), M, M + 31 );
        check_hip( hipDeviceSynchronize(), "sync" );
        std::vector<int32_t> sc = d_scores.to_host(); std::vector<nvbio_uint2> sk = d_sinks.to_host(), so = d_src.to_host();
        std::vector<uint16_t> cg = d_cig.to_host(); std::vector<uint32_t> ln = d_len.to_host();
        uint32_t gapped = 0;
        for (uint32_t j = 0; j < J; ++j)
        {
            int32_t ws; uint32_t wsrc[2], wsnk[2], wl; uint16_t wc[STRIDE];
            orc_full_sw_traceback( 2, swv, &pats[j*M], M, &text[wb[j]], we[j] - wb[j], -(1 << 30), &ws, wsrc, wsnk, wc, STRIDE, &wl );
            REQUIRE( sc[j] == ws && sk[j].x == wsnk[0] && sk[j].y == wsnk[1] && so[j].x == wsrc[0] && so[j].y == wsrc[1] && ln[j] == wl );
            for (uint32_t k = 0; k < wl && k < STRIDE; ++k) REQUIRE( cg[(size_t)j*STRIDE + k] == wc[k] );
            gapped += wl > 1;
        }
        printf( "host shim: %u full-matrix Smith-Waterman tracebacks equal to the oracle (%u with more than one CIGAR run)\n", J, gapped );
    }
    // error behaviour: an unsupported band throws with the C-ABI status
    bool threw = false;
    try { aln::batch_banded_alignment_score<9>( aln::make_gotoh_aligner<aln::LOCAL>( scheme ), batch, d_scores.data(), d_sinks.data() ); }
    catch (const error& e) { threw = (e.status == NVBIO_ERR_UNSUPPORTED); }
    REQUIRE( threw );
    printf( "host shim ok: %llu filter hits, %u x 3 banded alignments equal to the oracle\n", (unsigned long long)total, J );
    return 0;
}
