// best_approx.hpp -- nvBowtie's best-approx single-end loop as a C++ HOST loop over the C ABI of libnvbio_amd (include/nvbio_amd.h).
//
// What it replaces: Aligner::best_approx + best_approx_score (nvBowtie/bowtie2/cuda/aligner_best_approx.h:39-207,363-667) -- the seeding
// passes with reseeding, and per seeding pass the extension loop select -> locate -> BestScoreStream -> banded DP -> score_reduce, including
// the several-hits-per-read phase the reference switches to once fewer than half a batch of reads are active (:487-510).  Every
// data-parallel step is a kernel behind the C ABI; this file holds no device code.  The host reads two counters per extension pass
// (active reads, selected hits) through pinned memory -- they size the next launches -- and nothing else; queues are allocated once per
// call, at their worst-case size (a batch of reads, BATCH_SIZE hits).
#pragma once
#include <nvbio_amd.h>
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace nvbio_amd_host {

struct BestApproxParams              // nvBowtie's defaults (bowtie2_cuda_driver.cu:86-141)
{
    uint32_t seed_len        = 22;
    uint32_t seed_freq       = 0;    // 0: S(1, 1.15): int( 1 + 1.15 sqrtf( read_len ) )
    uint32_t max_hits        = 100;
    uint32_t rep_seeds       = 1000;
    uint32_t max_effort      = 15;
    uint32_t max_effort_init = 15;
    uint32_t min_ext         = 30;
    uint32_t max_ext         = 400;
    uint32_t max_reseed      = 2;
    uint32_t band            = 31;
    uint32_t top_seed        = 0;
    uint32_t batch_size      = 0;    // BATCH_SIZE of the reference's multi-hit rule; 0: the number of reads of the call
    uint32_t multi_hit       = 1;    // 0: always one hit per read and pass
};

struct BestApproxStats { uint64_t n_extensions = 0; uint32_t passes = 0, multi_passes = 0, seeding_passes = 0; };

namespace detail {
inline void ok(nvbio_status st) { if (st != NVBIO_OK) throw std::runtime_error( std::string( "nvbio_amd: " ) + nvbio_amd_last_error() ); }
inline void hip(hipError_t e) { if (e != hipSuccess) throw std::runtime_error( std::string( "hip: " ) + hipGetErrorString( e ) ); }
struct DevBuf
{
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { hip( hipMalloc( &p, bytes ? bytes : 16 ) ); }
    ~DevBuf() { if (p) (void)hipFree( p ); }
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    template <typename T> T* as() const { return (T*)p; }
};
}

// stored_reads4_dev: the reads as nvBowtie stores them (io::REVERSE), 4-bit packed, read r at symbols [r * read_len, (r+1) * read_len);
// quals_dev: one byte per stored symbol or NULL; best_dev [4 n_reads] int32 (16-byte aligned) / best_rc_dev [n_reads]: see
// nvbio_score_reduce_effort.  worst_score = the scheme's min_score( read_len ) (init_alignments' threshold).
inline BestApproxStats best_approx(int device, nvbio_fm_index_t fmi, const uint32_t* genome2_dev, uint32_t genome_len, const uint32_t* stored_reads4_dev,
                                   const uint8_t* quals_dev, uint32_t n_reads, uint32_t read_len, nvbio_alignment_type aln_type, const nvbio_gotoh_scheme& scheme,
                                   int32_t worst_score, const BestApproxParams& prm, int32_t* best_dev, uint8_t* best_rc_dev, hipStream_t stream)
{
    using namespace detail;
    BestApproxStats stats;
    const uint32_t R = n_reads, M = read_len;
    if (R == 0) return stats;
    hip( hipSetDevice( device ) );
    const uint32_t L = prm.seed_len < M ? prm.seed_len : M;
    const uint32_t S = prm.seed_freq ? prm.seed_freq : (uint32_t)(int32_t)(1.0f + 1.15f * sqrtf( (float)M ));      // SimpleFunc (params.h:87-100)
    const uint32_t retry_stride = S / (prm.max_reseed + 1u);
    const uint32_t max_effort_init = prm.max_effort_init > prm.max_effort ? prm.max_effort_init : prm.max_effort;
    const uint32_t max_ext = prm.max_ext > prm.max_effort ? prm.max_ext : prm.max_effort;
    const uint32_t BATCH = prm.batch_size ? prm.batch_size : R;
    const uint32_t spr_max = M >= L ? (M - L) / S + 1u : 0u;
    if (spr_max == 0) { ok( nvbio_best_approx_init( device, R, worst_score, best_dev, best_rc_dev, stream ) ); return stats; }

    uint32_t cap = 0; ok( nvbio_seed_hits_capacity( spr_max, prm.max_hits, &cap ) );
    const uint64_t hits_cap = (uint64_t)(BATCH > R ? BATCH : R);
    DevBuf read_index( 4ull * (R + 1) ), queue_a( 4ull * R ), queue_b( 4ull * R ), offs( 4ull * R ), fw( 8ull * R * spr_max ), rc( 8ull * R * spr_max ),
           deques( 8ull * R * cap ), sizes( 4ull * R ), reseed( R ), trys( 4ull * R ), active_a( 4ull * R ), active_b( 4ull * R ), hits_first( 4ull * R ),
           hits_count( 4ull * R ), h_read( 4ull * hits_cap ), h_seed( 4ull * hits_cap ), h_loc( 4ull * hits_cap ), h_score( 4ull * hits_cap ),
           h_sink( 4ull * hits_cap ), pos( 4ull * hits_cap ), j_read( 4ull * hits_cap ), j_flags( hits_cap ), j_wb( 4ull * hits_cap ), j_we( 4ull * hits_cap ),
           j_scores( 4ull * hits_cap ), j_sinks( 8ull * hits_cap ), counts( 16 );
    uint32_t* h_counts = nullptr; hip( hipHostMalloc( (void**)&h_counts, 16, hipHostMallocDefault ) );
    struct Pinned { uint32_t* p; ~Pinned() { (void)hipHostFree( p ); } } pinned = { h_counts };

    {   // the read batch's sequence_index
        std::vector<uint32_t> ri( R + 1 );
        for (uint32_t r = 0; r <= R; ++r) ri[r] = r * M;
        hip( hipMemcpyAsync( read_index.p, ri.data(), 4ull * (R + 1), hipMemcpyHostToDevice, stream ) );
        hip( hipStreamSynchronize( stream ) );
    }
    ok( nvbio_best_approx_init( device, R, worst_score, best_dev, best_rc_dev, stream ) );

    auto fetch_counts = [&](uint32_t words) {
        hip( hipMemcpyAsync( h_counts, counts.p, 4ull * words, hipMemcpyDeviceToHost, stream ) );
        hip( hipStreamSynchronize( stream ) );
    };

    const uint32_t* queue = nullptr;              // seed_queues: the reads of this seeding pass (NULL: all of them)
    uint32_t nq = R;
    uint32_t* queue_bufs[2] = { queue_a.as<uint32_t>(), queue_b.as<uint32_t>() };
    for (uint32_t seeding_pass = 0; seeding_pass <= prm.max_reseed && nq; ++seeding_pass)
    {
        const uint32_t first = seeding_pass * retry_stride;
        if (M < L + first) break;
        const uint32_t spr = (M - L - first) / S + 1u;
        ++stats.seeding_passes;
        nvbio_seed_hits_params sp = { spr, first, S, L, M, prm.max_hits, prm.rep_seeds, prm.max_effort, prm.min_ext, max_ext };
        ok( nvbio_seed_hits_capacity( spr, prm.max_hits, &cap ) );
        // the seeds of the queued reads, both match_range calls of the exact mapper, the deques
        ok( nvbio_read_queue_begin( device, queue, nq, M, first, prm.top_seed, max_effort_init, offs.as<uint32_t>(), active_a.as<uint32_t>(), trys.as<uint32_t>(), stream ) );
        nvbio_string_set qs = { stored_reads4_dev, 4u, offs.as<uint32_t>(), 0u, L, M, nq * spr, spr, S, nullptr };
        ok( nvbio_fm_match( fmi, &qs, NVBIO_FM_SCAN_FORWARD, fw.as<nvbio_uint2>(), nullptr, stream ) );
        ok( nvbio_fm_match( fmi, &qs, NVBIO_FM_COMPLEMENT,   rc.as<nvbio_uint2>(), nullptr, stream ) );
        hip( hipMemsetAsync( sizes.p, 0, 4ull * R, stream ) );
        hip( hipMemsetAsync( reseed.p, 0, R, stream ) );
        ok( nvbio_seed_hits_map( device, fw.as<nvbio_uint2>(), rc.as<nvbio_uint2>(), queue, nq, &sp, deques.as<nvbio_uint2>(), sizes.as<uint32_t>(),
                                 reseed.as<uint8_t>(), stream ) );

        // the extension loop (best_approx_score)
        uint32_t* active_in = active_a.as<uint32_t>(); uint32_t* active_out = active_b.as<uint32_t>();
        uint32_t n_active = nq, n_ext = 0;
        while (n_active && n_ext < max_ext)
        {
            uint32_t n_multi = 1;
            if (prm.multi_hit && n_active <= BATCH / 2u)
            {
                const uint32_t left = max_ext - n_ext < 4096u ? max_ext - n_ext : 4096u;
                n_multi = BATCH / n_active < left ? BATCH / n_active : left;
                if (n_multi < 1u) n_multi = 1u;
            }
            nvbio_hit_queues hq = { nullptr, h_read.as<uint32_t>(), h_seed.as<uint32_t>(), h_loc.as<uint32_t>(), h_score.as<int32_t>(), h_sink.as<uint32_t>(), 0u };
            ok( nvbio_seed_hits_select_multi( device, active_in, n_active, trys.as<uint32_t>(), cap, n_multi, deques.as<nvbio_uint2>(), sizes.as<uint32_t>(),
                                              active_out, hits_first.as<uint32_t>(), hits_count.as<uint32_t>(), &hq, counts.as<uint32_t>(), stream ) );
            fetch_counts( 2 );
            const uint32_t n_out = h_counts[0], n_hits = h_counts[1];
            if (n_out == 0) break;
            hq.n = n_hits;
            ok( nvbio_fm_locate( fmi, hq.hit_loc_dev, n_hits, pos.as<uint32_t>(), stream ) );
            ok( nvbio_seed_hits_loc( device, pos.as<uint32_t>(), &hq, stream ) );
            ok( nvbio_score_stream_flatten( device, &hq, read_index.as<uint32_t>(), prm.band, genome_len, 1u, j_read.as<uint32_t>(), j_flags.as<uint8_t>(),
                                            j_wb.as<uint32_t>(), j_we.as<uint32_t>(), stream ) );
            nvbio_alignment_batch batch = { stored_reads4_dev, 4u, read_index.as<uint32_t>(), quals_dev, j_read.as<uint32_t>(), j_flags.as<uint8_t>(), genome2_dev, 2u,
                                            j_wb.as<uint32_t>(), j_we.as<uint32_t>(), n_hits, M, 0u };
            ok( nvbio_banded_gotoh_score( device, prm.band, aln_type, &scheme, &batch, j_scores.as<int32_t>(), j_sinks.as<nvbio_uint2>(), stream ) );
            ok( nvbio_score_stream_output( device, &hq, j_scores.as<int32_t>(), j_sinks.as<nvbio_uint2>(), j_wb.as<uint32_t>(), -65536, stream ) );
            ok( nvbio_score_reduce_effort_multi( device, active_out, n_out, hits_first.as<uint32_t>(), hits_count.as<uint32_t>(), &hq, M, n_ext, &sp,
                                                 best_dev, best_rc_dev, trys.as<uint32_t>(), sizes.as<uint32_t>(), stream ) );
            n_ext += n_multi;
            stats.n_extensions += n_hits; ++stats.passes; if (n_multi > 1u) ++stats.multi_passes;
            std::swap( active_in, active_out );
            n_active = n_out;
        }
        // the reads that asked for reseeding go round again
        uint32_t* next = queue_bufs[seeding_pass & 1u];
        ok( nvbio_read_queue_filter( device, queue, nq, reseed.as<uint8_t>(), next, counts.as<uint32_t>(), stream ) );
        fetch_counts( 1 );
        queue = next; nq = h_counts[0];
    }
    hip( hipStreamSynchronize( stream ) );
    return stats;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The PAIRED-END form (Aligner::best_approx, aligner_best_approx_paired.h:84-200 and its best_approx_score, :590-1000): for anchor = mate 1, then
// mate 2: the seeding passes and the extension loop of the single-end form over the ANCHOR mate's seed hits, where a selected hit is scored
// as a pair -- anchor band-aligned against a threshold that tightens with the pairs found so far, the opposite mate by full-matrix DP in its
// fragment window for the hits whose anchor passed, score_reduce_paired keeping the best two pairs (or per-mate bests while unpaired).
// Three counters per extension pass through pinned memory (active reads, selected hits, hits whose anchor passed).
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct PairedParams { uint32_t policy = NVBIO_PE_POLICY_FR, min_frag_len = 0, max_frag_len = 500, overlap = 1, unpaired = 1; };
struct PairedStats  { uint64_t n_extensions = 0, n_opposite = 0; uint32_t passes = 0, multi_passes = 0; };

// stored_reads4_dev[m]: mate m+1 of every pair, stored reversed, 4-bit packed, uniform length read_len[m]; best_a_dev / best_o_dev: [n_reads][2][4] int32
// (see nvbio_pe_params); worst_score[m] = scheme.min_score( read_len[m] ).
inline PairedStats best_approx_paired(int device, nvbio_fm_index_t fmi, const uint32_t* genome2_dev, uint32_t genome_len, const uint32_t* const stored_reads4_dev[2],
                                      const uint8_t* const quals_dev[2], uint32_t n_reads, const uint32_t read_len[2], nvbio_alignment_type aln_type,
                                      const nvbio_gotoh_scheme& scheme, const int32_t worst_score[2], const BestApproxParams& prm, const PairedParams& pe,
                                      int32_t* best_a_dev, int32_t* best_o_dev, hipStream_t stream)
{
    using namespace detail;
    PairedStats stats;
    const uint32_t R = n_reads;
    if (R == 0) return stats;
    hip( hipSetDevice( device ) );
    const uint32_t Mmax = read_len[0] > read_len[1] ? read_len[0] : read_len[1];
    const uint32_t max_effort_init = prm.max_effort_init > prm.max_effort ? prm.max_effort_init : prm.max_effort;
    const uint32_t max_ext = prm.max_ext > prm.max_effort ? prm.max_ext : prm.max_effort;
    const uint32_t BATCH = prm.batch_size ? prm.batch_size : R;
    const uint64_t hits_cap = (uint64_t)(BATCH > R ? BATCH : R);
    uint32_t spr_max = 1, cap_max = 0;
    for (int m = 0; m < 2; ++m)
    {
        const uint32_t M = read_len[m], L = prm.seed_len < M ? prm.seed_len : M;
        const uint32_t S = prm.seed_freq ? prm.seed_freq : (uint32_t)(int32_t)(1.0f + 1.15f * sqrtf( (float)M ));
        const uint32_t spr = M >= L ? (M - L) / S + 1u : 0u;
        if (spr > spr_max) spr_max = spr;
    }
    ok( nvbio_seed_hits_capacity( spr_max, prm.max_hits, &cap_max ) );
    DevBuf read_index0( 4ull * (R + 1) ), read_index1( 4ull * (R + 1) ), queue_a( 4ull * R ), queue_b( 4ull * R ), offs( 4ull * R ), fw( 8ull * R * spr_max ),
           rc( 8ull * R * spr_max ), deques( 8ull * R * cap_max ), sizes( 4ull * R ), reseed( R ), trys( 4ull * R ), active_a( 4ull * R ), active_b( 4ull * R ),
           hits_first( 4ull * R ), hits_count( 4ull * R ), h_read( 4ull * hits_cap ), h_seed( 4ull * hits_cap ), h_loc( 4ull * hits_cap ), h_score( 4ull * hits_cap ),
           h_sink( 4ull * hits_cap ), h_oscore( 4ull * hits_cap ), h_oloc( 4ull * hits_cap ), h_osink( 4ull * hits_cap ), pos( 4ull * hits_cap ),
           j_read( 4ull * hits_cap ), j_flags( hits_cap ), j_wb( 4ull * hits_cap ), j_we( 4ull * hits_cap ), j_min( 4ull * hits_cap ), j_scores( 4ull * hits_cap ),
           j_sinks( 8ull * hits_cap ), valid( hits_cap ), oqueue( 4ull * hits_cap ), counts( 16 );
    uint32_t* h_counts = nullptr; hip( hipHostMalloc( (void**)&h_counts, 16, hipHostMallocDefault ) );
    struct Pinned { uint32_t* p; ~Pinned() { (void)hipHostFree( p ); } } pinned = { h_counts };
    DevBuf* read_index[2] = { &read_index0, &read_index1 };
    for (int m = 0; m < 2; ++m)
    {
        std::vector<uint32_t> ri( R + 1 );
        for (uint32_t r = 0; r <= R; ++r) ri[r] = r * read_len[m];
        hip( hipMemcpyAsync( read_index[m]->p, ri.data(), 4ull * (R + 1), hipMemcpyHostToDevice, stream ) );
        hip( hipStreamSynchronize( stream ) );
    }
    ok( nvbio_pe_init( device, R, worst_score[0], worst_score[1], best_a_dev, best_o_dev, stream ) );
    auto fetch_counts = [&](uint32_t words, uint32_t at = 0) {
        hip( hipMemcpyAsync( h_counts + at, counts.as<uint32_t>() + at, 4ull * words, hipMemcpyDeviceToHost, stream ) );
        hip( hipStreamSynchronize( stream ) );
    };

    for (uint32_t anchor = 0; anchor < 2; ++anchor)
    {
        const uint32_t a = anchor, o = 1u - anchor;
        const uint32_t M = read_len[a], Mo = read_len[o];
        const uint32_t L = prm.seed_len < M ? prm.seed_len : M;
        const uint32_t S = prm.seed_freq ? prm.seed_freq : (uint32_t)(int32_t)(1.0f + 1.15f * sqrtf( (float)M ));
        const uint32_t retry_stride = S / (prm.max_reseed + 1u);
        if (M < L) continue;
        nvbio_pe_params pp = { anchor, M, Mo, scheme.match * (int32_t)M, scheme.match * (int32_t)Mo, worst_score[a], worst_score[o], NVBIO_SCORE_MIN, -65536,
                               scheme.match, scheme.txt_gap_open, scheme.txt_gap_ext, prm.band, genome_len, pe.policy, pe.min_frag_len, pe.max_frag_len,
                               pe.overlap, pe.unpaired, prm.max_effort, prm.min_ext, max_ext };
        const uint32_t* queue = nullptr; uint32_t nq = R;
        uint32_t* queue_bufs[2] = { queue_a.as<uint32_t>(), queue_b.as<uint32_t>() };
        for (uint32_t seeding_pass = 0; seeding_pass <= prm.max_reseed && nq; ++seeding_pass)
        {
            const uint32_t first = seeding_pass * retry_stride;
            if (M < L + first) break;
            const uint32_t spr = (M - L - first) / S + 1u;
            uint32_t cap = 0; ok( nvbio_seed_hits_capacity( spr, prm.max_hits, &cap ) );
            nvbio_seed_hits_params sp = { spr, first, S, L, M, prm.max_hits, prm.rep_seeds, prm.max_effort, prm.min_ext, max_ext };
            ok( nvbio_read_queue_begin( device, queue, nq, M, first, prm.top_seed, max_effort_init, offs.as<uint32_t>(), active_a.as<uint32_t>(), trys.as<uint32_t>(), stream ) );
            nvbio_string_set qs = { stored_reads4_dev[a], 4u, offs.as<uint32_t>(), 0u, L, M, nq * spr, spr, S, nullptr };
            ok( nvbio_fm_match( fmi, &qs, NVBIO_FM_SCAN_FORWARD, fw.as<nvbio_uint2>(), nullptr, stream ) );
            ok( nvbio_fm_match( fmi, &qs, NVBIO_FM_COMPLEMENT,   rc.as<nvbio_uint2>(), nullptr, stream ) );
            hip( hipMemsetAsync( sizes.p, 0, 4ull * R, stream ) );
            hip( hipMemsetAsync( reseed.p, 0, R, stream ) );
            ok( nvbio_seed_hits_map( device, fw.as<nvbio_uint2>(), rc.as<nvbio_uint2>(), queue, nq, &sp, deques.as<nvbio_uint2>(), sizes.as<uint32_t>(),
                                     reseed.as<uint8_t>(), stream ) );
            uint32_t* active_in = active_a.as<uint32_t>(); uint32_t* active_out = active_b.as<uint32_t>();
            uint32_t n_active = nq, n_ext = 0;
            while (n_active && n_ext < max_ext)
            {
                uint32_t n_multi = 1;
                if (prm.multi_hit && n_active <= BATCH / 2u)
                {
                    const uint32_t left = max_ext - n_ext < 4096u ? max_ext - n_ext : 4096u;
                    n_multi = BATCH / n_active < left ? BATCH / n_active : left;
                    if (n_multi < 1u) n_multi = 1u;
                }
                nvbio_hit_queues hq = { nullptr, h_read.as<uint32_t>(), h_seed.as<uint32_t>(), h_loc.as<uint32_t>(), h_score.as<int32_t>(), h_sink.as<uint32_t>(), 0u };
                ok( nvbio_seed_hits_select_multi( device, active_in, n_active, trys.as<uint32_t>(), cap, n_multi, deques.as<nvbio_uint2>(), sizes.as<uint32_t>(),
                                                  active_out, hits_first.as<uint32_t>(), hits_count.as<uint32_t>(), &hq, counts.as<uint32_t>(), stream ) );
                fetch_counts( 2 );
                const uint32_t n_out = h_counts[0], n_hits = h_counts[1];
                if (n_out == 0) break;
                hq.n = n_hits;
                ok( nvbio_fm_locate( fmi, hq.hit_loc_dev, n_hits, pos.as<uint32_t>(), stream ) );
                ok( nvbio_seed_hits_loc( device, pos.as<uint32_t>(), &hq, stream ) );
                // anchor: band-aligned against the pair-derived threshold
                ok( nvbio_pe_anchor_flatten( device, &pp, &hq, best_a_dev, best_o_dev, j_read.as<uint32_t>(), j_flags.as<uint8_t>(), j_wb.as<uint32_t>(),
                                             j_we.as<uint32_t>(), j_min.as<int32_t>(), stream ) );
                nvbio_alignment_batch ab = { stored_reads4_dev[a], 4u, read_index[a]->as<uint32_t>(), quals_dev[a], j_read.as<uint32_t>(), j_flags.as<uint8_t>(), genome2_dev,
                                             2u, j_wb.as<uint32_t>(), j_we.as<uint32_t>(), n_hits, M, 0u };
                ok( nvbio_banded_gotoh_score( device, prm.band, aln_type, &scheme, &ab, j_scores.as<int32_t>(), j_sinks.as<nvbio_uint2>(), stream ) );
                ok( nvbio_pe_anchor_output( device, &pp, &hq, j_scores.as<int32_t>(), j_sinks.as<nvbio_uint2>(), j_wb.as<uint32_t>(), j_min.as<int32_t>(),
                                            h_oscore.as<int32_t>(), valid.as<uint8_t>(), stream ) );
                // opposite mate: full-matrix DP for the hits whose anchor passed
                ok( nvbio_select_flagged_indices( device, valid.as<uint8_t>(), n_hits, oqueue.as<uint32_t>(), counts.as<uint32_t>() + 2, stream ) );
                fetch_counts( 1, 2 );
                const uint32_t n_opp = h_counts[2];
                if (n_opp)
                {
                    ok( nvbio_pe_opposite_flatten( device, &pp, oqueue.as<uint32_t>(), n_opp, &hq, best_a_dev, best_o_dev, j_read.as<uint32_t>(), j_flags.as<uint8_t>(),
                                                   j_wb.as<uint32_t>(), j_we.as<uint32_t>(), j_min.as<int32_t>(), stream ) );
                    nvbio_alignment_batch ob = { stored_reads4_dev[o], 4u, read_index[o]->as<uint32_t>(), quals_dev[o], j_read.as<uint32_t>(), j_flags.as<uint8_t>(),
                                                 genome2_dev, 2u, j_wb.as<uint32_t>(), j_we.as<uint32_t>(), n_opp, Mo, 0u };
                    ok( nvbio_full_gotoh_score( device, aln_type, 0 /* pattern blocking */, &scheme, &ob, Mo, pe.max_frag_len, j_min.as<int32_t>(),
                                                j_scores.as<int32_t>(), j_sinks.as<nvbio_uint2>(), nullptr, 0, stream ) );
                    ok( nvbio_pe_opposite_output( device, &pp, oqueue.as<uint32_t>(), n_opp, j_scores.as<int32_t>(), j_sinks.as<nvbio_uint2>(), j_wb.as<uint32_t>(),
                                                  j_we.as<uint32_t>(), j_min.as<int32_t>(), h_oscore.as<int32_t>(), h_oloc.as<uint32_t>(), h_osink.as<uint32_t>(), stream ) );
                }
                ok( nvbio_pe_score_reduce( device, &pp, active_out, n_out, hits_first.as<uint32_t>(), hits_count.as<uint32_t>(), &hq, h_oscore.as<int32_t>(),
                                           h_oloc.as<uint32_t>(), h_osink.as<uint32_t>(), n_ext, best_a_dev, best_o_dev, trys.as<uint32_t>(), sizes.as<uint32_t>(), stream ) );
                n_ext += n_multi;
                stats.n_extensions += n_hits; stats.n_opposite += n_opp; ++stats.passes; if (n_multi > 1u) ++stats.multi_passes;
                std::swap( active_in, active_out );
                n_active = n_out;
            }
            uint32_t* next = queue_bufs[seeding_pass & 1u];
            ok( nvbio_read_queue_filter( device, queue, nq, reseed.as<uint8_t>(), next, counts.as<uint32_t>(), stream ) );
            fetch_counts( 1 );
            queue = next; nq = h_counts[0];
        }
    }
    hip( hipStreamSynchronize( stream ) );
    return stats;
}

} // namespace nvbio_amd_host
