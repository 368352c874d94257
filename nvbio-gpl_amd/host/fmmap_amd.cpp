// fmmap_amd.cpp -- the reference's smallest complete seed-and-extend caller (examples/fmmap/fmmap.cu:217-390: extract seeds ->
// FMIndexFilter::rank -> locate -> hit_to_diagonal -> banded alignment of the window around each diagonal -> best score per
// read) as a plain C++ host program over the C ABI of include/nvbio_amd.h: no Python, no torch, no device code of its own.
// Seeding follows nvBowtie's exact-seed policy (both strands, seed length 22, interval int(1 + 1.15 sqrt(read_len)),
// mapping_inl.h:193-282, bowtie2_cuda_driver.cu:86-141), scoring its default end-to-end scheme (scoring_inl.h:99-114).
//
//   fmmap_amd --genome G.u32 --genome-len N --reads R.u32 --n-reads R --read-len M [--out best.bin] [--steps K] [--kmer k]
//   fmmap_amd --synthetic --genome-len N --n-reads R --read-len M [--steps K] [--kmer k]
//
// --genome: 2-bit big-endian packed words (io::SequenceData<DNA>); --reads: 4-bit big-endian packed words (io::SequenceData<DNA_N>),
// reads of equal length back to back.  --out: per read int32 score, int64 end position (-1: none), uint8 strand, as three arrays.
// Prints one JSON line: step time and a checksum of the results.
#include <nvbio_amd/nvbio_amd.hpp>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace nvbio_amd;

static std::vector<uint32_t> read_words(const char* path)
{
    FILE* f = fopen( path, "rb" );
    if (!f) { fprintf( stderr, "cannot open %s\n", path ); exit( 2 ); }
    fseek( f, 0, SEEK_END ); const long bytes = ftell( f ); fseek( f, 0, SEEK_SET );
    std::vector<uint32_t> w( (size_t)bytes / 4 + 8, 0u );
    if (fread( w.data(), 1, (size_t)bytes, f ) != (size_t)bytes) { fprintf( stderr, "short read on %s\n", path ); exit( 2 ); }
    fclose( f );
    return w;
}

struct Rng { uint64_t s; uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; } };

int main(int argc, char** argv)
{
    const char *genome_path = nullptr, *reads_path = nullptr, *out_path = nullptr;
    uint64_t genome_len = 0; uint32_t n_reads = 0, read_len = 150, steps = 3, kmer = 17, algo_flags = 0; bool synthetic = false, canonical = true, poison = false;
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf( stderr, "%s needs a value\n", a.c_str() ); exit( 2 ); } return argv[++i]; };
        if      (a == "--genome")     genome_path = val();
        else if (a == "--reads")      reads_path = val();
        else if (a == "--out")        out_path = val();
        else if (a == "--genome-len") genome_len = (uint64_t)atof( val() );
        else if (a == "--n-reads")    n_reads = (uint32_t)atof( val() );
        else if (a == "--read-len")   read_len = (uint32_t)atoi( val() );
        else if (a == "--steps")      steps = (uint32_t)atoi( val() );
        else if (a == "--kmer")       kmer = (uint32_t)atoi( val() );
        else if (a == "--synthetic")  synthetic = true;
        else if (a == "--no-canonical") canonical = false;
        else if (a == "--poison")     poison = true;                                // overwrite the score / sink arrays before every scoring call (no result may come from the step before)
        else if (a == "--algo-flags") algo_flags = (uint32_t)atoi( val() );       // nvbio_alignment_batch::algo_flags of the extension (NVBIO_ALN_*: A/B runs)
        else { fprintf( stderr, "unknown argument %s\n", a.c_str() ); return 2; }
    }
    if (!genome_len || !n_reads || (!synthetic && !(genome_path && reads_path))) { fprintf( stderr, "usage: see the head of fmmap_amd.cpp\n" ); return 2; }
    const uint32_t N = (uint32_t)genome_len, R = n_reads, M = read_len;

    try
    {
        // ---- inputs ----
        std::vector<uint32_t> h_genome, h_reads;
        if (synthetic)
        {
            Rng g = { 0x9E3779B97F4A7C15ull };
            h_genome.resize( (size_t)(N + 15) / 16 + 8 );
            for (auto& w : h_genome) w = (uint32_t)g.next();
            auto sym = [&](uint64_t i) -> uint32_t { return (h_genome[i >> 4] >> (30 - 2 * (i & 15))) & 3u; };
            h_reads.assign( ((size_t)R * M + 7) / 8 + 8, 0u );
            for (uint32_t r = 0; r < R; ++r)
            {
                const uint64_t p = g.next() % (N - M - 8);
                const bool rc = g.next() & 1;
                for (uint32_t k = 0; k < M; ++k)
                {
                    uint32_t c = rc ? 3u - sym( p + M - 1 - k ) : sym( p + k );
                    if (g.next() % 100 == 0) c = (c + 1 + g.next() % 3) & 3u;              // 1 % substitutions
                    const uint64_t i = (uint64_t)r * M + k;
                    h_reads[i >> 3] |= c << (28 - 4 * (i & 7));
                }
            }
        }
        else { h_genome = read_words( genome_path ); h_reads = read_words( reads_path ); }
        device_vector<uint32_t> d_genome( h_genome ), d_reads( h_reads );

        // ---- index: built on the GPU, full suffix array + direct table ----
        const auto tb0 = std::chrono::steady_clock::now();
        // (odd k: the canonical table with 16-byte entries, one seed pass for both strands; --no-canonical or even k: the direct table, one pass per strand)
        canonical = canonical && (kmer & 1u) && kmer >= 3u && kmer <= 22u && 22u - kmer <= 7u;      // the canonical table serves seeds of k .. k + 7 symbols
        fm_index fmi( d_genome.data(), N, 0, kmer, 0, /*sa_int*/ 1, canonical ? (uint32_t)NVBIO_FM_TABLE_CANONICAL_WIDE : 0u );
        check_hip( hipDeviceSynchronize(), "sync" );
        const double build_s = std::chrono::duration<double>( std::chrono::steady_clock::now() - tb0 ).count();

        // ---- the step ----
        const uint32_t L = 22, S = (uint32_t)(1.0 + 1.15 * std::sqrt( (double)M )), BAND = 31;
        const string_set seeds = string_set::seeds( d_reads.data(), 4, M, R, L, S );
        const uint32_t spr = seeds.c.seeds_per_string, n_seeds = seeds.size();
        std::unique_ptr<SeedPass> pass0, pass1; std::unique_ptr<SeedPassBoth> both;
        if (canonical) both.reset( new SeedPassBoth( seeds ) );
        else { pass0.reset( new SeedPass( seeds ) ); pass1.reset( new SeedPass( seeds ) ); }
        SeedPass* pass[2] = { pass0.get(), pass1.get() };
        device_vector<uint64_t> keys( 2ull * n_seeds + 128ull * (R / (spr <= 64u ? 64u / spr : 1u) + 1u) ), best( R );   // room for the two-strand pass's 128 keys per tile
        device_vector<uint32_t> offs( R + 1 ), rid, wb, we, n_unique( 1 );
        device_vector<uint8_t>  flags, rc( R );
        device_vector<int32_t>  scores, best_score( R );
        device_vector<nvbio_uint2> sinks;
        device_vector<int64_t>  best_pos( R );
        device_vector<uint64_t> slots;
        {
            std::vector<uint32_t> h( R + 1 ); for (uint32_t r = 0; r <= R; ++r) h[r] = r * M;
            offs.assign( h.data(), h.size() );
        }
        const aln::QualityGotohScheme scheme = { 0, 6, 6, 5, 3, 5, 3 };                     // end-to-end defaults, constant quality >= 40
        const nvbio_gotoh_scheme fs = scheme.flat();
        uint32_t* h_counts = nullptr;
        check_hip( hipHostMalloc( (void**)&h_counts, 4 * sizeof(uint32_t), 0 ), "hipHostMalloc" );
        hipEvent_t ev[2]; for (auto& e : ev) check_hip( hipEventCreate( &e ), "hipEventCreate" );

        uint64_t n_cand = 0; double step_ms = 0;
        for (uint32_t it = 0; it < steps + 1; ++it)                                          // the first pass warms up
        {
            check_hip( hipDeviceSynchronize(), "sync" );
            const auto t0 = std::chrono::steady_clock::now();
            uint64_t n = 0;
            // seeds on several SA rows (repeats): the ordinary scan + locate, then sort + unique of their diagonals, appended to keys
            auto residual = [&](const nvbio_uint2* ranges, const uint32_t* ids, const uint32_t nr, const uint32_t strand) {
                slots.resize( nr );
                uint64_t n_hits = 0;
                check( nvbio_fm_filter_scan( fmi.handle(), ranges, nr, slots.data(), &n_hits, 0 ) );
                if (n + n_hits > keys.size()) keys.resize( n + n_hits );
                check( nvbio_fm_filter_locate_diagonals( fmi.handle(), ranges, slots.data(), nullptr, nr, 0, n_hits, spr, S, L, M, strand, ids, keys.data() + n, 0 ) );
                check( nvbio_sort_unique_keys( 0, keys.data() + n, n_hits, n_unique.data(), nullptr, 0, 0 ) );
                uint32_t nu = 0;
                check_hip( hipMemcpy( &nu, n_unique.data(), sizeof(uint32_t), hipMemcpyDeviceToHost ), "hipMemcpy" );
                n += nu;
            };
            if (canonical)
            {
                // seeds -> diagonals of both strands in one launch; short repeats leave their keys directly
                both->enact( fmi, NVBIO_FM_INLINE_HITS( 4 ), M );
                check_hip( hipMemcpyAsync( h_counts, both->counts(), 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, 0 ), "hipMemcpyAsync" );
                check_hip( hipEventRecord( ev[0], 0 ), "hipEventRecord" );
                check_hip( hipEventSynchronize( ev[0] ), "hipEventSynchronize" );
                const uint32_t nk = h_counts[0];
                check_hip( hipMemcpyAsync( keys.data(), both->keys(), (size_t)nk * sizeof(uint64_t), hipMemcpyDeviceToDevice, 0 ), "hipMemcpyAsync" );
                n = nk;
                if (h_counts[1]) residual( both->residual_ranges(), both->residual_ids(), h_counts[1], 0u );
                if (h_counts[2]) residual( both->residual_ranges() + both->capacity(), both->residual_ids() + both->capacity(), h_counts[2], 1u );
            }
            else
            {
                // seeds -> diagonals, both strands enqueued before either count is awaited
                for (uint32_t strand = 0; strand < 2; ++strand)
                {
                    pass[strand]->enact( fmi, strand ? (NVBIO_FM_SCAN_FORWARD | NVBIO_FM_COMPLEMENT) : 0u, M, strand );
                    check_hip( hipMemcpyAsync( h_counts + 2 * strand, pass[strand]->counts(), 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, 0 ), "hipMemcpyAsync" );
                    check_hip( hipEventRecord( ev[strand], 0 ), "hipEventRecord" );
                }
                for (uint32_t strand = 0; strand < 2; ++strand)
                {
                    check_hip( hipEventSynchronize( ev[strand] ), "hipEventSynchronize" );
                    const uint32_t nk = h_counts[2 * strand], nr = h_counts[2 * strand + 1];
                    check_hip( hipMemcpyAsync( keys.data() + n, pass[strand]->keys(), (size_t)nk * sizeof(uint64_t), hipMemcpyDeviceToDevice, 0 ), "hipMemcpyAsync" );
                    n += nk;
                    if (nr) residual( pass[strand]->residual_ranges(), pass[strand]->residual_ids(), nr, strand );
                }
            }
            // diagonals -> windows -> banded Gotoh -> best per read
            rid.resize( n ); wb.resize( n ); we.resize( n ); flags.resize( n ); scores.resize( n ); sinks.resize( n );
            check_hip( hipMemsetAsync( best.data(), 0, (size_t)R * sizeof(uint64_t), 0 ), "hipMemsetAsync" );
            if (n)
            {
                check( nvbio_diagonals_to_windows( 0, keys.data(), n, BAND, M, N, rid.data(), flags.data(), wb.data(), we.data(), 0 ) );
                nvbio_alignment_batch b; memset( &b, 0, sizeof(b) );
                b.reads_dev = d_reads.data(); b.read_bits = 4; b.read_offsets_dev = offs.data(); b.read_id_dev = rid.data(); b.flags_dev = flags.data();
                b.text_dev = d_genome.data(); b.text_bits = 2; b.win_begin_dev = wb.data(); b.win_end_dev = we.data(); b.n = (uint32_t)n; b.max_read_len = M; b.algo_flags = algo_flags;
                if (poison)
                {
                    check_hip( hipMemsetAsync( scores.data(), 0x5A, (size_t)n * sizeof(int32_t), 0 ), "hipMemsetAsync" );
                    check_hip( hipMemsetAsync( sinks.data(), 0x5A, (size_t)n * sizeof(nvbio_uint2), 0 ), "hipMemsetAsync" );
                }
                check( nvbio_banded_gotoh_score( 0, BAND, NVBIO_SEMI_GLOBAL, &fs, &b, scores.data(), sinks.data(), 0 ) );
                check( nvbio_best_candidate_reduce( 0, keys.data(), scores.data(), sinks.data(), wb.data(), n, best.data(), 0 ) );
            }
            check( nvbio_best_candidate_unpack( 0, best.data(), R, best_score.data(), best_pos.data(), rc.data(), 0 ) );
            check_hip( hipDeviceSynchronize(), "sync" );
            if (it > 0) step_ms += std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count();
            n_cand = n;
        }
        step_ms /= steps;

        // ---- results ----
        const std::vector<int32_t> hs = best_score.to_host();
        const std::vector<int64_t> hp = best_pos.to_host();
        const std::vector<uint8_t> hr = rc.to_host();
        const int32_t min_score = (int32_t)(-0.6f + -0.6f * (float)M);
        uint64_t checksum = 1469598103934665603ull, aligned = 0;
        for (uint32_t r = 0; r < R; ++r)
        {
            checksum = (checksum ^ (uint64_t)(uint32_t)hs[r]) * 1099511628211ull;
            checksum = (checksum ^ (uint64_t)hp[r]) * 1099511628211ull;
            checksum = (checksum ^ hr[r]) * 1099511628211ull;
            aligned += hs[r] >= min_score;
        }
        if (out_path)
        {
            FILE* f = fopen( out_path, "wb" );
            if (!f) { fprintf( stderr, "cannot write %s\n", out_path ); return 2; }
            fwrite( hs.data(), sizeof(int32_t), R, f ); fwrite( hp.data(), sizeof(int64_t), R, f ); fwrite( hr.data(), 1, R, f );
            fclose( f );
        }
        printf( "{\"program\": \"fmmap_amd (C++ host over the C ABI)\", \"genome_len\": %u, \"reads\": %u, \"read_len\": %u, \"kmer_table\": %u, \"canonical_table\": %s, "
                "\"index_build_s\": %.3f, \"ms_per_step\": %.3f, \"reads_per_s\": %.1f, \"candidates\": %llu, \"aligned_fraction\": %.6f, "
                "\"checksum\": \"%016llx\"}\n",
                N, R, M, kmer, canonical ? "true" : "false", build_s, step_ms, R / (step_ms * 1e-3), (unsigned long long)n_cand, (double)aligned / R, (unsigned long long)checksum );
        (void)hipHostFree( h_counts );
    }
    catch (const std::exception& e) { fprintf( stderr, "fmmap_amd: %s\n", e.what() ); return 1; }
    return 0;
}
