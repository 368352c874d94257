// nvbio_amd.hpp -- C++ host-side mirror of the reference's operator API for the seed-and-extend
// path, over the C ABI of include/nvbio_amd.h.  Header-only; no HIP code in here (it only needs
// the HIP runtime API for device buffers), so it compiles with g++ or hipcc.
//
// It keeps the reference's names, argument meaning and ownership rules:
//   nvbio_amd::fm_index            ~ nvbio::fm_index + io::FMIndexDataDevice (nvbio/fmindex/fmindex.h:320-361,
//                                    nvbio/io/fmindex/fmindex.h:263-330): a storage-free view + its device owner
//   match / locate                 ~ the free functions of nvbio/fmindex/fmindex.h:370-557, batched over a set
//   FMIndexFilter<amd_device_tag>  ~ nvbio::FMIndexFilter<device_tag,fm_index> (nvbio/fmindex/filter.h:136-231):
//                                    rank(index, string_set) -> n_hits ; locate(begin, end, hits) ; owns ranges/slots
//   aln::SimpleGotohScheme, aln::GotohAligner<TYPE,scheme>, aln::make_gotoh_aligner<TYPE>()
//                                  ~ nvbio/alignment/utils.h:103-123, alignment.h:437-462
//   aln::BatchedBandedAlignmentScore<BAND, stream, AmdDeviceScheduler>
//                                  ~ nvbio/alignment/batched.h:298 / batched_banded_inl.h:128-157:
//                                    min_temp_storage / max_temp_storage / enact(stream, temp_size, temp)
//   aln::BatchedBandedAlignmentTraceback<BAND, CHECKPOINTS, stream, AmdDeviceScheduler>
//                                  ~ nvbio/alignment/batched.h:420-437; output = Alignment + nvBowtie's io::Cigar runs
//   aln::batch_banded_alignment_score<BAND>(aligner, batch, scores, sinks)
//                                  ~ nvbio/alignment/batched.h:185
// Errors: the reference surfaces CUDA failures as nvbio::cuda_error exceptions
// (nvbio/basic/cuda/arch_inl.h:228-237); this shim throws nvbio_amd::error carrying the C-ABI status.
//
// The reference's stream concept is a set of device functors (init_context / load_strings /
// output); a functor cannot cross a C ABI, so the AMD scheduler accepts streams that expose their
// jobs as flat arrays: `const nvbio_alignment_batch& batch() const`, `int32_t* scores()`,
// `nvbio_uint2* sinks()`, `aligner()`.  aln::FlatAlignmentStream below is such a stream;
// INTEGRATION.md shows how nvBowtie's BestScoreStream and sw-benchmark's stream fill one.
#pragma once
#include <nvbio_amd.h>
#include <hip/hip_runtime_api.h>
#include <stdexcept>
#include <string>
#include <vector>
#include <cstdint>

namespace nvbio_amd {

struct error : std::runtime_error
{
    nvbio_status status;
    error(nvbio_status s, const std::string& what) : std::runtime_error( what ), status( s ) {}
};
inline void check(nvbio_status s)
{
    if (s != NVBIO_OK) throw error( s, std::string( "nvbio_amd: " ) + nvbio_amd_last_error() );
}
inline void check_hip(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw error( NVBIO_ERR_HIP, std::string( what ) + ": " + hipGetErrorString( e ) );
}

struct amd_device_tag {};

// minimal owning device array (the role nvbio::vector<device_tag,T> plays in the reference)
template <typename T>
class device_vector
{
public:
    device_vector() : m_ptr( nullptr ), m_size( 0 ), m_cap( 0 ) {}
    explicit device_vector(size_t n) : m_ptr( nullptr ), m_size( 0 ), m_cap( 0 ) { resize( n ); }
    device_vector(const std::vector<T>& h) : m_ptr( nullptr ), m_size( 0 ), m_cap( 0 ) { assign( h.data(), h.size() ); }
    device_vector(const device_vector&) = delete;
    device_vector& operator=(const device_vector&) = delete;
    ~device_vector() { if (m_ptr) (void)hipFree( m_ptr ); }
    void resize(size_t n)
    {
        if (n > m_cap)
        {
            T* p = nullptr;
            check_hip( hipMalloc( (void**)&p, (n ? n : 1) * sizeof(T) ), "hipMalloc" );
            if (m_ptr) { if (m_size) check_hip( hipMemcpy( p, m_ptr, m_size * sizeof(T), hipMemcpyDeviceToDevice ), "hipMemcpy" ); (void)hipFree( m_ptr ); }
            m_ptr = p; m_cap = n;
        }
        m_size = n;
    }
    void assign(const T* h, size_t n) { resize( n ); if (n) check_hip( hipMemcpy( m_ptr, h, n * sizeof(T), hipMemcpyHostToDevice ), "hipMemcpy H2D" ); }
    std::vector<T> to_host() const
    {
        std::vector<T> h( m_size );
        if (m_size) check_hip( hipMemcpy( h.data(), m_ptr, m_size * sizeof(T), hipMemcpyDeviceToHost ), "hipMemcpy D2H" );
        return h;
    }
    T*       data()       { return m_ptr; }
    const T* data() const { return m_ptr; }
    size_t   size() const { return m_size; }
private:
    T* m_ptr; size_t m_size, m_cap;
};

// a set of strings in device memory (string-set concept flattened; see nvbio_string_set)
struct string_set
{
    nvbio_string_set c;
    uint32_t size() const { return c.n; }
    // concatenated set with n+1 offsets (io::SequenceData layout)
    static string_set concatenated(const void* symbols, uint32_t bits, const uint32_t* offsets, uint32_t n)
    { string_set s; s.c = { symbols, bits, offsets, 1u, 0u, 0u, n }; return s; }
    // fixed-length infixes addressed by their start (seeds inside a read stream)
    static string_set infixes(const void* symbols, uint32_t bits, const uint32_t* starts, uint32_t len, uint32_t n)
    { string_set s; s.c = { symbols, bits, starts, 0u, len, len, n }; return s; }
    // the uniformly spaced seeds of n_strings equal-length strings laid out back to back
    // (uniform_seeds_functor; nvBowtie mapping_inl.h:485-556): n_strings * seeds_per_string queries
    static string_set seeds(const void* symbols, uint32_t bits, uint32_t string_len, uint32_t n_strings,
                            uint32_t seed_len, uint32_t seed_interval)
    {
        const uint32_t spr = (string_len - seed_len) / seed_interval + 1u;
        string_set s; s.c = { symbols, bits, nullptr, 0u, seed_len, string_len, n_strings * spr, spr, seed_interval }; return s;
    }
    // n strings of equal length laid out back to back
    static string_set uniform(const void* symbols, uint32_t bits, uint32_t len, uint32_t n)
    { string_set s; s.c = { symbols, bits, nullptr, 0u, len, len, n }; return s; }
};

// FM-index resident in HBM (owner of the C-ABI handle)
class fm_index
{
public:
    typedef uint32_t    index_type;
    typedef nvbio_uint2 range_type;

    // wrap arrays already in device memory (reference layout; the caller keeps ownership)
    fm_index(const nvbio_fm_index_view& view, int device = 0, uint32_t kmer_len = 0, hipStream_t stream = 0)
        : m_h( nullptr ), m_device( device ) { check( nvbio_fm_index_create( &view, device, kmer_len, stream, &m_h ) ); }
    // build on the GPU from a 2-bit packed text in device memory
    // (table_flags: NVBIO_FM_TABLE_*, e.g. NVBIO_FM_TABLE_CANONICAL for the two-strand seed pass)
    fm_index(const uint32_t* text2_dev, uint32_t length, int device = 0, uint32_t kmer_len = 0, hipStream_t stream = 0, uint32_t sa_int = 16,
             uint32_t table_flags = 0)
        : m_h( nullptr ), m_device( device )
    {
        const nvbio_fm_build_options opts = { kmer_len, sa_int, 0u, 0u, table_flags, 0u };
        check( nvbio_fm_index_build( text2_dev, length, device, &opts, stream, &m_h ) );
    }
    // load the reference's on-disk index files (<prefix>.bwt / .sa as written by nvBWT, read by io::FMIndexDataHost::load,
    // nvbio/io/fmindex/fmindex_impl.cu:111-252); sa_path may be NULL
    fm_index(const char* bwt_path, const char* sa_path, int device = 0, uint32_t kmer_len = 0, hipStream_t stream = 0)
        : m_h( nullptr ), m_device( device ) { check( nvbio_fm_index_load( bwt_path, sa_path, device, kmer_len, stream, &m_h ) ); }
    void save(const char* bwt_path, const char* sa_path = nullptr, hipStream_t stream = 0) const { check( nvbio_fm_index_save( m_h, bwt_path, sa_path, stream ) ); }
    bool supports_direct() const { int yes = 0; check( nvbio_fm_index_supports_direct( m_h, &yes ) ); return yes != 0; }
    uint32_t canonical_kmer() const { return (uint32_t)nvbio_fm_index_is_canonical( m_h ); }    // 0: no canonical table
    bool canonical() const { return canonical_kmer() != 0; }
    fm_index(const fm_index&) = delete;
    fm_index& operator=(const fm_index&) = delete;
    ~fm_index() { if (m_h) (void)nvbio_fm_index_destroy( m_h ); }

    index_type length()  const { return view().length; }
    index_type primary() const { return view().primary; }
    index_type L2(uint32_t c) const { return view().L2[c]; }
    index_type count(uint32_t c) const { const nvbio_fm_index_view v = view(); return v.L2[c+1] - v.L2[c]; }
    nvbio_fm_index_view view() const { nvbio_fm_index_view v; check( nvbio_fm_index_get_view( m_h, &v ) ); return v; }
    nvbio_fm_index_t handle() const { return m_h; }
    int device() const { return m_device; }
private:
    nvbio_fm_index_t m_h;
    int              m_device;
};

// batched forms of the free functions of nvbio/fmindex/fmindex.h:370-557
inline void match(const fm_index& fmi, const string_set& patterns, nvbio_uint2* ranges_dev, hipStream_t stream = 0)
{ check( nvbio_fm_match( fmi.handle(), &patterns.c, 0u, ranges_dev, nullptr, stream ) ); }
inline void match_reverse(const fm_index& fmi, const string_set& patterns, nvbio_uint2* ranges_dev, hipStream_t stream = 0)
{ check( nvbio_fm_match( fmi.handle(), &patterns.c, NVBIO_FM_SCAN_FORWARD, ranges_dev, nullptr, stream ) ); }
// nvbio::hamming_backtrack over a batch with the reference benchmark's counting delegate (nvbio/fmindex/backtrack.h:51-157,
// nvbio-test/fmindex_test.cu:720-800): counts_dev[i] = occurrences of pattern i within `mismatches` substitutions outside its
// exactly matched last `seed` symbols.  reference_quirks: count as the reference's code does (see nvbio_amd.h).
inline void hamming_backtrack(const fm_index& fmi, const string_set& patterns, uint32_t seed, uint32_t mismatches, uint32_t* counts_dev,
                              uint32_t* n_ranges_dev = nullptr, nvbio_uint2* ranges_dev = nullptr, uint32_t max_ranges = 0,
                              bool reference_quirks = false, hipStream_t stream = 0)
{
    check( nvbio_fm_hamming_backtrack( fmi.handle(), &patterns.c, seed, mismatches, reference_quirks ? NVBIO_BACKTRACK_REFERENCE_QUIRKS : 0u,
                                       counts_dev, n_ranges_dev, ranges_dev, max_ranges, stream ) );
}
inline void locate(const fm_index& fmi, const uint32_t* rows_dev, uint32_t n, uint32_t* pos_dev, hipStream_t stream = 0)
{ check( nvbio_fm_locate( fmi.handle(), rows_dev, n, pos_dev, stream ) ); }
inline void locate_ssa_iterator(const fm_index& fmi, const uint32_t* rows_dev, uint32_t n, nvbio_uint2* jt_dev, hipStream_t stream = 0)
{ check( nvbio_fm_locate_init( fmi.handle(), rows_dev, n, jt_dev, stream ) ); }
inline void lookup_ssa_iterator(const fm_index& fmi, const nvbio_uint2* jt_dev, uint32_t n, uint32_t* pos_dev, hipStream_t stream = 0)
{ check( nvbio_fm_locate_lookup( fmi.handle(), jt_dev, n, pos_dev, stream ) ); }

// the whole seed pass of one strand in one call (nvbio_fm_match_seed_diagonals): owner of its output arrays and scratch.
// keys(): the diagonal keys of the seeds that end on one SA row, in seed order, duplicates within a read dropped;
// residual_ranges() / residual_ids(): the seeds on several rows, for nvbio_fm_filter_scan + nvbio_fm_filter_locate_diagonals;
// counts(): device pointer to { number of keys, number of residual seeds }
class SeedPass
{
public:
    explicit SeedPass(const string_set& seeds) : m_seeds( seeds )
    {
        const size_t n = seeds.size();
        m_keys.resize( n ); m_ranges.resize( n ); m_ids.resize( n ); m_counts.resize( 4 );
        uint64_t bytes = 0;
        check( nvbio_fm_match_seed_diagonals_temp_bytes( &m_seeds.c, &bytes ) );
        m_temp.resize( bytes );
    }
    void enact(const fm_index& fmi, uint32_t flags, uint32_t read_len, uint32_t strand, hipStream_t stream = 0)
    {
        check( nvbio_fm_match_seed_diagonals( fmi.handle(), &m_seeds.c, flags, read_len, strand, m_keys.data(), m_ranges.data(), m_ids.data(),
                                              m_counts.data(), m_temp.data(), m_temp.size(), stream ) );
    }
    const uint64_t*    keys()            const { return m_keys.data(); }
    const nvbio_uint2* residual_ranges() const { return m_ranges.data(); }
    const uint32_t*    residual_ids()    const { return m_ids.data(); }
    const uint32_t*    counts()          const { return m_counts.data(); }
private:
    string_set                 m_seeds;
    device_vector<uint64_t>    m_keys;
    device_vector<nvbio_uint2> m_ranges;
    device_vector<uint32_t>    m_ids, m_counts;
    device_vector<uint8_t>     m_temp;
};

// the seed pass of BOTH strands in one launch over the canonical table (nvbio_fm_match_seed_diagonals_both; handles built with
// NVBIO_FM_TABLE_CANONICAL): keys() of both strands tile by tile; the residual seeds of the forward strand in [0, counts()[1]) of
// residual_ranges() / residual_ids(), of the reverse strand in [capacity(), capacity() + counts()[2]).
// flags: e.g. NVBIO_FM_INLINE_HITS(4), which keeps the residual lists empty on unique-ish genomes
class SeedPassBoth
{
public:
    explicit SeedPassBoth(const string_set& seeds) : m_seeds( seeds )
    {
        const size_t n = seeds.size();
        uint64_t bytes = 0, cap = 0;
        check( nvbio_fm_match_seed_diagonals_both_keys_capacity( &m_seeds.c, &cap ) );
        m_keys.resize( cap ? cap : 1 ); m_ranges.resize( 2 * n ); m_ids.resize( 2 * n ); m_counts.resize( 6 );
        check( nvbio_fm_match_seed_diagonals_both_temp_bytes( &m_seeds.c, &bytes ) );
        m_temp.resize( bytes );
    }
    void enact(const fm_index& fmi, uint32_t flags, uint32_t read_len, hipStream_t stream = 0)
    {
        check( nvbio_fm_match_seed_diagonals_both( fmi.handle(), &m_seeds.c, flags, read_len, m_keys.data(), m_ranges.data(), m_ids.data(),
                                                   capacity(), m_counts.data(), m_temp.data(), m_temp.size(), stream ) );
    }
    uint32_t           capacity()        const { return (uint32_t)m_seeds.size(); }
    const uint64_t*    keys()            const { return m_keys.data(); }
    const nvbio_uint2* residual_ranges() const { return m_ranges.data(); }
    const uint32_t*    residual_ids()    const { return m_ids.data(); }
    const uint32_t*    counts()          const { return m_counts.data(); }
private:
    string_set                 m_seeds;
    device_vector<uint64_t>    m_keys;
    device_vector<nvbio_uint2> m_ranges;
    device_vector<uint32_t>    m_ids, m_counts;
    device_vector<uint8_t>     m_temp;
};

template <typename system_tag> class FMIndexFilter;

template <>
class FMIndexFilter<amd_device_tag>
{
public:
    typedef fm_index    fm_index_type;
    typedef nvbio_uint2 range_type;
    typedef nvbio_uint2 hit_type;       // (text_pos, query_id)

    FMIndexFilter() : m_index( nullptr ), m_n_queries( 0 ), m_n_occurrences( 0 ) {}

    // enact the filter; returns the total number of hits (filter_inl.h:261-293).
    // allow_direct: let searches that collapse to one SA row finish on the text (nvbio_fm_match_direct) when the index
    // holds the full SA and the text -- same hits from locate(), faster; ranges() then holds a text position, not an SA
    // row, for the queries flagged in direct(), so leave it off if the caller reads ranges() itself.
    uint64_t rank(const fm_index& index, const string_set& set, hipStream_t stream = 0, bool allow_direct = false)
    {
        m_index = &index; m_n_queries = set.size();
        m_ranges.resize( m_n_queries ); m_slots.resize( m_n_queries );
        int can = 0;
        if (allow_direct) check( nvbio_fm_index_supports_direct( index.handle(), &can ) );
        m_is_direct = can != 0;
        if (m_is_direct)
        {
            m_direct.resize( m_n_queries );
            check( nvbio_fm_match_direct( index.handle(), &set.c, 0u, m_ranges.data(), m_direct.data(), stream ) );
            check( nvbio_fm_filter_scan( index.handle(), m_ranges.data(), m_n_queries, m_slots.data(), &m_n_occurrences, stream ) );
        }
        else
            check( nvbio_fm_filter_rank( index.handle(), &set.c, 0u, m_ranges.data(), m_slots.data(), &m_n_occurrences, stream ) );
        return m_n_occurrences;
    }
    // enumerate the hits [begin,end) into caller-owned device memory (filter_inl.h:299-393)
    void locate(uint64_t begin, uint64_t end, hit_type* hits_dev, hipStream_t stream = 0)
    {
        if (m_is_direct)
            check( nvbio_fm_filter_locate_direct( m_index->handle(), m_ranges.data(), m_slots.data(), m_direct.data(), m_n_queries, begin, end, hits_dev, stream ) );
        else
            check( nvbio_fm_filter_locate( m_index->handle(), m_ranges.data(), m_slots.data(), m_n_queries, begin, end, hits_dev, stream ) );
    }
    const uint8_t* direct() const { return m_is_direct ? m_direct.data() : nullptr; }
    uint32_t n_queries() const { return m_n_queries; }
    uint64_t n_hits()    const { return m_n_occurrences; }
    const range_type* ranges() const { return m_ranges.data(); }
    const uint64_t*   slots()  const { return m_slots.data(); }
private:
    const fm_index*            m_index;
    uint32_t                   m_n_queries;
    uint64_t                   m_n_occurrences;
    device_vector<range_type>  m_ranges;
    device_vector<uint64_t>    m_slots;
    device_vector<uint8_t>     m_direct;
    bool                       m_is_direct = false;
};

namespace aln {

enum AlignmentType { GLOBAL = NVBIO_GLOBAL, LOCAL = NVBIO_LOCAL, SEMI_GLOBAL = NVBIO_SEMI_GLOBAL };

struct SimpleGotohScheme
{
    SimpleGotohScheme() {}
    SimpleGotohScheme(int32_t match, int32_t mm, int32_t gap_open, int32_t gap_ext)
        : m_match( match ), m_mismatch( mm ), m_gap_open( gap_open ), m_gap_ext( gap_ext ) {}
    nvbio_gotoh_scheme flat() const { return { m_match, -m_mismatch, -m_mismatch, m_gap_open, m_gap_ext, m_gap_open, m_gap_ext }; }
    int32_t m_match, m_mismatch, m_gap_open, m_gap_ext;
};
// nvBowtie's SmithWatermanScoringScheme<QualCost,ConstantCost> as data (scoring.h:206-330)
struct QualityGotohScheme
{
    int32_t match, mm_min, mm_max, read_gap_const, read_gap_coeff, ref_gap_const, ref_gap_coeff;
    nvbio_gotoh_scheme flat() const
    { return { match, mm_min, mm_max, -read_gap_const - read_gap_coeff, -read_gap_coeff, -ref_gap_const - ref_gap_coeff, -ref_gap_coeff }; }
};

template <AlignmentType T, typename scheme_type>
struct GotohAligner
{
    static const AlignmentType TYPE = T;
    GotohAligner() {}
    GotohAligner(const scheme_type& s) : scheme( s ) {}
    scheme_type scheme;
};
template <AlignmentType T, typename scheme_type>
GotohAligner<T,scheme_type> make_gotoh_aligner(const scheme_type& s) { return GotohAligner<T,scheme_type>( s ); }

// aln::SimpleSmithWatermanScheme / SmithWatermanAligner (nvbio/alignment/utils.h:81-98, alignment.h:508-545): linear gaps
struct SimpleSmithWatermanScheme
{
    SimpleSmithWatermanScheme() {}
    SimpleSmithWatermanScheme(int32_t match, int32_t mm, int32_t del, int32_t ins)
        : m_match( match ), m_mismatch( mm ), m_deletion( del ), m_insertion( ins ) {}
    nvbio_sw_scheme flat_sw() const { return { m_match, m_mismatch, m_deletion, m_insertion }; }
    // the Gotoh form (open = extension), defined when deletion == insertion: what the banded traceback takes
    nvbio_gotoh_scheme flat() const { return { m_match, -m_mismatch, -m_mismatch, m_deletion, m_deletion, m_deletion, m_deletion }; }
    int32_t m_match, m_mismatch, m_deletion, m_insertion;
};
template <AlignmentType T, typename scheme_type>
struct SmithWatermanAligner
{
    static const AlignmentType TYPE = T;
    SmithWatermanAligner() {}
    SmithWatermanAligner(const scheme_type& s) : scheme( s ) {}
    scheme_type scheme;
};
template <AlignmentType T, typename scheme_type>
SmithWatermanAligner<T,scheme_type> make_smith_waterman_aligner(const scheme_type& s) { return SmithWatermanAligner<T,scheme_type>( s ); }

// the reference's edit-distance aligner: the Smith-Waterman kernels with EditDistanceSWScheme (match 0, mismatch -1,
// insertion = deletion = -1; ed/ed_utils.h:36-43, ed/ed_banded_inl.h:37-69, ed/ed_inl.h:60-168) -- pinned against the
// reference in tests/golden/ed_golden.npz (banded) and sw_golden.npz (full matrix)
struct EditDistanceScheme : SimpleSmithWatermanScheme
{
    EditDistanceScheme() : SimpleSmithWatermanScheme( 0, -1, -1, -1 ) {}
};
template <AlignmentType T>
SmithWatermanAligner<T,EditDistanceScheme> make_edit_distance_aligner() { return SmithWatermanAligner<T,EditDistanceScheme>( EditDistanceScheme() ); }

namespace detail {
// scoring entry points by aligner family
template <AlignmentType T, typename S>
nvbio_status banded_score(const GotohAligner<T,S>& a, int device, uint32_t band, const nvbio_alignment_batch* b, int32_t* sc, nvbio_uint2* sk, hipStream_t s)
{ const nvbio_gotoh_scheme f = a.scheme.flat(); return nvbio_banded_gotoh_score( device, band, (nvbio_alignment_type)T, &f, b, sc, sk, s ); }
template <AlignmentType T, typename S>
nvbio_status banded_score(const SmithWatermanAligner<T,S>& a, int device, uint32_t band, const nvbio_alignment_batch* b, int32_t* sc, nvbio_uint2* sk, hipStream_t s)
{ const nvbio_sw_scheme f = a.scheme.flat_sw(); return nvbio_banded_sw_score( device, band, (nvbio_alignment_type)T, &f, b, sc, sk, s ); }
template <AlignmentType T, typename S>
nvbio_status full_score(const GotohAligner<T,S>& a, int device, int tb, const nvbio_alignment_batch* b, uint32_t mp, uint32_t mt, const int32_t* ms,
                        int32_t* sc, nvbio_uint2* sk, void* temp, uint64_t temp_size, hipStream_t s)
{ const nvbio_gotoh_scheme f = a.scheme.flat(); return nvbio_full_gotoh_score( device, (nvbio_alignment_type)T, tb, &f, b, mp, mt, ms, sc, sk, temp, temp_size, s ); }
template <AlignmentType T, typename S>
nvbio_status full_score(const SmithWatermanAligner<T,S>& a, int device, int tb, const nvbio_alignment_batch* b, uint32_t mp, uint32_t mt, const int32_t* ms,
                        int32_t* sc, nvbio_uint2* sk, void* temp, uint64_t temp_size, hipStream_t s)
{ const nvbio_sw_scheme f = a.scheme.flat_sw(); return nvbio_full_sw_score( device, (nvbio_alignment_type)T, tb, &f, b, mp, mt, ms, sc, sk, temp, temp_size, s ); }
template <AlignmentType T, typename S>
nvbio_status banded_traceback(const GotohAligner<T,S>& a, int device, uint32_t band, const nvbio_alignment_batch* b, int32_t* sc, nvbio_uint2* src, nvbio_uint2* sk,
                              uint16_t* cig, uint32_t stride, uint32_t* lens, uint32_t flags, void* temp, uint64_t temp_size, hipStream_t s)
{ const nvbio_gotoh_scheme f = a.scheme.flat(); return nvbio_banded_gotoh_traceback( device, band, (nvbio_alignment_type)T, &f, b, sc, src, sk, cig, stride, lens, flags, temp, temp_size, s ); }
template <AlignmentType T, typename S>
nvbio_status banded_traceback(const SmithWatermanAligner<T,S>& a, int device, uint32_t band, const nvbio_alignment_batch* b, int32_t* sc, nvbio_uint2* src, nvbio_uint2* sk,
                              uint16_t* cig, uint32_t stride, uint32_t* lens, uint32_t flags, void* temp, uint64_t temp_size, hipStream_t s)
{ const nvbio_sw_scheme f = a.scheme.flat_sw(); return nvbio_banded_sw_traceback( device, band, (nvbio_alignment_type)T, &f, b, sc, src, sk, cig, stride, lens, flags, temp, temp_size, s ); }
template <AlignmentType T, typename S>
nvbio_status full_traceback(const GotohAligner<T,S>& a, int device, const nvbio_alignment_batch* b, uint32_t mp, uint32_t mt, const int32_t* ms, int32_t* sc, nvbio_uint2* src,
                            nvbio_uint2* sk, uint16_t* cig, uint32_t stride, uint32_t* lens, uint32_t flags, void* temp, uint64_t temp_size, hipStream_t s)
{ const nvbio_gotoh_scheme f = a.scheme.flat(); return nvbio_full_gotoh_traceback( device, (nvbio_alignment_type)T, &f, b, mp, mt, ms, sc, src, sk, cig, stride, lens, flags, temp, temp_size, s ); }
template <AlignmentType T, typename S>
nvbio_status full_traceback(const SmithWatermanAligner<T,S>& a, int device, const nvbio_alignment_batch* b, uint32_t mp, uint32_t mt, const int32_t* ms, int32_t* sc, nvbio_uint2* src,
                            nvbio_uint2* sk, uint16_t* cig, uint32_t stride, uint32_t* lens, uint32_t flags, void* temp, uint64_t temp_size, hipStream_t s)
{ const nvbio_sw_scheme f = a.scheme.flat_sw(); return nvbio_full_sw_traceback( device, (nvbio_alignment_type)T, &f, b, mp, mt, ms, sc, src, sk, cig, stride, lens, flags, temp, temp_size, s ); }
} // namespace detail

struct AmdDeviceScheduler {};

// a stream of alignment jobs in flat form (see nvbio_alignment_batch)
template <typename aligner_t>
struct FlatAlignmentStream
{
    typedef aligner_t aligner_type;
    FlatAlignmentStream(const aligner_t& a, const nvbio_alignment_batch& b, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                        uint32_t max_pattern_len, uint32_t max_text_len)
        : m_aligner( a ), m_batch( b ), m_scores( scores_dev ), m_sinks( sinks_dev ), m_max_p( max_pattern_len ), m_max_t( max_text_len ) {}
    const aligner_t& aligner() const { return m_aligner; }
    uint32_t size() const { return m_batch.n; }
    uint32_t max_pattern_length() const { return m_max_p; }
    uint32_t max_text_length()    const { return m_max_t; }
    const nvbio_alignment_batch& batch() const { return m_batch; }
    int32_t*     scores() const { return m_scores; }
    nvbio_uint2* sinks()  const { return m_sinks; }
    aligner_t m_aligner; nvbio_alignment_batch m_batch; int32_t* m_scores; nvbio_uint2* m_sinks; uint32_t m_max_p, m_max_t;
};

template <uint32_t BAND_LEN, typename stream_type, typename scheduler = AmdDeviceScheduler>
struct BatchedBandedAlignmentScore
{
    typedef typename stream_type::aligner_type aligner_type;
    // no temporary storage, as the reference's thread schedulers (batched_banded_inl.h:100-104,138-142)
    static uint64_t min_temp_storage(uint32_t, uint32_t, uint32_t) { return 0u; }
    static uint64_t max_temp_storage(uint32_t, uint32_t, uint32_t) { return 0u; }
    void enact(stream_type stream, uint64_t temp_size = 0u, uint8_t* temp = nullptr, int device = 0, hipStream_t s = 0)
    {
        (void)temp_size; (void)temp;
        check( detail::banded_score( stream.aligner(), device, BAND_LEN, &stream.batch(), stream.scores(), stream.sinks(), s ) );
    }
};

// the reference's staged scheduler tag (nvbio/alignment/batched.h): 32-row windows with the min_score exit.  The stream must also
// offer min_scores() (int32 per job, device; nullptr = min_score() for all) -- the context->min_score of the reference's stream.
struct DeviceStagedThreadScheduler {};
template <typename aligner_t>
struct FlatStagedAlignmentStream : FlatAlignmentStream<aligner_t>
{
    FlatStagedAlignmentStream(const aligner_t& a, const nvbio_alignment_batch& b, const int32_t* min_scores_dev, int32_t min_score_all,
                              int32_t* scores_dev, nvbio_uint2* sinks_dev, uint32_t max_pattern_len, uint32_t max_text_len)
        : FlatAlignmentStream<aligner_t>( a, b, scores_dev, sinks_dev, max_pattern_len, max_text_len ), m_min_scores( min_scores_dev ), m_min_score( min_score_all ) {}
    const int32_t* min_scores() const { return m_min_scores; }
    int32_t        min_score()  const { return m_min_score; }
    const int32_t* m_min_scores; int32_t m_min_score;
};
template <uint32_t BAND_LEN, typename stream_type>
struct BatchedBandedAlignmentScore<BAND_LEN,stream_type,DeviceStagedThreadScheduler>
{
    typedef typename stream_type::aligner_type aligner_type;
    // the reference asks for one band of short2 checkpoints per queue slot (batched_banded_inl.h:176-198); the band stays in registers here
    static uint64_t min_temp_storage(uint32_t, uint32_t, uint32_t) { return 0u; }
    static uint64_t max_temp_storage(uint32_t, uint32_t, uint32_t) { return 0u; }
    void enact(stream_type stream, uint64_t temp_size = 0u, uint8_t* temp = nullptr, int device = 0, hipStream_t s = 0)
    {
        (void)temp_size; (void)temp;
        const nvbio_gotoh_scheme f = stream.aligner().scheme.flat();
        check( nvbio_banded_gotoh_score_staged( device, BAND_LEN, (nvbio_alignment_type)aligner_type::TYPE, &f, &stream.batch(),
                                                stream.min_scores(), stream.min_score(), stream.scores(), stream.sinks(), s ) );
    }
};

template <typename stream_type, typename scheduler = AmdDeviceScheduler>
struct BatchedAlignmentScore
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64_t min_temp_storage(uint32_t max_pattern_len, uint32_t max_text_len, uint32_t stream_size)
    {
        uint64_t bytes = 0; nvbio_alignment_batch b = {}; b.n = stream_size;
        check( nvbio_full_gotoh_temp_bytes( &b, max_pattern_len, max_text_len, 1, &bytes ) );
        return bytes;
    }
    static uint64_t max_temp_storage(uint32_t p, uint32_t t, uint32_t n) { return min_temp_storage( p, t, n ); }
    void enact(stream_type stream, uint64_t temp_size = 0u, uint8_t* temp = nullptr, int device = 0, hipStream_t s = 0,
               bool text_blocking = true, const int32_t* min_scores_dev = nullptr)
    {
        check( detail::full_score( stream.aligner(), device, text_blocking ? 1 : 0, &stream.batch(),
                                   stream.max_pattern_length(), stream.max_text_length(), min_scores_dev,
                                   stream.scores(), stream.sinks(), temp, temp_size, s ) );
    }
};

// a traceback stream in flat form: the jobs of a FlatAlignmentStream plus where each job's Alignment
// {score, source, sink} and io::Cigar elements go (the role of nvBowtie's BestTracebackStream::output, which
// copies context->backtracer into pipeline.cigar, traceback_inl.h:115-160)
template <typename aligner_t>
struct FlatTracebackStream : FlatAlignmentStream<aligner_t>
{
    FlatTracebackStream(const aligner_t& a, const nvbio_alignment_batch& b, int32_t* scores_dev, nvbio_uint2* sources_dev,
                        nvbio_uint2* sinks_dev, uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                        bool sinks_given = false)
        : FlatAlignmentStream<aligner_t>( a, b, scores_dev, sinks_dev, b.max_read_len, 0 ), m_sources( sources_dev ),
          m_cigars( cigars_dev ), m_stride( cigar_stride ), m_lens( cigar_lens_dev ), m_given( sinks_given ) {}
    nvbio_uint2* sources() const { return m_sources; }
    uint16_t*    cigars()  const { return m_cigars; }
    uint32_t     cigar_stride() const { return m_stride; }
    uint32_t*    cigar_lens()   const { return m_lens; }
    bool         sinks_given()  const { return m_given; }
    nvbio_uint2* m_sources; uint16_t* m_cigars; uint32_t m_stride; uint32_t* m_lens; bool m_given;
};

// aln::BatchedBandedAlignmentTraceback<BAND_LEN,CHECKPOINTS,stream,scheduler> (nvbio/alignment/batched.h:420-437).
// CHECKPOINTS is accepted for source compatibility and ignored: the direction vectors of the whole band are kept
// in scratch instead of being recomputed between checkpoints (same result, see include/nvbio_amd.h).
template <uint32_t BAND_LEN, uint32_t CHECKPOINTS, typename stream_type, typename scheduler = AmdDeviceScheduler>
struct BatchedBandedAlignmentTraceback
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64_t min_temp_storage(uint32_t max_pattern_len, uint32_t, uint32_t stream_size)
    {
        uint64_t bytes = 0; nvbio_alignment_batch b = {}; b.n = stream_size; b.max_read_len = max_pattern_len;
        check( nvbio_banded_gotoh_traceback_temp_bytes( &b, BAND_LEN, &bytes ) );
        return bytes;
    }
    static uint64_t max_temp_storage(uint32_t p, uint32_t t, uint32_t n) { return min_temp_storage( p, t, n ); }
    void enact(stream_type stream, uint64_t temp_size = 0u, uint8_t* temp = nullptr, int device = 0, hipStream_t s = 0)
    {
        check( detail::banded_traceback( stream.aligner(), device, BAND_LEN, &stream.batch(),
                                         stream.scores(), stream.sources(), stream.sinks(), stream.cigars(), stream.cigar_stride(),
                                         stream.cigar_lens(), stream.sinks_given() ? NVBIO_TRACEBACK_SINKS_GIVEN : 0u,
                                         temp, temp_size, s ) );
    }
};

// aln::BatchedAlignmentTraceback<CHECKPOINTS,stream,scheduler> (nvbio/alignment/batched.h:395-411): full-matrix traceback
// of a FlatTracebackStream (x = text, y = pattern); CHECKPOINTS accepted and ignored as for the banded class
template <uint32_t CHECKPOINTS, typename stream_type, typename scheduler = AmdDeviceScheduler>
struct BatchedAlignmentTraceback
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64_t min_temp_storage(uint32_t max_pattern_len, uint32_t max_text_len, uint32_t stream_size)
    {
        uint64_t bytes = 0; nvbio_alignment_batch b = {}; b.n = stream_size;
        check( nvbio_full_gotoh_traceback_temp_bytes( &b, max_pattern_len, max_text_len, &bytes ) );
        return bytes;
    }
    static uint64_t max_temp_storage(uint32_t p, uint32_t t, uint32_t n) { return min_temp_storage( p, t, n ); }
    void enact(stream_type stream, uint32_t max_pattern_len, uint32_t max_text_len, uint64_t temp_size = 0u, uint8_t* temp = nullptr,
               int device = 0, hipStream_t s = 0, const int32_t* min_scores_dev = nullptr)
    {
        check( detail::full_traceback( stream.aligner(), device, &stream.batch(), max_pattern_len, max_text_len,
                                       min_scores_dev, stream.scores(), stream.sources(), stream.sinks(), stream.cigars(),
                                       stream.cigar_stride(), stream.cigar_lens(), stream.sinks_given() ? NVBIO_TRACEBACK_SINKS_GIVEN : 0u,
                                       temp, temp_size, s ) );
    }
};

// convenience function (nvbio/alignment/batched.h:185): banded scores of a flat batch
template <uint32_t BAND_LEN, typename aligner_type>
void batch_banded_alignment_score(const aligner_type& aligner, const nvbio_alignment_batch& batch,
                                  int32_t* scores_dev, nvbio_uint2* sinks_dev, int device = 0, hipStream_t s = 0)
{
    typedef FlatAlignmentStream<aligner_type> stream_type;
    BatchedBandedAlignmentScore<BAND_LEN,stream_type> batched;
    batched.enact( stream_type( aligner, batch, scores_dev, sinks_dev, 0, 0 ), 0u, nullptr, device, s );
}

// aln::Best2Sink<int32>( distinct_dist ) (nvbio/alignment/sink.h:96-116) as a batch: best and second-best distinct alignment
// of every job (Gotoh aligners; the int32 kernels report cell by cell in the reference's order)
template <uint32_t BAND_LEN, AlignmentType T, typename scheme_type>
void batch_banded_alignment_score_best2(const GotohAligner<T,scheme_type>& aligner, const nvbio_alignment_batch& batch, uint32_t distinct_dist,
                                        int32_t* scores_dev, nvbio_uint2* sinks_dev, int32_t* scores2_dev, nvbio_uint2* sinks2_dev,
                                        int device = 0, hipStream_t s = 0)
{
    const nvbio_gotoh_scheme f = aligner.scheme.flat();
    check( nvbio_banded_gotoh_score_best2( device, BAND_LEN, (nvbio_alignment_type)T, &f, &batch, distinct_dist,
                                           scores_dev, sinks_dev, scores2_dev, sinks2_dev, s ) );
}
template <AlignmentType T, typename scheme_type>
void batch_alignment_score_best2(const GotohAligner<T,scheme_type>& aligner, const nvbio_alignment_batch& batch, uint32_t max_pattern_len,
                                 uint32_t max_text_len, uint32_t distinct_dist, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                 int32_t* scores2_dev, nvbio_uint2* sinks2_dev, bool text_blocking = false,
                                 const int32_t* min_scores_dev = nullptr, int device = 0, hipStream_t s = 0)
{
    const nvbio_gotoh_scheme f = aligner.scheme.flat();
    check( nvbio_full_gotoh_score_best2( device, (nvbio_alignment_type)T, text_blocking ? 1 : 0, &f, &batch, max_pattern_len, max_text_len,
                                         min_scores_dev, distinct_dist, scores_dev, sinks_dev, scores2_dev, sinks2_dev, s ) );
}

} // namespace aln
} // namespace nvbio_amd
