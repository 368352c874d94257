/* nvbio_amd.h -- C ABI of the MI355X-native seed-and-extend core.
 *
 * This is the drop-in boundary for the hot path of NVBIO / nvBowtie: FM-index
 * rank / match / locate over a 2-bit packed BWT and batched (banded and full-matrix)
 * Gotoh scoring.  The reference has no C ABI; its operator API for this path is three
 * template concepts.  Every entry point below names the reference
 * interface it stands in for (file:line relative to the reference tree); the C++
 * shim that plugs these calls back under the reference's own class names is in
 * nvbio-gpl_amd/host/, and INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - plain C types only; every `*_dev` / "device pointer" argument is a pointer into the
 *     HBM of the GPU the handle/stream lives on; the caller owns those buffers.
 *   - all entry points return an nvbio_status (0 = ok) and never throw; a message for the
 *     last failure on the calling thread is available from nvbio_amd_last_error().
 *   - work is enqueued on the caller's stream (a hipStream_t passed as void*; NULL = the
 *     default stream) and is asynchronous; nvbio_amd_stream_synchronize() (or the caller's
 *     own hipStreamSynchronize) waits for it.  Calls on one handle from several host
 *     threads are safe as long as each thread uses its own stream and output buffers.
 *   - scratch a call needs beyond the caller's buffers (job lists, boundary columns, scan temporaries) comes from device blocks the
 *     library allocates with hipMalloc and KEEPS, per (device, stream), for the next call on that stream: after the first calls at a
 *     given size no call allocates.  nvbio_amd_release_scratch() gives the idle blocks back (it synchronises their streams).
 *   - results are bit-identical to the reference's CPU path: SA ranges, SA rows and text
 *     positions as uint32, scores as int32, sinks as (text, pattern) uint32 pairs.
 */
#ifndef NVBIO_AMD_H
#define NVBIO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVBIO_AMD_VERSION 100   /* 0.1.0 */

typedef enum
{
    NVBIO_OK              = 0,
    NVBIO_ERR_INVALID     = 1,  /* bad argument (null pointer, unsupported band/bits, ...)   */
    NVBIO_ERR_HIP         = 2,  /* a HIP runtime call failed                                   */
    NVBIO_ERR_NOMEM       = 3,  /* device or host allocation failed                            */
    NVBIO_ERR_UNSUPPORTED = 4,  /* valid in the reference but not built here                   */
    NVBIO_ERR_NO_DEVICE   = 5   /* no usable gfx950 device: the library has NO CPU fallback    */
} nvbio_status;

int         nvbio_amd_version(void);
const char* nvbio_amd_last_error(void);
/* number of visible HIP devices and the gcnArchName of one of them (e.g. "gfx950:sramecc+:xnack-") */
nvbio_status nvbio_amd_device_count(int* count);
nvbio_status nvbio_amd_device_arch(int device, char* name, uint32_t name_len);
nvbio_status nvbio_amd_stream_synchronize(int device, void* stream);
/* hipFree every scratch block that no call is using (see "Conventions"); each behind a synchronisation of the stream it served */
nvbio_status nvbio_amd_release_scratch(void);

/* -------------------------------------------------------------------------------------------
 * pair of uint32, layout-compatible with the reference's uint2 (SA ranges, hits, sinks)
 * ------------------------------------------------------------------------------------------- */
typedef struct { uint32_t x, y; } nvbio_uint2;

/* -------------------------------------------------------------------------------------------
 * FM-index
 * ------------------------------------------------------------------------------------------- */

/* Storage-free view of an FM-index resident in HBM: the members of nvbio::fm_index
 * (nvbio/fmindex/fmindex.h:320-361) in the production layout of io::FMIndexDataDevice
 * (nvbio/io/fmindex/fmindex.h:75-177, :294-295):
 *   bwt_occ : 32-byte records; record k = 4 words of 2-bit big-endian BWT symbols [64k,64k+64)
 *             followed by 4 words occ{A,C,G,T} = counts in BWT[0,64k)   (fmindex_impl.cu:300-313)
 *   ssa     : ssa[j] = SA[sa_int j], ssa[0] = 0xFFFFFFFF                (ssa_inl.h:254-301,477-495)
 *             sa_int = 16 is the reference's production layout (SA_INT, io/fmindex/fmindex.h:86);
 *             indices built here may sample more densely (a power of two down to 1 = the full
 *             suffix array, 12 GB for 3 Gbp): positions returned by locate() do not depend on it
 *   L2      : L2[c] = number of symbols < c, L2[4] = length             (fmindex.h:335-336)       */
typedef struct
{
    uint32_t        length;          /* n: text symbols (the BWT holds n symbols, '$' is implicit)  */
    uint32_t        primary;         /* BWT-matrix row of '$'                                       */
    uint32_t        L2[5];
    const uint32_t* bwt_occ_dev;     /* device pointer, 32-byte aligned                             */
    uint64_t        bwt_occ_words;   /* = 2 * ceil4(ceil(n/16))                                     */
    const uint32_t* ssa_dev;         /* device pointer, may be NULL (match-only index)              */
    uint64_t        ssa_words;       /* = n/sa_int + 1                                              */
    uint32_t        sa_int;          /* SA sampling interval: power of two in [1,64]; 0 means 16    */
} nvbio_fm_index_view;

typedef struct nvbio_fm_index_s* nvbio_fm_index_t;     /* opaque handle */

/* Wrap an index that is already in HBM (the caller keeps ownership of bwt_occ/ssa, as with the
 * reference's storage-free views).  kmer_len > 0 additionally builds, on the GPU, a table with
 * the SA range of every kmer_len-mer (4^kmer_len x 8 bytes, owned by the handle) which match()
 * uses to replace its first kmer_len backward-search steps with one lookup; results are
 * identical with and without it.  kmer_len = 0 disables it; values up to 17 are accepted
 * (k = 12: 128 MiB, k = 14: 2 GiB, k = 16: 32 GiB, k = 17: 128 GiB -- sized for 288 GB of HBM).
 * A handle that also holds the full suffix array and the text (nvbio_fm_index_build with sa_int = 1) keeps the last TWO
 * levels of the table: level kmer_len - 1 as the table of match(), and level kmer_len as the table of the direct seed pass
 * (nvbio_fm_match_direct, nvbio_fm_match_seed_diagonals): the entry of a k-mer with ONE occurrence rewritten to that
 * occurrence's text position and the 15 text symbols to its left, that of a k-mer with 2..7 occurrences pointing to a
 * 32/64-byte group with the same for each of them -- a 22-mer seed of a 3 Gbp genome is then matched AND located by the
 * table gather alone (texts up to 3.22 G symbols; longer ones keep positions without the context).
 * Replaces: constructing nvbio::fm_index / io::FMIndexDataDevice (nvbio/io/fmindex/fmindex_impl.cu:740-816). */
nvbio_status nvbio_fm_index_create(const nvbio_fm_index_view* view, int device, uint32_t kmer_len,
                                   void* stream, nvbio_fm_index_t* out);

/* Build the whole index on the GPU from a 2-bit big-endian packed text in HBM (layout of
 * io::SequenceData<DNA>, nvbio/io/sequence/sequence_traits.h:32-38): suffix sort, BWT, occ,
 * interleave, SSA.  All arrays are owned by the handle.  Replaces, for synthetic / benchmark
 * references, the offline nvBWT + FMIndexDataHost::load path (nvBWT/nvBWT.cu,
 * nvbio/io/fmindex/fmindex_impl.cu:111-331) and SSA_index_multiple's builder (ssa_inl.h:273-470).
 * Texts with repeats longer than options->max_lcp symbols are rejected with NVBIO_ERR_UNSUPPORTED
 * rather than sorted slowly. */
typedef struct
{
    uint32_t kmer_len;   /* k of the k-mer SA-range table, 0..17 (0 = none)                                  */
    uint32_t sa_int;     /* SA sampling interval, power of two in [1,64]; 0 = 16 (the reference's SA_INT)     */
    uint32_t max_lcp;    /* give up on texts with repeats longer than this many symbols; 0 = 4096            */
    uint32_t verify;     /* 1: also keep the inverse suffix array (4(n+1) bytes) and a copy of the text, so that
                            match() can finish a search whose range has collapsed to ONE row by comparing the
                            rest of the pattern with the text at SA[row] and jumping to ISA[position] (3 gathers
                            instead of one per remaining symbol).  Requires sa_int = 1.  Results are identical. */
    uint32_t table_flags;     /* NVBIO_FM_TABLE_*: which form of the k-mer tables a direct-capable handle keeps (A/B measurements and
                                 tests; results are identical with every combination)                                              */
    uint32_t bucket_symbols;  /* suffix sort: 0 = choose the number of prefix symbols used for bucketing from the text length;
                                 1 + v forces v in 0..4 (tests run the bucketed path on small texts this way)                      */
} nvbio_fm_build_options;

enum
{
    NVBIO_FM_TABLE_NO_DIRECT  = 1,  /* keep the plain SA-range table only, even if the handle holds the full SA and the text      */
    NVBIO_FM_TABLE_NO_CONTEXT = 2,  /* direct table, format 1: one-row entries hold the position only (the rest of a seed is then
                                       verified with a gather from the text)                                                      */
    NVBIO_FM_TABLE_NO_GROUPS  = 4,  /* no groups for k-mers with 2..7 occurrences (they keep their SA range and take rank steps)  */
    NVBIO_FM_TABLE_CANONICAL_WIDE = 16, /* NVBIO_FM_TABLE_CANONICAL with 16-byte entries: a k-mer with TWO occurrences (of either orientation) has
                                       both rows in its entry, so that only k-mers with three or more take the second gather (1.07 instead
                                       of 1.25 sectors per seed window on a 3 Gbp text; 128 GiB at k = 17)                         */
    NVBIO_FM_TABLE_CANONICAL  = 8   /* instead of the direct table: ONE table for a k-mer and its reverse complement (kmer_len odd; 64 GiB at
                                       k = 17 where the direct table takes 128), serving nvbio_fm_match_seed_diagonals_both; the plain table of
                                       (kmer_len - 1)-mers is kept for match().  Needs sa_int = 1.                                */
};

nvbio_status nvbio_fm_index_build(const uint32_t* text2_dev, uint32_t length, int device,
                                  const nvbio_fm_build_options* options /* NULL = defaults */,
                                  void* stream, nvbio_fm_index_t* out);

/* Load an index from the reference's on-disk files (written by nvBWT, nvBWT/nvBWT.cu:303-342; read by
 * io::FMIndexDataHost::load, nvbio/io/fmindex/fmindex_impl.cu:111-252,333-...):
 *   <bwt_path>  uint32 primary; uint32 cumulative symbol counts[4] (last = n); packed 2-bit BWT words
 *   <sa_path>   uint32 primary; uint32 counts[4]; uint32 SA_INT; uint32 n; ssa[1..]   (optional, may be NULL)
 * The occurrence table is rebuilt and interleaved on the GPU (the reference does it on the host,
 * fmindex_impl.cu:254-331).  All arrays are owned by the handle. */
nvbio_status nvbio_fm_index_load(const char* bwt_path, const char* sa_path, int device, uint32_t kmer_len,
                                 void* stream, nvbio_fm_index_t* out);
/* Write the index in those formats (the role of nvBWT for indices built by nvbio_fm_index_build);
 * sa_path may be NULL.  The reference's reader accepts .sa files with SA_INT = 16 only. */
nvbio_status nvbio_fm_index_save(nvbio_fm_index_t index, const char* bwt_path, const char* sa_path, void* stream);

nvbio_status nvbio_fm_index_destroy(nvbio_fm_index_t index);
nvbio_status nvbio_fm_index_get_view(nvbio_fm_index_t index, nvbio_fm_index_view* view);
/* copy the index arrays into caller buffers in HBM (either may be NULL); sizes from get_view */
nvbio_status nvbio_fm_index_export(nvbio_fm_index_t index, uint32_t* bwt_occ_out_dev, uint32_t* ssa_out_dev, void* stream);
/* bytes of HBM owned by the handle (k-mer table, and the index arrays if it was built here) */
nvbio_status nvbio_fm_index_device_bytes(nvbio_fm_index_t index, uint64_t* bytes);

/* A set of query strings in HBM: the string-set concept FMIndexFilter::rank consumes
 * (nvbio/fmindex/filter.h:52-231) flattened to arrays.
 *   symbols      packed big-endian words (symbol_bits 2 or 4: PackedStream<..,true>,
 *                nvbio/basic/packedstream_inl.h:33-75; 4-bit = io::SequenceData<DNA_N>) or one
 *                symbol per byte (symbol_bits 8).  Symbols > 3 are 'N'.
 *   offsets      if offsets_are_ranges: n+1 entries, string i = [offsets[i], offsets[i+1])
 *                else if non-NULL: n entries, string i = [offsets[i], offsets[i]+fixed_len)
 *                else: string i = [i*stride, i*stride + fixed_len)                             */
typedef struct
{
    const void*     symbols_dev;
    uint32_t        symbol_bits;
    const uint32_t* offsets_dev;
    uint32_t        offsets_are_ranges;
    uint32_t        fixed_len;
    uint32_t        stride;
    uint32_t        n;
    /* seed enumeration (uniform_seeds_functor, nvbio/strings/seeds.h; nvBowtie mapping_inl.h:485-556):
     * if seeds_per_string > 0 the set is the seeds of a set of n / seeds_per_string strings: query i is
     * seed j = i % seeds_per_string of string r = i / seeds_per_string,
     *   [ base(r) + j*seed_interval, + fixed_len ),  base(r) = offsets_dev ? offsets_dev[r] : r*stride   */
    uint32_t        seeds_per_string;
    uint32_t        seed_interval;
    /* ragged seed sets (reads of different lengths, each seeded at its own interval: nvBowtie computes both per read,
     * mapping_inl.h:507-529, `read_len = range.y - range.x`, `seed_freq( read_len )`): seed_intervals_dev[r] = the seed interval of
     * string r (n / seeds_per_string entries), or NULL.  With it offsets_dev holds n_strings + 1 entries (string r =
     * [offsets[r], offsets[r+1]) ), seeds_per_string is the LARGEST number of seeds of a string -- the stride of seed ids, seed id =
     * r * seeds_per_string + j -- and seed j of string r exists iff j * interval_r + fixed_len <= its length; a seed id without a seed
     * matches nothing (the empty range (1, 0)).  seed_interval is ignored. */
    const uint32_t* seed_intervals_dev;
} nvbio_string_set;

enum
{
    NVBIO_FM_SCAN_FORWARD = 1,   /* consume symbols 0..len-1 (match_reverse, fmindex_inl.h:247-278; nvBowtie
                                    match_range, mapping_inl.h:73-86); default is len-1..0 (match, :181-239) */
    NVBIO_FM_COMPLEMENT   = 2,   /* search the complement (c < 4 ? 3-c : c), as nvBowtie's rc seeds
                                    (mapping_inl.h:264-279)                                                   */
    NVBIO_FM_NO_KMER_TABLE = 4,  /* step every symbol through rank() even if the handle has a table         */
    NVBIO_FM_NO_VERIFY     = 8,  /* never take the SA/ISA verification shortcut even if the handle has it   */
    NVBIO_FM_COUNT_SECTORS = 16, /* nvbio_fm_match_seed_diagonals only: an accounting launch (see there)        */
    NVBIO_FM_NO_PIPELINE   = 32  /* nvbio_fm_match_seed_diagonals only: the plain kernel, one tile at a time,
                                    also where the software-pipelined one applies (A/B; same results)         */
};

/* ranges_dev[i] = SA range (inclusive; empty iff x > y) of query i: nvbio::match / match_reverse
 * (nvbio/fmindex/fmindex_inl.h:181-278), including the N rule (-> (1,0)) and the reference's
 * early-exit values for patterns that stop matching.
 * blocks_dev (optional): number of distinct 32-byte bwt_occ records the reference's algorithm
 * touches for query i (the algorithmic-traffic unit of SURVEY.md 8d); requires NO_KMER_TABLE. */
nvbio_status nvbio_fm_match(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t flags,
                            nvbio_uint2* ranges_dev, uint32_t* blocks_dev, void* stream);

/* rank(fmi, k, c) for n (row, symbol) pairs: nvbio/fmindex/fmindex_inl.h:27-47 */
nvbio_status nvbio_fm_rank(nvbio_fm_index_t index, const uint32_t* rows_dev, const uint8_t* syms_dev, uint32_t n,
                           uint32_t* out_dev, void* stream);
/* rank4(fmi, k): counts of all four symbols, fmindex_inl.h:96-123 (out_dev: 4 words per row) */
nvbio_status nvbio_fm_rank4(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n,
                            uint32_t* out_dev, void* stream);

/* The GENERIC rank dictionary of the reference (nvbio/fmindex/rank_dictionary.h; dispatch_rank over plain words,
 * rank_dictionary_inl.h:206-336): 2-bit big-endian text in 32- or 64-bit words (PackedStream<const uint32*|const uint64*,uint8,2,true>),
 * a separate occurrence table occ[4 k + c] = # c in text[0, k K) (build_occurrence_table<K>, :33-66), indices and counts of 32 or 64 bits
 * -- the layouts of the reference's own rank test (nvbio-test/rank_test.cu:83-227: uint32 / K = 64; uint64 / K = 128 with 64-bit indices),
 * and the one to use beyond 2^32 symbols.  (The production layout, 32-byte records of BWT + occ, is served by nvbio_fm_rank.)
 *   nvbio_rank_dictionary_occ_entries : entries (of index_bits each) the occurrence table needs = 4 ceil(length / K)
 *   nvbio_rank_dictionary_build       : fills occ_out_dev; counts[c] (host) = total occurrences of c; synchronizes
 *   nvbio_rank_dictionary_rank        : out[q] = rank( dict, idx[q], sym[q] ) = occurrences of sym[q] in text[0, idx[q]] (inclusive);
 *                                       idx = all ones (-1) gives 0 (:278-279); idx / out are index_bits wide
 *   nvbio_rank_dictionary_rank4       : out[4 q + c] = rank( dict, idx[q], c ) for the four symbols (rank4, :294-309)            */
typedef struct
{
    const void* text_dev;
    uint32_t    word_bits;     /* 32 or 64 */
    const void* occ_dev;       /* NULL for nvbio_rank_dictionary_build */
    uint32_t    index_bits;    /* 32 or 64 */
    uint32_t    K;             /* symbols per block: a power of two, a multiple of the symbols per word */
    uint64_t    length;        /* symbols */
} nvbio_rank_dictionary;
nvbio_status nvbio_rank_dictionary_occ_entries(const nvbio_rank_dictionary* dict, uint64_t* entries);
nvbio_status nvbio_rank_dictionary_build(int device, const nvbio_rank_dictionary* dict, void* occ_out_dev, uint64_t counts[4], void* stream);
nvbio_status nvbio_rank_dictionary_rank(int device, const nvbio_rank_dictionary* dict, const void* idx_dev, const uint8_t* syms_dev, uint32_t n,
                                        void* out_dev, void* stream);
nvbio_status nvbio_rank_dictionary_rank4(int device, const nvbio_rank_dictionary* dict, const void* idx_dev, uint32_t n, void* out_dev, void* stream);

/* out_dev[i] = basic_inv_psi(fmi, rows_dev[i]): one LF step, the row of the suffix one symbol to the
 * left (nvbio/fmindex/fmindex_inl.h:286-309); rows_dev == out_dev is allowed */
nvbio_status nvbio_fm_basic_inv_psi(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n,
                                    uint32_t* out_dev, void* stream);

/* pos_dev[i] = text position of SA row rows_dev[i]: nvbio::locate (fmindex_inl.h:360-394).
 * rows_dev == pos_dev is allowed (nvBowtie locates in place, locate_inl.h:113-138). */
nvbio_status nvbio_fm_locate(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n,
                             uint32_t* pos_dev, void* stream);
/* match() + locate() fused for the searches that end in a single text occurrence (the common case of a seed
 * pass: a 22-mer of a 3 Gbp genome).  As nvbio_fm_match, except that a search whose SA range has collapsed to
 * ONE row before the pattern is exhausted is finished on the text to the left of SA[row] instead of by one
 * rank step per remaining symbol: 2 dependent gathers instead of (remaining symbols + the later SA lookup).
 * For such a query  direct_dev[i] = 1  and  ranges_dev[i] = (pos, pos)  where pos is the TEXT POSITION of the
 * occurrence -- exactly locate(fmi, row) of the one row the reference's match() would end in -- or, if the rest
 * of the pattern does not match, direct_dev[i] = 0 and the empty range (1,0).  Every other query gets
 * direct_dev[i] = 0 and its SA range as from nvbio_fm_match.  Range sizes (and so nvbio_fm_filter_scan) are
 * unaffected; nvbio_fm_filter_locate_direct expands the mix into the same hits, in the same order, as
 * nvbio_fm_filter_locate does on the plain ranges.  Needs the handle to hold the full suffix array and the text
 * (nvbio_fm_index_build with sa_int = 1), else NVBIO_ERR_UNSUPPORTED. */
nvbio_status nvbio_fm_match_direct(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t flags,
                                   nvbio_uint2* ranges_dev, uint8_t* direct_dev, void* stream);
/* *yes = 1 iff the handle can serve nvbio_fm_match_direct (it holds the full suffix array and the text) */
nvbio_status nvbio_fm_index_supports_direct(nvbio_fm_index_t index, int* yes);
nvbio_status nvbio_fm_filter_locate_direct(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                           const uint8_t* direct_dev, uint32_t n_queries, uint64_t begin, uint64_t end,
                                           nvbio_uint2* hits_dev, void* stream);

/* nvbio_fm_filter_locate[_direct] and nvbio_hits_to_diagonals (below) in one pass: the expansion writes each hit's
 * diagonal key (read << 34 | strand << 33 | diagonal + 1024) instead of the (position, query) pair; direct_dev may
 * be NULL (plain ranges).  keys_dev[h - begin] equals nvbio_hits_to_diagonals of the hit nvbio_fm_filter_locate writes.
 * query_ids_dev (optional): the seed id of query i when the ranges are a compacted subset of a seed set (the residual
 * list of nvbio_fm_match_seed_diagonals); NULL = query i is seed i.  Bit 31 of a query id flips `strand` for that query (seed ids are
 * below 2^31), so that the two residual lists of nvbio_fm_match_seed_diagonals_both can go through one call. */
nvbio_status nvbio_fm_filter_locate_diagonals(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                              const uint8_t* direct_dev, uint32_t n_queries, uint64_t begin, uint64_t end,
                                              uint32_t seeds_per_read, uint32_t seed_interval, uint32_t seed_len, uint32_t read_len,
                                              uint32_t strand, const uint32_t* query_ids_dev, uint64_t* keys_dev, void* stream);

/* Approximate matching by backtracking under the Hamming distance: nvbio::hamming_backtrack (nvbio/fmindex/backtrack.h:51-157),
 * the kernel of the reference's "approximate search" benchmark (nvbio-test/fmindex_test.cu:739-800; the building block of
 * nvBowtie's approximate seed mapper, mapping_inl.h:114-184).  The last seed_len symbols of every query are matched exactly, the
 * rest may differ from the text in up to `mismatches` positions.
 *   counts_dev[i]   = total number of occurrences (the reference benchmark's CountDelegate: sum of the reported range sizes)
 *   n_ranges_dev[i] = number of SA ranges the delegate receives (optional)
 *   ranges_dev      = (optional) the first max_ranges of them per query, [i * max_ranges + k], in the reference's order
 * The traversal uses the reference benchmark's 128-entry stack; a query that would overflow it makes the call return
 * NVBIO_ERR_UNSUPPORTED (the reference writes past its array there).  Synchronises the stream.
 * flags: by default a branch stops when it reaches the start of the pattern, as the reference documents.  The reference's code
 * does not (backtrack.h:110-116 falls through at l == 0): a branch arriving there with all mismatches used reports its range
 * twice, and one arriving with some left continues through the symbols that PRECEDE the query in its stream, every branch it
 * spawns there ending in a report.  NVBIO_BACKTRACK_REFERENCE_QUIRKS reproduces exactly that (queries must then not start at
 * symbol 0 of the stream; pinned against the reference's own code over PackedStream patterns, tests/golden/bt_golden.npz). */
enum { NVBIO_BACKTRACK_REFERENCE_QUIRKS = 1 };
nvbio_status nvbio_fm_hamming_backtrack(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t seed_len, uint32_t mismatches, uint32_t flags,
                                        uint32_t* counts_dev, uint32_t* n_ranges_dev, nvbio_uint2* ranges_dev, uint32_t max_ranges, void* stream);

/* The whole seed pass of one strand, for handles that hold the full suffix array and the text: nvBowtie's
 * match_range over every seed (mapping_inl.h:73-86,193-282) followed, for every seed that ends on ONE SA row, by what
 * FMIndexFilter::rank's scan, FMIndexFilter::locate (filter_inl.h:193-252), hit_to_diagonal (examples/fmmap/fmmap.cu:92-117)
 * and the removal of adjacent duplicate diagonals would do with it: the hit's diagonal key goes straight to keys_dev
 * (as nvbio_fm_filter_locate_diagonals writes it), IN SEED ORDER; a key equal to the previous key of the same read is
 * dropped (consecutive seeds of a read that agree on the diagonal).  Seeds that end on several rows go to the residual list
 * (residual_ranges_dev[r], residual_ids_dev[r] = seed id, in arbitrary order) for nvbio_fm_filter_scan +
 * nvbio_fm_filter_locate_diagonals.  Capacities: seeds->n entries each.  counts_dev[0] = keys written, counts_dev[1] =
 * residual seeds (zeroed by the call).  The set of keys equals that of the plain operators over the same seeds.
 * seeds->n must be a whole number of strings (n = strings x seeds_per_string).
 * flags: NVBIO_FM_SCAN_FORWARD / NVBIO_FM_COMPLEMENT / NVBIO_FM_NO_KMER_TABLE as nvbio_fm_match; bits 16..31, if non-zero, cap
 * the launch at that many x 64 workgroups (a tuning / testing knob: results do not depend on it).
 * NVBIO_FM_COUNT_SECTORS: same results, and counts_dev (then 4 words, 8-byte aligned) also receives, as a uint64 in words
 * 2..3, the number of distinct 64-byte sectors the searches gathered from the index arrays (table entry, group, bwt_occ
 * records, SA word, text words): the unit in which the pass's memory traffic is accounted (a slower kernel instantiation,
 * for measurement harnesses).
 * temp_dev / temp_bytes: optional caller scratch (nvbio_fm_match_seed_diagonals_temp_bytes); if NULL the library allocates and
 * frees scratch itself (blocks kept per stream: see "Conventions"). */
nvbio_status nvbio_fm_match_seed_diagonals_temp_bytes(const nvbio_string_set* seeds, uint64_t* bytes);
nvbio_status nvbio_fm_match_seed_diagonals(nvbio_fm_index_t index, const nvbio_string_set* seeds, uint32_t flags, uint32_t read_len,
                                           uint32_t strand, uint64_t* keys_dev, nvbio_uint2* residual_ranges_dev, uint32_t* residual_ids_dev,
                                           uint32_t* counts_dev, void* temp_dev, uint64_t temp_bytes, void* stream);

/* Both strands of every seed in ONE pass (handles built with NVBIO_FM_TABLE_CANONICAL): the outputs of
 *   nvbio_fm_match_seed_diagonals( flags = 0, strand = 0 )  and
 *   nvbio_fm_match_seed_diagonals( flags = NVBIO_FM_SCAN_FORWARD | NVBIO_FM_COMPLEMENT, strand = 1 )
 * i.e. match() of every seed and of its reverse complement (nvBowtie maps both, mapping_inl.h:288-414) + locate() of the searches that end
 * on one row + hit_to_diagonal + the adjacent-duplicate removal, from one table gather per seed window: a k-mer and its reverse
 * complement share an entry that lists the occurrences of both orientations with the text on either side of each.
 *   keys_dev[0 .. counts_dev[0])        diagonal keys of both strands (read << 34 | strand << 33 | diagonal + 1024), grouped by tile of
 *                                       reads: a tile's forward keys in seed order, then its reverse-strand keys in seed order
 *   residual_*_dev[0 .. counts_dev[1])                              forward-strand searches that ended on several rows (range, seed id)
 *   residual_*_dev[residual_capacity .. + counts_dev[2])            the same for the reverse strand
 * keys_dev: nvbio_fm_match_seed_diagonals_both_keys_capacity() entries (256 per tile of 64 / seeds_per_string reads: with
 * NVBIO_FM_INLINE_HITS a seed can leave several keys, a tile at most 64 per strand -- slightly more than 2 * seeds->n when 64 is not a
 * multiple of seeds_per_string); residual arrays: 2 * residual_capacity entries, residual_capacity >= seeds->n.  counts_dev: 4 uint32
 * (6, 8-byte aligned, with NVBIO_FM_COUNT_SECTORS: the distinct 64-byte sectors gathered from the index as a uint64 at counts_dev + 4).
 * Seeds: packed 2 or 4 bits, fixed length in [kmer_len, kmer_len + 7], at most 64 per read.  flags: NVBIO_FM_COUNT_SECTORS, the grid
 * knob (bits 16..31) of nvbio_fm_match_seed_diagonals, and NVBIO_FM_INLINE_HITS(h), h in 2..4: a search that ends on up to h rows leaves
 * ALL their diagonal keys in keys_dev (behind its tile's one-row keys, no duplicate removal) instead of a residual entry -- what
 * FMIndexFilter's scan + locate would add for it, without the trip; only larger ranges reach the residual lists. */
#define NVBIO_FM_INLINE_HITS(h) (((uint32_t)(h) & 15u) << 8)
/* NVBIO_FM_DEFER_HEAVY: the searches the table cannot answer (a k-mer with more than 8 occurrences; more hits on a strand than
 * NVBIO_FM_INLINE_HITS) do not run inside the pass -- where every wave that holds one such window waits for its ten dependent gathers --
 * but are collected and run as a dense launch of their own behind it: same keys and residual entries (their keys follow the others in
 * keys_dev, without the adjacent-duplicate removal).  For repeat-rich references; on a unique-ish one it costs three short launches.
 * Ragged reads (seeds->seed_intervals_dev): read_len is ignored, every read's length comes from its offsets. */
#define NVBIO_FM_DEFER_HEAVY 64u
nvbio_status nvbio_fm_match_seed_diagonals_both_temp_bytes(const nvbio_string_set* seeds, uint64_t* bytes);
nvbio_status nvbio_fm_match_seed_diagonals_both_keys_capacity(const nvbio_string_set* seeds, uint64_t* n_keys);
nvbio_status nvbio_fm_match_seed_diagonals_both(nvbio_fm_index_t index, const nvbio_string_set* seeds, uint32_t flags, uint32_t read_len,
                                                uint64_t* keys_dev, nvbio_uint2* residual_ranges_dev, uint32_t* residual_ids_dev,
                                                uint32_t residual_capacity, uint32_t* counts_dev, void* temp_dev, uint64_t temp_bytes,
                                                void* stream);
/* the k of the canonical two-strand table the handle holds (built with NVBIO_FM_TABLE_CANONICAL: it serves seeds of k .. k + 7 symbols),
 * 0 if it holds none */
int nvbio_fm_index_is_canonical(nvbio_fm_index_t index);

/* the two-phase form nvBowtie uses (locate_init / locate_lookup kernels, locate_inl.h:144-201):
 * jt_dev[i] = locate_ssa_iterator(rows[i]) = (sampled row, steps)  (fmindex_inl.h:404-437)
 * pos_dev[i] = lookup_ssa_iterator(jt[i]) = ssa[j/sa_int] + t      (fmindex_inl.h:445-460)
 * (the intermediate pair equals the reference's only for sa_int = 16; positions always do)      */
nvbio_status nvbio_fm_locate_init(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n,
                                  nvbio_uint2* jt_dev, void* stream);
nvbio_status nvbio_fm_locate_lookup(nvbio_fm_index_t index, const nvbio_uint2* jt_dev, uint32_t n,
                                    uint32_t* pos_dev, void* stream);

/* FMIndexFilter<device_tag>::rank (nvbio/fmindex/filter_inl.h:261-293): ranges + inclusive scan of
 * the range sizes (uint64).  *n_hits receives the total (host value; this call synchronizes). */
nvbio_status nvbio_fm_filter_rank(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t flags,
                                  nvbio_uint2* ranges_dev, uint64_t* slots_dev, uint64_t* n_hits, void* stream);
/* the second half of FMIndexFilter::rank on its own (filter_inl.h:279-292): the inclusive scan of the sizes of
 * ranges that are already in HBM (e.g. from nvbio_fm_match) and the total; synchronizes like filter_rank. */
nvbio_status nvbio_fm_filter_scan(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, uint32_t n_queries,
                                  uint64_t* slots_dev, uint64_t* n_hits, void* stream);
/* FMIndexFilter<device_tag>::locate (filter_inl.h:299-393): hits_dev[h-begin] = (text_pos, query_id)
 * for the global hit indices h in [begin, end). */
nvbio_status nvbio_fm_filter_locate(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                    uint32_t n_queries, uint64_t begin, uint64_t end,
                                    nvbio_uint2* hits_dev, void* stream);

/* -------------------------------------------------------------------------------------------
 * seed hits -> candidate windows: the two index-arithmetic functors between FMIndexFilter::locate
 * and the banded aligner in the reference's smallest seed-and-extend caller (examples/fmmap/fmmap.cu)
 * ------------------------------------------------------------------------------------------- */

/* hit_to_diagonal (examples/fmmap/fmmap.cu:92-117) for uniformly enumerated seeds: for hit
 * (text_pos, seed_id), read = seed_id / seeds_per_read, seed offset p = (seed_id % seeds_per_read) *
 * seed_interval (on the reverse-complement strand p -> read_len - p - seed_len), diagonal = text_pos - p.
 * keys_dev[h] = read << 34 | strand << 33 | (diagonal + 1024): sortable, one per hit.               */
nvbio_status nvbio_hits_to_diagonals(int device, const nvbio_uint2* hits_dev, uint64_t n_hits, uint32_t seeds_per_read,
                                     uint32_t seed_interval, uint32_t seed_len, uint32_t read_len, uint32_t strand,
                                     uint64_t* keys_dev, void* stream);

/* Sort candidate keys and drop duplicates, in place: what fmmap does with its diagonals before extending them
 * (examples/fmmap/fmmap.cu:320-344: sort_by_key + unique), for host compositions that have no device sort of their own.
 * *n_out_dev (device) = number of distinct keys, left at the front of keys_dev in ascending order.
 * temp_dev / temp_bytes: optional caller scratch (nvbio_sort_unique_keys_temp_bytes). */
nvbio_status nvbio_sort_unique_keys_temp_bytes(uint64_t n, uint64_t* bytes);
nvbio_status nvbio_sort_unique_keys(int device, uint64_t* keys_dev, uint64_t n, uint32_t* n_out_dev, void* temp_dev, uint64_t temp_bytes, void* stream);

/* genome_infixes (examples/fmmap/fmmap.cu:169-196) with nvBowtie's window rule (BestScoreStream::init_context,
 * nvBowtie/bowtie2/cuda/score_inl.h:100-106): g_pos = max(diagonal,0); begin = g_pos > band/2 ? g_pos - band/2 : 0;
 * end = min(begin + band + read_len, genome_len); flags = strand ? REVERSE|COMPLEMENT : 0.                */
nvbio_status nvbio_diagonals_to_windows(int device, const uint64_t* keys_dev, uint64_t n, uint32_t band, uint32_t read_len,
                                        uint32_t genome_len, uint32_t* read_id_dev, uint8_t* flags_dev,
                                        uint32_t* win_begin_dev, uint32_t* win_end_dev, void* stream);

/* Ragged read batches (reads of different lengths: io::SequenceData's sequence_index; nvBowtie takes every read's length from its
 * range, mapping_inl.h:507-529, score_inl.h:100-106): the same operators with read_offsets_dev (n_reads + 1 symbol offsets) in place
 * of one read_len -- and, where a threshold depends on the read's length (MinScoreFunc, scoring.h:117-129), min_scores_dev[r] =
 * scheme.min_score( length of read r ) computed by the caller. */
nvbio_status nvbio_diagonals_to_windows_ragged(int device, const uint64_t* keys_dev, uint64_t n, uint32_t band, const uint32_t* read_offsets_dev,
                                               uint32_t genome_len, uint32_t* read_id_dev, uint8_t* flags_dev,
                                               uint32_t* win_begin_dev, uint32_t* win_end_dev, void* stream);
nvbio_status nvbio_fm_filter_locate_diagonals_ragged(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                                     const uint8_t* direct_dev, uint32_t n_queries, uint64_t begin, uint64_t end,
                                                     uint32_t seeds_per_read, uint32_t seed_len, const uint32_t* read_offsets_dev,
                                                     const uint32_t* seed_intervals_dev, uint32_t strand, const uint32_t* query_ids_dev,
                                                     uint64_t* keys_dev, void* stream);
nvbio_status nvbio_traceback_best_batch_ragged(int device, const uint64_t* best_dev, const int64_t* best_wb_dev, uint32_t n_reads,
                                               const uint32_t* read_offsets_dev, uint32_t band, uint32_t genome_len, const int32_t* min_scores_dev,
                                               uint8_t* flags_dev, uint32_t* win_begin_dev, uint32_t* win_end_dev, int32_t* scores_dev,
                                               nvbio_uint2* sinks_dev, void* stream);
/* distinct_dist = (length of the candidate's read) / 2, worst_score = min_scores_dev[read] - 1 */
nvbio_status nvbio_second_candidate_reduce_ragged(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                                  const uint32_t* win_begin_dev, uint64_t n, const uint64_t* best_dev,
                                                  const uint32_t* read_offsets_dev, const int32_t* min_scores_dev, uint64_t* second_dev, void* stream);
/* perfect_score = match x (length of the read), min_score = min_scores_dev[read], monotone = (match == 0) */
nvbio_status nvbio_mapq_ragged(int device, const uint64_t* best_dev, const uint64_t* second_dev, uint32_t n_reads, int32_t version, int32_t match,
                               const uint32_t* read_offsets_dev, const int32_t* min_scores_dev, int32_t* second_scores_dev, uint8_t* mapq_dev,
                               void* stream);

/* The residual seeds of a seed pass (nvbio_fm_match_seed_diagonals[_both]: searches that ended on several SA rows) under a seed-hit cap, in
 * ONE call: the first `cap` rows of every range (nvBowtie bounds repeats with its max_hits deque, mapping_inl.h:242-244; this is the order-free
 * stand-in the default pipeline uses) are located and turned into diagonal keys, as nvbio_fm_filter_scan + nvbio_fm_filter_locate_diagonals
 * over the capped ranges would -- without the scan, its host synchronisation and the global sort a duplicate removal would take: the
 * entries are sorted by seed id, so that consecutive seeds of a read are neighbours, and a key equal to the one the previous entry leaves
 * at the same row is dropped (the seeds of a read inside a repeat list the same loci row for row).  The SET of keys equals that of the
 * two-call form; a duplicate that survives costs a repeated extension, nothing else.
 *   ids_dev[e]        seed id of entry e, bit 31 = reverse strand (the convention of nvbio_fm_filter_locate_diagonals' query_ids_dev)
 *   keys_dev          room for n * cap keys; *n_keys_dev (device) = keys written
 *   read_offsets_dev / seed_intervals_dev   ragged reads (both or neither; then seed_interval and read_len are ignored)
 * Needs the full suffix array (sa_int = 1).  1 <= cap <= 64, n * cap < 2^31. */
nvbio_status nvbio_fm_residual_diagonals(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint32_t* ids_dev, uint32_t n, uint32_t cap,
                                         uint32_t seeds_per_read, uint32_t seed_interval, uint32_t seed_len, uint32_t read_len,
                                         const uint32_t* read_offsets_dev, const uint32_t* seed_intervals_dev,
                                         uint64_t* keys_dev, uint32_t* n_keys_dev, void* stream);

/* -------------------------------------------------------------------------------------------
 * nvBowtie's scoring stream, as data: what a specialisation of aln::BatchedBandedAlignmentScore for
 * bowtie2::cuda::BestScoreStream (nvBowtie/bowtie2/cuda/score_inl.h:44-136) hands over instead of per-item callbacks.
 * The arrays are the members of the stream's pipeline object (pipeline_states.h:49-115, scoring_queues.h:211-289), all in HBM:
 *   idx_queue_dev    pipeline.idx_queue: work item i scores hit idx_queue[i] (NULL: hit i)            (score_inl.h:89)
 *   hit_read_id_dev  pipeline.scoring_queues.hits.read_id
 *   hit_seed_dev     pipeline.scoring_queues.hits.seed, one packed_seed word per hit (defs.h:162-172: pos_in_read:12,
 *                    index_dir:1, rc:1, top_flag:1 from bit 0; only rc is read)
 *   hit_loc_dev      pipeline.scoring_queues.hits.loc                                                  (score_inl.h:100)
 *   hit_score_dev / hit_sink_dev   pipeline.scoring_queues.hits.score / .sink: the stream's output   (score_inl.h:127-129)
 *   n                pipeline.hits_queue_size = stream.size()
 * ------------------------------------------------------------------------------------------- */
typedef struct
{
    const uint32_t* idx_queue_dev;
    uint32_t*       hit_read_id_dev;   /* written by nvbio_seed_hits_select, read by the scoring stream calls */
    uint32_t*       hit_seed_dev;
    uint32_t*       hit_loc_dev;
    int32_t*        hit_score_dev;
    uint32_t*       hit_sink_dev;
    uint32_t        n;
} nvbio_hit_queues;

/* BestScoreStream::init_context + the orientation of load_strings for every work item (score_inl.h:85-115, alignment_utils.h:277-302):
 * the four per-job arrays of an nvbio_alignment_batch.  read_index_dev = the read batch's sequence_index (n_reads + 1 symbol
 * offsets, io::SequenceData); band_len = the stream's m_band_len; genome_len = pipeline.genome_length; reads_reversed = 1 when
 * the read batch was loaded with io::REVERSE, as nvBowtie does (nvBowtie.cpp:322,337,356): a forward hit then reads the stored
 * stream backwards (NVBIO_READ_REVERSE) and a reverse-complemented one forwards, complemented (NVBIO_READ_COMPLEMENT).
 * (context->min_score = max(second best, score_limit) is not needed: whole-pattern banded scoring never reads it,
 * gotoh_banded_inl.h:610-622 is the windowed form only.) */
nvbio_status nvbio_score_stream_flatten(int device, const nvbio_hit_queues* hits, const uint32_t* read_index_dev, uint32_t band_len,
                                        uint32_t genome_len, uint32_t reads_reversed, uint32_t* read_id_dev, uint8_t* flags_dev,
                                        uint32_t* win_begin_dev, uint32_t* win_end_dev, void* stream);
/* BestScoreStream::output for every work item (score_inl.h:119-133): hit.score = max( score, worst_score ) (scheme_type::worst_score
 * = -65536, scoring.h:223-224), hit.sink = window begin + sink.x, scattered through idx_queue. */
nvbio_status nvbio_score_stream_output(int device, const nvbio_hit_queues* hits, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                       const uint32_t* win_begin_dev, int32_t worst_score, void* stream);

/* -------------------------------------------------------------------------------------------
 * nvBowtie's seed-hit bookkeeping (the data-parallel kernels of its best-approx extension loop, aligner_best_approx.h:453-666):
 * per-read deques of seed hits, the selection of the next SA row, the effort-limited best / second-best reduction.
 * One SeedHit = 8 bytes: x = range_begin, y = range_delta:20 | pos_in_read:10 | rc:1 | indexdir:1 (seed_hit.h:45-218; the SA range is
 * exclusive at its end).  A read's deque is the reference's priority_deque over an interval heap (nvbio/basic/priority_deque.h):
 * element 0 the largest range, element 1 the smallest; `capacity` entries per read, deque r at deques_dev + r * capacity.
 * ------------------------------------------------------------------------------------------- */
typedef struct
{
    uint32_t seeds_per_read;   /* seeds mapped per read and strand                                                               */
    uint32_t first_offset;     /* stored offset of seed 0: retry * (seed_freq / (max_reseed + 1))         (mapping_inl.h:520,528) */
    uint32_t seed_interval;    /* seed_freq( read_len )                                                                          */
    uint32_t seed_len;
    uint32_t read_len;         /* reads of one length (< 1024: SeedHit keeps positions in 10 bits)                               */
    uint32_t max_hits;         /* cap of the deque: a full deque drops its largest range before every push  (:242-244; default 100) */
    uint32_t rep_seeds;        /* reseed when the mean range size reaches this                              (:549; default 1000)    */
    uint32_t max_effort;       /* failed extensions in a row that end a read's search                       (reduce.h:72-97; 15)   */
    uint32_t min_ext;          /* ... counted only from this many extensions on                             (default 30)           */
    uint32_t max_ext;          /* hard limit of extensions per read                                         (default 400)          */
} nvbio_seed_hits_params;

/* entries per read a deque array needs for these parameters: min( 2 x seeds_per_read, max_hits ) + 1 */
nvbio_status nvbio_seed_hits_capacity(uint32_t seeds_per_read, uint32_t max_hits, uint32_t* capacity);

/* seed_mapper<EXACT_MAPPING>::enact + the bookkeeping of map_kernel (nvBowtie/bowtie2/cuda/mapping_inl.h:193-282,485-556) for reads whose
 * seeds have been matched: fw_ranges_dev / rc_ranges_dev [n_reads x seeds_per_read] = nvbio_fm_match over the seeds of the STORED
 * (reversed) reads with NVBIO_FM_SCAN_FORWARD, and with NVBIO_FM_COMPLEMENT (reverse scan) -- the two match_range calls of the
 * mapper (USE_REVERSE_INDEX 0).  Work item t is read read_queue_dev[t] (NULL: read t).  For every seed, forward hit then
 * reverse-complemented hit are pushed in the reference's order under its max_hits rule; sizes_dev[r] = hits kept;
 * reseed_dev[r] (optional) = the reseeding decision `range_count == 0 || range_sum >= rep_seeds * range_count`. */
nvbio_status nvbio_seed_hits_map(int device, const nvbio_uint2* fw_ranges_dev, const nvbio_uint2* rc_ranges_dev, const uint32_t* read_queue_dev,
                                 uint32_t n_reads, const nvbio_seed_hits_params* params, nvbio_uint2* deques_dev, uint32_t* sizes_dev,
                                 uint8_t* reseed_dev, void* stream);

/* nvBowtie's APPROXIMATE seed mapper, seed_mapper<APPROX_MAPPING> (mapping_inl.h:114-184,288-342): every seed of the stored (reversed) read is
 * searched four times -- forwards in `index` and backwards in `reverse_index` (the FM-index of the REVERSED text, nvBowtie's rfmi), as it
 * stands and complemented -- each search matching its first half exactly and allowing one substitution in the rest; the two searches that
 * start from the seed's near end also report the exact match.  Every non-empty range is pushed to the read's deque under the max_hits
 * rule, in the reference's order, with the reference's flags (position in the read, strand, index direction).  reads_dev: reads of
 * params->read_len symbols back to back (read_bits 2 / 4 / 8).  Deques need nvbio_seed_hits_approx_capacity entries per read. */
nvbio_status nvbio_seed_hits_approx_capacity(uint32_t seeds_per_read, uint32_t seed_len, uint32_t max_hits, uint32_t* capacity);
nvbio_status nvbio_seed_hits_map_approx(nvbio_fm_index_t index, nvbio_fm_index_t reverse_index, const void* reads_dev, uint32_t read_bits,
                                        const uint32_t* read_queue_dev, uint32_t n_reads, const nvbio_seed_hits_params* params,
                                        nvbio_uint2* deques_dev, uint32_t* sizes_dev, uint8_t* reseed_dev, void* stream);

/* select_kernel (select_inl.h:62-130): every active read (active_in_dev[t] = read id | top_flag << 31, the reference's packed_read) whose
 * search has not stopped (trys_dev[read] != 0; NULL = never) and which has a hit left takes a slot of the output queue: its next SA row
 * goes to hits->hit_loc_dev[slot], with hits->hit_read_id_dev[slot], hits->hit_seed_dev[slot] = packed_seed( pos_in_read, index_dir, rc,
 * top_flag ) and active_out_dev[slot]; the row is popped off the front of the read's top (smallest) range in place, an exhausted top
 * range being dropped first (which clears the top flag).  *count_dev = slots written (slot order is arbitrary, as in the reference). */
nvbio_status nvbio_seed_hits_select(int device, const uint32_t* active_in_dev, uint32_t n_active, const uint32_t* trys_dev, uint32_t capacity,
                                    nvbio_uint2* deques_dev, uint32_t* sizes_dev, uint32_t* active_out_dev, const nvbio_hit_queues* hits,
                                    uint32_t* count_dev, void* stream);

/* the tail of nvBowtie's locate kernels (locate_inl.h:127-136,188-198): hit.loc = located position - seed.pos_in_read, uint32 arithmetic,
 * for the hits->n selected hits (positions_dev from nvbio_fm_locate over hits->hit_loc_dev) */
nvbio_status nvbio_seed_hits_loc(int device, const uint32_t* positions_dev, const nvbio_hit_queues* hits, void* stream);

/* score_reduce_kernel with ReduceBestApproxContext (reduce_inl.h:65-140, reduce.h:55-99), one hit per active read: slot i of the scoring
 * queue (hits->hit_score_dev / hit_loc_dev / hit_seed_dev [i]) belongs to read active_dev[i] & 0x7FFFFFFF.
 *   best_dev[4 r ..]  = { a1 score, a1 locus, a2 score, a2 locus } (io::BestAlignments: scores start at the read's worst score, loci at
 *                       0xFFFFFFFF = unaligned, aligner.h:279-301); best_rc_dev[r] = a1 strand | a2 strand << 1
 *   a hit at a locus already held is skipped; a better one becomes a1 (a1 moves to a2); a worse one that beats a2 and is `distinct`
 *   from a1 (other strand or more than read_len / 2 away, nvbio/io/alignments_inl.h:26-38) becomes a2 -- either resets trys_dev[r] to
 *   max_effort; anything else is a failure: from min_ext extensions on (n_ext = extensions before this pass) a hit that is not from
 *   the top seed decrements trys_dev[r], and at zero, or at max_ext extensions, the read's deque is erased (sizes_dev[r] = 0). */
nvbio_status nvbio_score_reduce_effort(int device, const uint32_t* active_dev, const nvbio_hit_queues* hits, uint32_t read_len, uint32_t n_ext,
                                       const nvbio_seed_hits_params* params, int32_t* best_dev, uint8_t* best_rc_dev, uint32_t* trys_dev,
                                       uint32_t* sizes_dev, void* stream);

/* The several-hits-per-read form of the two calls above (select_multi_kernel, select_inl.h:268-437; score_reduce_kernel's loop over a read's
 * hits, reduce_inl.h:94-134): once fewer than half a batch of reads are active nvBowtie takes up to n_multi = BATCH_SIZE / active SA rows per read
 * and pass (aligner_best_approx.h:487-510).  A read's hits take CONSECUTIVE slots of the hit queue in selection order:
 * hits_first_dev[s] / hits_count_dev[s] for the read in slot s of active_out_dev; counts_dev[0] = reads written, counts_dev[1] = hits written
 * (<= n_active * n_multi).  The reduction runs the one-hit rule over every read's hits in that order with n_ext + i as the extension
 * count of hit i (ReduceBestApproxContext::failure( idx, .. ), reduce.h:82-92). */
nvbio_status nvbio_seed_hits_select_multi(int device, const uint32_t* active_in_dev, uint32_t n_active, const uint32_t* trys_dev, uint32_t capacity,
                                          uint32_t n_multi, nvbio_uint2* deques_dev, uint32_t* sizes_dev, uint32_t* active_out_dev,
                                          uint32_t* hits_first_dev, uint32_t* hits_count_dev, const nvbio_hit_queues* hits, uint32_t* counts_dev,
                                          void* stream);
nvbio_status nvbio_score_reduce_effort_multi(int device, const uint32_t* active_dev, uint32_t n_active, const uint32_t* hits_first_dev,
                                             const uint32_t* hits_count_dev, const nvbio_hit_queues* hits, uint32_t read_len, uint32_t n_ext,
                                             const nvbio_seed_hits_params* params, int32_t* best_dev, uint8_t* best_rc_dev, uint32_t* trys_dev,
                                             uint32_t* sizes_dev, void* stream);

/* The read queues of nvBowtie's best-approx loop as calls, so that a host loop over this ABI needs no device code of its own
 * (aligner_best_approx.h:77,148-207,363-450):
 *   nvbio_best_approx_init      best_dev[4 r ..] = { worst, -1, worst, -1 }, best_rc_dev[r] = 0: init_alignments( reads, threshold_score )
 *   nvbio_read_queue_begin      for the n reads of queue_dev (NULL: reads 0..n-1), any of: seed_offsets_dev[t] = read * read_len + first_offset (the
 *                               offsets of a seeding pass's seed set), active_dev[t] = read | top_seed << 31 (packed_read), trys_dev[read] =
 *                               max_effort_init (select_init)
 *   nvbio_read_queue_filter     queue_out_dev = the reads of queue_dev (NULL: 0..n-1) with read_flags_dev[read] != 0, in order (the reads that asked
 *                               for reseeding go round again); *count_dev (device) = how many */
nvbio_status nvbio_best_approx_init(int device, uint32_t n_reads, int32_t worst_score, int32_t* best_dev, uint8_t* best_rc_dev, void* stream);
nvbio_status nvbio_read_queue_begin(int device, const uint32_t* queue_dev, uint32_t n, uint32_t read_len, uint32_t first_offset, uint32_t top_seed,
                                    uint32_t max_effort_init, uint32_t* seed_offsets_dev, uint32_t* active_dev, uint32_t* trys_dev, void* stream);
nvbio_status nvbio_read_queue_filter(int device, const uint32_t* queue_dev, uint32_t n, const uint8_t* read_flags_dev, uint32_t* queue_out_dev,
                                     uint32_t* count_dev, void* stream);

/* -------------------------------------------------------------------------------------------
 * nvBowtie's PAIRED-END best-approx loop, its data-parallel steps (aligner_best_approx_paired.h:84-200,590-1000): the anchor mate's seed hits
 * are walked as in the single-end loop (deques, select, locate: the calls above); a selected hit's anchor is band-aligned against a
 * threshold derived from the best PAIRS found so far (BestAnchorScoreStream, score_inl.h:143-274; compute_target_score,
 * alignment_utils.h:93-102), the hits whose anchor passes get the opposite mate aligned by full-matrix DP in the window the fragment
 * constraints allow (BestOppositeScoreStream, score_inl.h:283-456) and score_reduce_paired_kernel keeps the best two pairs, or the best two
 * alignments of each mate while no pair has been found (reduce_inl.h:157-290).
 * An alignment is 4 int32: { score, position (-1 = none), sink offset, rc | mate << 1 | paired << 2 } (io::Alignment, alignments.h:71-117);
 * best_a_dev / best_o_dev hold two per read pair each (pipeline.best_alignments / best_alignments_o).  Reads of both mates are stored reversed.
 * ------------------------------------------------------------------------------------------- */
typedef struct
{
    uint32_t anchor;                              /* 0: mate 1 is the anchor of this pass over the batch, 1: mate 2          */
    uint32_t anchor_len, opposite_len;            /* read lengths (uniform)                                                   */
    int32_t  anchor_perfect_score, opposite_perfect_score;   /* scheme.perfect_score( len ) = match * len                    */
    int32_t  anchor_min_score, opposite_min_score;           /* scheme.min_score( len )                                      */
    int32_t  score_limit;                         /* scheme.score_limit(): NVBIO_SCORE_MIN for the Smith-Waterman scheme     */
    int32_t  worst_score;                         /* scheme_type::worst_score = -65536                                       */
    int32_t  match, txt_gap_open, txt_gap_ext;    /* for aln::max_text_gaps (utils_inl.h:145-167)                            */
    uint32_t band, genome_len;
    uint32_t policy, min_frag_len, max_frag_len, overlap, unpaired;   /* NVBIO_PE_POLICY_*, minins, maxins, pe_overlap, pe_unpaired */
    uint32_t max_effort, min_ext, max_ext;
} nvbio_pe_params;
nvbio_status nvbio_pe_init(int device, uint32_t n_reads, int32_t worst_score_mate1, int32_t worst_score_mate2, int32_t* best_a_dev, int32_t* best_o_dev, void* stream);
/* per selected hit: the anchor's banded window + orientation + its min_score (INT32_MAX for a locus already held, whose window is empty) */
nvbio_status nvbio_pe_anchor_flatten(int device, const nvbio_pe_params* params, const nvbio_hit_queues* hits, const int32_t* best_a_dev, const int32_t* best_o_dev,
                                     uint32_t* read_id_dev, uint8_t* flags_dev, uint32_t* win_begin_dev, uint32_t* win_end_dev, int32_t* min_scores_dev,
                                     void* stream);
/* hit.score = score >= min_score ? score : worst_score, hit.sink; hit_opposite_score_dev[i] = worst_score; valid_dev[i] = the anchor passed */
nvbio_status nvbio_pe_anchor_output(int device, const nvbio_pe_params* params, const nvbio_hit_queues* hits, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                    const uint32_t* win_begin_dev, const int32_t* min_scores_dev, int32_t* hit_opposite_score_dev, uint8_t* valid_dev,
                                    void* stream);
/* for the hits queue_dev[0..n) whose anchor passed: the opposite mate's window, orientation and min_score (the empty window where it cannot run) */
nvbio_status nvbio_pe_opposite_flatten(int device, const nvbio_pe_params* params, const uint32_t* queue_dev, uint32_t n, const nvbio_hit_queues* hits,
                                       const int32_t* best_a_dev, const int32_t* best_o_dev, uint32_t* read_id_dev, uint8_t* flags_dev,
                                       uint32_t* win_begin_dev, uint32_t* win_end_dev, int32_t* min_scores_dev, void* stream);
nvbio_status nvbio_pe_opposite_output(int device, const nvbio_pe_params* params, const uint32_t* queue_dev, uint32_t n, const int32_t* scores_dev,
                                      const nvbio_uint2* sinks_dev, const uint32_t* win_begin_dev, const uint32_t* win_end_dev, const int32_t* min_scores_dev,
                                      int32_t* hit_opposite_score_dev, uint32_t* hit_opposite_loc_dev, uint32_t* hit_opposite_sink_dev, void* stream);
/* score_reduce_paired_kernel over every active read's hits (hits_first / hits_count as nvbio_score_reduce_effort_multi) */
nvbio_status nvbio_pe_score_reduce(int device, const nvbio_pe_params* params, const uint32_t* active_dev, uint32_t n_active, const uint32_t* hits_first_dev,
                                   const uint32_t* hits_count_dev, const nvbio_hit_queues* hits, const int32_t* hit_opposite_score_dev,
                                   const uint32_t* hit_opposite_loc_dev, const uint32_t* hit_opposite_sink_dev, uint32_t n_ext, int32_t* best_a_dev,
                                   int32_t* best_o_dev, uint32_t* trys_dev, uint32_t* sizes_dev, void* stream);
/* queue_out_dev = the indices i in [0, n) with flags_dev[i] != 0, in order; *count_dev (device) = how many */
nvbio_status nvbio_select_flagged_indices(int device, const uint8_t* flags_dev, uint32_t n, uint32_t* queue_out_dev, uint32_t* count_dev, void* stream);

/* Two conversions a binding of sw-benchmark's stream needs (sw-benchmark/sw-benchmark.cu:70-209): its reference text is packed
 * 2-bit LITTLE-endian (REF_BIG_ENDIAN = false, :64-65; the library reads the big-endian layout of io::SequenceData<DNA>), and its
 * output() stores `sink.score` into an int16 array (:197). */
nvbio_status nvbio_text_2bit_le_to_be(int device, const uint32_t* in_dev, uint32_t n_words, uint32_t* out_dev, void* stream);
nvbio_status nvbio_scores_to_int16(int device, const int32_t* scores_dev, uint32_t n, int16_t* out_dev, void* stream);

/* -------------------------------------------------------------------------------------------
 * Gotoh scoring
 * ------------------------------------------------------------------------------------------- */
typedef enum { NVBIO_GLOBAL = 0, NVBIO_LOCAL = 1, NVBIO_SEMI_GLOBAL = 2 } nvbio_alignment_type;   /* alignment.h:242 */

/* The Gotoh scoring-scheme concept (nvbio/alignment/alignment.h:437-449) as data.
 * aln::SimpleGotohScheme(match, mismatch, open, ext) (nvbio/alignment/utils.h:103-123) is
 *   { match, -mismatch, -mismatch, open, ext, open, ext };
 * nvBowtie's SmithWatermanScoringScheme<QualCost,ConstantCost> (nvBowtie/bowtie2/cuda/scoring.h:206-330) is
 *   { m_match, mmp_min, mmp_max, -(read_gap_const+read_gap_coeff), -read_gap_coeff,
 *     -(ref_gap_const+ref_gap_coeff), -ref_gap_coeff }
 * with mismatch(q) = -( mm_min + int( float(min(q,40))/40.0f * (mm_max-mm_min) ) ) (scoring.h:84-88). */
typedef struct
{
    int32_t match;
    int32_t mm_min, mm_max;              /* mismatch penalties (positive) at quality 0 / >= 40 */
    int32_t pat_gap_open, pat_gap_ext;   /* signed scores */
    int32_t txt_gap_open, txt_gap_ext;
} nvbio_gotoh_scheme;

/* The linear-gap Smith-Waterman scheme, aln::SimpleSmithWatermanScheme(match, mismatch, deletion, insertion)
 * (nvbio/alignment/utils.h:81-98), all four as signed scores: `deletion` is charged for a text symbol aligned to no pattern
 * symbol, `insertion` for a pattern symbol aligned to no text symbol (sw/sw_banded_inl.h:369-370,398-431).  The
 * edit-distance aligner is this scheme with (0, -1, -1, -1) (EditDistanceSWScheme, ed/ed_utils.h). */
typedef struct
{
    int32_t match, mismatch, deletion, insertion;
} nvbio_sw_scheme;

#define NVBIO_SCORE_MIN (-(1 << 30))     /* Field_traits<int32>::min(), BestSink's initial score (numbers.h:738-742) */

/* A batch of alignment jobs: the stream concept of aln::Batched[Banded]AlignmentScore
 * (pattern_length / text_length / load_strings; nvbio/alignment/batched.h:274-298, sw-benchmark.cu:70-209,
 * nvBowtie score_inl.h:44-136 + alignment_utils.h:194-308) flattened to arrays.
 *   job i aligns pattern = read[ read_id ? read_id[i] : i ], optionally reversed / complemented
 *   (flags bit0 / bit1: ReadStream, nvbio/io/utils.h:150-168), against text[ win_begin[i], win_end[i] ).   */
typedef struct
{
    const void*     reads_dev;         /* packed big-endian words (read_bits 2, 4) or bytes (8)            */
    uint32_t        read_bits;
    const uint32_t* read_offsets_dev;  /* n_reads+1 symbol offsets (io::SequenceData sequence_index)        */
    const uint8_t*  quals_dev;         /* one byte per read symbol, or NULL (trivial_quality_string -> 0)   */
    const uint32_t* read_id_dev;       /* n entries or NULL                                                 */
    const uint8_t*  flags_dev;         /* n entries or NULL                                                 */
    const void*     text_dev;          /* packed 2-bit big-endian words (text_bits 2) or bytes (8)          */
    uint32_t        text_bits;
    const uint32_t* win_begin_dev;     /* n entries: window begin (symbol index into text)                  */
    const uint32_t* win_end_dev;       /* n entries: window end (exclusive)                                 */
    uint32_t        n;
    uint32_t        max_read_len;      /* the stream's max_pattern_length() (batched.h stream concept), or 0 if
                                          unknown.  It lets the library pick 16-bit packed kernels when every
                                          score provably fits: a job whose pattern is LONGER than a non-zero
                                          max_read_len is rejected (score NVBIO_SCORE_MIN, sink (-1,-1)) rather
                                          than scored in registers that could wrap.                          */
    uint32_t        algo_flags;        /* NVBIO_ALN_*: which of the library's exact shortcuts / kernel variants a
                                          call may use (0 = all; A/B measurements and tests -- results are
                                          identical with every combination)                                  */
} nvbio_alignment_batch;


enum
{
    NVBIO_ALN_NO_UNGAPPED_SCORE     = 1,   /* every job through the DP (no ungapped end-to-end shortcut, banded and full matrix) */
    NVBIO_ALN_NO_THIRD_CHANCE       = 2,   /* three-mismatch jobs go to the DP                                                  */
    NVBIO_ALN_NO_PACKED_DP          = 4,   /* int32 kernels only                                                                */
    NVBIO_ALN_FORCE_PACKED_DP       = 8,   /* packed full-matrix kernel also for small batches                                  */
    NVBIO_ALN_NO_UNGAPPED_TRACEBACK = 16,  /* every traceback through the direction-vector DP                                   */
    NVBIO_ALN_PK_THREE_WAVES        = 32,  /* packed band-31 kernel built for 3 waves per SIMD (168 VGPRs; the prologue state spills)
                                              instead of 2 (256 VGPRs, nothing spills: the default since the row loop issues without
                                              idle slots and no longer gains from a third wave)                                  */
    NVBIO_ALN_NO_SECOND_CHANCE      = 128, /* two-mismatch jobs go to the DP (A/B: what the check costs inside the first pass)           */
    NVBIO_ALN_NO_NARROW_SCORE       = 512, /* end-to-end full-matrix scoring: every job the shortcut cannot settle through the DP over the whole window
                                              (no band-31 attempt around the best diagonal with its run test)                    */
    NVBIO_ALN_NO_BAND_ROUTE         = 1024, /* full-matrix end-to-end traceback: the jobs whose paths stay within 7 diagonals of their sink keep the
                                              row-restricted full-matrix kernel instead of the band-15 traceback kernel           */
    NVBIO_ALN_PK_STRIPE8            = 256, /* packed full-matrix scoring of end-to-end jobs (match = 0): the general kernel, 8 pattern columns per
                                              stripe, instead of the end-to-end one that sweeps 16 (A/B)                          */
    NVBIO_ALN_RAGGED_READS          = 4096, /* a HINT, the only flag that is not an A/B switch: the batch's reads differ in length.  The packed band-31
                                              end-to-end kernel -- two alignments per lane -- then runs a build that takes a lane's two alignments
                                              in one pass whatever their lengths (the shorter one starts late); without the hint such a lane takes
                                              two passes.  Results are identical either way.                                        */
    NVBIO_ALN_NO_LENGTH_SORT        = 8192, /* with NVBIO_ALN_RAGGED_READS: the DP's job list in batch order instead of ascending read length (A/B)   */
    NVBIO_ALN_NO_QUALITY_SHORTCUT   = 2048, /* band-31 end-to-end scoring of reads WITH base qualities under a quality-dependent mismatch penalty
                                              (nvBowtie's default ramp): every job through the DP, as before round 3 (A/B)          */
    NVBIO_ALN_NO_GAP_CHANCE         = 65536, /* band-31 end-to-end scoring: jobs without a near-clean diagonal (reads with an indel) go to the DP instead of the
                                               exact evaluation of their one-gap alignments (gap_chance_e2e31_kernel) (A/B)                               */
    NVBIO_ALN_NO_COOPERATIVE_DP     = 32768, /* full-matrix GLOBAL / SEMI_GLOBAL scoring: one lane per job and the boundary column in memory even where the
                                               several-lanes-per-job kernel (boundary in registers, no scratch) applies (A/B)                           */
    NVBIO_ALN_NO_F16_DP             = 16384, /* packed band-31 DP with 16-bit INTEGER lanes even where the binary16 lanes (exact while every score is an integer of
                                               magnitude <= 2040; one operation less per cell) would do (A/B)                                           */
    NVBIO_ALN_NO_NARROW_TRACEBACK   = 64   /* band-31 end-to-end traceback: every DP over the whole band (no band-15 route for the jobs
                                              whose optimal paths provably stay within 7 diagonals of the sink)                  */
};

enum { NVBIO_READ_REVERSE = 1, NVBIO_READ_COMPLEMENT = 2 };

/* Best candidate per read after the extension (the per-read score reduction of examples/fmmap/fmmap.cu:367-376, with a
 * total order): for candidate i of read keys_dev[i] >> 34,
 *   sel = max(scores[i] + 2^20, 0) << 34 | strand << 33 | (win_begin[i] + sinks[i].x)     (end position, hit.sink)
 * and best_dev[read] = max(best_dev[read], sel) by 64-bit atomic max.  The caller zero-initialises best_dev;
 * a read without candidates keeps 0.  */
nvbio_status nvbio_best_candidate_reduce(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                         const uint32_t* win_begin_dev, uint64_t n, uint64_t* best_dev, void* stream);

/* ... and back: scores_dev[r] (NVBIO_SCORE_MIN without a candidate), end_pos_dev[r] (-1 without), rc_dev[r] from best_dev[r] */
nvbio_status nvbio_best_candidate_unpack(int device, const uint64_t* best_dev, uint32_t n_reads, int32_t* scores_dev,
                                         int64_t* end_pos_dev, uint8_t* rc_dev, void* stream);

/* The window of every read's best candidate, for the traceback of the best alignments (nvBowtie's banded_traceback_best re-aligns the
 * best alignment inside the window it was scored in, traceback_inl.h:191-247): for every candidate i whose selection key equals
 * best_dev[read] (the final result of nvbio_best_candidate_reduce over ALL candidates),
 *   best_wb_dev[read] = max( best_wb_dev[read], win_begin_dev[i] ),  best_locus_dev[read] = max( ., max( diagonal of keys_dev[i], 0 ) )
 * by 64-bit atomic max (several candidates can tie on the whole key: the largest window begin wins).  The caller initialises both
 * arrays to -1; best_locus_dev may be NULL. */
nvbio_status nvbio_best_candidate_windows(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                          const uint32_t* win_begin_dev, uint64_t n, const uint64_t* best_dev, int64_t* best_wb_dev,
                                          int64_t* best_locus_dev, void* stream);

/* ... and the traceback batch of all reads from them (one job per read, job r = read r; the stream of banded_traceback_best):
 * flags (strand: NVBIO_READ_REVERSE | NVBIO_READ_COMPLEMENT), window [best_wb, min( best_wb + band + read_len, genome_len )), and
 * the score and sink the scoring pass already found there (sink = (end position - window begin, read_len): for
 * NVBIO_TRACEBACK_SINKS_GIVEN).  A read without a candidate, or whose best score is below min_score, gets the empty window
 * [0, 0), score NVBIO_SCORE_MIN and sink (-1, -1): the traceback reports nothing for it (cigar length 0). */
nvbio_status nvbio_traceback_best_batch(int device, const uint64_t* best_dev, const int64_t* best_wb_dev, uint32_t n_reads, uint32_t read_len,
                                        uint32_t band, uint32_t genome_len, int32_t min_score, uint8_t* flags_dev, uint32_t* win_begin_dev,
                                        uint32_t* win_end_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream);

/* The second-best alignment per read, as nvBowtie's score_reduce keeps it (nvBowtie/bowtie2/cuda/reduce_inl.h:65-140:
 * best a1 and a second a2 that must be `distinct` from a1 -- io::distinct_alignments, nvbio/io/alignments_inl.h:26-38: the other
 * strand, or more than distinct_dist = read_len / 2 positions away -- and score above the read's threshold; candidates at a
 * location already held are skipped).  The reference's loop depends on the order the extension results arrive in; this is
 * that loop applied to the candidates in descending order of the selection key, in two order-free passes: run
 * nvbio_best_candidate_reduce over ALL candidates first, then this call over the same arrays:
 *   second_dev[read] = max over the read's candidates with scores[i] > worst_score, sel != best_dev[read] and
 *                      distinct( best end position / strand, candidate end position / strand, distinct_dist ) of sel.
 * (Positions are end positions, hit.sink, where the reference compares hit.loc, the diagonal's start: the two differ by
 * read_len +- band.)  The caller zero-initialises second_dev; 0 = no second alignment (BestAlignments::has_second()). */
nvbio_status nvbio_second_candidate_reduce(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                           const uint32_t* win_begin_dev, uint64_t n, const uint64_t* best_dev,
                                           uint32_t distinct_dist, int32_t worst_score, uint64_t* second_dev, void* stream);

/* Bowtie2's mapping quality of every read from its best and second-best selection keys: BowtieMapq2 (version 2, the
 * evaluator nvBowtie instantiates, bowtie2_cuda_driver.cu:277,600) or BowtieMapq3 (version 3), single-end form of
 * nvBowtie/bowtie2/cuda/mapq.h:32-297.  perfect_score = scheme.perfect_score(read_len) (match * read_len, scoring.h:274),
 * min_score = scheme.min_score(read_len) (MinScoreFunc, scoring.h:117-129), monotone = (match bonus == 0) (scoring.h:339).
 * second_dev may be NULL (no second alignments); second_scores_dev (optional) receives the second-best scores
 * (NVBIO_SCORE_MIN where there is none).  Reads without a candidate get 0. */
typedef struct
{
    int32_t version;         /* 2 or 3 */
    int32_t monotone;
    int32_t perfect_score;
    int32_t min_score;
} nvbio_mapq_params;
nvbio_status nvbio_mapq(int device, const uint64_t* best_dev, const uint64_t* second_dev, uint32_t n_reads,
                        const nvbio_mapq_params* params, int32_t* second_scores_dev, uint8_t* mapq_dev, void* stream);

/* Paired-end: the genome window in which the opposite mate of an anchored mate is aligned (full-matrix DP),
 * BestOppositeScoreStream::init_context (nvBowtie/bowtie2/cuda/score_inl.h:389-425) with frame_opposite_mate
 * (alignment_utils.h:52-88).  g_pos = the anchor hit's locus (hit.loc), anchor_rc its strand, anchor = 0 if mate 1 is
 * the anchor, opposite_gapped_len = opposite mate length + aln::max_text_gaps(aligner, min_score, length)
 * (nvbio/alignment/utils_inl.h:141-162).  flags_dev = how the opposite mate is read (NVBIO_READ_REVERSE |
 * NVBIO_READ_COMPLEMENT when it aligns reverse-complemented); valid_dev = 0 where the window is empty or starts
 * past the genome end (the reference skips such hits). */
enum { NVBIO_PE_POLICY_FF = 0, NVBIO_PE_POLICY_FR = 1, NVBIO_PE_POLICY_RF = 2, NVBIO_PE_POLICY_RR = 3 };
nvbio_status nvbio_opposite_mate_windows(int device, const uint32_t* g_pos_dev, const uint8_t* anchor_rc_dev, uint32_t n,
                                         uint32_t anchor_len, uint32_t opposite_gapped_len, uint32_t anchor, uint32_t policy,
                                         uint32_t min_frag_len, uint32_t max_frag_len, uint32_t overlap, uint32_t genome_len,
                                         uint32_t* win_begin_dev, uint32_t* win_end_dev, uint8_t* flags_dev, uint8_t* valid_dev,
                                         void* stream);

/* scores_dev[i], sinks_dev[i] = BestSink<int32> (score, (text_end, pattern_end)) after
 * aln::banded_alignment_score<band>( GotohAligner<type>, pattern, quals, text, min_score, sink )
 * (nvbio/alignment/gotoh/gotoh_banded_inl.h:397-688; batched form batched_banded_inl.h:34-157).
 * band must be 3, 7, 15 or 31 (the instantiations nvBowtie dispatches, score_inl.h:468-508).
 * Jobs with text_len < pattern_len report nothing: score NVBIO_SCORE_MIN, sink (-1,-1).       */
nvbio_status nvbio_banded_gotoh_score(int device, uint32_t band, nvbio_alignment_type type,
                                      const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                      int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream);

/* The same through the reference's staged scheduler: BatchedBandedAlignmentScore<band, stream, DeviceStagedThreadScheduler>
 * (nvbio/alignment/batched_banded_inl.h:165-236; work unit StagedAlignmentUnitBase / BandedScoreUnit, batched_stream.h:117-285),
 * i.e. the windowed banded_alignment_score (banded_inl.h:179-208, gotoh/gotoh_banded_inl.h:703-727) over 32-row windows.  What
 * differs from nvbio_banded_gotoh_score in the RESULT: a job stops at a window's end when max(band) < min_score + rows left x match
 * (:610-622, the sign as the reference has it: exact for a zero match bonus, stricter than "cannot reach min_score" otherwise) -- LOCAL keeps what it reported so far, GLOBAL / SEMI_GLOBAL report nothing (NVBIO_SCORE_MIN, sink (-1,-1)) -- and the
 * band passes through int16 checkpoints clamped at -32736 between windows.  min_score of job i = min_scores_dev[i] (the stream's
 * context->min_score), or min_score for every job when min_scores_dev is NULL.  Instantiated for read_bits/text_bits 4/2, 2/2, 8/8. */
nvbio_status nvbio_banded_gotoh_score_staged(int device, uint32_t band, nvbio_alignment_type type,
                                             const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                             const int32_t* min_scores_dev, int32_t min_score,
                                             int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream);

/* Banded Gotoh traceback: aln::banded_alignment_traceback<band,CHECKPOINTS> / BatchedBandedAlignmentTraceback
 * (nvbio/alignment/banded_inl.h:354-483, gotoh/gotoh_banded_inl.h:730-950; nvBowtie banded_traceback_best,
 * traceback_inl.h:191-247) with nvBowtie's run-length Backtracker as the backtracer
 * (nvBowtie/bowtie2/cuda/alignment_utils.h:115-157).  Per job i:
 *   scores_dev[i], sinks_dev[i]   as nvbio_banded_gotoh_score;
 *   sources_dev[i]                Alignment::source: (text, pattern) cell where the alignment starts;
 *   cigars_dev[i*cigar_stride..]  io::Cigar elements (uint16: type in bits 0-1 -- 0 substitution/match,
 *                                 1 insertion, 2 deletion, 3 soft clip -- length in bits 2-15; nvbio/io/alignments.h:48-66)
 *                                 in BACKTRACKING order, exactly the sequence the Backtracker receives:
 *                                 [clip(pattern_len - sink.y)] ops... [clip(source.y)];
 *   cigar_lens_dev[i]             number of elements produced (elements beyond cigar_stride are dropped, the
 *                                 count is not); 0 with source = sink = (-1,-1) when nothing was reported;
 *                                 0xFFFFFFFF when the pattern is longer than batch->max_read_len (job skipped).
 * batch->max_read_len is REQUIRED here (the stream's max_pattern_length(): it sizes the scratch).
 * The reference recomputes the direction vectors from int16 checkpoints; this call returns
 * NVBIO_ERR_UNSUPPORTED for (scheme, max_read_len) combinations whose scores could leave that range, where the
 * reference's own result is truncation-dependent.
 * flags: NVBIO_TRACEBACK_SINKS_GIVEN -- scores_dev / sinks_dev already hold what nvbio_banded_gotoh_score returns
 * for this very batch (nvBowtie traces alignments it has scored, traceback_inl.h:103-113); the scoring pass is skipped.
 * Implementation note: jobs whose optimum is reached by the diagonal through the sink alone are traced without a
 * DP (the reference's tie rule makes their traceback all substitutions); the others run the DP once, writing
 * their direction vectors to scratch.  Results do not depend on which route a job takes.
 * temp_dev / temp_bytes: optional caller scratch (nvbio_banded_gotoh_traceback_temp_bytes); if NULL the library
 * allocates scratch (kept per stream) and processes the batch in as many launches as 16 GiB allow. */
enum { NVBIO_TRACEBACK_SINKS_GIVEN = 1 };
nvbio_status nvbio_banded_gotoh_traceback_temp_bytes(const nvbio_alignment_batch* batch_host_sizes, uint32_t band, uint64_t* bytes);
nvbio_status nvbio_banded_gotoh_traceback(int device, uint32_t band, nvbio_alignment_type type,
                                          const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                          int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                          uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                          uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream);

/* The same for the linear-gap SmithWatermanAligner (and so for the EditDistanceAligner's scheme (0, -1, -1, -1)):
 * aln::banded_alignment_traceback<band, MAX_PATTERN_LEN, CHECKPOINTS>( SmithWatermanAligner<type>, ... ) (nvbio/alignment/banded_inl.h:354-417
 * over sw/sw_banded_inl.h:281-520 and its walk, :741-797), deletion and insertion costs unequal or not.  As the reference's code behaves,
 * the direction vectors of this aligner carry no SINK (sw_banded_inl.h:420-468): a LOCAL walk does not stop where the score reaches 0
 * but runs on to the first pattern row, so that source.y = 0 for every traced job. */
nvbio_status nvbio_banded_sw_traceback(int device, uint32_t band, nvbio_alignment_type type,
                                       const nvbio_sw_scheme* scheme, const nvbio_alignment_batch* batch,
                                       int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                       uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                       uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream);

/* Full-matrix Gotoh traceback: aln::alignment_traceback<..,CHECKPOINTS> / BatchedAlignmentTraceback
 * (nvbio/alignment/alignment_inl.h:355-517, gotoh/gotoh_inl.h:458-538,1573-1640; nvBowtie traceback_best,
 * traceback_inl.h:249-275) with nvBowtie's run-length Backtracker: outputs as nvbio_banded_gotoh_traceback, with
 * x = text and y = pattern coordinates and INSERTION = pattern symbol without text, DELETION = text symbol without
 * pattern.  The scoring is the pattern-blocking pass (the one alignment_traceback itself runs), min_scores_dev
 * (optional) its stripe early exit.  max_pattern_len / max_text_len must bound every job (they size the scratch: a
 * boundary column plus 4 bits per DP cell, for whole waves of 64 jobs: temp_bytes must hold at least 64 jobs); a longer job is
 * skipped and flagged with cigar_lens = 0xFFFFFFFF.
 * flags: NVBIO_TRACEBACK_SINKS_GIVEN as above (scores / sinks from nvbio_full_gotoh_score with text_blocking = 0 and
 * the same min_scores).  NVBIO_ERR_UNSUPPORTED when scores could leave the reference's int16 checkpoints. */
nvbio_status nvbio_full_gotoh_traceback_temp_bytes(const nvbio_alignment_batch* batch_host_sizes, uint32_t max_pattern_len,
                                                   uint32_t max_text_len, uint64_t* bytes);
nvbio_status nvbio_full_gotoh_traceback(int device, nvbio_alignment_type type, const nvbio_gotoh_scheme* scheme,
                                        const nvbio_alignment_batch* batch, uint32_t max_pattern_len, uint32_t max_text_len,
                                        const int32_t* min_scores_dev,
                                        int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                        uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                        uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream);

/* The same for the linear-gap SmithWatermanAligner (and the EditDistanceAligner's scheme (0, -1, -1, -1)), deletion and insertion costs
 * unequal or not: aln::alignment_traceback<MAX_PATTERN_LEN, MAX_TEXT_LEN, CHECKPOINTS>( SmithWatermanAligner<type>, ... )
 * (nvbio/alignment/alignment_inl.h:355-517 over sw/sw_inl.h:306-392 -- the direction of a cell, SINK where a LOCAL score is 0 --,
 * :1476-1600 -- checkpoints and submatrices, swept in stripes of 16 pattern columns -- and its walk, :1644-1694).  Score and sink are
 * those of nvbio_full_sw_score with text_blocking = 0 (run here unless NVBIO_TRACEBACK_SINKS_GIVEN); scratch as
 * nvbio_full_gotoh_traceback_temp_bytes. */
nvbio_status nvbio_full_sw_traceback(int device, nvbio_alignment_type type, const nvbio_sw_scheme* scheme,
                                     const nvbio_alignment_batch* batch, uint32_t max_pattern_len, uint32_t max_text_len,
                                     const int32_t* min_scores_dev,
                                     int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                     uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                     uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream);

/* nvBowtie finish_alignment (nvBowtie/bowtie2/cuda/traceback_inl.h:536-705) for a batch that has been traced back by
 * either traceback call: from each job's CIGAR (as written by the traceback: backtracking order), its read as aligned and
 * its text window,
 *   ed_dev[i]   = edit distance (mismatches + inserted + deleted symbols; soft clips not counted), the NM of the alignment
 *                 (0 when nothing was traced, 0xFFFFFFFF when the CIGAR had been truncated to cigar_stride);
 *   mds_dev     (optional) = the MDS byte stream nvBowtie keeps per read (nvbio/io/alignments.h:37-43): 2 length bytes, then
 *                 [MDS_MATCH, run <= 255] / [MDS_MISMATCH, read symbol] / [MDS_INSERTION | MDS_DELETION, length, symbols...]
 *                 (soft clips are recorded as insertions, as the reference does); mds_lens_dev[i] = its length (bytes beyond
 *                 mds_stride are dropped, the count is not).
 * sources_dev = Alignment::source of the traceback (source.x = where the alignment starts in the text window). */
nvbio_status nvbio_finish_alignment(int device, const nvbio_alignment_batch* batch, const nvbio_uint2* sources_dev,
                                    const uint16_t* cigars_dev, uint32_t cigar_stride, const uint32_t* cigar_lens_dev,
                                    uint32_t* ed_dev, uint8_t* mds_dev, uint32_t mds_stride, uint32_t* mds_lens_dev, void* stream);

/* full-matrix Gotoh: aln::alignment_score / BatchedAlignmentScore (nvbio/alignment/gotoh/gotoh_inl.h:444-1256,
 * batched_inl.h:39-77).  text_blocking != 0 selects TextBlockingTag (sw-benchmark), 0 the default
 * PatternBlockingTag; min_scores_dev (optional) enables the reference's stripe early exit.
 * temp_dev / temp_bytes: optional caller scratch (see nvbio_full_gotoh_temp_bytes); if NULL the
 * library allocates and frees scratch itself (blocks kept per stream: see "Conventions"). */
nvbio_status nvbio_full_gotoh_temp_bytes(const nvbio_alignment_batch* batch_host_sizes, uint32_t max_pattern_len,
                                         uint32_t max_text_len, int text_blocking, uint64_t* bytes);
nvbio_status nvbio_full_gotoh_score(int device, nvbio_alignment_type type, int text_blocking,
                                    const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                    uint32_t max_pattern_len, uint32_t max_text_len,
                                    const int32_t* min_scores_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                    void* temp_dev, uint64_t temp_bytes, void* stream);

/* The Myers bit-vector aligner: aln::banded_alignment_score<band>( aln::EditDistanceAligner<TYPE, aln::MyersTag<5> >, ... )
 * (nvbio/alignment/myers/myers_banded_inl.h:172-315), the aligner examples/fmmap/fmmap.cu:346-359 instantiates (SEMI_GLOBAL, band 31).
 * scores_dev[i] = MINUS the edit distance found inside the band that slides down the window's main diagonal, sinks_dev[i] = (text column
 * where it ends, pattern length); NVBIO_SCORE_MIN / (-1,-1) when nothing is reported (text shorter than the pattern, or nothing reaches
 * min_score).  As the reference's code behaves, min_score is truncated to an int16 (:258) -- Field_traits<int32>::min() becomes 0 and
 * only distance-0 columns are reported; pass e.g. -32768 to see every distance.  band <= 32; GLOBAL and SEMI_GLOBAL only; text symbols
 * must be < 4 (the reference's match-vector set leaves the N vector uninitialised). */
nvbio_status nvbio_banded_myers_score(int device, uint32_t band, nvbio_alignment_type type, const nvbio_alignment_batch* batch, int32_t min_score,
                                      int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream);

/* Scoring into aln::Best2Sink<int32>( distinct_dist ) (nvbio/alignment/sink.h:96-116, sink_inl.h:55-83) instead of BestSink: the
 * best alignment in scores_dev / sinks_dev and a second one, ending more than distinct_dist text positions from the first, in
 * scores2_dev / sinks2_dev (NVBIO_SCORE_MIN / (-1,-1) when there is none).  The reference's sink does not demote the old best
 * when a new one arrives, so its result depends on the order of the reports; the kernels report cell by cell in the
 * reference's order (band rows, then columns; 8-wide stripes in the full matrix).  int32 kernels only; reads 2/4/8 bit over
 * 2-bit text, or bytes over bytes. */
nvbio_status nvbio_banded_gotoh_score_best2(int device, uint32_t band, nvbio_alignment_type type,
                                            const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch, uint32_t distinct_dist,
                                            int32_t* scores_dev, nvbio_uint2* sinks_dev, int32_t* scores2_dev, nvbio_uint2* sinks2_dev,
                                            void* stream);
nvbio_status nvbio_full_gotoh_score_best2(int device, nvbio_alignment_type type, int text_blocking,
                                          const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                          uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_scores_dev,
                                          uint32_t distinct_dist, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                          int32_t* scores2_dev, nvbio_uint2* sinks2_dev, void* stream);

/* The other two aligner families of the reference behind the same batch interface:
 * aln::SmithWatermanAligner<TYPE, scheme> and aln::EditDistanceAligner<TYPE> (nvbio/alignment/alignment.h:366-401,508-545).
 *   banded : banded_alignment_score<BAND> -> sw/sw_banded_inl.h:281-520 (the edit-distance aligner runs the same code with
 *            EditDistanceSWScheme, ed/ed_banded_inl.h:37-69: the aligner of examples/fmmap and of nvBowtie --scoring ed);
 *   full   : alignment_score -> sw/sw_inl.h:396-1300, ed/ed_inl.h:60-168.  These two families sweep the matrix in stripes of
 *            16 cells (sw_bandlen_selector, sw/sw_inl.h:1322-1325) where Gotoh uses 8, which decides LOCAL ties and where the
 *            min-score early exit is tested; both are reproduced.
 * Arguments, outputs and scratch exactly as nvbio_banded_gotoh_score / nvbio_full_gotoh_score
 * (nvbio_full_gotoh_temp_bytes sizes the scratch for either).  Qualities are ignored (constant mismatch score). */
nvbio_status nvbio_banded_sw_score(int device, uint32_t band, nvbio_alignment_type type,
                                   const nvbio_sw_scheme* scheme, const nvbio_alignment_batch* batch,
                                   int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream);
nvbio_status nvbio_full_sw_score(int device, nvbio_alignment_type type, int text_blocking,
                                 const nvbio_sw_scheme* scheme, const nvbio_alignment_batch* batch,
                                 uint32_t max_pattern_len, uint32_t max_text_len,
                                 const int32_t* min_scores_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                 void* temp_dev, uint64_t temp_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NVBIO_AMD_H */
