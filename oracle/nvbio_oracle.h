/* oracle/nvbio_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's CPU algorithm for the seed-and-extend hot
 * path (FM-index rank / match / locate, banded and full-matrix Gotoh scoring).
 * It is the checker for the HIP path: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product library (libnvbio_amd.so)
 * never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_*.py)
 *   (a) against the committed golden vectors in tests/golden/, which were produced by the
 *       reference's own host code (oracle/ref/nvbio_ref.cpp -> oracle/_ref/libnvbio_ref.so,
 *       generator tests/golden/make_golden.py), and which include the known answers of
 *       nvbio-test/alignment_test.cu:709-828 and nvbio-test/packedstream_test.cpp:143-172;
 *   (b) live against oracle/_ref/libnvbio_ref.so whenever that library is present
 *       (development container only).
 *
 * File:line citations are relative to the reference tree (/root/reference).
 */
#ifndef NVBIO_ORACLE_H
#define NVBIO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* --- packed streams (nvbio/basic/packedstream_inl.h:33-75; big-endian within the word) --- */
uint8_t orc_get2(const uint32_t* words, uint64_t i);            /* 2-bit symbol i  */
uint8_t orc_get4(const uint32_t* words, uint64_t i);            /* 4-bit symbol i  */
void    orc_pack2(const uint8_t* syms, uint64_t n, uint32_t* words);   /* words must be zeroed/own  */
void    orc_pack4(const uint8_t* syms, uint64_t n, uint32_t* words);

/* --- 2-bit popcounts (nvbio/basic/popcount_inl.h:230-236,318-341) --- */
uint32_t orc_popc_2bit(uint32_t x, uint32_t c);
uint32_t orc_popc_2bit_hi(uint32_t x, uint32_t c, uint32_t i);  /* all but the first (low) i symbols */

/* --- FM-index view (nvbio/fmindex/fmindex.h:320-361; production layout nvbio/io/fmindex/fmindex.h:75-177) --- */
typedef struct
{
    uint32_t        length;     /* n: symbols in the text (BWT has n symbols, no '$')          */
    uint32_t        primary;    /* row of '$'                                                    */
    uint32_t        L2[5];      /* L2[c] = # symbols < c                                         */
    const uint32_t* bwt_occ;    /* 32-byte records: 4 words BWT (64 symbols) + 4 words occ{A,C,G,T} */
    const uint32_t* ssa;        /* ssa[j] = SA[16 j]; ssa[0] = 0xFFFFFFFF; (n+16)/16 entries      */
} orc_fm_index;

/* number of uint32 words of the packed BWT, padded to a multiple of 4 (fmindex_test.cu:436-438) */
uint32_t orc_bwt_words(uint32_t n);

/* suffix array of text[0,n) (one symbol per byte, values 0..3) in the reference's convention:
 * sa[0] = n (the empty suffix), sa[1..n] = suffixes in lexicographic order, a proper prefix
 * sorting before its extensions (nvbio/fmindex/bwt.h:28-37).  Any correct SA gives the same
 * index, so this is a plain radix+comparison sort, not SA-IS.  sa must hold n+1 entries. */
void orc_suffix_sort(const uint8_t* text, uint32_t n, uint32_t* sa);

/* build the production index from text + SA:
 *   bwt/primary   nvbio/fmindex/bwt.h:41-53
 *   occ (K=64)    nvbio/fmindex/rank_dictionary_inl.h:33-66
 *   interleave    nvbio/io/fmindex/fmindex_impl.cu:300-313
 *   ssa (K=16)    nvbio/fmindex/ssa_inl.h:254-301 (entry 0 = -1)
 * bwt_occ must hold 2*orc_bwt_words(n) words, ssa (n+16)/16 words.  Returns primary. */
uint32_t orc_fm_build(const uint8_t* text, uint32_t n, const uint32_t* sa,
                      uint32_t* bwt_occ, uint32_t* ssa, uint32_t L2[5]);

/* rank_dictionary level (nvbio/fmindex/rank_dictionary_inl.h:338-479): k indexes the BWT text */
uint32_t orc_dict_rank(const orc_fm_index* f, uint32_t k, uint32_t c);
void     orc_dict_rank2(const orc_fm_index* f, uint32_t l, uint32_t r, uint32_t c, uint32_t out[2]);
void     orc_dict_rank4(const orc_fm_index* f, uint32_t k, uint32_t out[4]);

/* fm_index level (nvbio/fmindex/fmindex_inl.h:27-173): k indexes BWT-matrix rows */
uint32_t orc_rank(const orc_fm_index* f, uint32_t k, uint32_t c);
void     orc_rank2(const orc_fm_index* f, uint32_t l, uint32_t r, uint32_t c, uint32_t out[2]);
void     orc_rank4(const orc_fm_index* f, uint32_t k, uint32_t out[4]);

/* match / match_reverse (fmindex_inl.h:181-278).  pattern: one symbol per byte, >3 = N.
 * If blocks != NULL, *blocks receives the number of distinct 32-byte bwt_occ records the
 * search touched (the algorithmic-bytes unit of the seed pass). */
void     orc_match(const orc_fm_index* f, const uint8_t* pattern, uint32_t len, int reverse,
                   uint32_t out[2], uint32_t* blocks);

/* locate family (fmindex_inl.h:286-460) */
uint32_t orc_basic_inv_psi(const orc_fm_index* f, uint32_t i);
void     orc_locate_ssa(const orc_fm_index* f, uint32_t i, uint32_t out[2]);   /* (row j, steps t) */
uint32_t orc_lookup_ssa(const orc_fm_index* f, const uint32_t jt[2]);
uint32_t orc_locate(const orc_fm_index* f, uint32_t i);

/* batched forms: OpenMP parallel for over work items, the shape of the reference's host
 * path (nvbio-test/fmindex_test.cu:375-407,879-890; nvbio/fmindex/filter_inl.h:193-252) */
void orc_match_batch(const orc_fm_index* f, const uint8_t* syms, const uint32_t* offsets, uint32_t n,
                     int reverse, uint32_t* ranges, uint32_t* blocks);
void orc_locate_batch(const orc_fm_index* f, const uint32_t* rows, uint32_t n, uint32_t* pos);

/* FMIndexFilter<host_tag> (nvbio/fmindex/filter_inl.h:193-252): ranges + inclusive scan of sizes,
 * then hits (text_pos, query_id) for global hit indices [begin,end) */
uint64_t orc_filter_rank(const orc_fm_index* f, const uint8_t* syms, const uint32_t* offsets, uint32_t n,
                         uint32_t* ranges, uint64_t* slots);
void     orc_filter_locate(const orc_fm_index* f, const uint32_t* ranges, const uint64_t* slots, uint32_t n,
                           uint64_t begin, uint64_t end, uint32_t* hits /* 2 per hit */);

/* --- Gotoh scoring --- */
enum { ORC_GLOBAL = 0, ORC_LOCAL = 1, ORC_SEMI_GLOBAL = 2 };     /* aln::AlignmentType, alignment.h:242 */

/* scoring scheme: the Gotoh scheme concept (alignment.h:437-449) with either a constant
 * mismatch (SimpleGotohScheme, nvbio/alignment/utils.h:103-123: mm_min == mm_max == -mismatch,
 * all four gap costs from gap_open/gap_ext) or nvBowtie's quality ramp
 * (nvBowtie/bowtie2/cuda/scoring.h:73-92,278-285):
 *     mismatch(q) = -( mm_min + int( float(min(q,40))/40.0f * (mm_max - mm_min) ) ). */
typedef struct
{
    int32_t match;          /* score of a match (constant)                       */
    int32_t mm_min, mm_max; /* mismatch PENALTIES (positive) at quality 0 / >=40 */
    int32_t pat_gap_open, pat_gap_ext;  /* signed scores (negative)              */
    int32_t txt_gap_open, txt_gap_ext;
} orc_gotoh_scheme;

int32_t orc_mismatch(const orc_gotoh_scheme* s, uint32_t q);

/* Field_traits<int32>::min() = -(1<<30), the BestSink initial score (nvbio/basic/numbers.h:738-742) */
#define ORC_SCORE_MIN (-(1 << 30))

/* banded Gotoh, any odd band (gotoh_banded_inl.h:397-646, entry :668-688).
 * pattern/text one symbol per byte; quals may be NULL (trivial_quality_string -> 0).
 * Returns 0 if text_len < pattern_len (nothing reported), else 1.
 * score/sink follow BestSink<int32> (sink_inl.h:31-49): init (ORC_SCORE_MIN, (-1,-1)). */
int orc_banded_gotoh(uint32_t band, int type, const orc_gotoh_scheme* s,
                     const uint8_t* pat, const uint8_t* quals, uint32_t M,
                     const uint8_t* txt, uint32_t N,
                     int32_t* score, uint32_t sink[2]);

/* the staged scheduler's windowed scoring (32-row windows, min_score early exit); 1 = ran to the end, 0 = stopped early */
int orc_banded_gotoh_staged(uint32_t band, int type, const orc_gotoh_scheme* s,
                            const uint8_t* pat, const uint8_t* quals, uint32_t M,
                            const uint8_t* txt, uint32_t N, int32_t min_score,
                            int32_t* score, uint32_t sink[2]);

/* banded Gotoh traceback: aln::banded_alignment_traceback (nvbio/alignment/banded_inl.h:354-417,
 * gotoh/gotoh_banded_inl.h:872-948) delivered to nvBowtie's run-length Backtracker
 * (nvBowtie/bowtie2/cuda/alignment_utils.h:115-157).  cigar: io::Cigar elements (type bits 0-1:
 * 0 M, 1 I, 2 D, 3 soft clip; length bits 2-15) in BACKTRACKING order: [clip(M - sink.y)] ops
 * [clip(source.y)].  ops (optional): one byte per step in the same order.  Returns 1 if an
 * alignment was traced, 0 if nothing was reported (source = sink = (-1,-1), no cigar). */
int orc_banded_gotoh_traceback(uint32_t band, int type, const orc_gotoh_scheme* s,
                               const uint8_t* pat, const uint8_t* quals, uint32_t M,
                               const uint8_t* txt, uint32_t N,
                               int32_t* score, uint32_t source[2], uint32_t sink[2],
                               uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len,
                               uint8_t* ops, uint32_t ops_cap, uint32_t* n_ops);

/* the same over nvBowtie-shaped packed inputs (see orc_banded_gotoh_packed_batch); cigars has
 * cigar_stride elements per alignment */
void orc_banded_gotoh_traceback_packed_batch(uint32_t band, int type, const orc_gotoh_scheme* s,
                                   const uint32_t* reads4, const uint32_t* read_offsets, const uint8_t* quals,
                                   const uint32_t* read_id, const uint8_t* flags,
                                   const uint32_t* genome2, const uint32_t* win_begin, const uint32_t* win_end,
                                   uint32_t n, int32_t* scores, uint32_t* sources, uint32_t* sinks,
                                   uint16_t* cigars, uint32_t cigar_stride, uint32_t* cigar_lens);

/* nvBowtie finish_alignment (traceback_inl.h:536-705): edit distance and MDS byte stream of a traced alignment;
 * pat = the read as aligned, txt = the text window, cigar in backtracking order, cigar_offset = source.x.
 * mds needs mds_cap >= the stream's length for exact output (match runs are extended in place). */
void orc_finish_alignment(const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                          const uint16_t* cigar, uint32_t cigar_len, uint32_t cigar_offset,
                          uint32_t* ed, uint8_t* mds, uint32_t mds_cap, uint32_t* mds_len);

/* full-matrix Gotoh traceback: aln::alignment_traceback (nvbio/alignment/alignment_inl.h:355-455, walk
 * gotoh/gotoh_inl.h:1573-1640) through nvBowtie's run-length Backtracker; x = text, y = pattern; cigar as in
 * orc_banded_gotoh_traceback.  Returns 1 if an alignment was traced, 0 if nothing was reported. */
int orc_full_gotoh_traceback(int type, const orc_gotoh_scheme* s,
                             const uint8_t* pat, const uint8_t* quals, uint32_t M,
                             const uint8_t* txt, uint32_t N, int32_t min_score,
                             int32_t* score, uint32_t source[2], uint32_t sink[2],
                             uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len);

/* full-matrix Gotoh, 8-column stripes with an int16 (short2) boundary column
 * (gotoh_inl.h:444-841 pattern blocking, :847-1256 text blocking; alignment_score_dispatch :1283-1330).
 * blocking: 0 = PatternBlockingTag (alignment_score default), 1 = TextBlockingTag (sw-benchmark).
 * Returns 0 when the stripe early-exit fires (max_score + missing*match < min_score), else 1. */
/* Best2Sink<int32>( distinct_dist ) in place of BestSink: out = { score1, sink1.x, sink1.y, score2, sink2.x, sink2.y } */
int orc_banded_gotoh_best2(uint32_t band, int type, const orc_gotoh_scheme* s, const uint8_t* pat, const uint8_t* quals, uint32_t M,
                           const uint8_t* txt, uint32_t N, uint32_t distinct_dist, int64_t out[6]);
int orc_full_gotoh_best2(int type, int blocking, const orc_gotoh_scheme* s, const uint8_t* pat, const uint8_t* quals, uint32_t M,
                         const uint8_t* txt, uint32_t N, int32_t min_score, uint32_t distinct_dist, int64_t out[6]);
/* linear-gap Smith-Waterman / edit-distance aligners (sw/sw_banded_inl.h, sw/sw_inl.h); sw = {match, mismatch, deletion, insertion} */
int orc_banded_sw(uint32_t band, int type, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                  int32_t* score, uint32_t sink[2]);
int orc_banded_sw_traceback(uint32_t band, int type, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                            int32_t* score, uint32_t source[2], uint32_t sink[2], uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len);
/* full-matrix traceback of the linear-gap Smith-Waterman aligner (pattern blocking); outputs as orc_full_gotoh_traceback */
int orc_full_sw_traceback(int type, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                          int32_t* score, uint32_t source[2], uint32_t sink[2], uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len);
int orc_full_sw(int type, int blocking, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                int32_t min_score, int32_t* score, uint32_t sink[2]);
int orc_full_gotoh(int type, int blocking, const orc_gotoh_scheme* s,
                   const uint8_t* pat, const uint8_t* quals, uint32_t M,
                   const uint8_t* txt, uint32_t N, int32_t min_score,
                   int32_t* score, uint32_t sink[2]);

/* batched forms (OpenMP over work items: batched_banded_inl.h:113-121, batched_inl.h:283-306) */
void orc_banded_gotoh_batch(uint32_t band, int type, const orc_gotoh_scheme* s,
                            const uint8_t* pats, const uint8_t* quals, const uint32_t* pat_off,
                            const uint8_t* txts, const uint32_t* txt_off, uint32_t n,
                            int32_t* scores, uint32_t* sinks);
void orc_full_gotoh_batch(int type, int blocking, const orc_gotoh_scheme* s,
                          const uint8_t* pats, const uint8_t* quals, const uint32_t* pat_off,
                          const uint8_t* txts, const uint32_t* txt_off, uint32_t n,
                          int32_t min_score, int32_t* scores, uint32_t* sinks);

/* nvBowtie-shaped batched extension over packed inputs (the CPU baseline of the extend pass):
 * work item i aligns read reads[read_id[i]] (4-bit packed, offsets in symbols; flags bit0 = reverse
 * the read, bit1 = complement it: alignment_utils.h:277-302) against genome[win_begin[i], win_end[i])
 * (2-bit packed) with band `band`. */
void orc_banded_gotoh_packed_batch(uint32_t band, int type, const orc_gotoh_scheme* s,
                                   const uint32_t* reads4, const uint32_t* read_offsets, const uint8_t* quals,
                                   const uint32_t* read_id, const uint8_t* flags,
                                   const uint32_t* genome2, const uint32_t* win_begin, const uint32_t* win_end,
                                   uint32_t n, int32_t* scores, uint32_t* sinks);

void orc_full_gotoh_many_to_one(int type, int blocking, const orc_gotoh_scheme* s, const uint8_t* pats, const uint8_t* quals, const uint32_t* po,
                                const uint8_t* text, uint32_t text_len, uint32_t n, int32_t min_score, int32_t* scores, uint32_t* sinks);

int orc_num_threads(void);
void orc_set_num_threads(int n);

/* hamming_backtrack (nvbio/fmindex/backtrack.h) with a counting delegate; see nvbio_oracle.c */
uint32_t orc_hamming_backtrack(const orc_fm_index* f, const uint8_t* stream, uint32_t begin, uint32_t len, uint32_t seed, uint32_t mismatches,
                               int quirks, uint32_t* count, uint32_t* ranges, uint32_t cap);
/* nvBowtie's best / second-best reduction for one read (reduce_inl.h:65-140) and Bowtie2's mapping quality (mapq.h:32-297) */
void orc_score_reduce(const int32_t* scores, const uint32_t* pos, const uint8_t* rc, uint32_t n, uint32_t read_len, int32_t worst_score,
                      int64_t out[8]);
int  orc_mapq(int version, int monotone, int32_t perfect_score, int32_t min_score, int32_t best_score, int has_second, int32_t second_score);

/* nvBowtie's seed-hit deques (seed_hit.h, seed_hit_deque_array.h, nvbio/basic/priority_deque.h + interval_heap.h), exact seed mapper
 * bookkeeping (mapping_inl.h:193-282,485-556), select_kernel (select_inl.h:62-130) and the effort-limited score_reduce
 * (reduce_inl.h:65-140 with ReduceBestApproxContext, reduce.h:55-99); see nvbio_oracle.c */
typedef struct { uint32_t begin; uint32_t bits; } orc_seed_hit;     /* bits: range_delta:20 | pos:10 | rc:1 | indexdir:1 */
void     orc_hit_deque_push(orc_seed_hit* a, uint32_t* n, orc_seed_hit x);
void     orc_hit_deque_pop_bottom(orc_seed_hit* a, uint32_t* n);
void     orc_hit_deque_pop_top(orc_seed_hit* a, uint32_t* n);
uint32_t orc_hit_deque_top(uint32_t n);
int      orc_map_exact_read(const uint32_t* fw, const uint32_t* rc, const uint32_t* seed_off, uint32_t n_seeds, uint32_t read_len, uint32_t seed_len,
                            uint32_t max_hits, uint32_t rep_seeds, orc_seed_hit* deque, uint32_t* deque_size);
int      orc_map_approx_read(const orc_fm_index* f, const orc_fm_index* rf, const uint8_t* stored, uint32_t read_len, const uint32_t* seed_off, uint32_t n_seeds,
                             uint32_t seed_len, uint32_t max_hits, uint32_t rep_seeds, orc_seed_hit* deque, uint32_t* deque_size);
int      orc_select_read(orc_seed_hit* deque, uint32_t* size, uint32_t* top_flag, uint32_t* sa_pos, uint32_t* packed_seed);
void     orc_hit_deque_run(const uint32_t* ops, const uint32_t* begins, const uint32_t* bits, uint32_t n_ops, uint32_t max_hits,
                           uint32_t* heap_out, uint32_t* size_out, uint32_t* out_rows);
int      orc_score_reduce_effort(int64_t best[6], uint32_t* trys, int32_t score, uint32_t g_pos, uint32_t read_rc, uint32_t top_flag, uint32_t read_len,
                                 uint32_t ext, uint32_t max_effort, uint32_t min_ext, uint32_t max_ext);

/* the Myers bit-vector aligner as banded_alignment_score<BAND> runs it (myers/myers_banded_inl.h:247-342); score = -(edit distance) */
int      orc_banded_myers(uint32_t band, int type, const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                          int32_t* score, uint32_t sink[2]);
/* the generic rank dictionary (rank_dictionary_inl.h:33-66,206-336,482-539): plain 32- / 64-bit words, separate occ table, any K,
 * 32- or 64-bit indices */
void     orc_rank_generic_build(const void* text, uint32_t word_bits, uint64_t length, uint32_t K, uint32_t index_bits, void* occ, uint64_t cnt[4]);
uint64_t orc_rank_generic(const void* text, uint32_t word_bits, const void* occ, uint32_t index_bits, uint32_t K, uint64_t i, uint32_t c);

#ifdef __cplusplus
}
#endif
#endif
