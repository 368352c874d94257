/* oracle/nvbio_oracle.c -- TEST INFRASTRUCTURE ONLY (see nvbio_oracle.h).
 *
 * A CPU restatement, in plain C, of what the reference computes on the seed-and-extend
 * hot path.  It follows the reference function by function (citations below, relative
 * to /root/reference) so that integer results are bit-identical; it does not share any
 * code with the HIP implementation in nvbio-gpl_amd/csrc (different data flow: the GPU
 * path works on packed words, k-mer tables and register bands).
 */
#include "nvbio_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* the OpenMP team size of the batch loops (bench.py sets it to the CPUs the process is really granted) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads( n );
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * packed streams: PackedStream<..,2,true> / <..,4,true>  (nvbio/basic/packedstream_inl.h:33-75)
 * symbol i of a 2-bit stream lives at bits [30-2(i&15), 31-2(i&15)] of word i>>4
 * ------------------------------------------------------------------------------------------ */
uint8_t orc_get2(const uint32_t* w, uint64_t i) { return (uint8_t)((w[i >> 4] >> (30u - 2u*(uint32_t)(i & 15u))) & 3u); }
uint8_t orc_get4(const uint32_t* w, uint64_t i) { return (uint8_t)((w[i >> 3] >> (28u - 4u*(uint32_t)(i & 7u))) & 15u); }

void orc_pack2(const uint8_t* s, uint64_t n, uint32_t* w)
{
    for (uint64_t i = 0; i < n; ++i)
    {
        const uint32_t sh = 30u - 2u*(uint32_t)(i & 15u);
        w[i >> 4] = (w[i >> 4] & ~(3u << sh)) | ((uint32_t)(s[i] & 3u) << sh);
    }
}
void orc_pack4(const uint8_t* s, uint64_t n, uint32_t* w)
{
    for (uint64_t i = 0; i < n; ++i)
    {
        const uint32_t sh = 28u - 4u*(uint32_t)(i & 7u);
        w[i >> 3] = (w[i >> 3] & ~(15u << sh)) | ((uint32_t)(s[i] & 15u) << sh);
    }
}

/* ------------------------------------------------------------------------------------------
 * popc_2bit (nvbio/basic/popcount_inl.h:230-236) and the "all but the first i symbols"
 * variant with its c == 0 correction (:318-341)
 * ------------------------------------------------------------------------------------------ */
uint32_t orc_popc_2bit(uint32_t x, uint32_t c)
{
    const uint32_t odd  = ((c & 2u) ? x : ~x) >> 1;
    const uint32_t even = ((c & 1u) ? x : ~x);
    return (uint32_t)__builtin_popcount( odd & even & 0x55555555u );
}
uint32_t orc_popc_2bit_hi(uint32_t mask, uint32_t c, uint32_t i)
{
    const uint32_t r = orc_popc_2bit( mask & ~((1u << (i << 1)) - 1u), c );
    return (c == 0) ? r - i : r;
}

uint32_t orc_bwt_words(uint32_t n) { return (((n + 15u) / 16u) + 3u) & ~3u; }

/* ------------------------------------------------------------------------------------------
 * suffix sorting (test-index construction only).  Convention of gen_sa (bwt.h:28-37).
 * ------------------------------------------------------------------------------------------ */
static const uint8_t* g_sort_text;
static uint32_t       g_sort_n;
#pragma omp threadprivate(g_sort_text, g_sort_n)

static int suffix_cmp_from(uint32_t a, uint32_t b, uint32_t skip)
{
    const uint8_t* t = g_sort_text;
    const uint32_t n = g_sort_n;
    uint32_t i = a + skip, j = b + skip;
    while (i < n && j < n)
    {
        if (t[i] != t[j]) return t[i] < t[j] ? -1 : 1;
        ++i; ++j;
    }
    /* the shorter suffix (the one that ran out first) is the smaller one */
    if (i >= n && j >= n) return (a > b) ? -1 : (a < b ? 1 : 0);
    return (i >= n) ? -1 : 1;
}
static int suffix_cmp16(const void* pa, const void* pb)
{
    const uint32_t a = *(const uint32_t*)pa, b = *(const uint32_t*)pb;
    /* keys are equal on the first 16 (padded) symbols: compare in full to honour ends */
    return suffix_cmp_from( a, b, 0 );
}

void orc_suffix_sort(const uint8_t* text, uint32_t n, uint32_t* sa)
{
    sa[0] = n;
    if (n == 0) return;

    /* 1. 16-mer key (32 bits, zero padded past the end) for every suffix */
    uint64_t* kv  = (uint64_t*)malloc( sizeof(uint64_t) * (size_t)n );
    uint64_t* tmp = (uint64_t*)malloc( sizeof(uint64_t) * (size_t)n );
    {
        uint32_t key = 0;
        /* rolling: key(i) = symbols i..i+15 */
        for (uint32_t j = 0; j < 16; ++j)
            key = (key << 2) | (j < n ? (uint32_t)(text[j] & 3u) : 0u);
        for (uint32_t i = 0; i < n; ++i)
        {
            kv[i] = ((uint64_t)key << 32) | i;
            const uint64_t nx = (uint64_t)i + 16u;
            key = (key << 2) | (nx < n ? (uint32_t)(text[nx] & 3u) : 0u);
        }
    }
    /* 2. LSD radix sort on the key (2 passes of 16 bits) */
    for (int pass = 0; pass < 2; ++pass)
    {
        const int shift = 32 + 16*pass;
        size_t* cnt = (size_t*)calloc( 65537, sizeof(size_t) );
        for (uint32_t i = 0; i < n; ++i) cnt[ ((kv[i] >> shift) & 0xFFFFu) + 1 ]++;
        for (uint32_t b = 0; b < 65536; ++b) cnt[b+1] += cnt[b];
        for (uint32_t i = 0; i < n; ++i) tmp[ cnt[ (kv[i] >> shift) & 0xFFFFu ]++ ] = kv[i];
        free( cnt );
        uint64_t* sw = kv; kv = tmp; tmp = sw;
    }
    uint32_t* out = sa + 1;
    for (uint32_t i = 0; i < n; ++i) out[i] = (uint32_t)(kv[i] & 0xFFFFFFFFu);

    /* 3. resolve groups of equal keys by full comparison */
    #pragma omp parallel
    {
        g_sort_text = text; g_sort_n = n;
        #pragma omp for schedule(dynamic, 4096)
        for (uint32_t i = 0; i < n; ++i)
        {
            if (i > 0 && (kv[i] >> 32) == (kv[i-1] >> 32)) continue;   /* not a group head */
            uint32_t e = i + 1;
            while (e < n && (kv[e] >> 32) == (kv[i] >> 32)) ++e;
            if (e - i > 1)
                qsort( out + i, e - i, sizeof(uint32_t), suffix_cmp16 );
        }
    }
    free( kv ); free( tmp );
}

/* ------------------------------------------------------------------------------------------
 * index construction
 * ------------------------------------------------------------------------------------------ */
uint32_t orc_fm_build(const uint8_t* text, uint32_t n, const uint32_t* sa,
                      uint32_t* bwt_occ, uint32_t* ssa, uint32_t L2[5])
{
    const uint32_t words = orc_bwt_words( n );
    uint8_t*  bwt = (uint8_t*)calloc( (size_t)n + 2u, 1 );
    uint32_t* bw  = (uint32_t*)calloc( words, sizeof(uint32_t) );
    uint32_t* occ = (uint32_t*)calloc( words, sizeof(uint32_t) );

    /* gen_bwt_from_sa (bwt.h:41-53): bwt[i] = T[SA[i]-1], the row with SA[i]==0 is primary
     * and is then squeezed out of the string */
    uint32_t primary = 0;
    for (uint32_t i = 0; i <= n; ++i)
    {
        if (sa[i] == 0) primary = i;
        else            bwt[i] = text[ sa[i] - 1 ] & 3u;
    }
    for (uint32_t i = primary; i < n; ++i) bwt[i] = bwt[i+1];
    orc_pack2( bwt, n, bw );

    /* build_occurrence_table<64> (rank_dictionary_inl.h:33-66) */
    uint32_t counters[4] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < n; ++i)
    {
        if ((i & 63u) == 0)
            for (uint32_t c = 0; c < 4; ++c) occ[ (i >> 6)*4 + c ] = counters[c];
        ++counters[ bwt[i] ];
    }
    L2[0] = 0;
    for (uint32_t c = 0; c < 4; ++c) L2[c+1] = L2[c] + counters[c];

    /* interleave (fmindex_impl.cu:300-313): record k = {bwt[4k..4k+3], occ[4k..4k+3]} */
    for (uint32_t w = 0; w < words; w += 4)
        for (uint32_t k = 0; k < 4; ++k)
        {
            bwt_occ[ w*2 + k     ] = bw[ w + k ];
            bwt_occ[ w*2 + 4 + k ] = occ[ w + k ];
        }

    /* SSA_index_multiple<16> (ssa_inl.h:254-301): ssa[i/16] = SA[i] for i % 16 == 0, ssa[0] = -1 */
    for (uint32_t i = 0; i <= n; i += 16)
    {
        ssa[i >> 4] = (i == 0) ? 0xFFFFFFFFu : sa[i];
        if (i > 0xFFFFFFFFu - 16u) break;
    }
    free( bwt ); free( bw ); free( occ );
    return primary;
}

/* ------------------------------------------------------------------------------------------
 * rank dictionary, production dispatch (rank_dictionary_inl.h:338-479):
 * record k: words [8k..8k+3] = BWT symbols [64k,64k+64), words [8k+4..8k+7] = occ{A,C,G,T}
 * ------------------------------------------------------------------------------------------ */
static inline uint32_t dict_popc(const uint32_t* rec, uint32_t i, uint32_t k, uint32_t c)
{
    /* popc(): :351-369 */
    const uint32_t m    = (i - k*64u) >> 4;
    const uint32_t i_16 = ~i & 15u;
    uint32_t x = 0;
    if (m > 0) x += orc_popc_2bit( rec[0], c );
    if (m > 1) x += orc_popc_2bit( rec[1], c );
    if (m > 2) x += orc_popc_2bit( rec[2], c );
    return x + orc_popc_2bit_hi( rec[m], c, i_16 );
}

static inline uint32_t dict_run1(const orc_fm_index* f, uint32_t i, uint32_t c, uint32_t* blocks)
{
    /* run(dict,i,c): :412-423 */
    if (i == 0xFFFFFFFFu) return 0u;
    const uint32_t  k   = i >> 6;
    const uint32_t* rec = f->bwt_occ + (size_t)k*8u;
    if (blocks) *blocks += 1;
    return rec[4 + c] + dict_popc( rec, i, k, c );
}

static inline void dict_run2(const orc_fm_index* f, uint32_t l, uint32_t r, uint32_t c, uint32_t out[2], uint32_t* blocks)
{
    /* run(dict,range,c): :425-448 */
    if (l == 0xFFFFFFFFu && r == 0xFFFFFFFFu) { out[0] = out[1] = 0; return; }
    if (l == 0xFFFFFFFFu || l == r)
    {
        const uint32_t v = dict_run1( f, r, c, blocks );
        out[0] = (l == 0xFFFFFFFFu) ? 0u : v; out[1] = v;
        return;
    }
    const uint32_t kl = l >> 6, kh = r >> 6;
    const uint32_t* rl = f->bwt_occ + (size_t)kl*8u;
    const uint32_t* rh = f->bwt_occ + (size_t)kh*8u;
    if (blocks) *blocks += (kl == kh) ? 1 : 2;

    const uint32_t outl = rl[4 + c];
    const uint32_t outh = (kl == kh) ? outl : rh[4 + c];

    /* popc2(): :375-409 */
    const uint32_t ml = (l - kl*64u) >> 4, mh = (r - kh*64u) >> 4;
    const uint32_t l_16 = ~l & 15u,        h_16 = ~r & 15u;
    uint32_t xl = 0;
    if (ml > 0) xl += orc_popc_2bit( rl[0], c );
    if (ml > 1) xl += orc_popc_2bit( rl[1], c );
    if (ml > 2) xl += orc_popc_2bit( rl[2], c );
    uint32_t       xh     = (kl == kh) ? xl : 0u;
    const uint32_t startm = (kl == kh) ? ml : 0u;
    if (mh > 0 && startm == 0) xh += orc_popc_2bit( rh[0], c );
    if (mh > 1 && startm <= 1) xh += orc_popc_2bit( rh[1], c );
    if (mh > 2 && startm <= 2) xh += orc_popc_2bit( rh[2], c );
    xl += orc_popc_2bit_hi( rl[ml], c, l_16 );
    xh += orc_popc_2bit_hi( rh[mh], c, h_16 );
    out[0] = outl + xl; out[1] = outh + xh;
}

uint32_t orc_dict_rank(const orc_fm_index* f, uint32_t k, uint32_t c) { return dict_run1( f, k, c, 0 ); }
void     orc_dict_rank2(const orc_fm_index* f, uint32_t l, uint32_t r, uint32_t c, uint32_t out[2]) { dict_run2( f, l, r, c, out, 0 ); }
void     orc_dict_rank4(const orc_fm_index* f, uint32_t i, uint32_t out[4])
{
    /* run4(): :450-462 computes the four counts through the byte count-table (bwt.h:78-89);
     * the table sums are per-symbol popcounts, restated here symbol by symbol */
    const uint32_t  k   = i >> 6;
    const uint32_t* rec = f->bwt_occ + (size_t)k*8u;
    for (uint32_t c = 0; c < 4; ++c)
        out[c] = rec[4 + c] + dict_popc( rec, i, k, c );
}

/* ------------------------------------------------------------------------------------------
 * fm_index level rank (fmindex_inl.h:27-173)
 * ------------------------------------------------------------------------------------------ */
static inline uint32_t fm_count(const orc_fm_index* f, uint32_t c) { return f->L2[c+1] - f->L2[c]; }

static inline uint32_t fm_rank1(const orc_fm_index* f, uint32_t k, uint32_t c, uint32_t* blocks)
{
    if (k == 0xFFFFFFFFu) return 0;
    if (k == f->length)   return fm_count( f, c );
    if (k >= f->primary)  --k;                          /* because $ is not in the bwt */
    return dict_run1( f, k, c, blocks );
}
static inline void fm_rank2(const orc_fm_index* f, uint32_t l, uint32_t r, uint32_t c, uint32_t out[2], uint32_t* blocks)
{
    if (l == r)            { out[0] = out[1] = fm_rank1( f, l, c, blocks ); return; }
    else if (l == 0xFFFFFFFFu) { out[0] = 0; out[1] = fm_rank1( f, r, c, blocks ); return; }
    if (r == f->length)    { out[0] = fm_rank1( f, l, c, blocks ); out[1] = fm_count( f, c ); return; }
    if (l >= f->primary) --l;
    if (r >= f->primary) --r;
    dict_run2( f, l, r, c, out, blocks );
}
uint32_t orc_rank(const orc_fm_index* f, uint32_t k, uint32_t c) { return fm_rank1( f, k, c, 0 ); }
void     orc_rank2(const orc_fm_index* f, uint32_t l, uint32_t r, uint32_t c, uint32_t out[2]) { fm_rank2( f, l, r, c, out, 0 ); }
void     orc_rank4(const orc_fm_index* f, uint32_t k, uint32_t out[4])
{
    if (k == 0xFFFFFFFFu) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    if (k == f->length)   { for (uint32_t c = 0; c < 4; ++c) out[c] = fm_count( f, c ); return; }
    if (k >= f->primary) --k;
    orc_dict_rank4( f, k, out );
}

/* ------------------------------------------------------------------------------------------
 * match / match_reverse (fmindex_inl.h:181-278)
 * ------------------------------------------------------------------------------------------ */
void orc_match(const orc_fm_index* f, const uint8_t* p, uint32_t len, int reverse, uint32_t out[2], uint32_t* blocks)
{
    uint32_t x = 0, y = f->length;
    if (blocks) *blocks = 0;
    for (uint32_t s = 0; s < len && x <= y; ++s)
    {
        const uint8_t c = reverse ? p[s] : p[len - 1u - s];
        if (c > 3) { out[0] = 1; out[1] = 0; return; }       /* an N: no match */
        uint32_t r[2];
        fm_rank2( f, x - 1u, y, c, r, blocks );
        x = f->L2[c] + r[0] + 1u;
        y = f->L2[c] + r[1];
    }
    out[0] = x; out[1] = y;
}

/* ------------------------------------------------------------------------------------------
 * locate family (fmindex_inl.h:286-460); SSA_index_multiple_context<16> (ssa_inl.h:477-495)
 * ------------------------------------------------------------------------------------------ */
static inline uint8_t fm_bwt(const orc_fm_index* f, uint32_t k)
{
    return (uint8_t)((f->bwt_occ[ (size_t)(k >> 6)*8u + ((k & 63u) >> 4) ] >> (30u - 2u*(k & 15u))) & 3u);
}
uint32_t orc_basic_inv_psi(const orc_fm_index* f, uint32_t i)
{
    if (i == f->primary) return 0;
    const uint32_t k = i < f->primary ? i : i - 1u;
    const uint8_t  c = fm_bwt( f, k );
    return f->L2[c] + dict_run1( f, k, c, 0 );
}
void orc_locate_ssa(const orc_fm_index* f, uint32_t i, uint32_t out[2])
{
    uint32_t j = i, t = 0;
    while (j & 15u)
    {
        if (j != f->primary)
        {
            const uint8_t c = j < f->primary ? fm_bwt( f, j ) : fm_bwt( f, j - 1u );
            j = f->L2[c] + fm_rank1( f, j, c, 0 );
        }
        else
            j = 0;
        ++t;
    }
    out[0] = j; out[1] = t;
}
uint32_t orc_lookup_ssa(const orc_fm_index* f, const uint32_t jt[2]) { return f->ssa[ jt[0] >> 4 ] + jt[1]; }
uint32_t orc_locate(const orc_fm_index* f, uint32_t i)
{
    uint32_t jt[2];
    orc_locate_ssa( f, i, jt );
    return orc_lookup_ssa( f, jt );
}

void orc_match_batch(const orc_fm_index* f, const uint8_t* syms, const uint32_t* off, uint32_t n,
                     int reverse, uint32_t* ranges, uint32_t* blocks)
{
    /* small batches stay on the calling thread: a 128-thread team costs far more than a handful of searches (per-read callers) */
    #pragma omp parallel for schedule(static) if(n > 512)
    for (int64_t q = 0; q < (int64_t)n; ++q)
        orc_match( f, syms + off[q], off[q+1] - off[q], reverse, ranges + 2*q, blocks ? blocks + q : 0 );
}
void orc_locate_batch(const orc_fm_index* f, const uint32_t* rows, uint32_t n, uint32_t* pos)
{
    #pragma omp parallel for schedule(static) if(n > 512)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        pos[i] = orc_locate( f, rows[i] );
}

/* FMIndexFilter<host_tag>::rank / ::locate (filter_inl.h:193-252; functors :26-118) */
uint64_t orc_filter_rank(const orc_fm_index* f, const uint8_t* syms, const uint32_t* off, uint32_t n,
                         uint32_t* ranges, uint64_t* slots)
{
    orc_match_batch( f, syms, off, n, 0, ranges, 0 );
    uint64_t sum = 0;
    for (uint32_t q = 0; q < n; ++q)
    {
        sum += (uint64_t)(uint32_t)(1u + ranges[2*q+1] - ranges[2*q]);    /* range_size: uint32 arithmetic */
        slots[q] = sum;
    }
    return n ? slots[n-1] : 0;
}
void orc_filter_locate(const orc_fm_index* f, const uint32_t* ranges, const uint64_t* slots, uint32_t n,
                       uint64_t begin, uint64_t end, uint32_t* hits)
{
    #pragma omp parallel for schedule(static)
    for (int64_t h = (int64_t)begin; h < (int64_t)end; ++h)
    {
        /* upper_bound( h, slots, n ) */
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = lo + (hi - lo)/2; if (slots[mid] <= (uint64_t)h) lo = mid + 1; else hi = mid; }
        const uint32_t slot  = lo;
        const uint64_t base  = slot ? slots[slot-1] : 0u;
        const uint32_t local = (uint32_t)((uint64_t)h - base);
        hits[ 2*(h - (int64_t)begin)     ] = orc_locate( f, ranges[2*slot] + local );
        hits[ 2*(h - (int64_t)begin) + 1 ] = slot;
    }
}

/* ------------------------------------------------------------------------------------------
 * scoring scheme
 * ------------------------------------------------------------------------------------------ */
int32_t orc_mismatch(const orc_gotoh_scheme* s, uint32_t q)
{
    /* QualCost::operator() (nvBowtie/bowtie2/cuda/scoring.h:84-88), negated by
     * SmithWatermanScoringScheme::mismatch (:280-281) */
    const int   qi   = (int)q < 40 ? (int)q : 40;
    const float frac = (float)(qi / 40.0f);
    return -( s->mm_min + (int)( frac * (float)(s->mm_max - s->mm_min) ) );
}

static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }

/* BestSink<int32> and, with two != 0, Best2Sink<int32> (sink.h:60-116, sink_inl.h:31-83): the second alignment must end more
 * than `dist` text positions away from the first (uint32 arithmetic as the reference's, wrap-around included); a new best does
 * not demote the old one */
typedef struct { int32_t score; uint32_t x, y; int two; uint32_t dist; int32_t score2; uint32_t x2, y2; } best_sink;
static inline void sink_init(best_sink* s)
{
    s->score = s->score2 = ORC_SCORE_MIN; s->x = s->y = s->x2 = s->y2 = 0xFFFFFFFFu; s->two = 0; s->dist = 0;
}
static inline void sink_report(best_sink* s, int32_t score, uint32_t x, uint32_t y)
{
    if (s->score <= score) { s->score = score; s->x = x; s->y = y; }     /* last maximum wins (sink_inl.h:40-49) */
    else if (s->two && s->score2 <= score && ((uint32_t)(x + s->dist) < s->x || x > (uint32_t)(s->x + s->dist)))
    { s->score2 = score; s->x2 = x; s->y2 = y; }                         /* sink_inl.h:69-83 */
}

/* ------------------------------------------------------------------------------------------
 * banded Gotoh (gotoh_banded_inl.h:397-646)
 * ------------------------------------------------------------------------------------------ */
#define ORC_MAX_BAND 64

/* the banded DP; dirs (optional, M*B bytes) receives every cell's direction vector
 * hdir | edir | fdir as GotohSubmatrixContext::new_cell stores it (gotoh_banded_inl.h:316-330) */
static int banded_core_w(uint32_t B, int type, const orc_gotoh_scheme* sc,
                         const uint8_t* pat, const uint8_t* quals, uint32_t M,
                         const uint8_t* txt, uint32_t N, best_sink* sink_p, uint8_t* dirs,
                         uint32_t window, int32_t min_score)
{
    best_sink sink = *sink_p;
    /* Reference_cache<BAND> (alignment_base_inl.h:66-90): bands 3,5,7,15 cache whole uint32
     * symbols, every other band (31 in production) a 2-bit packed stream, so a cached symbol is
     * re-read modulo 4 (the out-of-range sentinel 255 becomes 3) */
    const int packed_cache = !(B == 3 || B == 5 || B == 7 || B == 15);

    uint32_t cache[ORC_MAX_BAND];
    for (uint32_t j = 0; j + 1 < B; ++j)
    {
        /* :432-433 reads text[j] unconditionally (out of bounds when N < BAND-1: undefined in
         * the reference); the sentinel is used here so that the result is defined */
        const uint32_t g = j < N ? txt[j] : 255u;
        cache[j] = packed_cache ? (g & 3u) : g;
    }

    const int32_t G_o = sc->pat_gap_open, G_e = sc->pat_gap_ext;
    const int32_t infimum = -32768 - imax( imax( G_o, G_e ), imax( sc->txt_gap_open, sc->txt_gap_ext ) );   /* :437-439 */

    int32_t H[ORC_MAX_BAND], F[ORC_MAX_BAND];
    H[0] = 0;                                                               /* init_row_zero :37-68 */
    for (uint32_t j = 1; j < B; ++j)
        H[j] = (type == ORC_GLOBAL) ? sc->txt_gap_open + (int32_t)(j-1)*sc->txt_gap_ext : 0;
    for (uint32_t j = 0; j < B; ++j) F[j] = infimum;

    /* direction vectors (alignment.h:326-336) */
    enum { D_SUB = 0, D_INS = 1, D_DEL = 2, D_SINK = 3, D_INS_EXT = 4, D_DEL_EXT = 8 };

    for (uint32_t i = 0; i < M; ++i)
    {
        /* the windowed form (:703-727), as the staged scheduler drives it (batched_stream.h:145-180): at the end of every
         * window short of the pattern's end, stop -- returning false, with whatever LOCAL cells were reported so far -- if no
         * band cell can still reach min_score (:610-622); otherwise the band goes through a short2 checkpoint clamped from
         * below at int16_min + 32 (GotohCheckpointedScoringContext::last_row / init, :166-176, :139-147).  The text cache the
         * next window reloads (:432-433) holds the same symbols the continuous loop carries. */
        if (window && i && (i % window) == 0)
        {
            int32_t mx = H[0];
            for (uint32_t j = 1; j < B; ++j) mx = imax( mx, H[j] );
            const int32_t thr = (int32_t)((uint32_t)min_score + (uint32_t)(M - i) * (uint32_t)sc->match);
            if (mx < thr) { *sink_p = sink; return 0; }
            for (uint32_t j = 0; j < B; ++j)
            {
                H[j] = (int16_t)imax( H[j], -32768 + 32 );
                F[j] = (int16_t)imax( F[j], -32768 + 32 );
            }
        }
        const uint8_t  q  = pat[i];
        const uint8_t  qq = quals ? quals[i] : 0;
        const int32_t  V  = sc->match;
        const int32_t  S  = orc_mismatch( sc, qq );
        uint8_t edir = D_SUB;

        /* j == 0 (:474-505) */
        {
            const int32_t ftop = F[1] + G_e, htop = H[1] + G_o;
            F[0] = imax( ftop, htop );
            const uint8_t fdir = ftop > htop ? D_DEL_EXT : D_SUB;
            const uint32_t g  = cache[0];
            const int32_t  d  = H[0] + ((g == q) ? V : S);
            int32_t        hi = imax( F[0], d );
            uint8_t      hdir = F[0] > d ? D_INS : D_SUB;
            if (type == ORC_LOCAL) { hi = imax( hi, 0 ); if (hi == 0) hdir = D_SINK; sink_report( &sink, hi, i+1, i+1 ); }
            H[0] = hi;
            if (dirs) dirs[(size_t)i*B] = hdir | D_SUB | fdir;
        }
        int32_t E = H[0] + G_o;                                             /* :507 */

        for (uint32_t j = 1; j + 1 < B; ++j)                                /* :509-566 */
        {
            const int32_t ftop = F[j+1] + G_e, htop = H[j+1] + G_o;
            F[j] = imax( ftop, htop );
            const uint8_t fdir = ftop > htop ? D_DEL_EXT : D_SUB;
            const uint32_t g = cache[j]; cache[j-1] = g;
            const int32_t  d  = H[j] + ((g == q) ? V : S);
            const int32_t  top = F[j], left = E;
            int32_t        hi = imax( imax( top, left ), d );
            uint8_t      hdir = top > left ? (top > d ? D_INS : D_SUB) : (left > d ? D_DEL : D_SUB);
            if (type == ORC_LOCAL) { hi = imax( hi, 0 ); if (hi == 0) hdir = D_SINK; sink_report( &sink, hi, i+j+1, i+1 ); }
            H[j] = hi;
            if (dirs) dirs[(size_t)i*B + j] = hdir | edir | fdir;
            const int32_t eleft = E + G_e, ediag = hi + G_o;                /* :562-565 */
            edir = eleft > ediag ? D_INS_EXT : D_SUB;
            E = imax( ediag, eleft );
        }

        /* new text character (:569-570) */
        const uint32_t g = ((uint64_t)i + B - 1u < N) ? txt[i + B - 1u] : 255u;
        cache[B-2] = packed_cache ? (g & 3u) : g;

        /* j == BAND-1 (:573-602) */
        {
            F[B-1] = infimum;
            const int32_t d  = H[B-1] + ((g == q) ? V : S);
            int32_t       hi = imax( E, d );
            uint8_t     hdir = E > d ? D_DEL : D_SUB;
            if (type == ORC_LOCAL) { hi = imax( hi, 0 ); if (hi == 0) hdir = D_SINK; sink_report( &sink, hi, i+B, i+1 ); }
            H[B-1] = hi;
            if (dirs) dirs[(size_t)i*B + B-1] = hdir | edir | D_SUB;
        }
    }

    if (type == ORC_GLOBAL)                                                  /* :629-630 */
        sink_report( &sink, H[B-1], M + B - 1u, M );
    else if (type == ORC_SEMI_GLOBAL)                                        /* :631-643 */
    {
        const uint32_t m = (uint32_t)imin( (int32_t)(M + B - 1u), (int32_t)N ) - (M - 1u);
        sink_report( &sink, H[0], M, M );
        for (uint32_t j = 1; j < B; ++j)
            if (j < m) sink_report( &sink, H[j], M + j, M );
    }
    *sink_p = sink;
    return 1;
}

static int banded_core(uint32_t B, int type, const orc_gotoh_scheme* sc,
                       const uint8_t* pat, const uint8_t* quals, uint32_t M,
                       const uint8_t* txt, uint32_t N, best_sink* sink_p, uint8_t* dirs)
{
    return banded_core_w( B, type, sc, pat, quals, M, txt, N, sink_p, dirs, 0u, 0 );
}

/* BatchedBandedAlignmentScore<BAND, stream, DeviceStagedThreadScheduler> for one job (batched_banded_inl.h:165-236,
 * batched_stream.h:117-285): the score through 32-row windows with the min_score early exit.  Returns 1 if every window
 * returned true, 0 if one stopped early (or text shorter than the pattern). */
int orc_banded_gotoh_staged(uint32_t B, int type, const orc_gotoh_scheme* sc,
                            const uint8_t* pat, const uint8_t* quals, uint32_t M,
                            const uint8_t* txt, uint32_t N, int32_t min_score,
                            int32_t* score, uint32_t sink_out[2])
{
    best_sink sink; sink_init( &sink );
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    if (B < 2 || B > ORC_MAX_BAND) return -1;
    if (N < M) return 0;
    const int r = banded_core_w( B, type, sc, pat, quals, M, txt, N, &sink, 0, 32u, min_score );
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    return r;
}

int orc_banded_gotoh(uint32_t B, int type, const orc_gotoh_scheme* sc,
                     const uint8_t* pat, const uint8_t* quals, uint32_t M,
                     const uint8_t* txt, uint32_t N,
                     int32_t* score, uint32_t sink_out[2])
{
    best_sink sink; sink_init( &sink );
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    if (B < 2 || B > ORC_MAX_BAND) return -1;
    if (N < M) return 0;                                                    /* :422-423 */
    banded_core( B, type, sc, pat, quals, M, txt, N, &sink, 0 );
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    return 1;
}

/* banded traceback: banded_alignment_traceback (nvbio/alignment/banded_inl.h:354-417) = score pass,
 * clip(pattern_len - sink.y), the walk of priv::banded_alignment_traceback
 * (gotoh/gotoh_banded_inl.h:872-948) over the direction vectors, clip(source.y) -- delivered to
 * nvBowtie's run-length Backtracker (nvBowtie/bowtie2/cuda/alignment_utils.h:115-157): io::Cigar
 * elements (type in bits 0-1, length in bits 2-15; nvbio/io/alignments.h:48-66) in BACKTRACKING
 * order.  The reference recomputes the direction vectors from int16 checkpoints every 16 rows; that
 * equals the single pass here as long as every score stays inside int16 (checked by the caller's
 * sizes; -32736 is the checkpoint clamp, :216-222).  ops (optional): the raw op per step.
 * Returns 1 if an alignment was traced, 0 if nothing was reported (no clip, no ops). */
int orc_banded_gotoh_traceback(uint32_t B, int type, const orc_gotoh_scheme* sc,
                               const uint8_t* pat, const uint8_t* quals, uint32_t M,
                               const uint8_t* txt, uint32_t N,
                               int32_t* score, uint32_t source[2], uint32_t sink_out[2],
                               uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len,
                               uint8_t* ops, uint32_t ops_cap, uint32_t* n_ops)
{
    best_sink best; sink_init( &best );
    *score = best.score; sink_out[0] = sink_out[1] = source[0] = source[1] = 0xFFFFFFFFu;
    *cigar_len = 0; if (n_ops) *n_ops = 0;
    if (B < 2 || B > ORC_MAX_BAND) return -1;
    uint8_t* dirs = (uint8_t*)malloc( (size_t)(M ? M : 1) * B );
    if (N >= M) banded_core( B, type, sc, pat, quals, M, txt, N, &best, dirs );
    *score = best.score;
    if (best.x == 0xFFFFFFFFu || best.y == 0xFFFFFFFFu) { free( dirs ); return 0; }   /* banded_inl.h:376-379 */
    sink_out[0] = best.x; sink_out[1] = best.y;

    uint32_t clen = 0, nops = 0; int prev = 255;
#define CIG_PUSH(type_, len_) do { if (clen < cigar_cap) cigar[clen] = (uint16_t)((type_) | ((len_) << 2)); ++clen; } while (0)
#define OP_PUSH(op_) do { if (ops && nops < ops_cap) ops[nops] = (uint8_t)(op_); ++nops; \
                          if (prev == (int)(op_)) { if (clen - 1 < cigar_cap) cigar[clen-1] += 4; } else { CIG_PUSH( op_, 1u ); prev = (int)(op_); } } while (0)
    if (M - best.y) CIG_PUSH( 3u, M - best.y );                             /* clip the end (:382) */

    int32_t entry = (int32_t)(best.x - best.y);                             /* :884-885 */
    int32_t row   = (int32_t)best.y - 1;
    int state = 0;                                                          /* HSTATE 0, ESTATE 1, FSTATE 2 */
    uint32_t sx, sy; int found = 0;
    while (row >= 0)
    {
        const uint8_t op = dirs[(size_t)row*B + entry];
        const uint8_t h_op = op & 3u;
        if (type == ORC_LOCAL && state == 0 && h_op == 3u)                  /* :898-906 */
        {
            sy = (uint32_t)row + 1u; sx = (uint32_t)entry + sy; found = 1;
            break;
        }
        if (state == 1)      { if ((op & 4u) == 0) state = 0; --entry; OP_PUSH( 2u ); }              /* E: DELETION */
        else if (state == 2) { if ((op & 8u) == 0) state = 0; ++entry; --row; OP_PUSH( 1u ); }       /* F: INSERTION */
        else
        {
            if (h_op == 2u) state = 1;
            else if (h_op == 1u) state = 2;
            else { --row; OP_PUSH( 0u ); }
        }
    }
    if (!found) { sy = 0; sx = (uint32_t)entry; }                            /* :945-947 with checkpoint 0 */
    if (sy) CIG_PUSH( 3u, sy );                                             /* clip the beginning (:413) */
#undef OP_PUSH
#undef CIG_PUSH
    source[0] = sx; source[1] = sy;
    *cigar_len = clen; if (n_ops) *n_ops = nops;
    free( dirs );
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * full-matrix Gotoh in 8-wide stripes (gotoh_inl.h)
 * ------------------------------------------------------------------------------------------ */
#define STRIPE 8u

/* text blocking (:847-1256): stripes over the text, a column of short2 over the pattern */
static int full_text_blocking(int type, const orc_gotoh_scheme* sc,
                              const uint8_t* pat, const uint8_t* quals, uint32_t M,
                              const uint8_t* txt, uint32_t N, int32_t min_score, best_sink* sink)
{
    const int32_t G_o = sc->pat_gap_open, G_e = sc->pat_gap_ext;
    const int32_t infimum = -32768 - imin( G_o, G_e );                      /* :1038 */
    int16_t* tx = (int16_t*)malloc( sizeof(int16_t) * 2u * ((size_t)M + 1u) );
    int16_t* ty = tx + M + 1u;

    /* GotohScoringContext::init (:56-74), TextBlockingTag branch */
    for (uint32_t i = 0; i < M; ++i)
    {
        tx[i] = (int16_t)((type != ORC_LOCAL) ? sc->txt_gap_open + sc->txt_gap_ext * (int32_t)i : 0);
        ty[i] = (int16_t)((type == ORC_LOCAL) ? 0 : infimum);
    }

    const uint32_t nb        = (N + STRIPE - 1u) / STRIPE;
    const uint32_t end_block = (STRIPE*nb > STRIPE) ? STRIPE*nb : STRIPE;   /* :1043-1045 (window_end == N) */
    uint8_t r_cache[STRIPE]; memset( r_cache, 0, sizeof(r_cache) );
    int32_t H[STRIPE+1], F[STRIPE+1];
    int     ok = 1;

    for (uint32_t block = 0; block < end_block; block += STRIPE)
    {
        const int last = (block + STRIPE >= end_block);
        for (uint32_t t = 0; t < STRIPE; ++t)
            if (block + t < N) r_cache[t] = txt[block + t];                 /* :1054-1056 / :1120-1126 */
        for (uint32_t j = 0; j <= STRIPE; ++j)
        {
            H[j] = (type == ORC_GLOBAL) ? (block + j > 0 ? G_o + G_e*(int32_t)(block + j - 1u) : 0) : 0;
            F[j] = infimum;
        }
        int32_t max_score = ORC_SCORE_MIN;
        int32_t temp_i    = H[0];

        for (uint32_t i = 0; i < M; ++i)
        {
            const uint8_t q  = pat[i];
            const uint8_t qq = quals ? quals[i] : 0;
            const int32_t m_i = sc->match, s_i = orc_mismatch( sc, qq );

            /* update_row (:852-969) */
            int32_t H_diag = temp_i;
            H[0] = temp_i = tx[i];
            int32_t E = ty[i];
            for (uint32_t j = 1; j <= STRIPE; ++j)
            {
                F[j] = imax( F[j] + G_e, H[j] + G_o );
                E    = imax( E + G_e, H[j-1] + G_o );
                const int32_t d  = H_diag + ((r_cache[j-1] == q) ? m_i : s_i);
                int32_t       hi = imax( imax( E, F[j] ), d );
                if (type == ORC_LOCAL) hi = imax( hi, 0 );
                H_diag = H[j];
                H[j]   = hi;
            }
            tx[i] = (int16_t)H[STRIPE]; ty[i] = (int16_t)E;                  /* short2 store :942 */
            max_score = imax( max_score, H[STRIPE] );
            if (type == ORC_LOCAL)
                for (uint32_t j = 1; j <= STRIPE; ++j)
                    if (!last || block + j <= N) sink_report( sink, H[j], block + j, i + 1u );
        }
        if (!last)
        {
            if (type == ORC_SEMI_GLOBAL)
                for (uint32_t j = 1; j <= STRIPE; ++j) sink_report( sink, H[j], block + j, M );
            const int32_t missing = (int32_t)(N - block - STRIPE);          /* :1106-1110 */
            if (max_score + missing * sc->match < min_score) { ok = 0; break; }
        }
        else
        {
            if (type == ORC_SEMI_GLOBAL)
            {
                for (uint32_t j = 1; j <= STRIPE; ++j)
                    if (block + j <= N) sink_report( sink, H[j], block + j, M );
            }
            else if (type == ORC_GLOBAL)
            {
                for (uint32_t j = 1; j <= STRIPE; ++j)
                    if (block + j == N) sink_report( sink, H[j], block + j, M );
            }
        }
    }
    free( tx );
    return ok;
}

/* pattern blocking (:444-841): stripes over the pattern, a column of short2 over the text */
/* dirs (optional, N*M bytes, row = text position): each cell's direction vector hdir | edir | fdir as
 * GotohSubmatrixContext::new_cell stores it (gotoh_inl.h:423-436; rules :503-538) */
static int full_pattern_blocking(int type, const orc_gotoh_scheme* sc,
                                 const uint8_t* pat, const uint8_t* quals, uint32_t M,
                                 const uint8_t* txt, uint32_t N, int32_t min_score, best_sink* sink, uint8_t* dirs)
{
    const int32_t G_o = sc->pat_gap_open, G_e = sc->pat_gap_ext;
    const int32_t infimum = -32768 - imin( G_o, G_e );
    int16_t* tx = (int16_t*)malloc( sizeof(int16_t) * 2u * ((size_t)N + 1u) );
    int16_t* ty = tx + N + 1u;

    for (uint32_t i = 0; i < N; ++i)                                        /* init (:56-74), PatternBlockingTag */
    {
        tx[i] = (int16_t)((type == ORC_GLOBAL) ? sc->txt_gap_open + sc->txt_gap_ext * (int32_t)i : 0);
        ty[i] = (int16_t)((type == ORC_LOCAL) ? 0 : infimum);
    }

    const uint32_t nb        = (M + STRIPE - 1u) / STRIPE;
    const uint32_t end_block = (STRIPE*nb > STRIPE) ? STRIPE*nb : STRIPE;
    uint8_t q_sym[STRIPE], q_qual[STRIPE];
    memset( q_sym, 0, sizeof(q_sym) ); memset( q_qual, 0, sizeof(q_qual) );
    int32_t H[STRIPE+1], F[STRIPE+1];
    int     ok = 1;

    for (uint32_t block = 0; block < end_block; block += STRIPE)
    {
        const int last = (block + STRIPE >= end_block);
        for (uint32_t t = 0; t < STRIPE; ++t)
            if (block + t < M) { q_sym[t] = pat[block + t]; q_qual[t] = quals ? quals[block + t] : 0; }
        for (uint32_t j = 0; j <= STRIPE; ++j)
        {
            H[j] = (type != ORC_LOCAL) ? (block + j > 0 ? G_o + G_e*(int32_t)(block + j - 1u) : 0) : 0;
            F[j] = infimum;
        }
        int32_t max_score = ORC_SCORE_MIN;
        int32_t temp_i    = H[0];

        for (uint32_t i = 0; i < N; ++i)
        {
            const uint8_t r_i = txt[i];
            int32_t H_diag = temp_i;                                        /* update_row (:458-575) */
            H[0] = temp_i = tx[i];
            int32_t E = ty[i];
            for (uint32_t j = 1; j <= STRIPE; ++j)
            {
                const int32_t ftop = F[j] + G_e, htop = H[j] + G_o;
                F[j] = imax( ftop, htop );
                const int32_t eleft = E + G_e, hleft = H[j-1] + G_o;
                E    = imax( eleft, hleft );
                const int32_t d  = H_diag + ((r_i == q_sym[j-1]) ? sc->match : orc_mismatch( sc, q_qual[j-1] ));
                int32_t       hi = imax( imax( E, F[j] ), d );
                if (type == ORC_LOCAL) hi = imax( hi, 0 );
                if (dirs && block + j <= M)
                {
                    /* SUBSTITUTION 0, INSERTION 1 (from E), DELETION 2 (from F), SINK 3, INSERTION_EXT 4, DELETION_EXT 8 */
                    const int32_t top = F[j], left = E;
                    uint8_t hdir = top > left ? (top > d ? 2u : 0u) : (left > d ? 1u : 0u);
                    if (type == ORC_LOCAL && hi == 0) hdir = 3u;
                    dirs[(size_t)i * M + (block + j - 1u)] = (uint8_t)(hdir | (eleft > hleft ? 4u : 0u) | (ftop > htop ? 8u : 0u));
                }
                H_diag = H[j];
                H[j]   = hi;
            }
            tx[i] = (int16_t)H[STRIPE]; ty[i] = (int16_t)E;
            max_score = imax( max_score, H[STRIPE] );
            if (type == ORC_LOCAL)
            {
                for (uint32_t j = 1; j <= STRIPE; ++j)
                    if (!last || block + j <= M) sink_report( sink, H[j], i + 1u, block + j );
            }
            else if (last && type == ORC_SEMI_GLOBAL)
            {
                /* save_boundary / save_Mth (utils_inl.h:169-262): the M-th column of this row */
                if (block + STRIPE >= M)
                    sink_report( sink, H[ ((M - 1u) & (STRIPE - 1u)) + 1u ], i + 1u, M );
            }
        }
        if (!last)
        {
            const int32_t missing = (int32_t)(M - block - STRIPE);          /* :706-710 */
            if (max_score + missing * sc->match < min_score) { ok = 0; break; }
        }
    }
    if (ok && type == ORC_GLOBAL)                                           /* :774-775 */
        sink_report( sink, H[ ((M - 1u) & (STRIPE - 1u)) + 1u ], N, M );
    free( tx );
    return ok;
}

/* ------------------------------------------------------------------------------------------
 * The linear-gap Smith-Waterman family (SmithWatermanAligner, EditDistanceAligner), restated on its own from
 * sw/sw_banded_inl.h:339-512 (banded) and sw/sw_inl.h:396-1215 (full matrix, stripes of 16: sw_bandlen_selector
 * :1322-1325; an int16 column of H; pattern blocking tests min_score after every stripe, :676-680, text blocking never).
 * sw = { match, mismatch, deletion, insertion } as signed scores; qualities play no part (SimpleSmithWatermanScheme).
 * ------------------------------------------------------------------------------------------ */
int orc_banded_sw(uint32_t B, int type, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                  int32_t* score, uint32_t sink_out[2])
{
    best_sink sink; sink_init( &sink );
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    if (N < M) return 0;                                                    /* :357-358 */
    const int32_t V = sw[0], S = sw[1], G = sw[2], I = sw[3];
    uint8_t cache[64];
    int32_t band[64];
    /* Reference_cache<31> keeps two bits per symbol (alignment/utils.h): what goes through it comes back as symbol & 3 */
    const uint8_t cmask = (B == 31u) ? 3u : 255u;
    for (uint32_t j = 0; j + 1u < B; ++j) cache[j] = (uint8_t)((j < N ? txt[j] : 255u) & cmask); /* :365-367 (undefined in the reference for N < B-1) */
    for (uint32_t j = 0; j < B; ++j) band[j] = (type == ORC_GLOBAL) ? (int32_t)j * G : 0;    /* init_row_zero :36-45 */
    for (uint32_t i = 0; i < M; ++i)
    {
        const uint8_t q = pat[i];
        int32_t hi;
        {
            const uint8_t g = cache[0];
            hi = imax( band[1] + G, band[0] + (g == q ? V : S) );
            if (type == ORC_LOCAL) { hi = imax( hi, 0 ); sink_report( &sink, hi, i + 1u, i + 1u ); }
            band[0] = hi;
        }
        for (uint32_t j = 1; j + 1u < B; ++j)
        {
            const uint8_t g = cache[j]; cache[j-1] = g;
            hi = imax( imax( band[j+1] + G, band[j-1] + I ), band[j] + (g == q ? V : S) );
            if (type == ORC_LOCAL) { hi = imax( hi, 0 ); sink_report( &sink, hi, i + j + 1u, i + 1u ); }
            band[j] = hi;
        }
        const uint8_t g = (i + B - 1u < N) ? txt[i + B - 1u] : 255u;        /* :452-453 */
        cache[B-2u] = (uint8_t)(g & cmask);
        hi = imax( band[B-2u] + I, band[B-1u] + (g == q ? V : S) );
        if (type == ORC_LOCAL) { hi = imax( hi, 0 ); sink_report( &sink, hi, i + B, i + 1u ); }
        band[B-1u] = hi;
    }
    if (type == ORC_GLOBAL) sink_report( &sink, band[B-1u], M + B - 1u, M );
    else if (type == ORC_SEMI_GLOBAL)
    {
        const uint32_t m = (M + B - 1u < N ? M + B - 1u : N) - (M - 1u);
        sink_report( &sink, band[0], M, M );
        for (uint32_t j = 1; j < B; ++j) if (j < m) sink_report( &sink, band[j], M + j, M );
    }
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    return 1;
}

/* banded traceback of the linear-gap Smith-Waterman aligner: banded_alignment_traceback (nvbio/alignment/banded_inl.h:354-417) over
 * sw/sw_banded_inl.h -- the score pass of orc_banded_sw with every cell's direction as SmithWatermanSubmatrixContext::new_cell stores
 * it (:420-468: INSERTION if top wins, DELETION if left wins, else SUBSTITUTION; strict `>`, top against left first; NO SINK mark, so
 * that a LOCAL walk runs on to row 0), then the walk of :741-797, run-length encoded as nvBowtie's Backtracker does.  Outputs as
 * orc_banded_gotoh_traceback.  Returns 1 if an alignment was traced, 0 if nothing was reported. */
int orc_banded_sw_traceback(uint32_t B, int type, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                            int32_t* score, uint32_t source[2], uint32_t sink_out[2],
                            uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len)
{
    best_sink sink; sink_init( &sink );
    *score = sink.score; sink_out[0] = sink_out[1] = source[0] = source[1] = 0xFFFFFFFFu; *cigar_len = 0;
    if (B < 2 || B > 64) return -1;
    if (N < M) return 0;
    const int32_t V = sw[0], S = sw[1], G = sw[2], I = sw[3];
    uint8_t cache[64];
    int32_t band[64];
    uint8_t* dirs = (uint8_t*)malloc( (size_t)(M ? M : 1) * B );
    const uint8_t cmask = (B == 31u) ? 3u : 255u;
    for (uint32_t j = 0; j + 1u < B; ++j) cache[j] = (uint8_t)((j < N ? txt[j] : 255u) & cmask);
    for (uint32_t j = 0; j < B; ++j) band[j] = (type == ORC_GLOBAL) ? (int32_t)j * G : 0;
    for (uint32_t i = 0; i < M; ++i)
    {
        const uint8_t q = pat[i];
        int32_t hi, top, left, diag;
        {
            const uint8_t g = cache[0];
            diag = band[0] + (g == q ? V : S); top = band[1] + G;
            hi = imax( top, diag );
            if (type == ORC_LOCAL) { hi = imax( hi, 0 ); sink_report( &sink, hi, i + 1u, i + 1u ); }
            band[0] = hi; dirs[(size_t)i*B] = top > diag ? 1u : 0u;
        }
        for (uint32_t j = 1; j + 1u < B; ++j)
        {
            const uint8_t g = cache[j]; cache[j-1] = g;
            diag = band[j] + (g == q ? V : S); top = band[j+1] + G; left = band[j-1] + I;
            hi = imax( imax( top, left ), diag );
            if (type == ORC_LOCAL) { hi = imax( hi, 0 ); sink_report( &sink, hi, i + j + 1u, i + 1u ); }
            band[j] = hi;
            dirs[(size_t)i*B + j] = top > left ? (top > diag ? 1u : 0u) : (left > diag ? 2u : 0u);
        }
        const uint8_t g = (i + B - 1u < N) ? txt[i + B - 1u] : 255u;
        cache[B-2u] = (uint8_t)(g & cmask);
        diag = band[B-1u] + (g == q ? V : S); left = band[B-2u] + I;
        hi = imax( left, diag );
        if (type == ORC_LOCAL) { hi = imax( hi, 0 ); sink_report( &sink, hi, i + B, i + 1u ); }
        band[B-1u] = hi; dirs[(size_t)i*B + B-1u] = left > diag ? 2u : 0u;
    }
    if (type == ORC_GLOBAL) sink_report( &sink, band[B-1u], M + B - 1u, M );
    else if (type == ORC_SEMI_GLOBAL)
    {
        const uint32_t m = (M + B - 1u < N ? M + B - 1u : N) - (M - 1u);
        sink_report( &sink, band[0], M, M );
        for (uint32_t j = 1; j < B; ++j) if (j < m) sink_report( &sink, band[j], M + j, M );
    }
    *score = sink.score;
    if (sink.x == 0xFFFFFFFFu || sink.y == 0xFFFFFFFFu) { free( dirs ); return 0; }
    sink_out[0] = sink.x; sink_out[1] = sink.y;
    uint32_t clen = 0; int prev = 255;
#define CIG_PUSH(type_, len_) do { if (clen < cigar_cap) cigar[clen] = (uint16_t)((type_) | ((len_) << 2)); ++clen; } while (0)
#define OP_PUSH(op_) do { if (prev == (int)(op_)) { if (clen - 1 < cigar_cap) cigar[clen-1] += 4; } else { CIG_PUSH( op_, 1u ); prev = (int)(op_); } } while (0)
    if (M - sink.y) CIG_PUSH( 3u, M - sink.y );
    int32_t entry = (int32_t)(sink.x - sink.y), row = (int32_t)sink.y - 1;
    while (row >= 0)
    {
        const uint8_t op = dirs[(size_t)row*B + entry];
        if (op == 2u)      { --entry; OP_PUSH( 2u ); }
        else if (op == 1u) { ++entry; --row; OP_PUSH( 1u ); }
        else               { --row; OP_PUSH( 0u ); }
    }
#undef OP_PUSH
#undef CIG_PUSH
    source[0] = (uint32_t)entry; source[1] = 0u;
    *cigar_len = clen;
    free( dirs );
    return 1;
}

#define SW_STRIPE 16u
/* dirs (pattern blocking only, may be NULL): the flow of cell (text row i, pattern column c) at dirs[i*M + c] -- 0 SUBSTITUTION,
 * 1 INSERTION, 2 DELETION, 3 SINK (a LOCAL cell of score 0), as SWSubmatrixContext::new_cell stores it (sw/sw_inl.h:371-388) from the
 * comparison of sw_alignment_score_dispatch (:484-490) */
static int full_sw_core(int type, int blocking, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                        int32_t min_score, int32_t* score, uint32_t sink_out[2], uint8_t* dirs)
{
    best_sink sink; sink_init( &sink );
    const int32_t V = sw[0], S = sw[1], G = sw[2], I = sw[3];
    /* the stripes run over `cols` symbols of one string, the int16 column over `rows` symbols of the other */
    const uint32_t rows = blocking ? M : N, cols = blocking ? N : M;
    const uint8_t* rstr = blocking ? pat : txt;
    const uint8_t* cstr = blocking ? txt : pat;
    const int32_t top_c  = blocking ? I : G;        /* from the previous row: that string advances alone          */
    const int32_t left_c = blocking ? G : I;
    int16_t* temp = (int16_t*)malloc( sizeof(int16_t) * ((size_t)rows + 1u) );
    for (uint32_t i = 0; i < rows; ++i)                                     /* SWScoringContext::init :56-71 */
        temp[i] = (int16_t)(blocking ? (type != ORC_LOCAL  ? I * (int32_t)(i + 1u) : 0)
                                     : (type == ORC_GLOBAL ? G * (int32_t)(i + 1u) : 0));
    const uint32_t nb        = (cols + SW_STRIPE - 1u) / SW_STRIPE;
    const uint32_t end_block = (SW_STRIPE*nb > SW_STRIPE) ? SW_STRIPE*nb : SW_STRIPE;
    uint8_t c_cache[SW_STRIPE]; memset( c_cache, 0, sizeof(c_cache) );
    int32_t band[SW_STRIPE+1];
    int     ok = 1;
    for (uint32_t block = 0; block < end_block; block += SW_STRIPE)
    {
        const int last = (block + SW_STRIPE >= end_block);
        for (uint32_t t = 0; t < SW_STRIPE; ++t) if (block + t < cols) c_cache[t] = cstr[block + t];
        for (uint32_t j = 0; j <= SW_STRIPE; ++j)                           /* :650 / :1083 */
            band[j] = blocking ? (type == ORC_GLOBAL ? G * (int32_t)(block + j) : 0)
                               : (type != ORC_LOCAL  ? I * (int32_t)(block + j) : 0);
        int32_t max_score = ORC_SCORE_MIN;
        int32_t temp_i    = band[0];
        for (uint32_t i = 0; i < rows; ++i)
        {
            const uint8_t r = rstr[i];
            int32_t prev = temp_i;
            band[0] = temp_i = temp[i];
            for (uint32_t j = 1; j <= SW_STRIPE; ++j)
            {
                const int32_t d  = prev + (c_cache[j-1] == r ? V : S);
                const int32_t top = band[j] + top_c, left = band[j-1] + left_c;
                int32_t       hi = imax( imax( top, left ), d );
                if (type == ORC_LOCAL) hi = imax( hi, 0 );
                prev = band[j]; band[j] = hi;
                if (dirs && !blocking && block + j <= M)
                {
                    const uint8_t dir = top > left ? (top > d ? 2u : 0u) : (left > d ? 1u : 0u);
                    dirs[(size_t)i * M + (block + j - 1u)] = (type == ORC_LOCAL && hi == 0) ? 3u : dir;
                }
            }
            temp[i] = (int16_t)band[SW_STRIPE];
            max_score = imax( max_score, band[SW_STRIPE] );
            if (type == ORC_LOCAL)
            {
                for (uint32_t j = 1; j <= SW_STRIPE; ++j)
                    if (!last || block + j <= cols)
                    {
                        if (blocking) sink_report( &sink, band[j], block + j, i + 1u );
                        else          sink_report( &sink, band[j], i + 1u, block + j );
                    }
            }
            else if (!blocking && last && type == ORC_SEMI_GLOBAL)          /* save_boundary: the M-th column */
                sink_report( &sink, band[ ((M - 1u) & (SW_STRIPE - 1u)) + 1u ], i + 1u, M );
        }
        if (blocking)
        {
            if (type == ORC_SEMI_GLOBAL)
            {
                for (uint32_t j = 1; j <= SW_STRIPE; ++j) if (!last || block + j <= N) sink_report( &sink, band[j], block + j, M );
            }
            else if (type == ORC_GLOBAL && last)
                for (uint32_t j = 1; j <= SW_STRIPE; ++j) if (block + j == N) sink_report( &sink, band[j], block + j, M );
        }
        else if (!last)
        {
            const int32_t missing = (int32_t)(M - block - SW_STRIPE);       /* :676-680 */
            if (max_score + missing * V < min_score) { ok = 0; break; }
        }
    }
    if (!blocking && ok && type == ORC_GLOBAL)                              /* save_Mth :745-746 */
        sink_report( &sink, band[ ((M - 1u) & (SW_STRIPE - 1u)) + 1u ], N, M );
    free( temp );
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    return ok;
}

int orc_full_sw(int type, int blocking, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                int32_t min_score, int32_t* score, uint32_t sink_out[2])
{
    return full_sw_core( type, blocking, sw, pat, M, txt, N, min_score, score, sink_out, NULL );
}

/* Full-matrix traceback of the linear-gap Smith-Waterman aligner: aln::alignment_traceback<..>( SmithWatermanAligner<type>, .. )
 * (nvbio/alignment/alignment_inl.h:355-455: score pass with checkpoints, clip, walk, implicit first row / column, clip) over
 * sw/sw_inl.h:1476-1600 (checkpoints every 64 pattern columns, the flow submatrix between two of them) and the walk :1644-1694:
 * one state, DELETION moves along the text only, INSERTION along the pattern only, a LOCAL walk stops at a SINK cell.
 * Outputs as orc_full_gotoh_traceback.  (The reference recomputes the submatrices from int16 checkpoints; equal to the single pass
 * here while every score fits int16.) */
int orc_full_sw_traceback(int type, const int32_t sw[4], const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score,
                          int32_t* score, uint32_t source[2], uint32_t sink_out[2], uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len)
{
    uint8_t* dirs = (uint8_t*)malloc( (size_t)(N ? N : 1) * (M ? M : 1) );
    uint32_t best[2];
    full_sw_core( type, 0, sw, pat, M, txt, N, min_score, score, best, dirs );
    sink_out[0] = sink_out[1] = source[0] = source[1] = 0xFFFFFFFFu; *cigar_len = 0;
    if (best[0] == 0xFFFFFFFFu || best[1] == 0xFFFFFFFFu) { free( dirs ); return 0; }
    sink_out[0] = best[0]; sink_out[1] = best[1];

    uint32_t clen = 0; int prev = 255;
#define CIG_PUSH(type_, len_) do { if (clen < cigar_cap) cigar[clen] = (uint16_t)((type_) | ((len_) << 2)); ++clen; } while (0)
#define OP_PUSH(op_) do { if (prev == (int)(op_)) { if (clen - 1 < cigar_cap) cigar[clen-1] += 4; } else { CIG_PUSH( op_, 1u ); prev = (int)(op_); } } while (0)
    if (M - best[1]) CIG_PUSH( 3u, M - best[1] );
    int32_t row = (int32_t)best[0], col = (int32_t)best[1] - 1;
    while (row > 0 && col >= 0)
    {
        const uint8_t op = dirs[(size_t)(row - 1) * M + col];
        if (type == ORC_LOCAL && op == 3u) break;
        if (op != 2u) --col;
        if (op != 1u) --row;
        OP_PUSH( op );
    }
    uint32_t sx = (uint32_t)row, sy = (uint32_t)(col + 1);
    if (type == ORC_SEMI_GLOBAL || type == ORC_GLOBAL)                       /* the implicit first row (alignment_inl.h:437-445) */
        if (sx == 0) for (; sy > 0; --sy) OP_PUSH( 1u );
    if (type == ORC_GLOBAL)                                                  /* ... and first column (:446-452) */
        if (sy == 0) for (; sx > 0; --sx) OP_PUSH( 2u );
    if (sy) CIG_PUSH( 3u, sy );
#undef OP_PUSH
#undef CIG_PUSH
    source[0] = sx; source[1] = sy;
    *cigar_len = clen;
    free( dirs );
    return 1;
}

/* the same two DPs reporting into a Best2Sink<int32>( distinct_dist ): out = { score1, sink1.x, sink1.y, score2, sink2.x, sink2.y } */
int orc_banded_gotoh_best2(uint32_t B, int type, const orc_gotoh_scheme* sc, const uint8_t* pat, const uint8_t* quals, uint32_t M,
                           const uint8_t* txt, uint32_t N, uint32_t distinct_dist, int64_t out[6])
{
    best_sink sink; sink_init( &sink ); sink.two = 1; sink.dist = distinct_dist;
    int ok = 0;
    if (B < 2 || B > ORC_MAX_BAND) return -1;
    if (N >= M) { banded_core( B, type, sc, pat, quals, M, txt, N, &sink, 0 ); ok = 1; }
    out[0] = sink.score; out[1] = sink.x; out[2] = sink.y; out[3] = sink.score2; out[4] = sink.x2; out[5] = sink.y2;
    return ok;
}
int orc_full_gotoh_best2(int type, int blocking, const orc_gotoh_scheme* sc, const uint8_t* pat, const uint8_t* quals, uint32_t M,
                         const uint8_t* txt, uint32_t N, int32_t min_score, uint32_t distinct_dist, int64_t out[6])
{
    best_sink sink; sink_init( &sink ); sink.two = 1; sink.dist = distinct_dist;
    const int ok = blocking ?
        full_text_blocking(    type, sc, pat, quals, M, txt, N, min_score, &sink ) :
        full_pattern_blocking( type, sc, pat, quals, M, txt, N, min_score, &sink, 0 );
    out[0] = sink.score; out[1] = sink.x; out[2] = sink.y; out[3] = sink.score2; out[4] = sink.x2; out[5] = sink.y2;
    return ok;
}

int orc_full_gotoh(int type, int blocking, const orc_gotoh_scheme* sc,
                   const uint8_t* pat, const uint8_t* quals, uint32_t M,
                   const uint8_t* txt, uint32_t N, int32_t min_score,
                   int32_t* score, uint32_t sink_out[2])
{
    best_sink sink; sink_init( &sink );
    const int ok = blocking ?
        full_text_blocking(    type, sc, pat, quals, M, txt, N, min_score, &sink ) :
        full_pattern_blocking( type, sc, pat, quals, M, txt, N, min_score, &sink, 0 );
    *score = sink.score; sink_out[0] = sink.x; sink_out[1] = sink.y;
    return ok;
}

/* full-matrix traceback: aln::alignment_traceback (nvbio/alignment/alignment_inl.h:355-455): pattern-blocking score
 * pass, clip(pattern_len - sink.y), the walk of priv::alignment_traceback (gotoh/gotoh_inl.h:1573-1640) over the
 * direction vectors, the implicit first row / column (:437-452), clip(source.y) -- delivered to nvBowtie's
 * run-length Backtracker as in orc_banded_gotoh_traceback (INSERTION = pattern symbol without text, DELETION = text
 * symbol without pattern).  Coordinates: x = text, y = pattern.  The reference recomputes 64-column blocks from int16
 * checkpoints; equal to the single pass here while every score fits int16. */
int orc_full_gotoh_traceback(int type, const orc_gotoh_scheme* sc,
                             const uint8_t* pat, const uint8_t* quals, uint32_t M,
                             const uint8_t* txt, uint32_t N, int32_t min_score,
                             int32_t* score, uint32_t source[2], uint32_t sink_out[2],
                             uint16_t* cigar, uint32_t cigar_cap, uint32_t* cigar_len)
{
    best_sink best; sink_init( &best );
    uint8_t* dirs = (uint8_t*)malloc( (size_t)(N ? N : 1) * (M ? M : 1) );
    full_pattern_blocking( type, sc, pat, quals, M, txt, N, min_score, &best, dirs );
    *score = best.score; sink_out[0] = sink_out[1] = source[0] = source[1] = 0xFFFFFFFFu; *cigar_len = 0;
    if (best.x == 0xFFFFFFFFu || best.y == 0xFFFFFFFFu) { free( dirs ); return 0; }
    sink_out[0] = best.x; sink_out[1] = best.y;

    uint32_t clen = 0; int prev = 255;
#define CIG_PUSH(type_, len_) do { if (clen < cigar_cap) cigar[clen] = (uint16_t)((type_) | ((len_) << 2)); ++clen; } while (0)
#define OP_PUSH(op_) do { if (prev == (int)(op_)) { if (clen - 1 < cigar_cap) cigar[clen-1] += 4; } else { CIG_PUSH( op_, 1u ); prev = (int)(op_); } } while (0)
    if (M - best.y) CIG_PUSH( 3u, M - best.y );
    int32_t row = (int32_t)best.x, col = (int32_t)best.y - 1;
    int state = 0;                                                          /* HSTATE 0, ESTATE 1, FSTATE 2 */
    int found = 0;
    while (row > 0 && col >= 0)
    {
        const uint8_t op = dirs[(size_t)(row - 1) * M + col];
        const uint8_t h_op = op & 3u;
        if (type == ORC_LOCAL && state == 0 && h_op == 3u) { found = 1; break; }
        if (state == 1)      { if ((op & 4u) == 0) state = 0; --col; OP_PUSH( 1u ); }          /* E: INSERTION */
        else if (state == 2) { if ((op & 8u) == 0) state = 0; --row; OP_PUSH( 2u ); }          /* F: DELETION  */
        else
        {
            if (h_op == 1u) state = 1;
            else if (h_op == 2u) state = 2;
            else { --col; --row; OP_PUSH( 0u ); }
        }
    }
    (void)found;
    uint32_t sx = (uint32_t)row, sy = (uint32_t)(col + 1);
    if (type == ORC_SEMI_GLOBAL || type == ORC_GLOBAL)                       /* the implicit first row (alignment_inl.h:437-445) */
        if (sx == 0) for (; sy > 0; --sy) OP_PUSH( 1u );
    if (type == ORC_GLOBAL)                                                  /* ... and first column (:446-452) */
        if (sy == 0) for (; sx > 0; --sx) OP_PUSH( 2u );
    if (sy) CIG_PUSH( 3u, sy );
#undef OP_PUSH
#undef CIG_PUSH
    source[0] = sx; source[1] = sy;
    *cigar_len = clen;
    free( dirs );
    return 1;
}

/* nvBowtie finish_alignment (nvBowtie/bowtie2/cuda/traceback_inl.h:536-705): from the CIGAR (stored backwards, as the
 * Backtracker leaves it), the read as aligned and the text window, the edit distance (mismatches + inserted + deleted
 * symbols; soft clips excluded) and the MDS byte stream: two length bytes, then tokens -- [MDS_MATCH 0, run <= 255],
 * [MDS_MISMATCH 1, read symbol], [MDS_INSERTION 2 | MDS_DELETION 3, length, symbols...] (soft clips are recorded as
 * insertions, :577-583).  cigar_offset = Alignment::source.x, where the alignment starts in the text window.
 * nvBowtie's device code cannot be compiled here: this restatement is pinned by its definition only. */
void orc_finish_alignment(const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N,
                          const uint16_t* cigar, uint32_t cigar_len, uint32_t cigar_offset,
                          uint32_t* ed_out, uint8_t* mds, uint32_t mds_cap, uint32_t* mds_len_out)
{
    uint32_t mds_len = 2, ed = 0; uint8_t mds_op = 4;                       /* MDS_INVALID */
#define MDS_PUT(v_) do { if (mds && mds_len < mds_cap) mds[mds_len] = (uint8_t)(v_); ++mds_len; } while (0)
    uint32_t j = 0, k = cigar_offset;
    for (uint32_t i = 0; i < cigar_len; ++i)
    {
        const uint32_t l = cigar[cigar_len - i - 1u] >> 2, t = cigar[cigar_len - i - 1u] & 3u;
        if (t != 0u)                                                        /* insertion 1, deletion 2, clip 3 */
        {
            mds_op = (t == 2u) ? 3 : 2;
            MDS_PUT( mds_op ); MDS_PUT( l );
        }
        for (uint32_t n = 0; n < l; ++n)
        {
            if (t != 2u) ++j;                                               /* substitution, insertion, clip consume the read */
            if (t == 0u || t == 2u) ++k;                                    /* substitution, deletion consume the text */
            const uint8_t readc = (j > 0 && j <= M) ? pat[j-1] : 255u;
            const uint8_t refc  = (k > 0 && k <= N) ? txt[k-1] : 255u;
            if (t == 0u)
            {
                if (readc == refc)
                {
                    if (mds_op == 0 && mds && mds_len - 1u < mds_cap && mds[mds_len-1] < 255) mds[mds_len-1]++;
                    else if (mds_op == 0 && !mds) { /* length-only pass: runs cannot be tracked without storage */ }
                    else { mds_op = 0; MDS_PUT( 0 ); MDS_PUT( 1 ); }
                }
                else { mds_op = 1; MDS_PUT( 1 ); MDS_PUT( readc ); ++ed; }
            }
            else
            {
                MDS_PUT( t == 2u ? refc : readc );
                if (t != 3u) ++ed;
            }
        }
    }
#undef MDS_PUT
    if (mds && mds_cap >= 2) { mds[0] = (uint8_t)(mds_len & 0xFF); mds[1] = (uint8_t)(mds_len >> 8); }
    *ed_out = ed; *mds_len_out = mds_len;
}

void orc_banded_gotoh_batch(uint32_t band, int type, const orc_gotoh_scheme* s,
                            const uint8_t* pats, const uint8_t* quals, const uint32_t* po,
                            const uint8_t* txts, const uint32_t* to, uint32_t n,
                            int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        orc_banded_gotoh( band, type, s, pats + po[i], quals ? quals + po[i] : 0, po[i+1] - po[i],
                          txts + to[i], to[i+1] - to[i], scores + i, sinks + 2*i );
}
void orc_full_gotoh_batch(int type, int blocking, const orc_gotoh_scheme* s,
                          const uint8_t* pats, const uint8_t* quals, const uint32_t* po,
                          const uint8_t* txts, const uint32_t* to, uint32_t n,
                          int32_t min_score, int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        orc_full_gotoh( type, blocking, s, pats + po[i], quals ? quals + po[i] : 0, po[i+1] - po[i],
                        txts + to[i], to[i+1] - to[i], min_score, scores + i, sinks + 2*i );
}

/* nvBowtie-shaped extension over packed inputs: ReadStream::operator[] (nvbio/io/utils.h:150-168)
 * + PackedStringLoader text window (alignment_utils.h:277-302) */
void orc_banded_gotoh_packed_batch(uint32_t band, int type, const orc_gotoh_scheme* s,
                                   const uint32_t* reads4, const uint32_t* read_offsets, const uint8_t* quals,
                                   const uint32_t* read_id, const uint8_t* flags,
                                   const uint32_t* genome2, const uint32_t* win_begin, const uint32_t* win_end,
                                   uint32_t n, int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel
    {
        uint32_t cap_p = 512, cap_t = 1024;
        uint8_t* p  = (uint8_t*)malloc( cap_p );
        uint8_t* pq = (uint8_t*)malloc( cap_p );
        uint8_t* t  = (uint8_t*)malloc( cap_t );
        #pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i)
        {
            const uint32_t rid   = read_id ? read_id[i] : (uint32_t)i;
            const uint32_t first = read_offsets[rid], len = read_offsets[rid+1] - first;
            const uint32_t tb = win_begin[i], tl = win_end[i] - tb;
            if (len > cap_p) { cap_p = 2*len; p = (uint8_t*)realloc( p, cap_p ); pq = (uint8_t*)realloc( pq, cap_p ); }
            if (tl  > cap_t) { cap_t = 2*tl;  t = (uint8_t*)realloc( t, cap_t ); }
            const int rev = flags ? (flags[i] & 1) : 0, comp = flags ? (flags[i] & 2) : 0;
            for (uint32_t k = 0; k < len; ++k)
            {
                const uint32_t idx = rev ? first + len - 1u - k : first + k;
                const uint8_t  c   = orc_get4( reads4, idx );
                p[k]  = comp ? (c < 4 ? 3 - c : c) : c;
                pq[k] = quals ? quals[idx] : 0;
            }
            for (uint32_t k = 0; k < tl; ++k) t[k] = orc_get2( genome2, (uint64_t)tb + k );
            orc_banded_gotoh( band, type, s, p, quals ? pq : 0, len, t, tl, scores + i, sinks + 2*i );
        }
        free( p ); free( pq ); free( t );
    }
}

void orc_banded_gotoh_traceback_packed_batch(uint32_t band, int type, const orc_gotoh_scheme* s,
                                   const uint32_t* reads4, const uint32_t* read_offsets, const uint8_t* quals,
                                   const uint32_t* read_id, const uint8_t* flags,
                                   const uint32_t* genome2, const uint32_t* win_begin, const uint32_t* win_end,
                                   uint32_t n, int32_t* scores, uint32_t* sources, uint32_t* sinks,
                                   uint16_t* cigars, uint32_t cigar_stride, uint32_t* cigar_lens)
{
    #pragma omp parallel
    {
        uint32_t cap_p = 512, cap_t = 1024;
        uint8_t* p  = (uint8_t*)malloc( cap_p );
        uint8_t* pq = (uint8_t*)malloc( cap_p );
        uint8_t* t  = (uint8_t*)malloc( cap_t );
        #pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i)
        {
            const uint32_t rid   = read_id ? read_id[i] : (uint32_t)i;
            const uint32_t first = read_offsets[rid], len = read_offsets[rid+1] - first;
            const uint32_t tb = win_begin[i], tl = win_end[i] - tb;
            if (len > cap_p) { cap_p = 2*len; p = (uint8_t*)realloc( p, cap_p ); pq = (uint8_t*)realloc( pq, cap_p ); }
            if (tl  > cap_t) { cap_t = 2*tl;  t = (uint8_t*)realloc( t, cap_t ); }
            const int rev = flags ? (flags[i] & 1) : 0, comp = flags ? (flags[i] & 2) : 0;
            for (uint32_t k = 0; k < len; ++k)
            {
                const uint32_t idx = rev ? first + len - 1u - k : first + k;
                const uint8_t  c   = orc_get4( reads4, idx );
                p[k]  = comp ? (c < 4 ? 3 - c : c) : c;
                pq[k] = quals ? quals[idx] : 0;
            }
            for (uint32_t k = 0; k < tl; ++k) t[k] = orc_get2( genome2, (uint64_t)tb + k );
            orc_banded_gotoh_traceback( band, type, s, p, quals ? pq : 0, len, t, tl, scores + i, sources + 2*i, sinks + 2*i,
                                        cigars + (size_t)i*cigar_stride, cigar_stride, cigar_lens + i, 0, 0, 0 );
        }
        free( p ); free( pq ); free( t );
    }
}

/* ------------------------------------------------------------------------------------------
 * hamming_backtrack (nvbio/fmindex/backtrack.h:51-157) with the reference benchmark's CountDelegate
 * (nvbio-test/fmindex_test.cu:720-737).  stream = the whole symbol stream the pattern lives in, pattern = stream + begin.
 * quirks = 0: a branch stops at the start of the pattern (the documented behaviour);
 * quirks = 1: the code's behaviour -- no stop at l == 0 (:110-116): second report when all mismatches are used, otherwise the walk
 *             continues through the symbols preceding the pattern (32-bit index arithmetic, as the benchmark's PackedStream), and
 *             match() over a length above 2^31 runs no step (its loop index is an int32, fmindex_inl.h:223).
 * Returns the number of ranges reported; *count = sum of their sizes (uint32); the first `cap` ranges go to ranges[2k], [2k+1].
 * ------------------------------------------------------------------------------------------ */
static void bt_match(const orc_fm_index* f, const uint8_t* stream, uint32_t begin, uint32_t len, uint32_t range[2])
{
    uint32_t x = range[0], y = range[1];
    for (int32_t i = (int32_t)(len - 1u); i >= 0 && x <= y; --i)
    {
        const uint8_t c = stream[(uint32_t)(begin + (uint32_t)i)];
        if (c > 3) { x = 1; y = 0; break; }
        uint32_t r[2]; fm_rank2( f, x - 1u, y, c, r, 0 );
        x = f->L2[c] + r[0] + 1u; y = f->L2[c] + r[1];
    }
    range[0] = x; range[1] = y;
}
uint32_t orc_hamming_backtrack(const orc_fm_index* f, const uint8_t* stream, uint32_t begin, uint32_t len, uint32_t seed, uint32_t mismatches,
                               int quirks, uint32_t* count, uint32_t* ranges, uint32_t cap)
{
    uint32_t total = 0, nr = 0;
#define BT_REPORT(X, Y) do { total += (Y) + 1u - (X); if (ranges && nr < cap) { ranges[2u*nr] = (X); ranges[2u*nr+1u] = (Y); } ++nr; } while (0)
    if (seed > len) seed = len;
    if (mismatches == 0 || seed == len)
    {
        uint32_t r[2] = { 0, f->length }; bt_match( f, stream, begin, len, r );
        if (r[0] <= r[1]) BT_REPORT( r[0], r[1] );
    }
    else
    {
        uint32_t root[2] = { 0, f->length }; bt_match( f, stream, begin + len - seed, seed, root );
        if (root[0] <= root[1])
        {
            static __thread uint32_t stack[4096][4];
            uint32_t sp = 0, lo[4], hi[4];
            orc_rank4( f, root[0] - 1u, lo ); orc_rank4( f, root[1], hi );
            const uint8_t c0 = stream[(uint32_t)(begin + len - seed - 1u)];
            for (uint32_t c = 0; c < 4; ++c)
                if (lo[c] < hi[c]) { stack[sp][0] = f->L2[c] + lo[c] + 1u; stack[sp][1] = f->L2[c] + hi[c]; stack[sp][2] = (c == c0) ? 0u : 1u; stack[sp][3] = len - seed - 1u; ++sp; }
            while (sp)
            {
                --sp;
                uint32_t range[2] = { stack[sp][0], stack[sp][1] };
                const uint32_t cost = stack[sp][2], l = stack[sp][3];
                if (l == 0u)
                {
                    if (range[0] <= range[1]) BT_REPORT( range[0], range[1] );
                    if (!quirks) continue;
                }
                if (cost < mismatches)
                {
                    orc_rank4( f, range[0] - 1u, lo ); orc_rank4( f, range[1], hi );
                    const uint8_t cp = stream[(uint32_t)(begin + (uint32_t)(l - 1u))];
                    for (uint32_t c = 0; c < 4; ++c)
                        if (lo[c] < hi[c] && sp < 4096u)
                        { stack[sp][0] = f->L2[c] + lo[c] + 1u; stack[sp][1] = f->L2[c] + hi[c]; stack[sp][2] = cost + (c == cp ? 0u : 1u); stack[sp][3] = l - 1u; ++sp; }
                }
                else
                {
                    bt_match( f, stream, begin, l, range );
                    if (range[0] <= range[1]) BT_REPORT( range[0], range[1] );
                }
            }
        }
    }
#undef BT_REPORT
    *count = total;
    return nr;
}

/* ------------------------------------------------------------------------------------------
 * nvBowtie's best / second-best bookkeeping and mapping quality, restated from device-only / thrust-bound sources
 * (PARITY UNPINNED: neither reduce_inl.h nor mapq.h compiles host-only in the development container)
 * ------------------------------------------------------------------------------------------ */
static int distinct_alignments(uint32_t pos1, int rc1, uint32_t pos2, int rc2, uint32_t dist)   /* nvbio/io/alignments_inl.h:26-38 */
{
    if (rc1 != rc2) return 1;
    return (pos1 >= pos2 - (pos2 < dist ? pos2 : dist) && pos1 <= pos2 + dist) ? 0 : 1;
}

/* score_reduce_kernel (nvBowtie/bowtie2/cuda/reduce_inl.h:65-140) for ONE read: candidates (score, position, strand) in the
 * order given; a1 / a2 start unaligned at worst_score (init_alignments_kernel, aligner.h:279-301).
 * out = { a1 aligned, a1 score, a1 pos, a1 rc, a2 aligned, a2 score, a2 pos, a2 rc } */
void orc_score_reduce(const int32_t* scores, const uint32_t* pos, const uint8_t* rc, uint32_t n, uint32_t read_len, int32_t worst_score,
                      int64_t out[8])
{
    int     a1_al = 0, a2_al = 0, a1_rc = 0, a2_rc = 0;
    int32_t a1_s = worst_score, a2_s = worst_score;
    uint32_t a1_p = 0xFFFFFFFFu, a2_p = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i)
    {
        if ((rc[i] == a1_rc && pos[i] == a1_p) || (rc[i] == a2_rc && pos[i] == a2_p)) continue;       /* :104-107 */
        if (scores[i] > a1_s)
        {
            a2_al = a1_al; a2_s = a1_s; a2_p = a1_p; a2_rc = a1_rc;                                   /* :113-114 */
            a1_al = 1; a1_s = scores[i]; a1_p = pos[i]; a1_rc = rc[i];
        }
        else if (scores[i] > a2_s && distinct_alignments( a1_p, a1_rc, pos[i], rc[i], read_len / 2u ))  /* :118-124 */
        {
            a2_al = 1; a2_s = scores[i]; a2_p = pos[i]; a2_rc = rc[i];
        }
    }
    out[0] = a1_al; out[1] = a1_s; out[2] = a1_p; out[3] = a1_rc;
    out[4] = a2_al; out[5] = a2_s; out[6] = a2_p; out[7] = a2_rc;
}

/* BowtieMapq3 / BowtieMapq2, single-end (nvBowtie/bowtie2/cuda/mapq.h:32-135 / :139-297) */
int orc_mapq(int version, int monotone, int32_t perfect_score, int32_t minimum_score, int32_t best_score, int has_second, int32_t second_score)
{
    const float max_score = (float)perfect_score, min_score = (float)minimum_score;
    if (version == 3)
    {
        static const int unpaired_one[11]         = { 43, 42, 41, 36, 32, 27, 20, 11, 4, 1, 0 };
        static const int unpaired_two_perfect[11] = { 2, 16, 23, 30, 31, 32, 34, 36, 38, 40, 42 };
        static const int unpaired_two[11][11] = {
            {  2,  2,  2,  1,  1, 0, 0, 0, 0, 0, 0 }, { 20, 14,  7,  3,  2, 1, 0, 0, 0, 0, 0 }, { 20, 16, 10,  6,  3, 1, 0, 0, 0, 0, 0 },
            { 20, 17, 13,  9,  3, 1, 1, 0, 0, 0, 0 }, { 21, 19, 15,  9,  5, 2, 2, 0, 0, 0, 0 }, { 22, 21, 16, 11, 10, 5, 0, 0, 0, 0, 0 },
            { 23, 22, 19, 16, 11, 0, 0, 0, 0, 0, 0 }, { 24, 25, 21, 30,  0, 0, 0, 0, 0, 0, 0 }, { 30, 26, 29,  0,  0, 0, 0, 0, 0, 0, 0 },
            { 30, 27,  0,  0,  0, 0, 0, 0, 0, 0, 0 }, { 30,  0,  0,  0,  0, 0, 0, 0, 0, 0, 0 } };
        const float norm_factor = 10.0f / (max_score - min_score);
        if ((float)best_score < min_score) return 0;
        const int best = imax( (int)max_score - best_score, 0 );
        int best_bin = (int)((float)best * norm_factor + 0.5f);
        best_bin = best_bin < 0 ? 0 : (best_bin > 10 ? 10 : best_bin);
        if (has_second)
        {
            const int diff = best_score - second_score;
            int diff_bin = (int)((float)diff * norm_factor + 0.5f);
            diff_bin = diff_bin < 0 ? 0 : (diff_bin > 10 ? 10 : diff_bin);
            return ((float)best == max_score) ? unpaired_two_perfect[best_bin] : unpaired_two[diff_bin][best_bin];
        }
        return ((float)best == max_score) ? 44 : unpaired_one[best_bin];
    }
    const float diff = max_score - min_score;
    const float best = (float)best_score;
    if (best < min_score) return 0;
    const float best_over = best - min_score;
    const float sb = (float)second_score;
    const float best_diff = has_second ? ((best < 0 ? -best : best) - (sb < 0 ? -sb : sb)) : 0.0f;
    const float bd = best_diff < 0 ? -best_diff : best_diff;
    if (monotone)
    {
        if (!has_second)
        {
            if      (best_over >= diff * 0.8f) return 42;
            else if (best_over >= diff * 0.7f) return 40;
            else if (best_over >= diff * 0.6f) return 24;
            else if (best_over >= diff * 0.5f) return 23;
            else if (best_over >= diff * 0.4f) return 8;
            else if (best_over >= diff * 0.3f) return 3;
            return 0;
        }
        if      (bd >= diff * 0.9f) return (best_over == diff) ? 39 : 33;
        else if (bd >= diff * 0.8f) return (best_over == diff) ? 38 : 27;
        else if (bd >= diff * 0.7f) return (best_over == diff) ? 37 : 26;
        else if (bd >= diff * 0.6f) return (best_over == diff) ? 36 : 22;
        else if (bd >= diff * 0.5f) { if (best_over == diff) return 35; if (best_over >= diff * 0.84f) return 25; if (best_over >= diff * 0.68f) return 16; return 5; }
        else if (bd >= diff * 0.4f) { if (best_over == diff) return 34; if (best_over >= diff * 0.84f) return 21; if (best_over >= diff * 0.68f) return 14; return 4; }
        else if (bd >= diff * 0.3f) { if (best_over == diff) return 32; if (best_over >= diff * 0.88f) return 18; if (best_over >= diff * 0.67f) return 15; return 3; }
        else if (bd >= diff * 0.2f) { if (best_over == diff) return 31; if (best_over >= diff * 0.88f) return 17; if (best_over >= diff * 0.67f) return 11; return 0; }
        else if (bd >= diff * 0.1f) { if (best_over == diff) return 30; if (best_over >= diff * 0.88f) return 12; if (best_over >= diff * 0.67f) return 7; return 0; }
        else if (bd > 0)            return (best_over >= diff * 0.67f) ? 6 : 2;
        return (best_over >= diff * 0.67f) ? 1 : 0;
    }
    if (!has_second)
    {
        if      (best_over >= diff * 0.8f) return 44;
        else if (best_over >= diff * 0.7f) return 42;
        else if (best_over >= diff * 0.6f) return 41;
        else if (best_over >= diff * 0.5f) return 36;
        else if (best_over >= diff * 0.4f) return 28;
        else if (best_over >= diff * 0.3f) return 24;
        return 22;
    }
    if      (bd >= diff * 0.9f) return 40;
    else if (bd >= diff * 0.8f) return 39;
    else if (bd >= diff * 0.7f) return 38;
    else if (bd >= diff * 0.6f) return 37;
    else if (bd >= diff * 0.5f) { if (best_over == diff) return 35; return (best_over >= diff * 0.50f) ? 25 : 20; }
    else if (bd >= diff * 0.4f) { if (best_over == diff) return 34; return (best_over >= diff * 0.50f) ? 21 : 19; }
    else if (bd >= diff * 0.3f) { if (best_over == diff) return 33; return (best_over >= diff * 0.5f) ? 18 : 16; }
    else if (bd >= diff * 0.2f) { if (best_over == diff) return 32; return (best_over >= diff * 0.5f) ? 17 : 12; }
    else if (bd >= diff * 0.1f) { if (best_over == diff) return 31; return (best_over >= diff * 0.5f) ? 14 : 9; }
    else if (bd > 0)            return (best_over >= diff * 0.5f) ? 11 : 2;
    return (best_over >= diff * 0.5f) ? 1 : 0;
}


/* =====================================================================================================================
 * nvBowtie's seed-hit deques, hit selection and effort-limited reduction.
 *
 * SeedHit (nvBowtie/bowtie2/cuda/seed_hit.h:45-218): 8 bytes, word 0 = range_begin, word 1 = range_delta:20 | pos:10 | rc:1 | indexdir:1
 * (bit 0 up), the SA range EXCLUSIVE at its end.  A read's hits live in a priority_deque<SeedHit, ., hit_compare>
 * (seed_hit_deque_array.h:145, seed_hit.h:220-229: f goes before s iff size(f) > size(s)), the interval heap of
 * nvbio/basic/priority_deque.h + interval_heap.h: element 0 is the "minimum" under that order -- the LARGEST range, what
 * pop_bottom() drops when the deque is full -- and top() = element 1 (element 0 of a one-element deque) the SMALLEST range, what
 * select pops rows from.  Elements of equal size are ordered by the heap's own moves, so those moves are restated one by one
 * (pinned on the reference's priority_deque itself: oracle/ref ref_hit_deque_run, tests/golden/deque_golden.npz).
 * ===================================================================================================================== */
static uint32_t hit_size(const orc_seed_hit* h) { return h->bits & 0xFFFFFu; }
static int hit_before(const orc_seed_hit* f, const orc_seed_hit* s) { return hit_size( f ) > hit_size( s ); }      /* hit_compare */
static void hit_swap(orc_seed_hit* a, uint32_t i, uint32_t j) { const orc_seed_hit t = a[i]; a[i] = a[j]; a[j] = t; }

/* sift_up<left_bound> (interval_heap.h:370-391): towards the root along the low (left) or high ends */
static void heap_up(orc_seed_hit* a, uint32_t i, int low, uint32_t limit)
{
    while (i >= limit)
    {
        const uint32_t par = ((i / 2u - 1u) | 1u) ^ (low ? 1u : 0u);
        if (low ? hit_before( &a[i], &a[par] ) : hit_before( &a[par], &a[i] )) { hit_swap( a, i, par ); i = par; }
        else break;
    }
}
/* sift_leaf_max / sift_leaf_min (:394-439): a leaf first against the other end of its interval, then up */
static void heap_leaf_high(orc_seed_hit* a, uint32_t n, uint32_t i, uint32_t limit)
{
    const uint32_t co = (i * 2u < n) ? i * 2u : (i ^ 1u);
    if (hit_before( &a[i], &a[co] )) { hit_swap( a, i, co ); heap_up( a, co, 1, limit ); }
    else heap_up( a, i, 0, limit );
}
static void heap_leaf_low(orc_seed_hit* a, uint32_t n, uint32_t i, uint32_t limit)
{
    uint32_t co = i | 1u;
    if (co >= n)
    {
        if (co == 1u) return;                                             /* a single element */
        co = (co / 2u - 1u) | 1u;
    }
    if (hit_before( &a[co], &a[i] )) { hit_swap( a, i, co ); heap_up( a, co, 0, limit ); }
    else heap_up( a, i, 1, limit );
}
/* sift_down<left_bound> (:442-503): the hole moves to a leaf along the smaller low / larger high children, then the leaf rules */
static void heap_down(orc_seed_hit* a, uint32_t n, uint32_t i, int low, uint32_t limit)
{
    const int32_t ep = (int32_t)(n / 2u) - ((low && (n & 3u) == 0u) ? 2 : 1);     /* one past the last element with two children (signed: -1 for tiny heaps) */
    while ((int32_t)i < ep)
    {
        uint32_t c = i * 2u + (low ? 2u : 1u);
        if (hit_before( &a[c + (low ? 2u : 0u)], &a[c + (low ? 0u : 2u)] )) c += 2u;
        hit_swap( a, i, c ); i = c;
    }
    if ((int32_t)i <= ep + (low ? 0 : 1))
    {
        uint32_t c = i * 2u + (low ? 2u : 1u);
        if (c < n)
        {
            if (!low && c + 1u < n && hit_before( &a[c], &a[c + 1u] ))
            {
                ++c; hit_swap( a, i, c ); i = c;
                heap_leaf_low( a, n, i, limit );
                return;
            }
            hit_swap( a, i, c ); i = c;
        }
    }
    if (low) heap_leaf_low( a, n, i, limit ); else heap_leaf_high( a, n, i, limit );
}

void orc_hit_deque_push(orc_seed_hit* a, uint32_t* n, orc_seed_hit x)           /* priority_deque::push -> push_interval_heap */
{
    a[*n] = x; ++*n;
    const uint32_t i = *n - 1u;
    if (i & 1u) heap_leaf_high( a, *n, i, 2u ); else heap_leaf_low( a, *n, i, 2u );
}
void orc_hit_deque_pop_bottom(orc_seed_hit* a, uint32_t* n)                      /* pop_interval_heap_min: drops the LARGEST range */
{
    const uint32_t last = *n - 1u;
    hit_swap( a, 0u, last );
    heap_down( a, last, 0u, 1, 2u );
    *n = last;
}
void orc_hit_deque_pop_top(orc_seed_hit* a, uint32_t* n)                         /* pop_interval_heap_max: drops the SMALLEST range */
{
    if (*n > 2u)
    {
        const uint32_t last = *n - 1u;
        hit_swap( a, 1u, last );
        heap_down( a, last, 1u, 0, 2u );
    }
    --*n;
}
uint32_t orc_hit_deque_top(uint32_t n) { return n > 1u ? 1u : 0u; }              /* priority_deque::maximum */

/* map_kernel's loop over the seeds of ONE read with seed_mapper<EXACT_MAPPING> (mapping_inl.h:485-556,193-282; USE_REVERSE_INDEX 0),
 * given what match_range returned for every seed: fw[j] = the forward scan over the stored (reversed) read, rc[j] = the reverse
 * scan complemented, both INCLUSIVE ranges, empty iff x > y (a seed with an N has both empty, :235-236).
 * seed j starts `seed_off[j]` symbols into the stored read.  Returns the reseeding decision (:547-556). */
int orc_map_exact_read(const uint32_t* fw, const uint32_t* rc, const uint32_t* seed_off, uint32_t n_seeds, uint32_t read_len, uint32_t seed_len,
                       uint32_t max_hits, uint32_t rep_seeds, orc_seed_hit* deque, uint32_t* deque_size)
{
    uint32_t n = 0, range_sum = 0, range_count = 0;
    for (uint32_t j = 0; j < n_seeds; ++j)
    {
        for (int strand = 0; strand < 2; ++strand)
        {
            const uint32_t x = strand ? rc[2 * j] : fw[2 * j], y = strand ? rc[2 * j + 1] : fw[2 * j + 1];
            if (x > y) continue;
            /* SeedHit::build_flags( STANDARD, FORWARD, read_range.y - pos - seed_len ) / ( COMPLEMENT, FORWARD, pos - read_range.x ) */
            const uint32_t pos = strand ? seed_off[j] : read_len - seed_off[j] - seed_len;
            orc_seed_hit h;
            h.begin = x;
            h.bits  = ((y + 1u - x) & 0xFFFFFu) | ((pos & 0x3FFu) << 20) | ((uint32_t)strand << 30);     /* inclusive_to_exclusive */
            if (n == max_hits) orc_hit_deque_pop_bottom( deque, &n );
            orc_hit_deque_push( deque, &n, h );
            range_sum += y - x + 1u; ++range_count;
        }
    }
    *deque_size = n;
    return (range_count == 0u || range_sum >= rep_seeds * range_count) ? 1 : 0;
}

/* select_kernel for ONE active read (select_inl.h:62-130): the next SA row of the read's top hit.
 * Returns 0 when the read leaves the active queue (no hits left); else 1 with *sa_pos, *packed_seed
 * (defs.h:162-172: pos_in_read:12 | index_dir:1 | rc:1 | top_flag:1) and the updated *top_flag.  (context.stop, i.e. trys == 0,
 * is the caller's test.) */
int orc_select_read(orc_seed_hit* deque, uint32_t* size, uint32_t* top_flag, uint32_t* sa_pos, uint32_t* packed_seed)
{
    if (*size == 0u) return 0;
    orc_seed_hit* hit = &deque[orc_hit_deque_top( *size )];
    if (hit_size( hit ) == 0u)                                           /* get_range().x >= get_range().y: exhausted */
    {
        orc_hit_deque_pop_top( deque, size );
        if (*size == 0u) return 0;
        hit = &deque[orc_hit_deque_top( *size )];
        *top_flag = 0u;
    }
    *sa_pos = hit->begin;                                                /* pop_front(): begin++, delta-- */
    hit->begin += 1u;
    hit->bits = (hit->bits & ~0xFFFFFu) | ((hit_size( hit ) - 1u) & 0xFFFFFu);
    const uint32_t pos = (hit->bits >> 20) & 0x3FFu, rcf = (hit->bits >> 30) & 1u, dir = (hit->bits >> 31) & 1u;
    *packed_seed = pos | (dir << 12) | (rcf << 13) | (*top_flag << 14);
    return 1;
}

/* score_reduce_kernel for ONE hit of one read with ReduceBestApproxContext (reduce_inl.h:65-140, reduce.h:55-99).
 * best = { a1 score, a1 pos, a1 rc, a2 score, a2 pos, a2 rc } (positions 0xFFFFFFFF = unaligned, scores start at worst_score:
 * aligner.h:279-301); *trys the read's effort counter; ext = extensions done before this pass + index of the hit in the pass.
 * Returns 1 if the read's deque must be erased (pipeline.hits.erase: the search stops). */
int orc_score_reduce_effort(int64_t best[6], uint32_t* trys, int32_t score, uint32_t g_pos, uint32_t read_rc, uint32_t top_flag, uint32_t read_len,
                            uint32_t ext, uint32_t max_effort, uint32_t min_ext, uint32_t max_ext)
{
    if ((read_rc == (uint32_t)best[2] && g_pos == (uint32_t)best[1]) || (read_rc == (uint32_t)best[5] && g_pos == (uint32_t)best[4])) return 0;
    if (score > (int32_t)best[0])
    {
        *trys = max_effort;                                              /* context.best_score */
        best[3] = best[0]; best[4] = best[1]; best[5] = best[2];
        best[0] = score; best[1] = g_pos; best[2] = read_rc;
    }
    else if (score > (int32_t)best[3] && distinct_alignments( (uint32_t)best[1], (int)best[2], g_pos, (int)read_rc, read_len / 2u ))
    {
        *trys = max_effort;                                              /* context.second_score */
        best[3] = score; best[4] = g_pos; best[5] = read_rc;
    }
    else if (*trys > 0u)                                                 /* context.failure */
    {
        if ((ext >= min_ext && top_flag == 0u && --*trys == 0u) || ext >= max_ext) return 1;
    }
    return 0;
}

/* a sequence of deque operations (the driver of oracle/ref ref_hit_deque_run, same op codes: 0 push under the max_hits rule,
 * 1 pop_top, 2 pop_bottom, 3 select's in-place row pop) -- for pinning the heap moves on the reference's container */
void orc_hit_deque_run(const uint32_t* ops, const uint32_t* begins, const uint32_t* bits, uint32_t n_ops, uint32_t max_hits,
                       uint32_t* heap_out, uint32_t* size_out, uint32_t* out_rows)
{
    orc_seed_hit* a = (orc_seed_hit*)malloc( ((size_t)n_ops + 1u) * sizeof(orc_seed_hit) );
    uint32_t n = 0;
    for (uint32_t k = 0; k < n_ops; ++k)
    {
        out_rows[k] = 0xFFFFFFFFu;
        if (ops[k] == 0u)
        {
            const orc_seed_hit h = { begins[k], bits[k] };
            if (n == max_hits) orc_hit_deque_pop_bottom( a, &n );
            orc_hit_deque_push( a, &n, h );
        }
        else if (ops[k] == 1u) { if (n) orc_hit_deque_pop_top( a, &n ); }
        else if (ops[k] == 2u) { if (n) orc_hit_deque_pop_bottom( a, &n ); }
        else
        {
            uint32_t top_flag = 1u, row = 0xFFFFFFFFu, seed = 0u;
            if (orc_select_read( a, &n, &top_flag, &row, &seed )) out_rows[k] = row;
        }
    }
    *size_out = n;
    for (uint32_t k = 0; k < n; ++k) { heap_out[2 * k] = a[k].begin; heap_out[2 * k + 1] = a[k].bits; }
    free( a );
}


/* ---------------------------------------------------------------------------------------------------------------------
 * nvBowtie's approximate seed mapper, seed_mapper<APPROX_MAPPING> (nvBowtie/bowtie2/cuda/mapping_inl.h:114-184,288-342): every seed is
 * searched four times -- forwards in the forward index and backwards in the index of the REVERSED text, as it stands and
 * complemented -- each search matching its first half exactly and allowing ONE substitution in the rest (`map`), the searches that
 * start from the seed's far end leaving the exact match to the other one.  Device-only source: parity unpinned; rank / rank4 / the
 * deque are pinned.
 * ------------------------------------------------------------------------------------------------------------------- */
/* match_range (mapping_inl.h:73-86): backward-search steps over query[begin, end) in that order */
static void map_match_range(const orc_fm_index* f, uint32_t range[2], const uint8_t* q, uint32_t begin, uint32_t end)
{
    for (uint32_t i = begin; i < end && range[0] <= range[1]; ++i)
    {
        const uint8_t c = q[i];
        if (c > 3) { range[0] = 1u; range[1] = 0u; return; }
        uint32_t r[2];
        orc_rank2( f, range[0] - 1u, range[1], c, r );
        range[0] = f->L2[c] + r[0] + 1u; range[1] = f->L2[c] + r[1];
    }
}
static void map_push(orc_seed_hit* deque, uint32_t* n, uint32_t max_hits, const uint32_t range[2], uint32_t flags, uint32_t* sum, uint32_t* count)
{
    orc_seed_hit h;
    h.begin = range[0];
    h.bits  = ((range[1] + 1u - range[0]) & 0xFFFFFu) | flags;
    if (*n == max_hits) orc_hit_deque_pop_bottom( deque, n );
    orc_hit_deque_push( deque, n, h );
    *sum += range[1] - range[0] + 1u; ++*count;
}
/* map<find_exact> (:114-184) */
static void map_one_mismatch(const orc_fm_index* f, const uint8_t* q, uint32_t len1, uint32_t len2, int find_exact, uint32_t flags,
                             orc_seed_hit* deque, uint32_t* n, uint32_t max_hits, uint32_t* sum, uint32_t* count)
{
    uint32_t base[2] = { 0u, f->length };
    map_match_range( f, base, q, 0u, len1 );
    for (uint32_t i = len1; i < len2 && base[0] <= base[1]; ++i)
    {
        const uint8_t c = q[i];
        uint32_t lo[4], hi[4];
        orc_rank4( f, base[0] - 1u, lo ); orc_rank4( f, base[1], hi );
        for (uint8_t sub = 0; sub < 4; ++sub)
            if (sub != c && hi[sub] > lo[sub])
            {
                uint32_t range[2] = { f->L2[sub] + lo[sub] + 1u, f->L2[sub] + hi[sub] };
                map_match_range( f, range, q, i + 1u, len2 );
                if (range[0] <= range[1]) map_push( deque, n, max_hits, range, flags, sum, count );
            }
        if (c < 4) { base[0] = f->L2[c] + lo[c] + 1u; base[1] = f->L2[c] + hi[c]; }
        else       { base[0] = 1u; base[1] = 0u; break; }
    }
    if (find_exact && base[0] <= base[1]) map_push( deque, n, max_hits, base, flags, sum, count );
}
/* map_kernel's loop over the seeds of ONE read with seed_mapper<APPROX_MAPPING>; stored = the read as nvBowtie stores it (reversed),
 * one symbol per byte; f = forward index, rf = the index of the reversed text.  Returns the reseeding decision. */
int orc_map_approx_read(const orc_fm_index* f, const orc_fm_index* rf, const uint8_t* stored, uint32_t read_len, const uint32_t* seed_off, uint32_t n_seeds,
                        uint32_t seed_len, uint32_t max_hits, uint32_t rep_seeds, orc_seed_hit* deque, uint32_t* deque_size)
{
    uint32_t n = 0, sum = 0, count = 0;
    uint8_t fq[64], rq[64], cfq[64], crq[64];
    for (uint32_t j = 0; j < n_seeds; ++j)
    {
        const uint32_t pos = seed_off[j];                                  /* pos - read_range.x */
        for (uint32_t k = 0; k < seed_len; ++k)
        {
            fq[k] = stored[pos + k]; rq[k] = stored[pos + seed_len - 1u - k];
            cfq[k] = fq[k] < 4 ? 3 - fq[k] : fq[k]; crq[k] = rq[k] < 4 ? 3 - rq[k] : rq[k];
        }
        /* flags: pos:10 << 20 | rc << 30 | indexdir << 31 (SeedHit::build_flags( readtype, indexdir, pos )) */
        map_one_mismatch( f,  fq,  seed_len / 2u,        seed_len, 1, (((read_len - pos - seed_len) & 0x3FFu) << 20),                         deque, &n, max_hits, &sum, &count );
        map_one_mismatch( rf, rq,  (seed_len + 1u) / 2u, seed_len, 0, (((read_len - pos - 1u) & 0x3FFu) << 20) | (1u << 31),                  deque, &n, max_hits, &sum, &count );
        map_one_mismatch( rf, cfq, seed_len / 2u,        seed_len, 1, (((pos + seed_len - 1u) & 0x3FFu) << 20) | (1u << 30) | (1u << 31),     deque, &n, max_hits, &sum, &count );
        map_one_mismatch( f,  crq, (seed_len + 1u) / 2u, seed_len, 0, ((pos & 0x3FFu) << 20) | (1u << 30),                                    deque, &n, max_hits, &sum, &count );
    }
    *deque_size = n;
    return (count == 0u || sum >= rep_seeds * count) ? 1 : 0;
}


/* ---------------------------------------------------------------------------------------------------------------------
 * The GENERIC rank dictionary (nvbio/fmindex/rank_dictionary_inl.h:33-66 build_occurrence_table, :206-336 dispatch_rank over plain
 * words, :482-539 rank / rank4): 2-bit big-endian text in 32- or 64-bit words (symbol i at bits [W-2-2(i mod W/2), +2) of word
 * i / (W/2)), occ[4 k + c] = # c in text[0, k K), indices and counts of 32 or 64 bits.
 * ------------------------------------------------------------------------------------------------------------------- */
static uint32_t gen_symbol(const void* text, uint32_t word_bits, uint64_t i)
{
    if (word_bits == 32) { const uint32_t w = ((const uint32_t*)text)[i >> 4]; return (w >> (30u - 2u * (uint32_t)(i & 15u))) & 3u; }
    const uint64_t w = ((const uint64_t*)text)[i >> 5]; return (uint32_t)(w >> (62u - 2u * (uint32_t)(i & 31u))) & 3u;
}
/* build_occurrence_table<K>: occ entries as index_bits-wide words; cnt[c] = totals */
void orc_rank_generic_build(const void* text, uint32_t word_bits, uint64_t length, uint32_t K, uint32_t index_bits, void* occ, uint64_t cnt[4])
{
    uint64_t counters[4] = { 0, 0, 0, 0 };
    for (uint64_t i = 0; i < length; ++i)
    {
        if ((i & (K - 1u)) == 0u)
            for (uint32_t c = 0; c < 4; ++c)
            {
                if (index_bits == 32) ((uint32_t*)occ)[(i / K) * 4u + c] = (uint32_t)counters[c];
                else                  ((uint64_t*)occ)[(i / K) * 4u + c] = counters[c];
            }
        ++counters[gen_symbol( text, word_bits, i )];
    }
    for (uint32_t c = 0; c < 4; ++c) cnt[c] = counters[c];
}
/* rank( dict, i, c ) = occurrences of c in text[0, i] (:276-292); i = all ones -> 0 */
uint64_t orc_rank_generic(const void* text, uint32_t word_bits, const void* occ, uint32_t index_bits, uint32_t K, uint64_t i, uint32_t c)
{
    const uint64_t minus1 = index_bits == 32 ? 0xFFFFFFFFull : ~0ull;
    if (i == minus1) return 0;
    const uint64_t k = i / K;
    uint64_t r = index_bits == 32 ? ((const uint32_t*)occ)[k * 4u + c] : ((const uint64_t*)occ)[k * 4u + c];
    for (uint64_t j = k * K; j <= i; ++j) r += gen_symbol( text, word_bits, j ) == c;      /* what the word-by-word popcounts add up to */
    return index_bits == 32 ? (uint32_t)r : r;
}

/* sw-benchmark's shape: every pattern against the whole of one text (sw-benchmark.cu:152,362-369), OpenMP over work items */
void orc_full_gotoh_many_to_one(int type, int blocking, const orc_gotoh_scheme* s, const uint8_t* pats, const uint8_t* quals, const uint32_t* po,
                                const uint8_t* text, uint32_t text_len, uint32_t n, int32_t min_score, int32_t* scores, uint32_t* sinks)
{
    #pragma omp parallel for schedule(dynamic,64)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        orc_full_gotoh( type, blocking, s, pats + po[i], quals ? quals + po[i] : 0, po[i+1] - po[i], text, text_len, min_score, scores + i, sinks + 2*i );
}


/* ---------------------------------------------------------------------------------------------------------------------
 * The Myers bit-vector aligner: aln::banded_alignment_score<BAND>( EditDistanceAligner<TYPE, MyersTag<5> >, ... ) =
 * banded_myers<BAND, 0, TYPE, 5> (nvbio/alignment/myers/myers_banded_inl.h:247-342) -- the aligner examples/fmmap/fmmap.cu:346-359
 * instantiates.  A band of BAND bits slides down the main diagonal: one diagonal step per text symbol while pattern symbols enter
 * the band, then horizontal steps; the score is MINUS the edit distance.  As the code behaves: `min_score` is taken as an int16
 * (:258: the caller's int32 is truncated -- Field_traits<int32>::min() becomes 0, so that only distance-0 cells get reported),
 * SEMI_GLOBAL reports every column of the horizontal phase that reaches it (BestSink keeps the last best), GLOBAL the final one.
 * Text symbols must be < 4 (the constructor of MyersBitVectors<5> leaves B[4] uninitialised, :139-144).
 * ------------------------------------------------------------------------------------------------------------------- */
static int myers_column(uint32_t band, uint32_t eq, uint32_t* VP, uint32_t* VN, int horizontal, int s)
{
    uint32_t X = eq | *VN;
    const uint32_t D0 = ((*VP + (X & *VP)) ^ *VP) | X;
    const uint32_t HN = *VP & D0;
    const uint32_t HP = *VN | ~(*VP | D0);
    X = D0 >> 1;
    *VN = X & HP;
    *VP = HN | ~(X | HP);
    if (!horizontal) return 1 - (int)((D0 >> (band - 1u)) & 1u);                  /* diagonal_column (:172-186) */
    return (int)((HP >> s) & 1u) - (int)((HN >> s) & 1u);                         /* horizontal_column (:194-208) */
}
int orc_banded_myers(uint32_t band, int type, const uint8_t* pat, uint32_t M, const uint8_t* txt, uint32_t N, int32_t min_score32,
                     int32_t* score, uint32_t sink[2])
{
    *score = -(1 << 30); sink[0] = sink[1] = 0xFFFFFFFFu;                         /* BestSink<int32>() */
    if (N < M) return 0;
    const int16_t min_score = (int16_t)min_score32;                              /* `const int16 min_score` (:258) */
    uint32_t B[5] = { 0, 0, 0, 0, 0 };
    uint32_t VP = 0xFFFFFFFFu, VN = 0u;
    int dist = 0;                                                                /* -C with C = 0 */
    const uint32_t last = (N - 1u < M) ? N - 1u : M;
    for (uint32_t i = 0; i < last; ++i)
    {
        for (int c = 0; c < 5; ++c) B[c] >>= 1;
        if (pat[i] < 5) B[pat[i]] |= 1u << (band - 1u);
        dist -= myers_column( band, B[txt[i]], &VP, &VN, 0, 0 );
    }
    int s = (int)band - 1 + (int)M - (int)last;
    for (uint32_t i = last; i < N && s >= 0; ++i)
    {
        for (int c = 0; c < 5; ++c) B[c] >>= 1;
        dist -= myers_column( band, B[txt[i]], &VP, &VN, 1, s );
        if (type == ORC_SEMI_GLOBAL && dist >= min_score && *score <= dist) { *score = dist; sink[0] = i + 1u; sink[1] = M; }
        --s;
    }
    if (type == ORC_GLOBAL && dist >= min_score && *score <= dist) { *score = dist; sink[0] = N; sink[1] = M; }
    return 1;
}
