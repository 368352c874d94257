// gotoh_full.hip -- batched full-matrix Gotoh scoring for gfx950 (sw-benchmark, opposite-mate windows).
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   gotoh_alignment_score_dispatch<8,TYPE,TextBlockingTag>      nvbio/alignment/gotoh/gotoh_inl.h:847-1256
//   gotoh_alignment_score_dispatch<8,TYPE,PatternBlockingTag>   nvbio/alignment/gotoh/gotoh_inl.h:444-841
//   boundary-column initialisation (GotohScoringContext::init)  gotoh_inl.h:56-74
//   save_boundary / save_Mth (semi-global / global reports)     nvbio/alignment/utils_inl.h:169-262
//   stripe early exit against min_score                         gotoh_inl.h:706-710,1106-1110
//   BestSink<int32>                                             nvbio/alignment/sink_inl.h:31-49
//   batched driver                                              nvbio/alignment/batched_inl.h:39-77
//
// The DP matrix is swept in stripes of 8 columns held in registers; the right-most column of a
// stripe is handed to the next stripe through a per-job boundary column of (H,E) pairs stored as
// two int16 -- exactly the reference's `short2` column, including its truncation -- laid out
// job-interleaved in HBM (element i of job t at [i * jobs + t]) so that the 64 lanes of a wave
// read and write 256 contiguous bytes per row.
#include "gotoh_common.h"
#include <type_traits>
#include "bitplanes.h"
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace nvbio_amd {

namespace {

// BestSink<int32>; with BEST2 also Best2Sink<int32>( dist ) (sink_inl.h:55-83): the second alignment ends more than dist text
// positions from the first, and a new best does not demote the old one
template <bool BEST2>
struct SinkT
{
    int32_t score; uint32_t x, y;
    int32_t score2; uint32_t x2, y2, dist;
    __device__ __forceinline__ void init(const uint32_t d = 0) { score = score2 = NVBIO_SCORE_MIN; x = y = x2 = y2 = 0xFFFFFFFFu; dist = d; }
    __device__ __forceinline__ void report(const int32_t s, const uint32_t sx, const uint32_t sy)
    {
        if (score <= s) { score = s; x = sx; y = sy; }          // last maximum wins
        else if (BEST2 && score2 <= s && ((uint32_t)(sx + dist) < x || sx > (uint32_t)(x + dist))) { score2 = s; x2 = sx; y2 = sy; }
    }
};
typedef SinkT<false> Sink;

// (H,E) boundary cell: two int16, truncating like make_vector<short>(H,E) (gotoh_inl.h:942)
__device__ __forceinline__ uint32_t pack_cell(const int32_t h, const int32_t e) { return ((uint32_t)h & 0xFFFFu) | ((uint32_t)e << 16); }
__device__ __forceinline__ int32_t  cell_h(const uint32_t c) { return (int32_t)(int16_t)(c & 0xFFFFu); }
__device__ __forceinline__ int32_t  cell_e(const uint32_t c) { return (int32_t)(int16_t)(c >> 16); }

constexpr int STRIPE = 8;

// TEXT_BLOCKING: stripes run over the text, the boundary column over the M pattern rows
// otherwise     : stripes run over the pattern, the boundary column over the N text rows
// WIDE: the linear-gap Smith-Waterman / edit-distance family (separate deletion / insertion terms, logical stripes of 16)
template <int TYPE, bool TEXT_BLOCKING, int RBITS, int TBITS, bool BEST2 = false, bool WIDE = false>
__global__ void __launch_bounds__(128)
full_gotoh_kernel(const BatchDev b, const SchemeDev sc, const uint32_t job_begin, const uint32_t jobs, const int32_t* __restrict__ min_scores,
                  uint32_t* __restrict__ column, int32_t* __restrict__ scores, uint2* __restrict__ sinks,
                  const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count,
                  const uint32_t distinct_dist = 0, int32_t* __restrict__ scores2 = nullptr, uint2* __restrict__ sinks2 = nullptr)
{
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // slot inside this launch
    if (t >= jobs) return;
    // with a job list (the jobs the ungapped pass could not settle) slot t is entry job_begin + t of the list
    if (job_list && job_begin + t >= *job_count) return;
    const uint32_t job = job_list ? job_list[job_begin + t] : job_begin + t;

    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = b.win_end[job] - tb;
    const int32_t  min_score = min_scores ? min_scores[job] : NVBIO_SCORE_MIN;

    SymbolReader<TBITS> trd( b.text );
    SymbolReader<RBITS> prd( b.reads );
    auto pattern = [&](const uint32_t i, uint32_t& q, uint32_t& qq) {
        const uint32_t idx = rev ? first + M - 1u - i : first + i;
        q = prd.get( idx );
        if (comp && q < 4u) q = 3u - q;
        qq = b.quals ? b.quals[idx] : 0u;
    };

    // F runs down the rows of a stripe, E along it: with pattern blocking the rows are text positions (F = the text advancing
    // alone), with text blocking they are pattern positions.  The Gotoh aligner charges the same terms to both (SchemeDev).
    const int32_t F_o = (WIDE && TEXT_BLOCKING) ? sc.ins_go : sc.pat_go, F_e = (WIDE && TEXT_BLOCKING) ? sc.ins_ge : sc.pat_ge;
    const int32_t E_o = !WIDE ? F_o : (TEXT_BLOCKING ? sc.pat_go : sc.ins_go), E_e = !WIDE ? F_e : (TEXT_BLOCKING ? sc.pat_ge : sc.ins_ge);
    const int32_t infimum = -32768 - (sc.pat_go < sc.pat_ge ? sc.pat_go : sc.pat_ge);   // gotoh_inl.h:634,1038
    const int32_t V = sc.match;
    constexpr bool wide = WIDE;                                 // logical stripes of 16 (the SW / edit-distance aligners)

    const uint32_t rows = TEXT_BLOCKING ? M : N;                // length of the boundary column
    const uint32_t cols = TEXT_BLOCKING ? N : M;                // extent the stripes cover
    uint32_t* col = column + t;                                 // element i at col[i * jobs]

    // GotohScoringContext::init (gotoh_inl.h:56-74)
    for (uint32_t i = 0; i < rows; ++i)
    {
        const int32_t x = TEXT_BLOCKING ? ((TYPE != NVBIO_LOCAL)  ? sc.txt_go + sc.txt_ge * (int32_t)i : 0)
                                        : ((TYPE == NVBIO_GLOBAL) ? sc.txt_go + sc.txt_ge * (int32_t)i : 0);
        const int32_t y = (TYPE == NVBIO_LOCAL) ? 0 : infimum;
        col[(size_t)i * jobs] = pack_cell( x, y );
    }

    SinkT<BEST2> sink; sink.init( distinct_dist );
    const uint32_t nb        = (cols + STRIPE - 1u) / STRIPE;
    const uint32_t end_block = (STRIPE * nb > (uint32_t)STRIPE) ? STRIPE * nb : (uint32_t)STRIPE;

    uint32_t c_sym[STRIPE];                                     // the stripe's symbols (text or pattern)
    int32_t  c_mm[STRIPE];                                      // pattern blocking: mismatch score per column
    #pragma unroll
    for (int j = 0; j < STRIPE; ++j) { c_sym[j] = 0; c_mm[j] = 0; }
    int32_t H[STRIPE + 1], F[STRIPE + 1];
    bool ok = true;

    for (uint32_t block = 0; block < end_block && ok; block += STRIPE)
    {
        const bool last = (block + STRIPE >= end_block);
        #pragma unroll
        for (int j = 0; j < STRIPE; ++j)
        {
            if (block + j < cols)
            {
                if (TEXT_BLOCKING) c_sym[j] = trd.get( tb + block + j );
                else { uint32_t q, qq; pattern( block + j, q, qq ); c_sym[j] = q; c_mm[j] = s_mm[qq < 63u ? qq : 63u]; }
            }
        }
        #pragma unroll
        for (int j = 0; j <= STRIPE; ++j)
        {
            const bool penal = TEXT_BLOCKING ? (TYPE == NVBIO_GLOBAL) : (TYPE != NVBIO_LOCAL);      // :1061-1066 / :676-681
            const int32_t t_o = WIDE ? sc.top_go : sc.pat_go, t_e = WIDE ? sc.top_ge : sc.pat_ge;
            H[j] = penal ? ((block + j > 0) ? t_o + t_e * (int32_t)(block + j - 1u) : 0) : 0;
            F[j] = infimum;
        }
        int32_t max_score = NVBIO_SCORE_MIN;
        int32_t temp_i    = H[0];

        for (uint32_t i = 0; i < rows; ++i)
        {
            uint32_t r_sym = 0; int32_t r_mm = 0;               // the row's symbol
            if (TEXT_BLOCKING) { uint32_t q, qq; pattern( i, q, qq ); r_sym = q; r_mm = s_mm[qq < 63u ? qq : 63u]; }
            else                 r_sym = trd.get( tb + i );

            // update_row (gotoh_inl.h:458-575 / :852-969)
            int32_t H_diag = temp_i;
            const uint32_t cell = col[(size_t)i * jobs];
            H[0] = temp_i = cell_h( cell );
            int32_t E = cell_e( cell );
            #pragma unroll
            for (int j = 1; j <= STRIPE; ++j)
            {
                F[j] = max2( F[j] + F_e, H[j] + F_o );
                E    = max2( E + E_e, H[j - 1] + E_o );
                const int32_t S = (c_sym[j - 1] == r_sym) ? V : (TEXT_BLOCKING ? r_mm : c_mm[j - 1]);
                int32_t hi = max3( E, F[j], H_diag + S );
                if (TYPE == NVBIO_LOCAL) hi = max2( hi, 0 );
                H_diag = H[j];
                H[j]   = hi;
            }
            col[(size_t)i * jobs] = pack_cell( H[STRIPE], E );
            max_score = max2( max_score, H[STRIPE] );

            if (TYPE == NVBIO_LOCAL && BEST2)
            {
                // Best2Sink depends on the order of the reports: cell by cell, as the reference (8-wide stripes only)
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j)
                    if (!last || block + j <= cols)
                    {
                        if (TEXT_BLOCKING) sink.report( H[j], block + j, i + 1u );
                        else               sink.report( H[j], i + 1u, block + j );
                    }
            }
            else if (TYPE == NVBIO_LOCAL)
            {
                // the row's cells are reported with j ascending and the LAST maximum wins: one packed
                // (score << 4 | j) max per cell and a single compare per row give the same sink
                int32_t key = -1;
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j)
                    if (!last || block + j <= cols) key = max2( key, (H[j] << 4) | j );
                if (key >= 0)
                {
                    const uint32_t j = (uint32_t)(key & 15);
                    const int32_t  v = key >> 4;
                    // 16-wide logical stripes report (stripe, row, column)-major while this sweep is 8 wide: a tie with a cell of
                    // the same logical stripe is won by the later ROW (the column is then larger too), any other tie by the newcomer
                    const uint32_t b_col = TEXT_BLOCKING ? sink.x : sink.y, b_row = TEXT_BLOCKING ? sink.y : sink.x;
                    const bool take = !wide || v != sink.score || ((b_col - 1u) >> 4) != (block >> 4) || i + 1u >= b_row;
                    if (take)
                    {
                        if (TEXT_BLOCKING) sink.report( v, block + j, i + 1u );
                        else               sink.report( v, i + 1u, block + j );
                    }
                }
            }
            else if (!TEXT_BLOCKING && last && TYPE == NVBIO_SEMI_GLOBAL)
            {
                // save_boundary -> save_Mth: the M-th column of this row (utils_inl.h:169-262)
                const uint32_t jm = ((M - 1u) & (STRIPE - 1u)) + 1u;
                int32_t v = 0;
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j) if ((uint32_t)j == jm) v = H[j];
                sink.report( v, i + 1u, M );
            }
        }

        if (!last)
        {
            if (TEXT_BLOCKING && TYPE == NVBIO_SEMI_GLOBAL)
            {
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j) sink.report( H[j], block + j, M );
            }
            const int32_t missing = (int32_t)(cols - block - STRIPE);
            // stripe early exit: after every 8-wide Gotoh stripe; after every 16-wide one for the SW family with pattern
            // blocking (sw/sw_inl.h:676-680), whose text-blocking sweep has no such test at all (:1063-1115)
            const bool test = !wide || (!TEXT_BLOCKING && (block & STRIPE));
            if (test && max_score + missing * V < min_score) ok = false;
        }
        else if (TEXT_BLOCKING)
        {
            if (TYPE == NVBIO_SEMI_GLOBAL)
            {
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j) if (block + j <= N) sink.report( H[j], block + j, M );
            }
            else if (TYPE == NVBIO_GLOBAL)
            {
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j) if (block + j == N) sink.report( H[j], block + j, M );
            }
        }
    }
    if (!TEXT_BLOCKING && ok && TYPE == NVBIO_GLOBAL)                                 // gotoh_inl.h:774-775
    {
        const uint32_t jm = ((M - 1u) & (STRIPE - 1u)) + 1u;
        int32_t v = 0;
        #pragma unroll
        for (int j = 1; j <= STRIPE; ++j) if ((uint32_t)j == jm) v = H[j];
        sink.report( v, N, M );
    }
    scores[job] = sink.score;
    sinks[job]  = make_uint2( sink.x, sink.y );
    if (BEST2) { scores2[job] = sink.score2; sinks2[job] = make_uint2( sink.x2, sink.y2 ); }
}

// ---------------------------------------------------------------------------------------------
// The COOPERATIVE full-matrix kernel: L lanes per job (gotoh_warp_inl.h:33-245 is the reference's several-threads-per-alignment form).
// full_gotoh_kernel gives a job to one lane and keeps the boundary column between stripes in memory: a batch of 100 k jobs (the sw-benchmark
// shape, BASELINE configs[0]) is 1.5 waves per SIMD, each waiting for its boundary cell row after row.  Here lane l of a job owns the W pattern
// columns [l W, (l + 1) W) for the WHOLE sweep over the text and runs l rows behind lane l - 1: at step t it computes text row t - l from the
// (H, E) its left neighbour left in that row one step earlier (one DPP row-shift each), so the boundary never leaves the registers, L times as
// many lanes are busy, and there is no scratch.  GLOBAL and SEMI_GLOBAL without the stripe early exit (min_scores = NULL): there the reference's
// result does not depend on how the matrix is swept -- the score of cell (N, M), or the last maximum of column M over ascending text positions --
// as long as no value leaves the int16 range of its boundary cells (the host checks (M + N) * largest step).  The two blockings differ only in
// which gap terms initialise which border (gotoh_inl.h:56-74 against :676-681 / :1061-1066): `text_blocking` picks them.
// ---------------------------------------------------------------------------------------------
template <int TYPE, int L, int W, int RBITS>
__global__ void __launch_bounds__(256)
full_gotoh_coop_kernel(const BatchDev b, const SchemeDev sc, const bool text_blocking, int32_t* __restrict__ scores, uint2* __restrict__ sinks)
{
    static_assert( TYPE != NVBIO_LOCAL && (L == 4 || L == 8 || L == 16), "GLOBAL / SEMI_GLOBAL; the lanes of a job inside one DPP row" );
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t gl  = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t l   = gl % L;
    const uint32_t jid = gl / L;
    const bool     valid = jid < b.n;
    const uint32_t job = valid ? jid : b.n - 1u;                // (lanes past the batch repeat its last job and report nothing)

    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = b.win_end[job] - tb;
    const bool     fits  = M >= 1u && M <= (uint32_t)(L * W);

    // border terms: a pattern prefix of length n scores po + pe (n - 1), a text prefix to + te (n - 1) (GLOBAL only; free otherwise)
    const int32_t po = text_blocking ? sc.txt_go : sc.pat_go, pe = text_blocking ? sc.txt_ge : sc.pat_ge;
    const int32_t to = text_blocking ? sc.pat_go : sc.txt_go, te = text_blocking ? sc.pat_ge : sc.txt_ge;
    const int32_t go = sc.pat_go, ge = sc.pat_ge, V = sc.match;
    const int32_t infimum = -32768 - (sc.pat_go < sc.pat_ge ? sc.pat_go : sc.pat_ge);

    // my columns: symbols, mismatch scores, the cells of the row above the matrix
    const uint32_t c0 = l * (uint32_t)W;
    uint32_t psym[W]; int32_t pmm[W], H[W], F[W];
    {
        SymbolReader<RBITS> prd( b.reads );
        #pragma unroll
        for (int k = 0; k < W; ++k)
        {
            const uint32_t j = c0 + (uint32_t)k;
            uint32_t q = 4u, qq = 0u;                            // (columns past the pattern: a symbol that matches nothing)
            if (j < M)
            {
                const uint32_t idx = rev ? first + M - 1u - j : first + j;
                q = prd.get( idx );
                if (comp && q < 4u) q = 3u - q;
                qq = b.quals ? b.quals[idx] : 0u;
            }
            psym[k] = q; pmm[k] = s_mm[qq < 63u ? qq : 63u];
            H[k] = po + pe * (int32_t)j;                         // prefix length j + 1
            F[k] = infimum;
        }
    }
    int32_t h_diag_left = c0 ? po + pe * (int32_t)(c0 - 1u) : 0;  // H( row above, my first column - 1 ): the corner is 0
    int32_t h_out = 0, e_out = infimum;                          // my last column's (H, E) of the row I computed last

    // where column M lives
    const uint32_t lm = fits ? (M - 1u) / (uint32_t)W : 0u, km = fits ? (M - 1u) % (uint32_t)W : 0u;
    int32_t  best = NVBIO_SCORE_MIN; uint32_t best_x = 0xFFFFFFFFu;

    const uint32_t* __restrict__ twords = (const uint32_t*)b.text;
    uint32_t steps = fits ? N + (uint32_t)L - 1u : 0u;          // every lane of the wave runs the longest job's steps
    #pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor( (int)steps, o ); steps = v > steps ? v : steps; }

    for (uint32_t t = 0; t < steps; ++t)
    {
        // the left neighbour's last column, from the step before (row_shr:1; lane 0 of a job takes the matrix's first column instead)
        int32_t h_in = __builtin_amdgcn_update_dpp( 0, h_out, 0x111, 0xF, 0xF, false );
        int32_t e_in = __builtin_amdgcn_update_dpp( 0, e_out, 0x111, 0xF, 0xF, false );
        const uint32_t i = t - l;                                // my text row (wraps while I wait for my turn)
        if (l == 0u) { h_in = (TYPE == NVBIO_GLOBAL) ? to + te * (int32_t)i : 0; e_in = infimum; }
        if (i < N && fits)
        {
            const uint32_t tp = tb + i;
            const uint32_t r_sym = (twords[tp >> 4] >> (30u - 2u * (tp & 15u))) & 3u;
            int32_t H_diag = h_diag_left;
            h_diag_left = h_in;
            int32_t hl = h_in, E = e_in;
            #pragma unroll
            for (int k = 0; k < W; ++k)
            {
                F[k] = max2( F[k] + ge, H[k] + go );
                E    = max2( E + ge, hl + go );
                const int32_t S = (psym[k] == r_sym) ? V : pmm[k];
                const int32_t hi = max3( E, F[k], H_diag + S );
                H_diag = H[k];
                H[k]   = hi; hl = hi;
            }
            h_out = hl; e_out = E;
            if (TYPE == NVBIO_SEMI_GLOBAL && l == lm)
            {
                int32_t v = 0;
                #pragma unroll
                for (int k = 0; k < W; ++k) if ((uint32_t)k == km) v = H[k];
                if (best <= v) { best = v; best_x = i + 1u; }    // ascending text positions, the last maximum wins
            }
        }
    }
    if (valid && (l == lm || !fits))
    {
        if (!fits) { if (l == 0u) { scores[job] = NVBIO_SCORE_MIN; sinks[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); } return; }
        if (TYPE == NVBIO_GLOBAL)
        {
            // cell (N, M); with text blocking an empty text reports nothing (no stripe holds column N = 0)
            int32_t v = 0;
            #pragma unroll
            for (int k = 0; k < W; ++k) if ((uint32_t)k == km) v = H[k];
            if (N == 0u && text_blocking) { scores[job] = NVBIO_SCORE_MIN; sinks[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); }
            else { scores[job] = v; sinks[job] = make_uint2( N, M ); }
        }
        else
        {
            scores[job] = best;
            sinks[job]  = best_x == 0xFFFFFFFFu ? make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ) : make_uint2( best_x, M );
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ungapped shortcut for end-to-end (SEMI_GLOBAL, match = 0) full-matrix scoring: the same argument as
// ungapped_e2e31_kernel (gotoh_banded.hip) over ALL diagonals of the window.  An alignment with a gap scores
// at most G = max(gap opens) < 0; the ungapped alignments of the whole pattern are the diagonals d = 0..N-M,
// U_d = -P * #{i : read[i] != text[i+d]}, ending in text column M+d.  If U* = max_d U_d > G, the optimum is
// U*, reached in exactly the columns with U_d = U*, and both blockings report end columns in ascending order
// with "last maximum wins": sink = (M + largest such d, M).  The stripe early exit (gotoh_inl.h:706-710,
// 1106-1110) must provably stay silent, or the job goes to the DP: with pattern blocking the stripe maximum is
// over all text positions of a pattern prefix, and every prefix of the optimal alignment scores >= U* (match =
// 0), so U* >= min_score suffices; with text blocking it is over the pattern rows of one text column, where only
// the first row is known to score >= -P, so min_score <= -P is required (stripes left of the alignment's start
// can otherwise trip the test even though the optimum passes it -- the reference does exit there).
// Anything else (U* <= G, N < M, windows over 528 symbols) is flagged for the DP as well.
// Planes live in registers: 17 + 17 words of text, 6 x 3 of pattern; the text window slides one bit per
// diagonal (v_alignbit), 32 diagonals per statically indexed outer step.
// ---------------------------------------------------------------------------------------------
// bit planes of a job of the full-matrix shortcuts: pattern rows 0..M-1 (pl / ph = low / high bit, pn = "is N", pm = rows that exist),
// window symbols 0..N-1 (tl / th, 544 bits; bits at and past N are whatever follows the window)
template <int RBITS>
__device__ __forceinline__ void full_planes(const BatchDev& b, const uint32_t first, const uint32_t M, const bool rev, const bool comp,
                                            const uint32_t tb, const uint32_t N,
                                            uint32_t (&pl)[6], uint32_t (&ph)[6], uint32_t (&pn)[6], uint32_t (&pm)[6],
                                            uint32_t (&tl)[18], uint32_t (&th)[18])
{
    // ---- pattern planes (bitplanes.h): bit i = row i ----
    uint64_t rlo[3], rhi[3], rn[3];
    {
        ReadWords<RBITS> rw;
        load_read_words<RBITS>( b.reads, first, M, rw );
        read_planes192<RBITS>( rw, first, M, rev, comp, rlo, rhi, rn );
    }
    #pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const int32_t left = (int32_t)M - 64 * k;
        const uint64_t mask = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
        pl[2*k] = (uint32_t)rlo[k]; pl[2*k+1] = (uint32_t)(rlo[k] >> 32);
        ph[2*k] = (uint32_t)rhi[k]; ph[2*k+1] = (uint32_t)(rhi[k] >> 32);
        pm[2*k] = (uint32_t)mask;   pm[2*k+1] = (uint32_t)(mask >> 32);
        pn[2*k] = (uint32_t)rn[k] & pm[2*k]; pn[2*k+1] = (uint32_t)(rn[k] >> 32) & pm[2*k+1];
    }

    // ---- text planes: 544 bits, bit k = window symbol k (34 packed words cover 528 symbols at any offset) ----
    {
        const uint32_t* __restrict__ twords = (const uint32_t*)b.text;
        const uint32_t toff = tb & 15u;
        const uint32_t tw0 = tb >> 4, tw_last = (tb + N - 1u) >> 4;
        uint32_t lo16[35], hi16[35];
        #pragma unroll
        for (int j = 0; j < 34; ++j)
        {
            const uint32_t widx = tw0 + (uint32_t)j;
            const uint32_t w = __brev( twords[widx < tw_last ? widx : tw_last] );
            uint32_t lo = (w >> 1) & 0x55555555u, hi = w & 0x55555555u;
            lo = (lo | (lo >> 1)) & 0x33333333u; lo = (lo | (lo >> 2)) & 0x0F0F0F0Fu; lo = (lo | (lo >> 4)) & 0x00FF00FFu; lo = (lo | (lo >> 8)) & 0xFFFFu;
            hi = (hi | (hi >> 1)) & 0x33333333u; hi = (hi | (hi >> 2)) & 0x0F0F0F0Fu; hi = (hi | (hi >> 4)) & 0x00FF00FFu; hi = (hi | (hi >> 8)) & 0xFFFFu;
            lo16[j] = lo; hi16[j] = hi;
        }
        lo16[34] = 0; hi16[34] = 0;
        uint32_t a[18], c[18];
        #pragma unroll
        for (int k = 0; k < 17; ++k) { a[k] = lo16[2*k] | (lo16[2*k+1] << 16); c[k] = hi16[2*k] | (hi16[2*k+1] << 16); }
        a[17] = 0; c[17] = 0;
        // drop the toff (< 16) symbols in front of the window
        #pragma unroll
        for (int k = 0; k < 17; ++k)
        {
            tl[k] = __builtin_amdgcn_alignbit( a[k + 1], a[k], toff );
            th[k] = __builtin_amdgcn_alignbit( c[k + 1], c[k], toff );
        }
        tl[17] = 0; th[17] = 0;
    }

}

// MODE 0: the pass over every job: best diagonal, settled or not; a job the second chance could still settle is flagged need_dp = 3 with its
//   best diagonal stashed in scores / sinks (inside this pass every wave paid for the second chance's own 350-diagonal scan whenever one of
//   its lanes asked for it, i.e. always: 15.8 -> 9 ms per 10 M opposite-mate windows with it moved out).
// MODE 1: the pass over the dense list of those jobs (job_list / job_count on the device): the second chance; each ends as 0 or 1.
template <int RBITS, int MODE = 0>
__global__ void __launch_bounds__(256)
ungapped_full_e2e_kernel(const BatchDev b, const int32_t P, const int32_t G, const int32_t gap_open, const int32_t gap_ext, const bool second_chance,
                         const int32_t* __restrict__ min_scores, const bool text_blocking,
                         int32_t* __restrict__ scores, uint2* __restrict__ sinks, uint8_t* __restrict__ need_dp, const bool stash = false,
                         const uint32_t* __restrict__ job_list = nullptr, const uint32_t* __restrict__ job_count = nullptr)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == 0 ? slot >= b.n : slot >= *job_count) return;
    const uint32_t job = MODE == 0 ? slot : job_list[slot];
    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = b.win_end[job] - tb;
    if (M == 0u || M > 161u || N < M || N > 528u) { need_dp[job] = 1; return; }

    uint32_t pl[6], ph[6], pn[6], pm[6], tl[18], th[18];
    full_planes<RBITS>( b, first, M, rev, comp, tb, N, pl, ph, pn, pm, tl, th );

    const uint32_t last_d = N - M;                               // diagonals 0..N-M end inside the window
    uint32_t best_cnt = 0xFFFFFFFFu, best_d = 0;
    if (MODE == 1)
    {
        // the first pass left this job's best diagonal in scores / sinks
        best_cnt = P > 0 ? (uint32_t)(-scores[job] / P) : 0u;
        best_d   = sinks[job].x - M;
    }
    #pragma unroll
    for (int wo = 0; wo < 17; ++wo)                              // 32 diagonals per step; d <= N - M <= 527
    {
        if (MODE == 1) break;
        if ((uint32_t)wo * 32u > last_d) break;
        uint32_t ql[7], qh[7];
        #pragma unroll
        for (int k = 0; k < 7; ++k) { ql[k] = (wo + k < 18) ? tl[wo + k] : 0u; qh[k] = (wo + k < 18) ? th[wo + k] : 0u; }
        const uint32_t d_end = ((uint32_t)wo * 32u + 31u < last_d) ? (uint32_t)wo * 32u + 31u : last_d;
        for (uint32_t d = (uint32_t)wo * 32u; d <= d_end; ++d)
        {
            uint32_t cnt = 0;
            #pragma unroll
            for (int k = 0; k < 6; ++k)
            {
                const uint32_t mm = (((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) & pm[k]) | pn[k];
                cnt += (uint32_t)__popc( mm );
            }
            if (cnt <= best_cnt) { best_cnt = cnt; best_d = d; }
            #pragma unroll
            for (int k = 0; k < 6; ++k)
            {
                ql[k] = __builtin_amdgcn_alignbit( ql[k + 1], ql[k], 1u );
                qh[k] = __builtin_amdgcn_alignbit( qh[k + 1], qh[k], 1u );
            }
            ql[6] >>= 1; qh[6] >>= 1;
        }
    }
    const int64_t U = -(int64_t)P * (int64_t)best_cnt;
    const int32_t min_score = min_scores ? min_scores[job] : NVBIO_SCORE_MIN;
    const bool exit_silent = text_blocking ? ((int64_t)min_score <= -(int64_t)P) : (U >= (int64_t)min_score);
    bool settled = U > (int64_t)G;
    if (!settled && second_chance && (int64_t)G - P < U && 2 * (int64_t)G < U && U >= (int64_t)min_score && exit_silent)
    {
        // Second chance, as in ungapped_e2e31_kernel: one gap plus one mismatch and two gaps score below U*, so only
        // single-gap, mismatch-free alignments could reach it: text gap (diagonal d-g then d) iff lead_{d-g} + tail_d >= M,
        // pattern gap (d then d-g) iff lead_d + tail_{d-g} + g >= M, for the gap lengths g with open + (g-1) ext >= U*.
        // The matrix has no band, so a pattern gap may also sit at the very start or end of the read: tail_d + g >= M or
        // lead_d + g >= M on any diagonal.  (second_chance requires equal pattern / text gap costs: the boundary
        // rows and columns charge one or the other depending on the blocking.)
        int32_t gmax = 0;
        while (gmax < 5 && (int64_t)gap_open + (int64_t)gmax * gap_ext >= U) ++gmax;
        if (MODE == 0 && gmax >= 1 && gmax <= 4)
        {
            need_dp[job] = 3; scores[job] = (int32_t)U; sinks[job] = make_uint2( M + best_d, M );      // for the second-chance launch
            return;
        }
        if (gmax >= 1 && gmax <= 4)
        {
            uint32_t lead_prev[4] = { 0, 0, 0, 0 }, tail_prev[4] = { 0, 0, 0, 0 };
            bool gapped = false;
            // A pattern gap at an end of the read also lets the rest of it lie on a diagonal that is not wholly inside the window:
            // e symbols of the read inserted before the window's first symbol (diagonal -e) or after its last one (diagonal
            // N - M + e), e <= gmax.  Those 2 gmax diagonals take part like the others, their rows without a text symbol counting as
            // mismatches: the ones before the window here, seeding lead / tail of "the diagonals before d", the ones after it at the end
            // of the loop below.
            #pragma unroll
            for (int e = 4; e >= 1; --e)
                if (e <= gmax)
                {
                    uint32_t last = 0;
                    #pragma unroll
                    for (int k = 0; k < 6; ++k)
                    {
                        const uint32_t l = __builtin_amdgcn_alignbit( tl[k], k ? tl[k - 1] : 0u, 32u - (uint32_t)e );      // window symbol r - e at bit r
                        const uint32_t h = __builtin_amdgcn_alignbit( th[k], k ? th[k - 1] : 0u, 32u - (uint32_t)e );
                        uint32_t mm = (((pl[k] ^ l) | (ph[k] ^ h)) & pm[k]) | pn[k];
                        if (k == 0) mm |= ((1u << e) - 1u) & pm[0];                      // rows 0 .. e-1 have no text symbol
                        if (mm) last = 32u * k + 31u - (uint32_t)__builtin_clz( mm );
                    }
                    const uint32_t tail = M - 1u - last;                                 // (row 0 always counts: `last` is defined)
                    if (tail + (uint32_t)gmax >= M) gapped = true;                       // rows g .. M-1 match for some e <= g <= gmax
                    #pragma unroll
                    for (int k = 3; k > 0; --k) { lead_prev[k] = lead_prev[k - 1]; tail_prev[k] = tail_prev[k - 1]; }
                    lead_prev[0] = 0u; tail_prev[0] = tail;
                }
            const uint32_t last_e = last_d + (uint32_t)gmax;                             // ... through the diagonals that end past the window
            #pragma unroll
            for (int wo = 0; wo < 17; ++wo)
            {
                if ((uint32_t)wo * 32u > last_e || gapped) break;
                uint32_t ql[7], qh[7];
                #pragma unroll
                for (int k = 0; k < 7; ++k) { ql[k] = (wo + k < 18) ? tl[wo + k] : 0u; qh[k] = (wo + k < 18) ? th[wo + k] : 0u; }
                const uint32_t d_end = ((uint32_t)wo * 32u + 31u < last_e) ? (uint32_t)wo * 32u + 31u : last_e;
                for (uint32_t d = (uint32_t)wo * 32u; d <= d_end && !gapped; ++d)
                {
                    // rows fm .. M-1 of a diagonal past last_d have no text symbol: mismatches
                    const uint32_t fm = d > last_d ? (d - last_d < M ? M - (d - last_d) : 0u) : M;
                    // first / last mismatching row, searched from the two ends and only as far as needed: on a diagonal that is not (nearly)
                    // the read's own, word 0 and the top word already hold mismatches, and a word is evaluated by the wave only while some
                    // lane is still looking -- 2-3 word evaluations per diagonal instead of 12 (the values are the same)
                    uint32_t first = M, last = 0xFFFFFFFFu;
                    #pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (__any( first == M ))
                        {
                            const uint32_t force = fm >= 32u * k + 32u ? 0u : (fm <= 32u * k ? 0xFFFFFFFFu : (0xFFFFFFFFu << (fm - 32u * k)));
                            const uint32_t mm = ((((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) | force) & pm[k]) | pn[k];
                            if (first == M && mm) first = 32u * k + (uint32_t)__builtin_ctz( mm );
                        }
                    #pragma unroll
                    for (int k = 5; k >= 0; --k)
                        if (__any( last == 0xFFFFFFFFu ))
                        {
                            const uint32_t force = fm >= 32u * k + 32u ? 0u : (fm <= 32u * k ? 0xFFFFFFFFu : (0xFFFFFFFFu << (fm - 32u * k)));
                            const uint32_t mm = ((((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) | force) & pm[k]) | pn[k];
                            if (last == 0xFFFFFFFFu && mm) last = 32u * k + 31u - (uint32_t)__builtin_clz( mm );
                        }
                    const uint32_t lead = first;
                    const uint32_t tail = (last == 0xFFFFFFFFu) ? M : M - 1u - last;
                    #pragma unroll
                    for (int g = 1; g <= 4; ++g)
                        if (g <= gmax)
                        {
                            if (lead + (uint32_t)g >= M || tail + (uint32_t)g >= M) gapped = true;      // pattern gap at an end of the read
                            // (diagonal d - g exists for every g <= gmax: the ones before the window were seeded above)
                            if (lead_prev[g - 1] + tail >= M) gapped = true;
                            if (lead + tail_prev[g - 1] + (uint32_t)g >= M) gapped = true;
                        }
                    #pragma unroll
                    for (int k = 3; k > 0; --k) { lead_prev[k] = lead_prev[k - 1]; tail_prev[k] = tail_prev[k - 1]; }
                    lead_prev[0] = lead; tail_prev[0] = tail;
                    #pragma unroll
                    for (int k = 0; k < 6; ++k)
                    {
                        ql[k] = __builtin_amdgcn_alignbit( ql[k + 1], ql[k], 1u );
                        qh[k] = __builtin_amdgcn_alignbit( qh[k + 1], qh[k], 1u );
                    }
                    ql[6] >>= 1; qh[6] >>= 1;
                }
            }
            settled = !gapped;
        }
    }
    if (settled && U >= (int64_t)min_score && exit_silent)
    {
        scores[job] = (int32_t)U; sinks[job] = make_uint2( M + best_d, M ); need_dp[job] = 0;
    }
    else
    {
        need_dp[job] = 1;
        if (stash) { scores[job] = (int32_t)U; sinks[job] = make_uint2( M + best_d, M ); }      // for the narrow route (narrow_jobs_kernel)
    }
}

// ---------------------------------------------------------------------------------------------
// The narrow route of end-to-end full-matrix scoring (pattern blocking, match = 0, one mismatch penalty P, gap of g symbols =
// open + (g-1) ext < 0): most jobs the shortcut cannot settle hold ONE good alignment next to the best diagonal d* (an indel, or a few
// mismatches more than the shortcut takes), and the matrix has 350 diagonals of which the DP would fill every cell.  Instead:
//   1. the band-31 kernel (two jobs per lane, gotoh_banded.hip) scores the 31 diagonals around d*: S_b, a lower bound of the optimum S*
//      (every path inside the band is a path of the matrix, with the same score and the same end);
//   2. an alignment that scores >= S_b holds at most n = |S_b| / min(P, |open|) events (mismatches, gaps) and at most
//      g = (|S_b| - |open|) / |ext| + 1 gap symbols: its rows that are neither split it into at most n + 1 runs of exact matches on one
//      diagonal each, the longest of at least (M - n - g) / (n + 1) rows, and it stays within g diagonals of that run's diagonal;
//   3. so if that bound is >= 16 and the only diagonals of the window (those that hang over its ends by up to g symbols included)
//      with a run of 16 exact matches lie within 15 - g of the band's centre, every alignment that scores >= S_b lies inside the band:
//      S* = S_b, and the band's last maximum is the matrix's (both sinks keep the last of equal scores, text ends ascending).
// A job that fails any of it -- S_b below min_score, a long run elsewhere, a bound below 16 -- takes the DP as before.
// ---------------------------------------------------------------------------------------------
struct FlagIs3 { __host__ __device__ __forceinline__ uint8_t operator()(const uint8_t v) const { return v == 3u ? 1u : 0u; } };

__global__ void __launch_bounds__(256)
narrow_jobs_kernel(const BatchDev b, const uint8_t* __restrict__ need_dp, const uint2* __restrict__ stashed,
                   uint32_t* __restrict__ wb2, uint32_t* __restrict__ we2, uint8_t* __restrict__ route)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= b.n) return;
    const uint32_t rid = b.read_id ? b.read_id[job] : job;
    const uint32_t M   = b.read_offsets[rid + 1] - b.read_offsets[rid];
    const uint32_t tb  = b.win_begin[job];
    const uint32_t N   = b.win_end[job] - tb;
    const bool ok = need_dp[job] == 1 && M >= 1u && M <= 161u && N <= 528u && N >= M + 30u;      // (what the shortcut kernel looked at, and room for a band)
    uint32_t c = 15u;
    if (ok)
    {
        const uint32_t d = stashed[job].x - M;                   // the best diagonal
        const uint32_t hi = N - M - 15u;
        c = d < 15u ? 15u : (d > hi ? hi : d);
    }
    wb2[job] = tb + c - 15u; we2[job] = tb + c + 15u + M;       // (a window of M + 30 symbols also where it is not used)
    route[job] = ok ? 1 : 0;
}

template <int RBITS>
__global__ void __launch_bounds__(256)
narrow_check_kernel(const BatchDev b, const int32_t P, const int32_t gap_open, const int32_t gap_ext, const int32_t* __restrict__ min_scores,
                    const uint32_t* __restrict__ wb2, const int32_t* __restrict__ scores, uint2* __restrict__ sinks, uint8_t* __restrict__ need_dp,
                    const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= *job_count) return;
    const uint32_t job   = job_list[slot];
    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = b.win_end[job] - tb;
    const int32_t  L     = scores[job];                          // the band's score
    const int32_t  min_score = min_scores ? min_scores[job] : NVBIO_SCORE_MIN;
    // the bounds of step 2
    const int32_t a    = -L;
    const int32_t unit = P < -gap_open ? P : -gap_open;
    const int32_t n_ev = unit > 0 ? a / unit : 1 << 20;
    const int32_t g    = a < -gap_open ? 0 : (gap_ext < 0 ? (a + gap_open) / (-gap_ext) + 1 : 1 << 20);
    bool ok = L <= 0 && L >= min_score && g <= 15 && n_ev + g < (int32_t)M && ((int32_t)M - n_ev - g) / (n_ev + 1) >= 16;
    if (!__any( ok )) return;                                    // (need_dp stays 1)

    uint32_t pl[6], ph[6], pn[6], pm[6], tl[18], th[18];
    full_planes<RBITS>( b, first, M, rev, comp, tb, N, pl, ph, pn, pm, tl, th );

    const uint32_t c  = wb2[job] - tb + 15u;                     // the band's centre diagonal
    const uint32_t hw = ok ? 15u - (uint32_t)g : 0u;
    const uint32_t z_lo = c - hw, z_hi = c + hw;                 // runs on these diagonals keep an alignment inside the band
    const uint32_t last_d = N - M;
    bool far = false;
    // a run of 16 matching rows in the 192-bit mismatch vector mm (rows >= M and rows without a text symbol are set)
    auto run16 = [&](const uint32_t (&mm)[6]) -> bool {
        uint32_t z[7];
        #pragma unroll
        for (int k = 0; k < 6; ++k) z[k] = ~mm[k] & pm[k];
        z[6] = 0;
        #pragma unroll
        for (int sh = 1; sh <= 4; sh <<= 1)
        {
            #pragma unroll
            for (int k = 0; k < 6; ++k) z[k] &= __builtin_amdgcn_alignbit( z[k + 1], z[k], (uint32_t)sh );
        }
        uint32_t nz = z[0] | z[1] | z[2] | z[3] | z[4] | z[5];  // runs of 8
        if (!__any( nz != 0u )) return false;
        #pragma unroll
        for (int k = 0; k < 6; ++k) z[k] &= __builtin_amdgcn_alignbit( z[k + 1], z[k], 8u );
        nz = z[0] | z[1] | z[2] | z[3] | z[4] | z[5];
        return nz != 0u;
    };
    // diagonals -15 .. -1: window symbol r - e under row r, rows 0 .. e-1 without a text symbol (all of them lie outside the zone)
    #pragma unroll 1
    for (int e = 15; e >= 1; --e)                                // (a loop, not 15 copies: unrolled, the kernel needed all 256 VGPRs)
    {
        uint32_t mm[6];
        #pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            const uint32_t l = __builtin_amdgcn_alignbit( tl[k], k ? tl[k - 1] : 0u, 32u - (uint32_t)e );
            const uint32_t h = __builtin_amdgcn_alignbit( th[k], k ? th[k - 1] : 0u, 32u - (uint32_t)e );
            mm[k] = (((pl[k] ^ l) | (ph[k] ^ h)) & pm[k]) | pn[k];
        }
        mm[0] |= (1u << e) - 1u;
        if (run16( mm ) && e <= g) far = true;
    }
    // diagonals 0 .. N - M + g: those past N - M end beyond the window, their last rows have no text symbol
    const uint32_t last_e = last_d + (ok ? (uint32_t)g : 0u);
    #pragma unroll
    for (int wo = 0; wo < 17; ++wo)                              // (statically indexed: tl / th are register arrays)
    {
        if (!__any( (uint32_t)wo * 32u <= last_e && !far && ok )) break;
        uint32_t ql[7], qh[7];
        #pragma unroll
        for (int k = 0; k < 7; ++k) { ql[k] = (wo + k < 18) ? tl[wo + k] : 0u; qh[k] = (wo + k < 18) ? th[wo + k] : 0u; }
        #pragma unroll 1
        for (uint32_t dd = 0; dd < 32u; ++dd)
        {
            const uint32_t d = (uint32_t)wo * 32u + dd;
            if (!__any( d <= last_e && !far && ok )) break;
            const uint32_t fm = d > last_d ? (d - last_d < M ? M - (d - last_d) : 0u) : M;      // rows fm .. M-1 have no text symbol
            uint32_t mm[6];
            #pragma unroll
            for (int k = 0; k < 6; ++k)
            {
                const uint32_t force = fm >= 32u * k + 32u ? 0u : (fm <= 32u * k ? 0xFFFFFFFFu : (0xFFFFFFFFu << (fm - 32u * k)));
                mm[k] = ((((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) | force) & pm[k]) | pn[k];
            }
            const bool outside = d < z_lo || d > z_hi;
            if (run16( mm ) && outside && d <= last_e) far = true;
            #pragma unroll
            for (int k = 0; k < 6; ++k)
            {
                ql[k] = __builtin_amdgcn_alignbit( ql[k + 1], ql[k], 1u );
                qh[k] = __builtin_amdgcn_alignbit( qh[k + 1], qh[k], 1u );
            }
            ql[6] >>= 1; qh[6] >>= 1;
        }
    }
    if (ok && !far)
    {
        // the band's sink in the matrix's coordinates: x = text symbols up to the alignment's end
        sinks[job] = make_uint2( (wb2[job] - tb) + sinks[job].x, M );
        need_dp[job] = 0;
    }
}


// ---------------------------------------------------------------------------------------------
// 16-bit packed pattern-blocking kernel: one lane owns TWO jobs of the same shape (M = max_pattern_len,
// N = max_text_len -- the opposite-mate windows of a read batch), one in each half of every register, as
// banded_gotoh_band31_pk_kernel does for the banded DP.  The boundary column already is the reference's int16
// (H,E) pair; here the registers are too, which is exact as long as no real score can leave +-16000 (host
// check) -- the infimum cells stay representable: -32768 - min(Go,Ge) + Ge >= -32768.  Equality of the row's
// text symbol with the stripe's 8 pattern symbols comes from one XOR on 2-bit packed symbols per job.
// Jobs of any other shape go through full_gotoh_kernel (a second job list).
// ---------------------------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s pk(const int a, const int b) { v2s r; r.x = (short)a; r.y = (short)b; return r; }
__device__ __forceinline__ v2s pk_max(const v2s a, const v2s b) { return __builtin_elementwise_max( a, b ); }
__device__ __forceinline__ v2s pk_bits(const uint32_t u) { return __builtin_bit_cast( v2s, u ); }
__device__ __forceinline__ uint32_t bits_pk(const v2s v) { return __builtin_bit_cast( uint32_t, v ); }

template <int TYPE, int RBITS>
__global__ void __launch_bounds__(128)
full_gotoh_pb_pk_kernel(const BatchDev b, const SchemeDev sc, const uint32_t M, const uint32_t N, const uint32_t pair_begin, const uint32_t pairs,
                        const int32_t* __restrict__ min_scores, uint2* __restrict__ column, int32_t* __restrict__ scores, uint2* __restrict__ sinks,
                        const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count)
{
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // pair slot inside this launch
    if (t >= pairs) return;
    const uint32_t n_jobs = *job_count;
    const uint32_t slot0 = 2u * (pair_begin + t);
    if (slot0 >= n_jobs) return;

    uint32_t job[2], first[2], tb[2]; bool rev[2], comp[2], valid[2]; int32_t min_score[2];
    #pragma unroll
    for (int u = 0; u < 2; ++u)
    {
        valid[u] = slot0 + u < n_jobs;
        job[u]   = job_list[valid[u] ? slot0 + u : slot0];
        const uint32_t rid = b.read_id ? b.read_id[job[u]] : job[u];
        first[u] = b.read_offsets[rid];
        const uint32_t fl = b.flags ? b.flags[job[u]] : 0u;
        rev[u]  = (fl & NVBIO_READ_REVERSE) != 0;
        comp[u] = (fl & NVBIO_READ_COMPLEMENT) != 0;
        tb[u]   = b.win_begin[job[u]];
        min_score[u] = min_scores ? min_scores[job[u]] : NVBIO_SCORE_MIN;
    }

    SymbolReader<2> trd0( b.text ), trd1( b.text );
    SymbolReader<RBITS> prd0( b.reads ), prd1( b.reads );

    const int32_t G_o = sc.pat_go, G_e = sc.pat_ge;
    const int32_t infimum = -32768 - (G_o < G_e ? G_o : G_e);
    const v2s GO = pk( G_o, G_o ), GE = pk( G_e, G_e );
    const v2s INF = pk( infimum, infimum ), ZERO = pk( 0, 0 ), K16 = pk( 16, 16 );
    const int32_t V = sc.match;

    uint2* col = column + t;                                    // element i at col[i * pairs]: (cell of job 0, cell of job 1)
    for (uint32_t i = 0; i < N; ++i)                             // GotohScoringContext::init (gotoh_inl.h:56-74)
    {
        const int32_t x = (TYPE == NVBIO_GLOBAL) ? sc.txt_go + sc.txt_ge * (int32_t)i : 0;
        const int32_t y = (TYPE == NVBIO_LOCAL) ? 0 : infimum;
        const uint32_t c = pack_cell( x, y );
        col[(size_t)i * pairs] = make_uint2( c, c );
    }

    Sink sink[2]; sink[0].init(); sink[1].init();
    bool alive[2] = { true, true };
    const uint32_t nb        = (M + STRIPE - 1u) / STRIPE;
    const uint32_t end_block = (STRIPE * nb > (uint32_t)STRIPE) ? STRIPE * nb : (uint32_t)STRIPE;
    const uint32_t jm = ((M - 1u) & (STRIPE - 1u)) + 1u;

    v2s c_mm[STRIPE], c_dv[STRIPE];
    #pragma unroll
    for (int j = 0; j < STRIPE; ++j) { c_mm[j] = ZERO; c_dv[j] = ZERO; }
    uint32_t cs[2] = { 0, 0 }, cnot[2] = { 0x5555u, 0x5555u };  // stripe symbols, 2 bits each; bit 2j set where column j can never match
    v2s H[STRIPE + 1], F[STRIPE + 1];

    for (uint32_t block = 0; block < end_block && (alive[0] || alive[1]); block += STRIPE)
    {
        const bool last = (block + STRIPE >= end_block);
        #pragma unroll
        for (int j = 0; j < STRIPE; ++j)
        {
            if (block + j < M)
            {
                int mm2[2];
                #pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const uint32_t idx = rev[u] ? first[u] + M - 1u - (block + j) : first[u] + block + j;
                    uint32_t q = u ? prd1.get( idx ) : prd0.get( idx );
                    if (comp[u] && q < 4u) q = 3u - q;
                    const uint32_t qq = b.quals ? b.quals[idx] : 0u;
                    mm2[u] = s_mm[qq < 63u ? qq : 63u];
                    cs[u]   = (cs[u] & ~(3u << (2 * j))) | ((q & 3u) << (2 * j));
                    cnot[u] = (cnot[u] & ~(1u << (2 * j))) | ((q > 3u ? 1u : 0u) << (2 * j));
                }
                c_mm[j] = pk( mm2[0], mm2[1] );
                c_dv[j] = pk( V - mm2[0], V - mm2[1] );
            }
        }
        #pragma unroll
        for (int j = 0; j <= STRIPE; ++j)
        {
            const int h0 = (TYPE != NVBIO_LOCAL) ? ((block + j > 0) ? G_o + G_e * (int32_t)(block + j - 1u) : 0) : 0;     // :676-681
            H[j] = pk( h0, h0 ); F[j] = INF;
        }
        v2s max_score = pk( -32768, -32768 );
        v2s temp_i    = H[0];

        for (uint32_t i = 0; i < N; ++i)
        {
            const uint32_t r0 = trd0.get( tb[0] + i ), r1 = trd1.get( tb[1] + i );
            // equality of the row's text symbol with the stripe's 8 pattern symbols, job 0 at even bits, job 1 at odd bits
            const uint32_t x0 = cs[0] ^ (r0 * 0x5555u), x1 = cs[1] ^ (r1 * 0x5555u);
            const uint32_t e0 = ~(x0 | (x0 >> 1)) & 0x5555u & ~cnot[0];
            const uint32_t e1 = ~(x1 | (x1 >> 1)) & 0x5555u & ~cnot[1];
            const uint32_t EQ = e0 | (e1 << 1);

            v2s H_diag = temp_i;
            const uint2 cell = col[(size_t)i * pairs];
            H[0] = temp_i = pk( cell_h( cell.x ), cell_h( cell.y ) );
            v2s E = pk( cell_e( cell.x ), cell_e( cell.y ) );
            v2s key = pk( -1, -1 );
            #pragma unroll
            for (int j = 1; j <= STRIPE; ++j)
            {
                const v2s f = pk_max( F[j] + GE, H[j] + GO );
                F[j] = f;
                E = pk_max( E + GE, H[j - 1] + GO );
                const uint32_t eq01 = ((EQ >> (2 * (j - 1))) & 1u) | (((EQ >> (2 * (j - 1) + 1)) & 1u) << 16);
                const v2s d = H_diag + c_mm[j - 1] + pk_bits( eq01 ) * c_dv[j - 1];
                v2s hi = pk_max( pk_max( E, f ), d );
                if (TYPE == NVBIO_LOCAL) hi = pk_max( hi, ZERO );
                H_diag = H[j];
                H[j]   = hi;
                if (TYPE == NVBIO_LOCAL && (!last || block + j <= M)) key = pk_max( key, hi * K16 + pk( j, j ) );
            }
            col[(size_t)i * pairs] = make_uint2( pack_cell( H[STRIPE].x, E.x ), pack_cell( H[STRIPE].y, E.y ) );
            max_score = pk_max( max_score, H[STRIPE] );

            if (TYPE == NVBIO_LOCAL)
            {
                const int k0 = key.x, k1 = key.y;
                if (alive[0] && k0 >= 0) sink[0].report( k0 >> 4, i + 1u, block + (uint32_t)(k0 & 15) );
                if (alive[1] && k1 >= 0) sink[1].report( k1 >> 4, i + 1u, block + (uint32_t)(k1 & 15) );
            }
            else if (last && TYPE == NVBIO_SEMI_GLOBAL)
            {
                v2s v = ZERO;
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j) if ((uint32_t)j == jm) v = H[j];
                if (alive[0]) sink[0].report( v.x, i + 1u, M );
                if (alive[1]) sink[1].report( v.y, i + 1u, M );
            }
        }
        if (!last)
        {
            const int32_t missing = (int32_t)(M - block - STRIPE);
            if ((int32_t)max_score.x + missing * V < min_score[0]) alive[0] = false;     // stripe early exit, per job
            if ((int32_t)max_score.y + missing * V < min_score[1]) alive[1] = false;
        }
    }
    if (TYPE == NVBIO_GLOBAL)
    {
        v2s v = ZERO;
        #pragma unroll
        for (int j = 1; j <= STRIPE; ++j) if ((uint32_t)j == jm) v = H[j];
        if (alive[0]) sink[0].report( v.x, N, M );
        if (alive[1]) sink[1].report( v.y, N, M );
    }
    #pragma unroll
    for (int u = 0; u < 2; ++u)
        if (valid[u]) { scores[job[u]] = sink[u].score; sinks[job[u]] = make_uint2( sink[u].x, sink[u].y ); }
}

// ---------------------------------------------------------------------------------------------
// The same for the end-to-end case with a zero match bonus (SEMI_GLOBAL, match = 0: nvBowtie's opposite-mate scoring), swept in
// stripes of SIXTEEN pattern columns: half the passes over the text, half the boundary-column traffic and half the per-row set-up per
// cell.  The reference tests its early exit after every 8 columns (gotoh_inl.h:676-680): the running maximum of column 8 of the stripe
// is kept beside that of column 16 and both tests are made at the stripe's end, in order -- nothing is reported before the last stripe,
// so a job that the first test would have stopped simply loses what the last stripe reported.  With match = 0 the diagonal term is
// one v_pk_mad_u16 on the mismatch flag; `Hg = H + GO` is kept beside `H` (it feeds F of the next row and E of the next column), and
// the row is ordered by hand so that no packed operation is consumed by the instruction behind it (see banded_gotoh_band31_pk_kernel):
// 10 issue slots per cell, against 13 + 3 idle ones, and 30 per row instead of 160 per 8 cells around them.
// ---------------------------------------------------------------------------------------------
// a packed pair from its 32-bit image (never element by element: see the note on the register arrays below)
__device__ __forceinline__ v2s pkw(const int a, const int b) { return pk_bits( ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16) ); }

template <int RBITS>
__global__ void __launch_bounds__(128)
full_gotoh_pb_pk16_kernel(const BatchDev b, const SchemeDev sc, const uint32_t M, const uint32_t N, const uint32_t pair_begin, const uint32_t pairs,
                          const int32_t* __restrict__ min_scores, uint2* __restrict__ column, int32_t* __restrict__ scores, uint2* __restrict__ sinks,
                          const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count)
{
    constexpr int W = 16;
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // pair slot inside this launch
    if (t >= pairs) return;
    const uint32_t n_jobs = *job_count;
    const uint32_t slot0 = 2u * (pair_begin + t);
    if (slot0 >= n_jobs) return;

    uint32_t job[2], first[2], tb[2]; bool rev[2], comp[2], valid[2]; int32_t min_score[2];
    #pragma unroll
    for (int u = 0; u < 2; ++u)
    {
        valid[u] = slot0 + u < n_jobs;
        job[u]   = job_list[valid[u] ? slot0 + u : slot0];
        const uint32_t rid = b.read_id ? b.read_id[job[u]] : job[u];
        first[u] = b.read_offsets[rid];
        const uint32_t fl = b.flags ? b.flags[job[u]] : 0u;
        rev[u]  = (fl & NVBIO_READ_REVERSE) != 0;
        comp[u] = (fl & NVBIO_READ_COMPLEMENT) != 0;
        tb[u]   = b.win_begin[job[u]];
        min_score[u] = min_scores ? min_scores[job[u]] : NVBIO_SCORE_MIN;
    }
    SymbolReader<RBITS> prd0( b.reads ), prd1( b.reads );
    const uint32_t* __restrict__ twords = (const uint32_t*)b.text;

    const int32_t G_o = sc.pat_go, G_e = sc.pat_ge;
    const int32_t infimum = -32768 - (G_o < G_e ? G_o : G_e);
    const v2s GO = pkw( G_o, G_o ), GE = pkw( G_e, G_e );
    const v2s INF = pkw( infimum, infimum );

    uint2* col = column + t;                                    // element i at col[i * pairs]: (cell of job 0, cell of job 1)
    {
        const uint32_t c = pack_cell( 0, infimum );             // GotohScoringContext::init (gotoh_inl.h:56-74), SEMI_GLOBAL
        for (uint32_t i = 0; i < N; ++i) col[(size_t)i * pairs] = make_uint2( c, c );
    }

    Sink sink[2]; sink[0].init(); sink[1].init();
    bool alive[2] = { true, true };
    const uint32_t nb8 = (M + 7u) / 8u ? (M + 7u) / 8u : 1u;    // the reference's blocks of 8 columns
    const uint32_t n16 = (nb8 + 1u) / 2u;
    const uint32_t jm  = ((M - 1u) & (W - 1u)) + 1u;            // the pattern's last column inside the last stripe

    // the text stream of a job: 16 rows per packed word, assembled from the two words that cover them (loaded one chunk ahead)
    uint32_t t_lo[2], t_hi[2];
    #pragma unroll
    for (int u = 0; u < 2; ++u) { t_lo[u] = tb[u] >> 4; t_hi[u] = (tb[u] + (N ? N - 1u : 0u)) >> 4; }

    // (register arrays of packed pairs are kept as 32-bit words: as 2 x i16 vectors the compiler splits them into halves wherever one
    // element is read, and glues them back with a v_perm_b32 per use)
    v2s c_mm[W];
    #pragma unroll
    for (int j = 0; j < W; ++j) c_mm[j] = pk_bits( 0u );
    v2s H[W + 1], Hg[W + 1], F[W + 1];

    for (uint32_t k = 0; k < n16 && (alive[0] || alive[1]); ++k)
    {
        const uint32_t block = k * W;
        const bool last = (k + 1u == n16);
        // the stripe's pattern symbols as bit planes (bit j = column block + j): low bit, high bit, "is N"
        uint32_t p0[2] = { 0, 0 }, p1[2] = { 0, 0 }, pn[2] = { 0, 0 };
        #pragma unroll
        for (int j = 0; j < W; ++j)
        {
            if (block + j < M)
            {
                int mm2[2];
                #pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const uint32_t idx = rev[u] ? first[u] + M - 1u - (block + j) : first[u] + block + j;
                    uint32_t q = u ? prd1.get( idx ) : prd0.get( idx );
                    if (comp[u] && q < 4u) q = 3u - q;
                    const uint32_t qq = b.quals ? b.quals[idx] : 0u;
                    mm2[u] = s_mm[qq < 63u ? qq : 63u];
                    p0[u] |= (q & 1u) << j; p1[u] |= ((q >> 1) & 1u) << j; pn[u] |= (q > 3u ? 1u : 0u) << j;
                }
                c_mm[j] = pk_bits( ((uint32_t)mm2[0] & 0xFFFFu) | ((uint32_t)mm2[1] << 16) );
            }
        }
        #pragma unroll
        for (int j = 0; j <= W; ++j)
        {
            const int h0 = (block + j > 0) ? G_o + G_e * (int32_t)(block + j - 1u) : 0;     // :676-681
            const uint32_t hb = ((uint32_t)h0 & 0xFFFFu) * 0x00010001u;
            H[j] = pk_bits( hb ); Hg[j] = pk_bits( hb ) + GO; F[j] = INF;
        }
        v2s max_mid = pkw( -32768, -32768 ), max_end = pkw( -32768, -32768 );
        v2s temp_i  = H[0];

        uint32_t wa[2], wb2[2];
        auto issue_text = [&](const uint32_t i0) {
            #pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                const uint32_t w = (tb[u] + i0) >> 4;
                wa[u]  = twords[w     < t_hi[u] ? w      : t_hi[u]];
                wb2[u] = twords[w + 1 < t_hi[u] ? w + 1u : t_hi[u]];
            }
        };
        issue_text( 0 );
        uint2 cell_next = col[0];

        // the sweep over the text, compiled twice: only the last stripe reports (the pattern's last column, picked out of the row as it
        // is computed -- reading it back out of the register array makes the compiler split the array into halves)
        auto sweep = [&](auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        for (uint32_t i0 = 0; i0 < N; i0 += 16u)
        {
            uint32_t tw[2];
            #pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                const uint32_t sh = ((tb[u] + i0) & 15u) * 2u;
                tw[u] = sh ? ((wa[u] << sh) | (wb2[u] >> (32u - sh))) : wa[u];
            }
            if (i0 + 16u < N) issue_text( i0 + 16u );
            const uint32_t r_end = (i0 + 16u < N) ? 16u : N - i0;
            for (uint32_t tt = 0; tt < r_end; ++tt)
            {
                const uint32_t i = i0 + tt;
                const uint2 cell = cell_next;
                if (i + 1u < N) cell_next = col[(size_t)(i + 1u) * pairs];
                // the row's mismatch flags: bit j / 16+j of NW = column block+j of job 0 / 1 does NOT match
                uint32_t nq[2];
                #pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const uint32_t r = tw[u] >> 30; tw[u] <<= 2;
                    nq[u] = (p0[u] ^ (0u - (r & 1u))) | (p1[u] ^ (0u - (r >> 1))) | pn[u];
                }
                const uint32_t NW = __builtin_amdgcn_perm( nq[1], nq[0], 0x05040100u );
                const v2s Hc = pk_bits( __builtin_amdgcn_perm( cell.y, cell.x, 0x05040100u ) );      // (H of job 0, H of job 1) of the boundary column
                const v2s Ec = pk_bits( __builtin_amdgcn_perm( cell.y, cell.x, 0x07060302u ) );
                const v2s Hd0 = temp_i;
                temp_i = Hc;

                // cell 1's off-chain part, cell 2's first three operations
                v2s hg = Hc + GO, eg = Ec + GE, E, fd, x, tj, d, f, hsel = Hc;
                {
                    const v2s x1 = F[1] + GE;
                    const v2s t1 = pk_bits( NW & 0x00010001u );
                    const v2s d1 = Hd0 + t1 * c_mm[0];
                    const v2s f1 = pk_max( x1, Hg[1] );
                    F[1] = f1;
                    fd = pk_max( f1, d1 );
                    x  = F[2] + GE;
                    tj = pk_bits( (NW >> 1) & 0x00010001u );
                }
                constexpr int Z = 0;
                __builtin_amdgcn_sched_barrier( Z );
                #pragma unroll
                for (int j = 1; j <= W; ++j)
                {
                    // the E chain of column j (E -> h -> hg) interleaved with what column j+1 and j+2 can compute without it
                    E = pk_max( eg, hg );
                    __builtin_amdgcn_sched_barrier( Z );
                    if (j + 1 <= W) d = H[j] + tj * c_mm[j];                 // H[j] still is the previous row's
                    __builtin_amdgcn_sched_barrier( Z );
                    const v2s h = pk_max( fd, E );
                    if (LAST && (uint32_t)j == jm) hsel = h;
                    __builtin_amdgcn_sched_barrier( Z );
                    if (j + 1 <= W) f = pk_max( x, Hg[j + 1] );
                    __builtin_amdgcn_sched_barrier( Z );
                    hg = h + GO;
                    __builtin_amdgcn_sched_barrier( Z );
                    if (j + 1 <= W) eg = E + GE;
                    __builtin_amdgcn_sched_barrier( Z );
                    if (j + 1 <= W) fd = pk_max( f, d );
                    __builtin_amdgcn_sched_barrier( Z );
                    if (j + 2 <= W) x = F[j + 2] + GE;
                    if (j + 2 <= W) tj = pk_bits( (NW >> (j + 1)) & 0x00010001u );
                    H[j] = h; Hg[j] = hg;
                    if (j + 1 <= W) F[j + 1] = f;
                    __builtin_amdgcn_sched_barrier( Z );
                }
                // the stripe's last column goes to the boundary column: (H, E) of job 0, of job 1
                col[(size_t)i * pairs] = make_uint2( __builtin_amdgcn_perm( bits_pk( E ), bits_pk( H[W] ), 0x05040100u ),
                                                     __builtin_amdgcn_perm( bits_pk( E ), bits_pk( H[W] ), 0x07060302u ) );
                max_end = pk_max( max_end, H[W] );
                max_mid = pk_max( max_mid, H[8] );
                if (LAST)
                {
                    const uint32_t v = bits_pk( hsel );
                    if (alive[0]) sink[0].report( (int32_t)(int16_t)(v & 0xFFFFu), i + 1u, M );
                    if (alive[1]) sink[1].report( (int32_t)v >> 16, i + 1u, M );
                }
            }
        }
        };
        if (last) sweep( std::true_type() ); else sweep( std::false_type() );
        // the reference's early exit (match = 0: nothing is still to be gained), first after column block + 8, then after block + 16
        if (2u * k + 1u < nb8)
        {
            const uint32_t mb = bits_pk( max_mid );
            if ((int32_t)(int16_t)(mb & 0xFFFFu) < min_score[0]) { if (alive[0] && last) sink[0].init(); alive[0] = false; }
            if (((int32_t)mb >> 16)              < min_score[1]) { if (alive[1] && last) sink[1].init(); alive[1] = false; }
        }
        if (!last)
        {
            const uint32_t eb = bits_pk( max_end );
            if ((int32_t)(int16_t)(eb & 0xFFFFu) < min_score[0]) alive[0] = false;
            if (((int32_t)eb >> 16)              < min_score[1]) alive[1] = false;
        }
    }
    #pragma unroll
    for (int u = 0; u < 2; ++u)
        if (valid[u]) { scores[job[u]] = sink[u].score; sinks[job[u]] = make_uint2( sink[u].x, sink[u].y ); }
}

// jobs of the batch's dominant shape that still need the DP go to the packed kernel, the other ones that need it
// to full_gotoh_kernel: two flag arrays for two DeviceSelect passes
__global__ void __launch_bounds__(256)
classify_shape_kernel(const BatchDev b, const uint32_t M0, const uint32_t N0, const uint8_t* __restrict__ need_dp,
                      uint8_t* __restrict__ to_packed, uint8_t* __restrict__ to_plain)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= b.n) return;
    const uint32_t rid = b.read_id ? b.read_id[job] : job;
    const uint32_t M = b.read_offsets[rid + 1] - b.read_offsets[rid];
    const uint32_t N = b.win_end[job] - b.win_begin[job];
    const bool need = need_dp ? need_dp[job] != 0 : true;
    const bool uni  = (M == M0 && N == N0);
    to_packed[job] = (need && uni) ? 1 : 0;
    to_plain[job]  = (need && !uni) ? 1 : 0;
}

static bool full_packed_ok(const int type, const SchemeDev& sc, const uint32_t M, const uint32_t N)
{
    if (M == 0 || N == 0) return false;
    if (sc.match < 0 || sc.mm_min < 0 || sc.mm_max < 0) return false;
    int64_t step = sc.match;
    const int64_t c[] = { sc.mm_min, sc.mm_max, -(int64_t)sc.pat_go, -(int64_t)sc.pat_ge, -(int64_t)sc.txt_go, -(int64_t)sc.txt_ge };
    for (int64_t v : c) { if (v < 0) return false; if (v > step) step = v; }       // gap terms must be <= 0, penalties >= 0
    if (step > 4096) return false;
    if (((int64_t)M + N) * step > 12000) return false;
    if (type == NVBIO_LOCAL && (int64_t)sc.match * M > 2000) return false;          // (score << 4 | column) must fit an int16
    return true;
}

static bool full_ungapped_ok(const SchemeDev& sc, const BatchDev& b, int32_t* P)
{
    if (sc.match != 0) return false;
    if (sc.mm_min < 0 || sc.mm_max < 0) return false;
    if (b.quals != nullptr && sc.mm_min != sc.mm_max) return false;
    if (sc.pat_go >= 0 || sc.txt_go >= 0 || sc.pat_ge > 0 || sc.txt_ge > 0) return false;
    *P = sc.mm_min;
    return true;
}

template <int TYPE, bool TB>
nvbio_status launch_bits(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t job_begin, uint32_t jobs,
                         const int32_t* min_scores, uint32_t* column, int32_t* scores, uint2* sinks, hipStream_t s,
                         const uint32_t* job_list, const uint32_t* job_count)
{
    const dim3 grid( (jobs + 127u) / 128u ), block( 128 );
#define NVB_GO(RB, TBITS) do { if (sc.wide) hipLaunchKernelGGL( (full_gotoh_kernel<TYPE,TB,RB,TBITS,false,true>), grid, block, 0, s, b, sc, job_begin, jobs, min_scores, column, scores, sinks, job_list, job_count ); \
                               else         hipLaunchKernelGGL( (full_gotoh_kernel<TYPE,TB,RB,TBITS>), grid, block, 0, s, b, sc, job_begin, jobs, min_scores, column, scores, sinks, job_list, job_count ); } while (0)
    if      (rbits == 4 && tbits == 2) NVB_GO(4, 2);
    else if (rbits == 2 && tbits == 2) NVB_GO(2, 2);
    else if (rbits == 8 && tbits == 2) NVB_GO(8, 2);
    else if (rbits == 8 && tbits == 8) NVB_GO(8, 8);
    else if (rbits == 4 && tbits == 8) NVB_GO(4, 8);
    else if (rbits == 2 && tbits == 8) NVB_GO(2, 8);
    else { set_error( "unsupported read_bits/text_bits %u/%u", rbits, tbits ); return NVBIO_ERR_INVALID; }
#undef NVB_GO
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

template <bool TB>
nvbio_status launch_type(int type, const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t job_begin, uint32_t jobs,
                         const int32_t* min_scores, uint32_t* column, int32_t* scores, uint2* sinks, hipStream_t s,
                         const uint32_t* job_list, const uint32_t* job_count)
{
    switch (type)
    {
    case NVBIO_GLOBAL:      return launch_bits<NVBIO_GLOBAL,TB>     ( b, sc, rbits, tbits, job_begin, jobs, min_scores, column, scores, sinks, s, job_list, job_count );
    case NVBIO_LOCAL:       return launch_bits<NVBIO_LOCAL,TB>      ( b, sc, rbits, tbits, job_begin, jobs, min_scores, column, scores, sinks, s, job_list, job_count );
    case NVBIO_SEMI_GLOBAL: return launch_bits<NVBIO_SEMI_GLOBAL,TB>( b, sc, rbits, tbits, job_begin, jobs, min_scores, column, scores, sinks, s, job_list, job_count );
    }
    set_error( "invalid alignment type %d", type );
    return NVBIO_ERR_INVALID;
}

} // anonymous namespace
} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" nvbio_status nvbio_full_gotoh_temp_bytes(const nvbio_alignment_batch* batch, uint32_t max_pattern_len, uint32_t max_text_len,
                                                    int text_blocking, uint64_t* bytes)
{
    NVB_REQUIRE( batch && bytes, "batch/bytes is NULL" );
    const uint64_t rows = text_blocking ? max_pattern_len : max_text_len;
    *bytes = (uint64_t)batch->n * (rows ? rows : 1u) * sizeof(uint32_t);
    return NVBIO_OK;
}

static nvbio_status full_score(int device, int type, int text_blocking, const SchemeDev sc, const nvbio_alignment_batch* batch,
                               uint32_t max_pattern_len, uint32_t max_text_len,
                               const int32_t* min_scores_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                               void* temp_dev, uint64_t temp_bytes, void* stream)
{
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev, "NULL output pointer" );
    const uint64_t rows = text_blocking ? max_pattern_len : max_text_len;
    NVB_REQUIRE( rows > 0, "max_pattern_len / max_text_len must bound the boundary column" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;

    // ---- the cooperative kernel (L lanes per job, no boundary column in memory): GLOBAL, and SEMI_GLOBAL where the end-to-end shortcut does not
    //      apply, without the stripe early exit; packed 2- / 4-bit reads in a 2-bit text; every value inside the int16 range of the reference's
    //      boundary cells.  Batches large enough for the two-jobs-per-lane kernel keep that one.
    {
        int64_t step = sc.match > 0 ? sc.match : -(int64_t)sc.match;
        const int64_t cc[] = { sc.mm_min, sc.mm_max, sc.pat_go, sc.pat_ge, sc.txt_go, sc.txt_ge };
        for (int64_t v : cc) { const int64_t a = v < 0 ? -v : v; if (a > step) step = a; }
        const bool pk_bits  = batch->text_bits == 2 && (batch->read_bits == 4 || batch->read_bits == 2);
        const bool e2e      = type == NVBIO_SEMI_GLOBAL && sc.match == 0 && !(b.algo & NVBIO_ALN_NO_UNGAPPED_SCORE);      // (the shortcut's ground)
        const bool pk_route = !text_blocking && (b.n >= 262144u || (b.algo & NVBIO_ALN_FORCE_PACKED_DP)) && !(b.algo & NVBIO_ALN_NO_PACKED_DP);
        const bool coop = (type == NVBIO_GLOBAL || type == NVBIO_SEMI_GLOBAL) && !e2e && !pk_route && min_scores_dev == nullptr && pk_bits && plain_gotoh( sc ) &&
                          max_pattern_len >= 1 && max_pattern_len <= 256u && ((int64_t)max_pattern_len + max_text_len + 2) * step <= 30000 &&
                          !(b.algo & NVBIO_ALN_NO_COOPERATIVE_DP);
        if (coop)
        {
#define NVB_COOP(TYPE_, L_, W_) do { const uint64_t lanes = (uint64_t)b.n * L_; const dim3 grid( (uint32_t)((lanes + 255u) / 256u) ), block( 256 ); \
            if (batch->read_bits == 4) hipLaunchKernelGGL( (full_gotoh_coop_kernel<TYPE_,L_,W_,4>), grid, block, 0, s, b, sc, text_blocking != 0, scores_dev, (uint2*)sinks_dev ); \
            else                       hipLaunchKernelGGL( (full_gotoh_coop_kernel<TYPE_,L_,W_,2>), grid, block, 0, s, b, sc, text_blocking != 0, scores_dev, (uint2*)sinks_dev ); } while (0)
#define NVB_COOP_SHAPE(TYPE_) do { const uint32_t mp = max_pattern_len; \
            if      (mp <=  32u) NVB_COOP( TYPE_, 4, 8 );  \
            else if (mp <=  64u) NVB_COOP( TYPE_, 4, 16 ); \
            else if (mp <= 100u) NVB_COOP( TYPE_, 4, 25 ); \
            else if (mp <= 128u) NVB_COOP( TYPE_, 8, 16 ); \
            else if (mp <= 152u) NVB_COOP( TYPE_, 8, 19 ); \
            else if (mp <= 200u) NVB_COOP( TYPE_, 8, 25 ); \
            else                 NVB_COOP( TYPE_, 8, 32 ); } while (0)
            if (type == NVBIO_GLOBAL) NVB_COOP_SHAPE( NVBIO_GLOBAL ); else NVB_COOP_SHAPE( NVBIO_SEMI_GLOBAL );
#undef NVB_COOP_SHAPE
#undef NVB_COOP
            NVB_HIP( hipGetLastError() );
            return NVBIO_OK;
        }
    }

    // ---- which jobs need which kernel --------------------------------------------------------------------------
    //  1. end-to-end shortcut (ungapped_full_e2e_kernel): settles the jobs whose best diagonal beats every gapped
    //     alignment and flags the rest (need_dp);
    //  2. pattern blocking, packable scheme: the flagged jobs of the batch's dominant shape (M = max_pattern_len,
    //     N = max_text_len) go to the 16-bit packed kernel two per lane (list A), the others to the int32 kernel (list B);
    //     otherwise every flagged job goes to the int32 kernel (list B).
    //  The lists and their lengths stay on the device.
    uint32_t *list_a = nullptr, *count_a = nullptr, *job_list = nullptr, *job_count = nullptr; void* aux = nullptr;
    int32_t P = 0;
    const bool packable_bits = batch->text_bits == 2 && (batch->read_bits == 4 || batch->read_bits == 2);
    const bool shortcut = type == NVBIO_SEMI_GLOBAL && packable_bits && plain_gotoh( sc ) && full_ungapped_ok( sc, b, &P ) && !(b.algo & NVBIO_ALN_NO_UNGAPPED_SCORE);
    // (two jobs per lane only pay when the lanes still fill the chip: 100 k jobs of the sw-benchmark shape ran 15-20 % slower packed)
    const bool packed   = !text_blocking && packable_bits && (b.n >= 262144u || (b.algo & NVBIO_ALN_FORCE_PACKED_DP)) && plain_gotoh( sc ) && full_packed_ok( type, sc, max_pattern_len, max_text_len ) &&
                          !(b.algo & NVBIO_ALN_NO_PACKED_DP);
    if (shortcut || packed)
    {
        size_t sel_bytes = 0;
        hipcub::CountingInputIterator<uint32_t> ids( 0u );
        NVB_HIP( hipcub::DeviceSelect::Flagged( nullptr, sel_bytes, ids, (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)b.n, s ) );
        const uint64_t flags_bytes = ((uint64_t)b.n + 255u) & ~255ull;
        const uint64_t list_bytes  = ((uint64_t)b.n * 4u + 255u) & ~255ull;
        // (the narrow route: one more flag array, its job list, the band windows; see narrow_jobs_kernel)
        const bool narrow = shortcut && !text_blocking && !(b.algo & (NVBIO_ALN_NO_NARROW_SCORE | NVBIO_ALN_NO_PACKED_DP)) && max_pattern_len <= 161u &&
                            banded31_packed_ok( sc, max_pattern_len ) && sc.pat_ge < 0;
        if (scratch_alloc( &aux, 4u * flags_bytes + 5u * list_bytes + 512u + sel_bytes, s ) != hipSuccess)
        {
            (void)hipGetLastError();
            set_error( "full Gotoh: out of device memory for the job lists" );
            return NVBIO_ERR_NOMEM;
        }
        uint8_t* need_dp   = (uint8_t*)aux;
        uint8_t* to_packed = need_dp + flags_bytes;
        uint8_t* to_plain  = to_packed + flags_bytes;
        list_a    = (uint32_t*)(to_plain + flags_bytes);
        job_list  = (uint32_t*)((uint8_t*)list_a + list_bytes);
        count_a   = (uint32_t*)((uint8_t*)job_list + list_bytes);
        job_count = count_a + 1;
        uint32_t* count_n  = count_a + 2;
        uint8_t*  route    = (uint8_t*)count_a + 256u;
        uint32_t* list_n   = (uint32_t*)(route + flags_bytes);
        uint32_t* wb2      = (uint32_t*)((uint8_t*)list_n + list_bytes);
        uint32_t* we2      = (uint32_t*)((uint8_t*)wb2 + list_bytes);
        void* sel_temp = (uint8_t*)we2 + list_bytes + 256u;
        const dim3 grid( (b.n + 255u) / 256u ), block( 256 );
        if (shortcut)
        {
            const int32_t G = sc.pat_go > sc.txt_go ? sc.pat_go : sc.txt_go;
            const bool second_chance = (sc.pat_go == sc.txt_go && sc.pat_ge == sc.txt_ge);
            if (batch->read_bits == 4) hipLaunchKernelGGL( (ungapped_full_e2e_kernel<4,0>), grid, block, 0, s, b, P, G, sc.pat_go, sc.pat_ge, second_chance, min_scores_dev, text_blocking != 0, scores_dev, (uint2*)sinks_dev, need_dp, narrow,
                                                           (const uint32_t*)nullptr, (const uint32_t*)nullptr );
            else                       hipLaunchKernelGGL( (ungapped_full_e2e_kernel<2,0>), grid, block, 0, s, b, P, G, sc.pat_go, sc.pat_ge, second_chance, min_scores_dev, text_blocking != 0, scores_dev, (uint2*)sinks_dev, need_dp, narrow,
                                                           (const uint32_t*)nullptr, (const uint32_t*)nullptr );
            if (second_chance)
            {
                // the jobs the second chance can still settle (need_dp == 3), compacted, through their own launch: every job ends as 0 or 1
                uint32_t* list_s = list_n; uint32_t* count_s = count_n + 1;                    // (the narrow route's list is built after this)
                hipcub::TransformInputIterator<uint8_t, FlagIs3, const uint8_t*> is3( need_dp, FlagIs3() );
                const hipError_t e3 = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, is3, list_s, count_s, (int)b.n, s );
                if (e3 != hipSuccess) { scratch_free( aux, s ); set_error( "DeviceSelect failed: %s", hipGetErrorString( e3 ) ); return NVBIO_ERR_HIP; }
                if (batch->read_bits == 4) hipLaunchKernelGGL( (ungapped_full_e2e_kernel<4,1>), grid, block, 0, s, b, P, G, sc.pat_go, sc.pat_ge, second_chance, min_scores_dev, text_blocking != 0, scores_dev, (uint2*)sinks_dev, need_dp, narrow,
                                                               (const uint32_t*)list_s, (const uint32_t*)count_s );
                else                       hipLaunchKernelGGL( (ungapped_full_e2e_kernel<2,1>), grid, block, 0, s, b, P, G, sc.pat_go, sc.pat_ge, second_chance, min_scores_dev, text_blocking != 0, scores_dev, (uint2*)sinks_dev, need_dp, narrow,
                                                               (const uint32_t*)list_s, (const uint32_t*)count_s );
            }
        }
        hipError_t e = hipSuccess;
        if (narrow)
        {
            // the unsettled jobs with room for a band around their best diagonal: band-31 score, then the run test (narrow_check_kernel)
            hipLaunchKernelGGL( narrow_jobs_kernel, grid, block, 0, s, b, (const uint8_t*)need_dp, (const uint2*)sinks_dev, wb2, we2, route );
            e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, route, list_n, count_n, (int)b.n, s );
            if (e == hipSuccess)
            {
                BatchDev b2 = b; b2.win_begin = wb2; b2.win_end = we2; b2.max_read_len = max_pattern_len;
                banded31_packed_launch( b2, sc, batch->read_bits, b.n, scores_dev, (uint2*)sinks_dev, list_n, count_n, s );
                if (batch->read_bits == 4) hipLaunchKernelGGL( (narrow_check_kernel<4>), grid, block, 0, s, b, P, sc.pat_go, sc.pat_ge, min_scores_dev, (const uint32_t*)wb2, (const int32_t*)scores_dev, (uint2*)sinks_dev, need_dp, (const uint32_t*)list_n, (const uint32_t*)count_n );
                else                       hipLaunchKernelGGL( (narrow_check_kernel<2>), grid, block, 0, s, b, P, sc.pat_go, sc.pat_ge, min_scores_dev, (const uint32_t*)wb2, (const int32_t*)scores_dev, (uint2*)sinks_dev, need_dp, (const uint32_t*)list_n, (const uint32_t*)count_n );
            }
        }
        if (packed)
        {
            hipLaunchKernelGGL( classify_shape_kernel, grid, block, 0, s, b, max_pattern_len, max_text_len, shortcut ? (const uint8_t*)need_dp : (const uint8_t*)nullptr,
                                to_packed, to_plain );
            e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, to_packed, list_a, count_a, (int)b.n, s );
            if (e == hipSuccess) e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, to_plain, job_list, job_count, (int)b.n, s );
        }
        else
        {
            list_a = nullptr; count_a = nullptr;
            e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, need_dp, job_list, job_count, (int)b.n, s );
        }
        if (e != hipSuccess) { scratch_free( aux, s ); set_error( "DeviceSelect failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    }

    // boundary columns: caller scratch if given, else stream-ordered scratch; jobs are processed in
    // as many launches as the scratch allows (at least one wave of jobs per launch)
    void*    owned = nullptr;
    uint32_t* column = (uint32_t*)temp_dev;
    uint64_t  cap_jobs;
    if (column)
    {
        cap_jobs = temp_bytes / (rows * sizeof(uint32_t));
        if (!(cap_jobs >= 64 || cap_jobs >= b.n))
        {
            if (aux) scratch_free( aux, s );
            set_error( "invalid argument: temp_bytes too small (see nvbio_full_gotoh_temp_bytes)" );
            return NVBIO_ERR_INVALID;
        }
    }
    else
    {
        cap_jobs = b.n;
        const uint64_t budget = 8ull << 30;                      // at most 8 GiB of scratch per launch
        if (cap_jobs * rows * sizeof(uint32_t) > budget) cap_jobs = budget / (rows * sizeof(uint32_t));
        if (cap_jobs < 64) cap_jobs = 64;
        if (scratch_alloc( &owned, cap_jobs * rows * sizeof(uint32_t), s ) != hipSuccess)
        {
            (void)hipGetLastError();
            if (aux) scratch_free( aux, s );
            set_error( "full Gotoh: out of device memory for %llu boundary columns", (unsigned long long)cap_jobs );
            return NVBIO_ERR_NOMEM;
        }
        column = (uint32_t*)owned;
    }
    nvbio_status st = NVBIO_OK;
    if (list_a)
    {
        // packed kernel over list A: a pair of jobs shares a lane and a column of (cell, cell) pairs -- the same bytes per job
        const uint64_t cap_pairs = cap_jobs / 2u ? cap_jobs / 2u : 1u;
        const uint64_t all_pairs = ((uint64_t)b.n + 1u) / 2u;
        for (uint64_t pb = 0; pb < all_pairs && st == NVBIO_OK; pb += cap_pairs)
        {
            const uint32_t pairs = (uint32_t)((all_pairs - pb) < cap_pairs ? (all_pairs - pb) : cap_pairs);
            const dim3 grid( (pairs + 127u) / 128u ), block( 128 );
#define NVB_PK(TYPE_, RB) hipLaunchKernelGGL( (full_gotoh_pb_pk_kernel<TYPE_,RB>), grid, block, 0, s, b, sc, max_pattern_len, max_text_len, (uint32_t)pb, pairs, \
                                              min_scores_dev, (uint2*)column, scores_dev, (uint2*)sinks_dev, (const uint32_t*)list_a, (const uint32_t*)count_a )
#define NVB_PK16(RB) hipLaunchKernelGGL( (full_gotoh_pb_pk16_kernel<RB>), grid, block, 0, s, b, sc, max_pattern_len, max_text_len, (uint32_t)pb, pairs, \
                                         min_scores_dev, (uint2*)column, scores_dev, (uint2*)sinks_dev, (const uint32_t*)list_a, (const uint32_t*)count_a )
            const bool wide16 = type == NVBIO_SEMI_GLOBAL && sc.match == 0 && !(b.algo & NVBIO_ALN_PK_STRIPE8);      // the end-to-end kernel, 16 columns per stripe
            if (batch->read_bits == 4)
            {
                if (wide16) NVB_PK16( 4 );
                else if (type == NVBIO_GLOBAL) NVB_PK( NVBIO_GLOBAL, 4 ); else if (type == NVBIO_LOCAL) NVB_PK( NVBIO_LOCAL, 4 ); else NVB_PK( NVBIO_SEMI_GLOBAL, 4 );
            }
            else
            {
                if (wide16) NVB_PK16( 2 );
                else if (type == NVBIO_GLOBAL) NVB_PK( NVBIO_GLOBAL, 2 ); else if (type == NVBIO_LOCAL) NVB_PK( NVBIO_LOCAL, 2 ); else NVB_PK( NVBIO_SEMI_GLOBAL, 2 );
            }
#undef NVB_PK16
#undef NVB_PK
            if (hipGetLastError() != hipSuccess) { set_error( "packed full Gotoh launch failed" ); st = NVBIO_ERR_HIP; }
        }
    }
    for (uint64_t begin = 0; begin < b.n && st == NVBIO_OK; begin += cap_jobs)
    {
        const uint32_t jobs = (uint32_t)((b.n - begin) < cap_jobs ? (b.n - begin) : cap_jobs);
        st = text_blocking ?
            launch_type<true> ( type, b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, min_scores_dev, column, scores_dev, (uint2*)sinks_dev, s, job_list, job_count ) :
            launch_type<false>( type, b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, min_scores_dev, column, scores_dev, (uint2*)sinks_dev, s, job_list, job_count );
    }
    if (owned) scratch_free( owned, s );
    if (aux)   scratch_free( aux, s );
    return st;
}

extern "C" nvbio_status nvbio_full_gotoh_score(int device, nvbio_alignment_type type, int text_blocking,
                                               const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                               uint32_t max_pattern_len, uint32_t max_text_len,
                                               const int32_t* min_scores_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                               void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    return full_score( device, type, text_blocking, scheme_dev( scheme ), batch, max_pattern_len, max_text_len, min_scores_dev,
                       scores_dev, sinks_dev, temp_dev, temp_bytes, stream );
}

// Best2Sink scoring: the int32 kernel reporting cell by cell (no shortcut, no packed kernel: both rely on BestSink's rule)
template <int TYPE, bool TB>
static nvbio_status launch_best2(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t job_begin, uint32_t jobs,
                                 const int32_t* min_scores, uint32_t* column, uint32_t dist, int32_t* scores, uint2* sinks, int32_t* scores2, uint2* sinks2, hipStream_t s)
{
    const dim3 grid( (jobs + 127u) / 128u ), block( 128 );
#define NVB_GO2(RB, TBITS) hipLaunchKernelGGL( (full_gotoh_kernel<TYPE,TB,RB,TBITS,true>), grid, block, 0, s, b, sc, job_begin, jobs, min_scores, column, scores, sinks, \
                                               (const uint32_t*)nullptr, (const uint32_t*)nullptr, dist, scores2, sinks2 )
    if      (rbits == 4 && tbits == 2) NVB_GO2(4, 2);
    else if (rbits == 2 && tbits == 2) NVB_GO2(2, 2);
    else if (rbits == 8 && tbits == 2) NVB_GO2(8, 2);
    else if (rbits == 8 && tbits == 8) NVB_GO2(8, 8);
    else { set_error( "Best2Sink scoring: read_bits/text_bits %u/%u not instantiated (4/2, 2/2, 8/2, 8/8)", rbits, tbits ); return NVBIO_ERR_UNSUPPORTED; }
#undef NVB_GO2
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_full_gotoh_score_best2(int device, nvbio_alignment_type type, int text_blocking,
                                                     const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                                     uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_scores_dev,
                                                     uint32_t distinct_dist, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                                     int32_t* scores2_dev, nvbio_uint2* sinks2_dev, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev && scores2_dev && sinks2_dev, "NULL output pointer" );
    NVB_REQUIRE( type == NVBIO_GLOBAL || type == NVBIO_LOCAL || type == NVBIO_SEMI_GLOBAL, "invalid alignment type" );
    const uint64_t rows = text_blocking ? max_pattern_len : max_text_len;
    NVB_REQUIRE( rows > 0, "max_pattern_len / max_text_len must bound the boundary column" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const SchemeDev sc = scheme_dev( scheme );
    uint64_t cap_jobs = b.n;
    const uint64_t budget = 8ull << 30;
    if (cap_jobs * rows * sizeof(uint32_t) > budget) cap_jobs = budget / (rows * sizeof(uint32_t));
    if (cap_jobs < 64) cap_jobs = 64;
    void* column = nullptr;
    if (scratch_alloc( &column, cap_jobs * rows * sizeof(uint32_t), s ) != hipSuccess)
    {
        (void)hipGetLastError();
        set_error( "full Gotoh: out of device memory for %llu boundary columns", (unsigned long long)cap_jobs );
        return NVBIO_ERR_NOMEM;
    }
    nvbio_status st = NVBIO_OK;
    for (uint64_t begin = 0; begin < b.n && st == NVBIO_OK; begin += cap_jobs)
    {
        const uint32_t jobs = (uint32_t)((b.n - begin) < cap_jobs ? (b.n - begin) : cap_jobs);
#define NVB_B2(TYPE_) (text_blocking ? launch_best2<TYPE_,true> ( b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, min_scores_dev, (uint32_t*)column, distinct_dist, scores_dev, (uint2*)sinks_dev, scores2_dev, (uint2*)sinks2_dev, s ) \
                                     : launch_best2<TYPE_,false>( b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, min_scores_dev, (uint32_t*)column, distinct_dist, scores_dev, (uint2*)sinks_dev, scores2_dev, (uint2*)sinks2_dev, s ))
        st = type == NVBIO_GLOBAL ? NVB_B2( NVBIO_GLOBAL ) : type == NVBIO_LOCAL ? NVB_B2( NVBIO_LOCAL ) : NVB_B2( NVBIO_SEMI_GLOBAL );
#undef NVB_B2
    }
    scratch_free( column, s );
    return st;
}

extern "C" nvbio_status nvbio_full_sw_score(int device, nvbio_alignment_type type, int text_blocking,
                                            const nvbio_sw_scheme* scheme, const nvbio_alignment_batch* batch,
                                            uint32_t max_pattern_len, uint32_t max_text_len,
                                            const int32_t* min_scores_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                            void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    // the boundary column runs over the pattern with text blocking, over the text otherwise
    return full_score( device, type, text_blocking, scheme_dev( scheme, text_blocking != 0 ), batch, max_pattern_len, max_text_len,
                       min_scores_dev, scores_dev, sinks_dev, temp_dev, temp_bytes, stream );
}
