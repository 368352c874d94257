// gotoh_full_traceback.hip -- batched full-matrix Gotoh traceback (score + CIGAR) for gfx950: the opposite mates of
// a paired-end batch, sw-benchmark-style alignments.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   alignment_traceback (driver: score pass, clip, walk, first row/column, clip)   nvbio/alignment/alignment_inl.h:355-455
//   direction vectors per cell (hdir | edir | fdir), pattern blocking             nvbio/alignment/gotoh/gotoh_inl.h:458-538, :423-436
//   priv::alignment_traceback (the H/E/F state walk)                              gotoh_inl.h:1573-1640
//   nvBowtie's run-length Backtracker and io::Cigar                               nvBowtie/bowtie2/cuda/alignment_utils.h:115-157
//   nvBowtie traceback_best (full DP, FULL_DP_CHECKPOINTS)                        nvBowtie/bowtie2/cuda/traceback_inl.h:249-275
//
// Same design as gotoh_traceback.hip: the reference checkpoints every 64 pattern columns as int16 and recomputes each
// block's direction vectors on the way back; here one forward pass (the pattern-blocking DP of gotoh_full.hip, 8-column
// stripes) writes every cell's direction nibble to HBM scratch -- one 32-bit word per (text row, stripe), job-interleaved
// -- and the walk reads them back.  A 150 x 500 job takes 38 KB; launches are chunked to the scratch granted.  Jobs
// whose optimum is reached by the diagonal through the sink alone are traced without a DP (the tie rule of
// gotoh_inl.h:529-531 resolves the diagonal's ties to SUBSTITUTION, as in the banded case).  Identical to the reference
// while scores fit its int16 checkpoints (host check).
#include "gotoh_common.h"
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace nvbio_amd {
namespace {

enum : uint32_t { D_SUB = 0u, D_INS = 1u, D_DEL = 2u, D_SINK = 3u, D_INS_EXT = 4u, D_DEL_EXT = 8u };
constexpr int STRIPE = 8;

struct Sink
{
    int32_t score; uint32_t x, y;
    __device__ __forceinline__ void init() { score = NVBIO_SCORE_MIN; x = y = 0xFFFFFFFFu; }
    __device__ __forceinline__ void report(const int32_t s, const uint32_t sx, const uint32_t sy)
    {
        if (score <= s) { score = s; x = sx; y = sy; }
    }
};
__device__ __forceinline__ uint32_t pack_cell(const int32_t h, const int32_t e) { return ((uint32_t)h & 0xFFFFu) | ((uint32_t)e << 16); }
__device__ __forceinline__ int32_t  cell_h(const uint32_t c) { return (int32_t)(int16_t)(c & 0xFFFFu); }
__device__ __forceinline__ int32_t  cell_e(const uint32_t c) { return (int32_t)(int16_t)(c >> 16); }

struct JobInfo
{
    uint32_t first, M, tb, N; bool rev, comp;
};
__device__ __forceinline__ JobInfo load_job(const BatchDev& b, const uint32_t job)
{
    JobInfo j;
    const uint32_t rid = b.read_id ? b.read_id[job] : job;
    j.first = b.read_offsets[rid];
    j.M     = b.read_offsets[rid + 1] - j.first;
    const uint32_t fl = b.flags ? b.flags[job] : 0u;
    j.rev  = (fl & NVBIO_READ_REVERSE) != 0;
    j.comp = (fl & NVBIO_READ_COMPLEMENT) != 0;
    j.tb   = b.win_begin[job];
    j.N    = b.win_end[job] - j.tb;
    return j;
}

// ---- forward DP with direction vectors + walk back ------------------------------------------------------------------
template <int TYPE, int RBITS, int TBITS>
__global__ void __launch_bounds__(128)
full_gotoh_traceback_kernel(const BatchDev b, const SchemeDev sc, const uint32_t max_M, const uint32_t max_N,
                            const uint32_t job_begin, const uint32_t jobs, const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count,
                            const int32_t* __restrict__ min_scores, uint32_t* __restrict__ column, uint32_t* __restrict__ dirs,
                            int32_t* __restrict__ scores, uint2* __restrict__ sources, uint2* __restrict__ sinks,
                            uint16_t* __restrict__ cigars, const uint32_t cigar_stride, uint32_t* __restrict__ cigar_lens,
                            const uint8_t* __restrict__ band_code = nullptr, const uint32_t given_sinks = 0u)
{
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= jobs) return;
    if (job_list && job_begin + t >= *job_count) return;
    const uint32_t job = job_list ? job_list[job_begin + t] : job_begin + t;
    // given_sinks (the linear-gap Smith-Waterman aligner): score and sink are the scoring pass's -- which sweeps the matrix in its own
    // logical stripes of 16 columns (sw/sw_inl.h:1322-1325), what decides LOCAL ties and the early exit -- and this kernel only has to
    // supply the direction vectors, which do not depend on the sweep
    if (given_sinks && (sinks[job].x == 0xFFFFFFFFu || sinks[job].y == 0xFFFFFFFFu))
    {
        sources[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); cigar_lens[job] = 0;
        return;
    }
    const JobInfo J = load_job( b, job );
    const uint32_t M = J.M, N = J.N;
    const int32_t min_score = min_scores ? min_scores[job] : NVBIO_SCORE_MIN;

    if (M > max_M || N > max_N)                                  // would overrun the scratch: skipped, flagged
    {
        scores[job] = NVBIO_SCORE_MIN; sinks[job] = sources[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu );
        cigar_lens[job] = 0xFFFFFFFFu;
        return;
    }

    SymbolReader<TBITS> trd( b.text );
    SymbolReader<RBITS> prd( b.reads );
    const int32_t G_o = sc.pat_go, G_e = sc.pat_ge;             // F: the text advances alone
    const int32_t I_o = sc.ins_go, I_e = sc.ins_ge;             // E: the pattern advances alone (= G for the Gotoh aligner; the Smith-Waterman
                                                                // aligner's insertion, sw/sw_inl.h:628-629), T_*: the stripe-top boundary
    const int32_t T_o = sc.top_go, T_e = sc.top_ge;
    const int32_t g_min = G_o < G_e ? G_o : G_e, i_min = I_o < I_e ? I_o : I_e;
    const int32_t infimum = -32768 - (g_min < i_min ? g_min : i_min);
    const int32_t V = sc.match;
    const uint32_t nst = (max_M + STRIPE - 1u) / STRIPE;        // direction words per text row

    // Restricted rows (end-to-end jobs whose score S* and sink are known, band_code[job] = 2 + G): every step of a path scores <= 0, so a
    // path that reaches S* holds gaps of at most G symbols in all and stays within G diagonals of the one it ends on; a stripe of 8
    // pattern columns then only needs the text rows within G of that diagonal -- 8 + 2 G + 1 of them instead of all N (an opposite-mate
    // window has 500).  Cells on such a path get their exact values from the restricted DP (their best predecessor is on one too),
    // every alternative a direction rule compares them with can only come out lower, never one that ties (a tie would be another
    // optimal path, inside the region as well): the traced path is the full DP's.  Cells outside the region read as -infinity.
    const uint32_t code = (TYPE == NVBIO_SEMI_GLOBAL && band_code) ? band_code[job] : 1u;
    const bool restricted = code >= 2u;
    const int64_t band_G = (int64_t)code - 2, delta = (int64_t)sinks[job].x - (int64_t)sinks[job].y;   // (the sink given by the scoring pass)
    auto rows_of = [&](const uint32_t block, int64_t& lo, int64_t& hi) {
        lo = 0; hi = (int64_t)N - 1;
        if (restricted)
        {
            lo = (int64_t)block + delta - band_G; hi = (int64_t)block + 7 + delta + band_G;
            if (lo < 0) lo = 0;
            if (hi > (int64_t)N - 1) hi = (int64_t)N - 1;
        }
    };

    // scratch layout: wave-major -- the 64 jobs of a wave own one contiguous block (boundary column: max_N lines of 256 bytes, direction
    // words: max_N * nst lines), so that a wave's working set is a few hundred KB of neighbouring pages whatever the number of jobs in
    // the launch (job-interleaved over the whole launch, the first version, put every line of a wave 0.86 MB from the next: one TLB
    // entry per access)
    const size_t wv = t >> 6, ln = t & 63u;
    uint32_t* col = column + wv * ((size_t)max_N * 64u) + ln;   // element i at col[i * 64]
    uint32_t* const dbase = dirs + wv * ((size_t)max_N * nst * 64u) + ln;
    if (!restricted)
        for (uint32_t i = 0; i < N; ++i)
        {
            const int32_t x = (TYPE == NVBIO_GLOBAL) ? sc.txt_go + sc.txt_ge * (int32_t)i : 0;
            const int32_t y = (TYPE == NVBIO_LOCAL) ? 0 : infimum;
            col[(size_t)i * 64u] = pack_cell( x, y );
        }

    Sink sink; sink.init();
    const uint32_t nb        = (M + STRIPE - 1u) / STRIPE;
    const uint32_t end_block = (STRIPE * nb > (uint32_t)STRIPE) ? STRIPE * nb : (uint32_t)STRIPE;
    const uint32_t jm = ((M - 1u) & (STRIPE - 1u)) + 1u;
    uint32_t c_sym[STRIPE]; int32_t c_mm[STRIPE];
    #pragma unroll
    for (int j = 0; j < STRIPE; ++j) { c_sym[j] = 0; c_mm[j] = 0; }
    int32_t H[STRIPE + 1], F[STRIPE + 1];
    bool ok = true;

    for (uint32_t block = 0; block < end_block && ok; block += STRIPE)
    {
        const bool last = (block + STRIPE >= end_block);
        #pragma unroll
        for (int j = 0; j < STRIPE; ++j)
            if (block + j < M)
            {
                const uint32_t idx = J.rev ? J.first + M - 1u - (block + j) : J.first + block + j;
                uint32_t q = prd.get( idx );
                if (J.comp && q < 4u) q = 3u - q;
                const uint32_t qq = b.quals ? b.quals[idx] : 0u;
                c_sym[j] = q; c_mm[j] = s_mm[qq < 63u ? qq : 63u];
            }
        #pragma unroll
        for (int j = 0; j <= STRIPE; ++j)
        {
            H[j] = (TYPE != NVBIO_LOCAL) ? ((block + j > 0) ? T_o + T_e * (int32_t)(block + j - 1u) : 0) : 0;
            F[j] = infimum;
        }
        int32_t max_score = NVBIO_SCORE_MIN;
        int32_t temp_i    = H[0];
        uint32_t* drow = dbase + (size_t)(block / STRIPE) * 64u;              // word of row i at drow[i * nst * 64]

        int64_t lo, hi, plo = 0, phi = -1;
        rows_of( block, lo, hi );
        if (restricted && block > 0u) rows_of( block - STRIPE, plo, phi );   // the rows the previous stripe left in `col`
        // the boundary column of the stripe at row i: the carried column of the previous stripe, -infinity where that stripe did not go,
        // the DP's first column for the first stripe
        // (restricted: rows are stored RELATIVE to the first row of the stripe that writes them, so that the lanes of a wave -- whose
        // bands lie at different text offsets -- still touch the same scratch lines in the same iteration)
        auto col_at = [&](const int64_t i) -> uint32_t {
            if (!restricted) return col[(size_t)i * 64u];
            if (block == 0u) return pack_cell( 0, infimum );
            return (i >= plo && i <= phi) ? col[(size_t)(i - plo) * 64u] : pack_cell( infimum, infimum );
        };
        if (restricted && lo > 0)
        {
            // the row above the first one of the region lies outside it for every column of this stripe
            #pragma unroll
            for (int j = 1; j <= STRIPE; ++j) { H[j] = infimum; F[j] = infimum; }
            temp_i = cell_h( col_at( lo - 1 ) );
        }

        for (int64_t ii = lo; ii <= hi; ++ii)
        {
            const uint32_t i = (uint32_t)ii;
            const uint32_t r_sym = trd.get( J.tb + i );
            int32_t H_diag = temp_i;
            const uint32_t cell = col_at( ii );
            H[0] = temp_i = cell_h( cell );
            int32_t E = cell_e( cell );
            uint32_t dw = 0;
            int32_t key = -1;
            #pragma unroll
            for (int j = 1; j <= STRIPE; ++j)
            {
                const int32_t ftop = F[j] + G_e, htop = H[j] + G_o;
                F[j] = max2( ftop, htop );
                const int32_t eleft = E + I_e, hleft = H[j - 1] + I_o;
                E = max2( eleft, hleft );
                const int32_t d = H_diag + ((c_sym[j - 1] == r_sym) ? V : c_mm[j - 1]);
                const int32_t top = F[j], left = E;
                int32_t hi = max3( left, top, d );
                uint32_t hdir = top > left ? (top > d ? D_DEL : D_SUB) : (left > d ? D_INS : D_SUB);   // gotoh_inl.h:529-531
                if (TYPE == NVBIO_LOCAL) { hi = max2( hi, 0 ); if (hi == 0) hdir = D_SINK; }
                H_diag = H[j];
                H[j]   = hi;
                dw |= (hdir | (eleft > hleft ? D_INS_EXT : 0u) | (ftop > htop ? D_DEL_EXT : 0u)) << (4 * (j - 1));
                if (TYPE == NVBIO_LOCAL && (!last || block + j <= M)) key = max2( key, (hi << 4) | j );
            }
            const size_t ri = restricted ? (size_t)(ii - lo) : (size_t)i;   // row index in the scratch
            // (in place: the slot written, row - lo, lies at or below the slot just read, row - plo, and below every slot still to be read)
            col[ri * 64u] = restricted ? pack_cell( max2( H[STRIPE], infimum ), max2( E, infimum ) ) : pack_cell( H[STRIPE], E );
            drow[ri * nst * 64u] = dw;
            max_score = max2( max_score, H[STRIPE] );
            if (TYPE == NVBIO_LOCAL)
            {
                if (key >= 0) sink.report( key >> 4, i + 1u, block + (uint32_t)(key & 15) );
            }
            else if (last && TYPE == NVBIO_SEMI_GLOBAL)
            {
                int32_t v = 0;
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j) if ((uint32_t)j == jm) v = H[j];
                sink.report( v, i + 1u, M );
            }
        }
        if (!last && !restricted && !given_sinks)
        {
            const int32_t missing = (int32_t)(M - block - STRIPE);
            if (max_score + missing * V < min_score) ok = false;
        }
    }
    if (ok && TYPE == NVBIO_GLOBAL)
    {
        int32_t v = 0;
        #pragma unroll
        for (int j = 1; j <= STRIPE; ++j) if ((uint32_t)j == jm) v = H[j];
        sink.report( v, N, M );
    }

    if (given_sinks) { sink.score = scores[job]; sink.x = sinks[job].x; sink.y = sinks[job].y; }
    scores[job] = sink.score;
    sinks[job]  = make_uint2( sink.x, sink.y );
    if (sink.x == 0xFFFFFFFFu || sink.y == 0xFFFFFFFFu)
    {
        sources[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); cigar_lens[job] = 0;
        return;
    }

    // ---- the walk (gotoh_inl.h:1599-1638), the implicit first row / column (alignment_inl.h:437-452) --------------
    uint16_t* cig = cigars + (size_t)job * cigar_stride;
    uint32_t  clen = 0, prev = 255u, run = 0;
    auto emit = [&](const uint32_t type, const uint32_t len) { if (clen < cigar_stride) cig[clen] = (uint16_t)(type | (len << 2)); ++clen; };
    auto push = [&](const uint32_t op) { if (op == prev) ++run; else { if (run) emit( prev, run ); prev = op; run = 1u; } };
    if (M - sink.y) emit( 3u, M - sink.y );

    int32_t row = (int32_t)sink.x, ccol = (int32_t)sink.y - 1;
    uint32_t state = 0;
    int32_t  w_row = -1, w_st = -1; uint32_t word = 0;
    while (row > 0 && ccol >= 0)
    {
        if (row != w_row || (ccol >> 3) != w_st)
        {
            w_row = row; w_st = ccol >> 3;
            int64_t slo, shi; rows_of( (uint32_t)w_st * STRIPE, slo, shi );
            const size_t ri = restricted ? (size_t)((int64_t)(row - 1) - slo) : (size_t)(row - 1);
            word = dbase[(ri * nst + (uint32_t)w_st) * 64u];
        }
        const uint32_t op = (word >> (4 * (ccol & 7))) & 15u, h_op = op & 3u;
        if (TYPE == NVBIO_LOCAL && state == 0u && h_op == D_SINK) break;
        if (state == 1u)      { if ((op & D_INS_EXT) == 0u) state = 0u; --ccol; push( D_INS ); }
        else if (state == 2u) { if ((op & D_DEL_EXT) == 0u) state = 0u; --row;  push( D_DEL ); }
        else
        {
            if (h_op == D_INS)      state = 1u;
            else if (h_op == D_DEL) state = 2u;
            else { --ccol; --row; push( D_SUB ); }
        }
    }
    uint32_t sx = (uint32_t)row, sy = (uint32_t)(ccol + 1);
    if (TYPE == NVBIO_SEMI_GLOBAL || TYPE == NVBIO_GLOBAL)
        if (sx == 0u) for (; sy > 0u; --sy) push( D_INS );
    if (TYPE == NVBIO_GLOBAL)
        if (sy == 0u) for (; sx > 0u; --sx) push( D_DEL );
    if (run) emit( prev, run );
    if (sy) emit( 3u, sy );
    sources[job]    = make_uint2( sx, sy );
    cigar_lens[job] = clen;
}

// ---- ungapped shortcut: the diagonal through the sink ---------------------------------------------------------------
// Given S* and the sink (x = text end, y = pattern end) of the scoring pass: if the k diagonal steps ending in the sink
// alone score S* -- LOCAL: the smallest such k; SEMI_GLOBAL: k = y, which needs x >= y -- every cell on the diagonal
// holds its prefix score, the diagonal ties for the maximum, ties resolve to SUBSTITUTION, a LOCAL walk stops at the
// first prefix score 0: the traceback is k substitutions.  GLOBAL always takes the DP (its first column is not free).
template <int TYPE, int RBITS, int TBITS>
__global__ void __launch_bounds__(256)
ungapped_full_traceback_kernel(const BatchDev b, const SchemeDev sc, const uint32_t max_M, const uint32_t max_N,
                               const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                               uint2* __restrict__ sources, uint16_t* __restrict__ cigars, const uint32_t cigar_stride,
                               uint32_t* __restrict__ cigar_lens, uint8_t* __restrict__ need_dp,
                               const int32_t gap_open_min = 0, const int32_t gap_ext_min = 0)
{
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= b.n) return;
    const JobInfo J = load_job( b, job );
    const uint2   sink = sinks[job];
    const int32_t best = scores[job];
    need_dp[job] = 0;
    if (J.M > max_M || J.N > max_N) { sources[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); cigar_lens[job] = 0xFFFFFFFFu; return; }
    if (sink.x == 0xFFFFFFFFu || sink.y == 0xFFFFFFFFu) { sources[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); cigar_lens[job] = 0; return; }
    // the DP, over the rows within G diagonals of the sink's when G can be bounded (end-to-end, match bonus 0, gap_ext_min > 0: see
    // full_gotoh_traceback_kernel): need_dp = 2 + G, else 1.  (One job over all rows in a launch of restricted ones is the launch's time.)
    auto dp_code = [&]() -> uint8_t {
        if (TYPE == NVBIO_SEMI_GLOBAL && gap_ext_min > 0 && sink.y == J.M && best <= 0)
        {
            const int32_t a = -best;
            const int32_t G = a < gap_open_min ? 0 : (a - gap_open_min) / gap_ext_min + 1;
            if (G <= 250) return (uint8_t)(2 + G);
        }
        return (uint8_t)1;
    };
    if (TYPE == NVBIO_GLOBAL || (TYPE == NVBIO_SEMI_GLOBAL && sink.x < sink.y)) { need_dp[job] = dp_code(); return; }

    SymbolReader<TBITS> trd( b.text );
    SymbolReader<RBITS> prd( b.reads );
    int32_t Q = 0; uint32_t k = 0;
    bool found = (TYPE == NVBIO_LOCAL) && (best == 0);
    const uint32_t kmax = sink.x < sink.y ? sink.x : sink.y;
    while (k < kmax && !found)
    {
        const uint32_t pc  = sink.y - 1u - k;                    // pattern column, text row sink.x - 1 - k
        const uint32_t idx = J.rev ? J.first + J.M - 1u - pc : J.first + pc;
        uint32_t q = prd.get( idx );
        if (J.comp && q < 4u) q = 3u - q;
        const uint32_t qq = b.quals ? b.quals[idx] : 0u;
        const uint32_t g  = trd.get( J.tb + sink.x - 1u - k );
        Q += (g == q) ? sc.match : s_mm[qq < 63u ? qq : 63u];
        ++k;
        if (TYPE == NVBIO_LOCAL && Q == best) found = true;
    }
    if (TYPE == NVBIO_SEMI_GLOBAL) found = (k == sink.y && Q == best);
    if (!found) { need_dp[job] = dp_code(); return; }

    uint16_t* cig = cigars + (size_t)job * cigar_stride;
    uint32_t  clen = 0;
    auto emit = [&](const uint32_t type, const uint32_t len) { if (clen < cigar_stride) cig[clen] = (uint16_t)(type | (len << 2)); ++clen; };
    if (J.M - sink.y) emit( 3u, J.M - sink.y );
    if (k)            emit( D_SUB, k );
    if (sink.y - k)   emit( 3u, sink.y - k );
    sources[job]    = make_uint2( sink.x - k, sink.y - k );
    cigar_lens[job] = clen;
}

// ---- finish_alignment: edit distance + MDS byte stream of a traced alignment (nvBowtie traceback_inl.h:536-705) --------
template <int RBITS, int TBITS>
__global__ void __launch_bounds__(256)
finish_alignment_kernel(const BatchDev b, const uint2* __restrict__ sources, const uint16_t* __restrict__ cigars, const uint32_t cigar_stride,
                        const uint32_t* __restrict__ cigar_lens, uint32_t* __restrict__ ed_out,
                        uint8_t* __restrict__ mds, const uint32_t mds_stride, uint32_t* __restrict__ mds_lens)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= b.n) return;
    const uint32_t clen = cigar_lens[job];
    if (clen == 0u || clen > cigar_stride || sources[job].x == 0xFFFFFFFFu)     // nothing traced, or a truncated CIGAR
    {
        ed_out[job] = (clen > cigar_stride && clen != 0u) ? 0xFFFFFFFFu : 0u;
        if (mds_lens) mds_lens[job] = 0u;
        return;
    }
    const JobInfo J = load_job( b, job );
    SymbolReader<TBITS> trd( b.text );
    SymbolReader<RBITS> prd( b.reads );
    const uint16_t* cig = cigars + (size_t)job * cigar_stride;
    uint8_t* out = mds ? mds + (size_t)job * mds_stride : nullptr;

    uint32_t mds_len = 2u, ed = 0u, mds_op = 4u;                 // MDS_INVALID
    uint32_t last_run = 0u, last_run_at = 0u;                    // the open MDS_MATCH run lives in a register, written when it closes
    auto put = [&](const uint32_t v) { if (out && mds_len < mds_stride) out[mds_len] = (uint8_t)v; ++mds_len; };
    auto close_run = [&]() { if (last_run && out && last_run_at < mds_stride) out[last_run_at] = (uint8_t)last_run; last_run = 0u; };

    uint32_t j = 0u, k = sources[job].x;
    for (uint32_t i = 0; i < clen; ++i)
    {
        const uint32_t el = cig[clen - i - 1u];
        const uint32_t l = el >> 2, t = el & 3u;
        if (t != 0u) { close_run(); mds_op = (t == 2u) ? 3u : 2u; put( mds_op ); put( l ); }
        for (uint32_t n = 0; n < l; ++n)
        {
            if (t != 2u) ++j;
            if (t == 0u || t == 2u) ++k;
            uint32_t readc = 255u, refc = 255u;
            if (t != 2u && j <= J.M)
            {
                readc = prd.get( J.rev ? J.first + J.M - j : J.first + j - 1u );
                if (J.comp && readc < 4u) readc = 3u - readc;
            }
            if ((t == 0u || t == 2u) && k <= J.N) refc = trd.get( J.tb + k - 1u );
            if (t == 0u)
            {
                if (readc == refc)
                {
                    if (mds_op == 0u && last_run < 255u) ++last_run;
                    else { close_run(); mds_op = 0u; put( 0u ); last_run_at = mds_len; last_run = 1u; put( 1u ); }
                }
                else { close_run(); mds_op = 1u; put( 1u ); put( readc ); ++ed; }
            }
            else
            {
                put( t == 2u ? refc : readc );
                if (t != 3u) ++ed;
            }
        }
    }
    close_run();
    if (out && mds_stride >= 2u) { out[0] = (uint8_t)(mds_len & 0xFFu); out[1] = (uint8_t)(mds_len >> 8); }
    ed_out[job] = ed;
    if (mds_lens) mds_lens[job] = mds_len;
}

// ---- the band route: restricted jobs through the banded traceback kernel ---------------------------------------------------
// A job whose optimal paths stay within G <= 7 diagonals of its sink's (need_dp = 2 + G, see full_gotoh_traceback_kernel) is a band-15
// problem: the banded traceback kernel (gotoh_traceback.hip) takes it over a window of M + 14 symbols around that diagonal, with the full
// matrix's rule for ties between the two gap moves.  The argument is the one of the restricted rows: every cell on an optimal path gets
// its exact value inside the band, every alternative a direction rule compares it with can only come out lower, never tie.  Needs the
// band inside the window (sink diagonal between 7 and N - M - 7); the others keep the restricted full-matrix kernel.
__global__ void __launch_bounds__(256)
tb_band_route_kernel(const BatchDev b, uint8_t* __restrict__ need_dp, const uint2* __restrict__ sinks,
                     uint32_t* __restrict__ wb2, uint32_t* __restrict__ we2, uint8_t* __restrict__ route)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= b.n) return;
    const JobInfo J = load_job( b, job );
    const uint32_t code = need_dp[job];
    const uint2 sink = sinks[job];
    bool ok = code >= 2u && code - 2u <= 7u && sink.y == J.M && sink.x >= sink.y && J.N >= J.M + 14u;
    uint32_t delta = 7u;
    if (ok)
    {
        delta = sink.x - sink.y;
        ok = delta >= 7u && delta + 7u <= J.N - J.M;
    }
    if (!ok) delta = 7u;
    wb2[job] = J.tb + delta - 7u; we2[job] = J.tb + delta + 7u + J.M;
    route[job] = ok ? 1 : 0;
    if (ok) need_dp[job] = 0;                                    // not for the full-matrix kernel's list
}

// sources and sinks of the routed jobs back in the window's coordinates
__global__ void __launch_bounds__(256)
tb_band_fixup_kernel(const BatchDev b, const uint32_t* __restrict__ wb2, const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count,
                     uint2* __restrict__ sources, uint2* __restrict__ sinks)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= *job_count) return;
    const uint32_t job = job_list[slot];
    const uint32_t off = wb2[job] - b.win_begin[job];
    if (sources[job].x != 0xFFFFFFFFu) sources[job].x += off;
    if (sinks[job].x   != 0xFFFFFFFFu) sinks[job].x   += off;
}

} // anonymous namespace
} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" nvbio_status nvbio_finish_alignment(int device, const nvbio_alignment_batch* batch, const nvbio_uint2* sources_dev,
                                               const uint16_t* cigars_dev, uint32_t cigar_stride, const uint32_t* cigar_lens_dev,
                                               uint32_t* ed_dev, uint8_t* mds_dev, uint32_t mds_stride, uint32_t* mds_lens_dev, void* stream)
{
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( sources_dev && cigars_dev && cigar_lens_dev && ed_dev, "NULL device pointer" );
    NVB_REQUIRE( mds_dev == nullptr || (mds_stride >= 2 && mds_lens_dev != nullptr), "mds_dev needs mds_stride >= 2 and mds_lens_dev" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    const dim3 grid( (b.n + 255u) / 256u ), block( 256 );
    const uint32_t rb = batch->read_bits, tbits = batch->text_bits;
    hipStream_t s = (hipStream_t)stream;
#define NVB_FIN(RB, TB) hipLaunchKernelGGL( (finish_alignment_kernel<RB,TB>), grid, block, 0, s, b, (const uint2*)sources_dev, cigars_dev, cigar_stride, \
                                            cigar_lens_dev, ed_dev, mds_dev, mds_stride, mds_lens_dev )
    if      (rb == 4 && tbits == 2) NVB_FIN( 4, 2 ); else if (rb == 2 && tbits == 2) NVB_FIN( 2, 2 );
    else if (rb == 8 && tbits == 2) NVB_FIN( 8, 2 ); else if (rb == 8 && tbits == 8) NVB_FIN( 8, 8 );
    else if (rb == 4 && tbits == 8) NVB_FIN( 4, 8 ); else NVB_FIN( 2, 8 );
#undef NVB_FIN
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

static inline uint64_t full_tb_bytes_per_job(const uint32_t max_M, const uint32_t max_N)
{
    return (uint64_t)max_N * sizeof(uint32_t) * (1u + (max_M + 7u) / 8u);       // boundary column + direction words
}

extern "C" nvbio_status nvbio_full_gotoh_traceback_temp_bytes(const nvbio_alignment_batch* batch, uint32_t max_pattern_len,
                                                              uint32_t max_text_len, uint64_t* bytes)
{
    NVB_REQUIRE( batch && bytes, "batch/bytes is NULL" );
    NVB_REQUIRE( max_pattern_len > 0 && max_text_len > 0, "max_pattern_len / max_text_len must be positive" );
    *bytes = (((uint64_t)batch->n + 63u) & ~63ull) * full_tb_bytes_per_job( max_pattern_len, max_text_len );   // whole waves of 64 jobs own scratch
    return NVBIO_OK;
}

// gotoh != NULL: the Gotoh aligner; sw != NULL: the linear-gap Smith-Waterman / edit-distance aligner (deletion and insertion may differ)
static nvbio_status full_traceback_impl(int device, nvbio_alignment_type type, const nvbio_gotoh_scheme* gotoh, const nvbio_sw_scheme* sw,
                                        const nvbio_alignment_batch* batch, uint32_t max_pattern_len, uint32_t max_text_len,
                                        const int32_t* min_scores_dev,
                                        int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                        uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                        uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( gotoh != nullptr || sw != nullptr, "scheme is NULL" );
    nvbio_gotoh_scheme as_gotoh;                                 // the magnitudes of the scheme, for the int16 bound below
    if (sw) as_gotoh = nvbio_gotoh_scheme{ sw->match, -sw->mismatch, -sw->mismatch, sw->deletion, sw->deletion, sw->insertion, sw->insertion };
    const nvbio_gotoh_scheme* scheme = gotoh ? gotoh : &as_gotoh;
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( type == NVBIO_GLOBAL || type == NVBIO_LOCAL || type == NVBIO_SEMI_GLOBAL, "invalid alignment type" );
    NVB_REQUIRE( scores_dev && sources_dev && sinks_dev && cigar_lens_dev, "NULL output pointer" );
    NVB_REQUIRE( cigars_dev != nullptr || cigar_stride == 0, "cigars_dev is NULL" );
    NVB_REQUIRE( max_pattern_len > 0 && max_text_len > 0, "max_pattern_len / max_text_len must bound the jobs (they size the scratch)" );
    {
        int64_t step = scheme->match < 0 ? -(int64_t)scheme->match : scheme->match;
        const int64_t c[] = { scheme->mm_min, scheme->mm_max, -(int64_t)scheme->pat_gap_open, -(int64_t)scheme->pat_gap_ext,
                              -(int64_t)scheme->txt_gap_open, -(int64_t)scheme->txt_gap_ext };
        for (int64_t v : c) { if (v < 0) v = -v; if (v > step) step = v; }
        if (((int64_t)max_pattern_len + max_text_len + 1) * step > 30000)
        {
            set_error( "full traceback: scores of %u x %u jobs under this scheme can overflow the reference's int16 checkpoints", max_pattern_len, max_text_len );
            return NVBIO_ERR_UNSUPPORTED;
        }
    }
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const SchemeDev sc = sw ? scheme_dev( sw, false ) : scheme_dev( gotoh );
    const uint32_t rb = batch->read_bits, tbits = batch->text_bits;

    // ---- 1. scoring pass (pattern blocking) unless handed over; 2. the ungapped shortcut; 3. job list ----
    // (Smith-Waterman aligner: score and sink always come from its own scoring pass -- the kernel below is told so -- and every job
    // takes the DP: the shortcut's tie rules are the Gotoh aligner's)
    if (sw && !(flags & NVBIO_TRACEBACK_SINKS_GIVEN))
        NVB_CHECK( nvbio_full_sw_score( device, type, 0, sw, batch, max_pattern_len, max_text_len, min_scores_dev, scores_dev, sinks_dev,
                                        nullptr, 0, stream ) );
    const bool shortcut = !(b.algo & NVBIO_ALN_NO_UNGAPPED_TRACEBACK) && !sw;
    uint32_t *job_list = nullptr, *job_count = nullptr; void* aux = nullptr; uint8_t* need_dp = nullptr;
    uint32_t *band_list = nullptr, *band_count = nullptr, *band_wb = nullptr, *band_we = nullptr; uint8_t* band_route = nullptr;
    // the row-restricted DP applies to nvBowtie's end-to-end mode (see full_gotoh_traceback_kernel)
    const int32_t go_min = -(sc.pat_go > sc.txt_go ? sc.pat_go : sc.txt_go), ge_min = -(sc.pat_ge > sc.txt_ge ? sc.pat_ge : sc.txt_ge);
    const bool narrow = type == NVBIO_SEMI_GLOBAL && sc.match == 0 && sc.mm_min >= 0 && sc.mm_max >= 0 && plain_gotoh( sc ) &&
                        ge_min > 0 && go_min >= ge_min && !(b.algo & NVBIO_ALN_NO_NARROW_TRACEBACK);
    const bool band_ok = narrow && shortcut && !(b.algo & NVBIO_ALN_NO_BAND_ROUTE);
    if (shortcut)
    {
        if (!(flags & NVBIO_TRACEBACK_SINKS_GIVEN))
            NVB_CHECK( nvbio_full_gotoh_score( device, type, 0, gotoh, batch, max_pattern_len, max_text_len, min_scores_dev, scores_dev, sinks_dev,
                                               nullptr, 0, stream ) );
        size_t sel_bytes = 0;
        hipcub::CountingInputIterator<uint32_t> ids( 0u );
        NVB_HIP( hipcub::DeviceSelect::Flagged( nullptr, sel_bytes, ids, (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)b.n, s ) );
        const uint64_t flags_bytes = ((uint64_t)b.n + 255u) & ~255ull;
        const uint64_t list_bytes  = ((uint64_t)b.n * 4u + 255u) & ~255ull;
        if (scratch_alloc( &aux, 2u * flags_bytes + 4u * list_bytes + 256u + sel_bytes, s ) != hipSuccess)
        {
            (void)hipGetLastError();
            set_error( "full traceback: out of device memory for the job list" );
            return NVBIO_ERR_NOMEM;
        }
        need_dp   = (uint8_t*)aux;
        job_list  = (uint32_t*)((uint8_t*)aux + flags_bytes);
        job_count = (uint32_t*)((uint8_t*)aux + flags_bytes + list_bytes);
        band_count = job_count + 1;
        band_route = (uint8_t*)job_count + 256u;
        band_list  = (uint32_t*)(band_route + flags_bytes);
        band_wb    = (uint32_t*)((uint8_t*)band_list + list_bytes);
        band_we    = (uint32_t*)((uint8_t*)band_wb + list_bytes);
        void* sel_temp = (uint8_t*)band_we + list_bytes;
        const dim3 grid( (b.n + 255u) / 256u ), block( 256 );
#define NVB_UNG(TYPE_, RB, TB) hipLaunchKernelGGL( (ungapped_full_traceback_kernel<TYPE_,RB,TB>), grid, block, 0, s, b, sc, max_pattern_len, max_text_len, \
                                                   (const int32_t*)scores_dev, (const uint2*)sinks_dev, (uint2*)sources_dev, cigars_dev, cigar_stride, cigar_lens_dev, need_dp, \
                                                   go_min, narrow ? ge_min : 0 )
#define NVB_UNG_BITS(TYPE_) \
        if      (rb == 4 && tbits == 2) NVB_UNG( TYPE_, 4, 2 ); else if (rb == 2 && tbits == 2) NVB_UNG( TYPE_, 2, 2 ); \
        else if (rb == 8 && tbits == 2) NVB_UNG( TYPE_, 8, 2 ); else if (rb == 8 && tbits == 8) NVB_UNG( TYPE_, 8, 8 ); \
        else if (rb == 4 && tbits == 8) NVB_UNG( TYPE_, 4, 8 ); else NVB_UNG( TYPE_, 2, 8 )
        if (type == NVBIO_GLOBAL) { NVB_UNG_BITS( NVBIO_GLOBAL ); } else if (type == NVBIO_LOCAL) { NVB_UNG_BITS( NVBIO_LOCAL ); } else { NVB_UNG_BITS( NVBIO_SEMI_GLOBAL ); }
#undef NVB_UNG_BITS
#undef NVB_UNG
        hipError_t e = hipSuccess;
        if (band_ok)
        {
            // the restricted jobs that fit a band of 15 leave the full-matrix list (see tb_band_route_kernel); launched below
            hipLaunchKernelGGL( tb_band_route_kernel, grid, block, 0, s, b, need_dp, (const uint2*)sinks_dev, band_wb, band_we, band_route );
            e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, band_route, band_list, band_count, (int)b.n, s );
        }
        if (e == hipSuccess) e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, need_dp, job_list, job_count, (int)b.n, s );
        if (e != hipSuccess) { scratch_free( aux, s ); set_error( "DeviceSelect failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    }

    // ---- 4. the DP with direction vectors + walk, over the job list (or every job) ----
    const uint64_t per_job = full_tb_bytes_per_job( max_pattern_len, max_text_len );
    void* owned = nullptr; uint8_t* scratch = (uint8_t*)temp_dev; uint64_t cap_jobs;
    if (scratch)
    {
        cap_jobs = (temp_bytes / per_job) & ~63ull;                  // whole waves of 64 jobs
        if (cap_jobs < 64)
        {
            if (aux) scratch_free( aux, s );
            set_error( "invalid argument: temp_bytes too small (see nvbio_full_gotoh_traceback_temp_bytes)" );
            return NVBIO_ERR_INVALID;
        }
    }
    else
    {
        cap_jobs = shortcut ? ((uint64_t)b.n + 3u) / 4u : b.n;
        if (cap_jobs < 16384u) cap_jobs = b.n < 16384u ? b.n : 16384u;
        const uint64_t budget = 8ull << 30;
        if (cap_jobs * per_job > budget) cap_jobs = budget / per_job;
        cap_jobs = (cap_jobs + 63u) & ~63ull;
        if (scratch_alloc( &owned, cap_jobs * per_job, s ) != hipSuccess)
        {
            (void)hipGetLastError();
            if (aux) scratch_free( aux, s );
            set_error( "full traceback: out of device memory for %llu direction matrices", (unsigned long long)cap_jobs );
            return NVBIO_ERR_NOMEM;
        }
        scratch = (uint8_t*)owned;
    }
    nvbio_status st = NVBIO_OK;
    if (band_ok)
    {
        BatchDev b2 = b; b2.win_begin = band_wb; b2.win_end = band_we; b2.max_read_len = max_pattern_len;
        st = banded15_full_ties_traceback( b2, sc, rb, tbits, b.n, band_list, band_count, (uint32_t*)scratch, cap_jobs * per_job,
                                           scores_dev, (uint2*)sources_dev, (uint2*)sinks_dev, cigars_dev, cigar_stride, cigar_lens_dev, s );
        if (st == NVBIO_OK)
            hipLaunchKernelGGL( tb_band_fixup_kernel, dim3( (b.n + 255u) / 256u ), dim3( 256 ), 0, s, b, (const uint32_t*)band_wb, (const uint32_t*)band_list,
                                (const uint32_t*)band_count, (uint2*)sources_dev, (uint2*)sinks_dev );
    }
    for (uint64_t begin = 0; begin < b.n && st == NVBIO_OK; begin += cap_jobs)
    {
        const uint32_t jobs = (uint32_t)((b.n - begin) < cap_jobs ? (b.n - begin) : cap_jobs);
        uint32_t* column = (uint32_t*)scratch;
        const uint64_t jobs64 = ((uint64_t)jobs + 63u) & ~63ull;                  // whole waves own scratch
        uint32_t* dirs   = column + (size_t)jobs64 * max_text_len;
        const dim3 grid( (jobs + 127u) / 128u ), block( 128 );
#define NVB_TB(TYPE_, RB, TB) hipLaunchKernelGGL( (full_gotoh_traceback_kernel<TYPE_,RB,TB>), grid, block, 0, s, b, sc, max_pattern_len, max_text_len, (uint32_t)begin, jobs, \
                                                  (const uint32_t*)job_list, (const uint32_t*)job_count, min_scores_dev, column, dirs, scores_dev, (uint2*)sources_dev, \
                                                  (uint2*)sinks_dev, cigars_dev, cigar_stride, cigar_lens_dev, (const uint8_t*)(narrow ? need_dp : nullptr), sw ? 1u : 0u )
#define NVB_TB_BITS(TYPE_) \
        if      (rb == 4 && tbits == 2) NVB_TB( TYPE_, 4, 2 ); else if (rb == 2 && tbits == 2) NVB_TB( TYPE_, 2, 2 ); \
        else if (rb == 8 && tbits == 2) NVB_TB( TYPE_, 8, 2 ); else if (rb == 8 && tbits == 8) NVB_TB( TYPE_, 8, 8 ); \
        else if (rb == 4 && tbits == 8) NVB_TB( TYPE_, 4, 8 ); else NVB_TB( TYPE_, 2, 8 )
        if (type == NVBIO_GLOBAL) { NVB_TB_BITS( NVBIO_GLOBAL ); } else if (type == NVBIO_LOCAL) { NVB_TB_BITS( NVBIO_LOCAL ); } else { NVB_TB_BITS( NVBIO_SEMI_GLOBAL ); }
#undef NVB_TB_BITS
#undef NVB_TB
        if (hipGetLastError() != hipSuccess) { set_error( "full traceback launch failed" ); st = NVBIO_ERR_HIP; }
    }
    if (owned) scratch_free( owned, s );
    if (aux)   scratch_free( aux, s );
    return st;
}

extern "C" nvbio_status nvbio_full_gotoh_traceback(int device, nvbio_alignment_type type, const nvbio_gotoh_scheme* scheme,
                                                   const nvbio_alignment_batch* batch, uint32_t max_pattern_len, uint32_t max_text_len,
                                                   const int32_t* min_scores_dev,
                                                   int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                                   uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                                   uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    return full_traceback_impl( device, type, scheme, nullptr, batch, max_pattern_len, max_text_len, min_scores_dev, scores_dev, sources_dev, sinks_dev,
                                cigars_dev, cigar_stride, cigar_lens_dev, flags, temp_dev, temp_bytes, stream );
}

extern "C" nvbio_status nvbio_full_sw_traceback(int device, nvbio_alignment_type type, const nvbio_sw_scheme* scheme,
                                                const nvbio_alignment_batch* batch, uint32_t max_pattern_len, uint32_t max_text_len,
                                                const int32_t* min_scores_dev,
                                                int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                                uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                                uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    return full_traceback_impl( device, type, nullptr, scheme, batch, max_pattern_len, max_text_len, min_scores_dev, scores_dev, sources_dev, sinks_dev,
                                cigars_dev, cigar_stride, cigar_lens_dev, flags, temp_dev, temp_bytes, stream );
}
