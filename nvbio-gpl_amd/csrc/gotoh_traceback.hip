// gotoh_traceback.hip -- batched banded Gotoh traceback (score + CIGAR) for gfx950.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   banded_alignment_traceback (driver: score pass, clip, walk, clip)   nvbio/alignment/banded_inl.h:354-417
//   direction vectors per cell (hdir | edir | fdir)                     nvbio/alignment/gotoh/gotoh_banded_inl.h:461-602, :316-330
//   priv::banded_alignment_traceback (the H/E/F state walk)             gotoh_banded_inl.h:872-948
//   DirectionVector / State encodings                                   nvbio/alignment/alignment.h:326-346
//   nvBowtie's run-length Backtracker and io::Cigar                     nvBowtie/bowtie2/cuda/alignment_utils.h:115-157, nvbio/io/alignments.h:48-66
//   batched driver                                                      nvbio/alignment/batched_banded_inl.h (BatchedBandedAlignmentTraceback)
//
// MI355X design.  The reference keeps one int16 checkpoint of the band every 16 rows and recomputes
// each 16-row block of direction vectors on the way back (twice the DP, plus a per-thread
// submatrix in local memory) because a K40 has 12 GB.  With 288 GB of HBM the direction vectors of
// the WHOLE band are simply written out during the one forward pass: a row of a band-31 DP is
// 31 x 4 bits = one 16-byte vector per lane, stored row-major / job-interleaved so that a wave
// writes 4 x 256 contiguous bytes per row (2.4 KB per 150 bp alignment; the launch is chunked to
// the scratch the caller grants).  The walk back then reads at most one row vector per step,
// again coalesced across the wave, and run-length encodes straight into the caller's CIGAR array.
// The two are identical as long as every score fits the reference's int16 checkpoints, which the
// host checks from the scheme and the batch's max_read_len (else NVBIO_ERR_UNSUPPORTED).
#include "gotoh_common.h"
#include "bitplanes.h"
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace nvbio_amd {

namespace {

enum : uint32_t { D_SUB = 0u, D_INS = 1u, D_DEL = 2u, D_SINK = 3u, D_INS_EXT = 4u, D_DEL_EXT = 8u };

template <int BAND, int TYPE, int RBITS, int TBITS>
__global__ void __launch_bounds__(128)
banded_gotoh_traceback_kernel(const BatchDev b, const SchemeDev sc, const uint32_t job_begin, const uint32_t jobs,
                              const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count,
                              uint32_t* __restrict__ dirs,
                              int32_t* __restrict__ scores, uint2* __restrict__ sources, uint2* __restrict__ sinks,
                              uint16_t* __restrict__ cigars, const uint32_t cigar_stride, uint32_t* __restrict__ cigar_lens,
                              const uint8_t* __restrict__ band_off = nullptr, const uint32_t full_ties = 0u)
{
    constexpr int WORDS = (BAND + 7) / 8;                        // 32-bit words of direction nibbles per row

    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // slot inside this launch
    if (t >= jobs) return;
    // with a job list (the jobs the ungapped pass could not settle) slot t of this launch is entry
    // job_begin + t of the list, and the list's length lives on the device
    if (job_list && job_begin + t >= *job_count) return;
    const uint32_t job = job_list ? job_list[job_begin + t] : job_begin + t;

    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    // band_off (a band-15 / band-7 launch over jobs of a band-31 batch, see the narrow-band route at the entry point): this job's band covers
    // columns [off, off + BAND) of the band the batch asked for -- the window begins `off` symbols later, sink and source move back by it
    const uint32_t off   = band_off ? band_off[job] : 0u;
    const uint32_t tb    = b.win_begin[job] + off;
    const uint32_t N     = b.win_end[job] - tb;

    int32_t  best   = NVBIO_SCORE_MIN;
    uint32_t best_x = 0xFFFFFFFFu, best_y = 0xFFFFFFFFu;

    if (M > b.max_read_len)                                      // would overrun the direction-vector scratch: skip, flagged
    {
        scores[job] = best; sinks[job] = sources[job] = make_uint2( best_x, best_y );
        cigar_lens[job] = 0xFFFFFFFFu;
        return;
    }

    if (N >= M)                                                  // else nothing is reported (gotoh_banded_inl.h:422-423)
    {
        constexpr bool PACKED = !(BAND == 3 || BAND == 5 || BAND == 7 || BAND == 15);

        SymbolReader<TBITS> trd( b.text );
        SymbolReader<RBITS> prd( b.reads );

        uint64_t cache_bits = 0;                                 // PACKED: symbol j at bits [2j,2j+1]
        uint32_t cache_raw[PACKED ? 1 : BAND - 1];
        #pragma unroll
        for (int j = 0; j < BAND - 1; ++j)
        {
            const uint32_t g = ((uint32_t)j < N) ? trd.get( tb + j ) : 255u;
            if (PACKED) cache_bits |= (uint64_t)(g & 3u) << (2 * j);
            else        cache_raw[j] = g;
        }

        const int32_t G_o = sc.pat_go, G_e = sc.pat_ge;             // F: the text advances alone
        const int32_t I_o = sc.ins_go, I_e = sc.ins_ge;             // E: the pattern advances alone (= G for the Gotoh aligner; the
                                                                    // Smith-Waterman aligner's insertion, sw_banded_inl.h:420-436)
        // the Smith-Waterman aligner's direction vectors carry no SINK: its LOCAL walk runs on to the first row (sw_banded_inl.h:420-436,758-790)
        const bool sink_marks = !sc.wide;
        const int32_t infimum = -32768 - max2( max2( G_o, G_e ), max2( sc.txt_go, sc.txt_ge ) );
        const int32_t V = sc.match;

        int32_t H[BAND], F[BAND];
        H[0] = 0;
        #pragma unroll
        for (int j = 1; j < BAND; ++j) H[j] = (TYPE == NVBIO_GLOBAL) ? sc.txt_go + (j - 1) * sc.txt_ge : 0;
        #pragma unroll
        for (int j = 0; j < BAND; ++j) F[j] = infimum;

        for (uint32_t i = 0; i < M; ++i)
        {
            const uint32_t pidx = rev ? first + M - 1u - i : first + i;
            uint32_t q = prd.get( pidx );
            if (comp && q < 4u) q = 3u - q;
            const uint32_t qq = b.quals ? b.quals[pidx] : 0u;
            const int32_t  S  = s_mm[qq < 63u ? qq : 63u];

            const uint32_t g_new = (i + (uint32_t)(BAND - 1) < N) ? trd.get( tb + i + (BAND - 1) ) : 255u;

            uint64_t eq_bits = 0;
            if (PACKED && q < 4u)
            {
                const uint64_t x = cache_bits ^ ((uint64_t)q * 0x5555555555555555ull);
                eq_bits = ~(x | (x >> 1)) & 0x5555555555555555ull;
            }

            uint32_t dw[WORDS];
            #pragma unroll
            for (int w = 0; w < WORDS; ++w) dw[w] = 0;

            int32_t  E = 0;
            uint32_t edir = D_SUB;
            int32_t  row_key = -1;
            #pragma unroll
            for (int j = 0; j < BAND; ++j)
            {
                // F and its direction (:476-480,513-517; column BAND-1: :575-576)
                int32_t f = infimum; uint32_t fdir = D_SUB;
                if (j < BAND - 1)
                {
                    const int32_t ftop = F[j + 1] + G_e, htop = H[j + 1] + G_o;
                    f = max2( ftop, htop );
                    fdir = ftop > htop ? D_DEL_EXT : D_SUB;
                }
                F[j] = f;

                bool eq;
                if (j == BAND - 1)   eq = (g_new == q);
                else if (PACKED)     eq = ((eq_bits >> (2 * j)) & 1ull) != 0;
                else                 eq = (cache_raw[j] == q);
                const int32_t d = H[j] + (eq ? V : S);

                int32_t h; uint32_t hdir;
                if (j == 0)             { h = max2( f, d ); hdir = f > d ? D_INS : D_SUB; }                   // :486-503
                else if (j == BAND - 1) { h = max2( E, d ); hdir = E > d ? D_DEL : D_SUB; }                   // :579-601
                else
                {
                    h = max3( f, E, d );                                                                      // :534-557
                    hdir = f > E ? (f > d ? D_INS : D_SUB) : (E > d ? D_DEL : D_SUB);
                    // full_ties (a band cut out of a FULL matrix, gotoh_full_traceback.hip): where the two gap moves tie above the
                    // diagonal one, the full-matrix reference asks for the deletion first and so ends on the insertion
                    // (gotoh_inl.h:529-531), the banded one the other way round
                    if (full_ties && f == E && f > d) hdir = D_INS;
                }
                if (TYPE == NVBIO_LOCAL)
                {
                    h = max2( h, 0 );
                    if (h == 0 && sink_marks) hdir = D_SINK;
                    row_key = max2( row_key, (h << 5) | j );
                }
                H[j] = h;
                dw[j >> 3] |= (hdir | edir | fdir) << (4 * (j & 7));

                // E for the next column and its direction (:507,560-565)
                if (j == 0) { E = h + I_o; edir = D_SUB; }
                else
                {
                    const int32_t eleft = E + I_e, ediag = h + I_o;
                    edir = eleft > ediag ? D_INS_EXT : D_SUB;
                    E = max2( ediag, eleft );
                }
            }
            #pragma unroll
            for (int w = 0; w < WORDS; ++w) dirs[((size_t)i * WORDS + w) * jobs + t] = dw[w];

            if (PACKED) cache_bits = (cache_bits >> 2) | ((uint64_t)(g_new & 3u) << (2 * (BAND - 2)));
            else
            {
                #pragma unroll
                for (int j = 0; j < BAND - 2; ++j) cache_raw[j] = cache_raw[j + 1];
                cache_raw[BAND - 2] = g_new;
            }

            if (TYPE == NVBIO_LOCAL)
            {
                const int32_t h = row_key >> 5;
                if (h >= best) { best = h; best_x = i + (uint32_t)(row_key & 31) + 1u; best_y = i + 1u; }
            }
        }

        if (TYPE == NVBIO_GLOBAL)                                // :629-630
        {
            if (best <= H[BAND - 1]) { best = H[BAND - 1]; best_x = M + BAND - 1; best_y = M; }
        }
        else if (TYPE == NVBIO_SEMI_GLOBAL)                      // :631-643
        {
            const uint32_t mb = M + (uint32_t)(BAND - 1);
            const uint32_t m  = (mb < N ? mb : N) - (M - 1u);
            #pragma unroll
            for (int j = 0; j < BAND; ++j)
                if (j == 0 || (uint32_t)j < m)
                    if (best <= H[j]) { best = H[j]; best_x = M + j; best_y = M; }
        }
    }

    scores[job] = best;
    sinks[job]  = make_uint2( best_x == 0xFFFFFFFFu ? best_x : best_x + off, best_y );
    if (best_x == 0xFFFFFFFFu || best_y == 0xFFFFFFFFu)         // banded_inl.h:376-379: nothing to trace
    {
        sources[job]    = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu );
        cigar_lens[job] = 0;
        return;
    }

    // ---- the walk back (gotoh_banded_inl.h:884-947), run-length encoded as nvBowtie's Backtracker does ----
    uint16_t* cig = cigars + (size_t)job * cigar_stride;
    uint32_t  clen = 0;
    auto emit = [&](const uint32_t type, const uint32_t len) {
        if (clen < cigar_stride) cig[clen] = (uint16_t)(type | (len << 2));
        ++clen;
    };
    if (M - best_y) emit( 3u, M - best_y );                      // clip the end of the pattern (banded_inl.h:382)

    int32_t  entry = (int32_t)(best_x - best_y);
    int32_t  row   = (int32_t)best_y - 1;
    uint32_t state = 0;                                          // HSTATE 0, ESTATE 1, FSTATE 2
    uint32_t prev = 255u, run = 0;
    uint32_t src_x = 0, src_y = 0;
    bool     found = false;

    int32_t  loaded_row = -1;
    uint32_t rw[WORDS];
    #pragma unroll
    for (int w = 0; w < WORDS; ++w) rw[w] = 0;

    while (row >= 0)
    {
        if (row != loaded_row)
        {
            #pragma unroll
            for (int w = 0; w < WORDS; ++w) rw[w] = dirs[((size_t)row * WORDS + w) * jobs + t];
            loaded_row = row;
        }
        uint32_t word = rw[0];
        #pragma unroll
        for (int w = 1; w < WORDS; ++w) if ((entry >> 3) == w) word = rw[w];
        const uint32_t op   = (word >> (4 * (entry & 7))) & 15u;
        const uint32_t h_op = op & 3u;

        if (TYPE == NVBIO_LOCAL && state == 0u && h_op == D_SINK)
        {
            src_y = (uint32_t)row + 1u; src_x = (uint32_t)entry + src_y; found = true;
            break;
        }
        uint32_t push = 255u;
        if (state == 1u)      { if ((op & D_INS_EXT) == 0u) state = 0u; --entry; push = D_DEL; }
        else if (state == 2u) { if ((op & D_DEL_EXT) == 0u) state = 0u; ++entry; --row; push = D_INS; }
        else
        {
            if (h_op == D_DEL)      state = 1u;
            else if (h_op == D_INS) state = 2u;
            else { --row; push = D_SUB; }
        }
        if (push != 255u)
        {
            if (push == prev) ++run;
            else { if (run) emit( prev, run ); prev = push; run = 1u; }
        }
    }
    if (run) emit( prev, run );
    if (!found) { src_y = 0u; src_x = (uint32_t)entry; }
    if (src_y) emit( 3u, src_y );                                // clip the beginning (banded_inl.h:413)

    sources[job]    = make_uint2( src_x + off, src_y );
    cigar_lens[job] = clen;
}

// ---------------------------------------------------------------------------------------------
// Ungapped shortcut.  Let S* and the sink come from the scoring pass, and let Q_k be the score of
// the k diagonal steps that end in the sink.  If the diagonal alone reaches S* (LOCAL: some Q_k = S*;
// otherwise H_0[entry] + Q_sink.y = S*), every cell on it holds exactly its prefix score -- more
// would beat the optimum, less would not reach it -- so at each of them the diagonal move ties for
// the maximum, and the reference's direction rule resolves ties to SUBSTITUTION (strict `>` at
// gotoh_banded_inl.h:553-556); a LOCAL walk stops at the first cell whose score is 0, i.e. at the
// smallest k with Q_k = S*.  The traceback of such a job is therefore k substitutions along the
// diagonal, known without running the DP; only the other jobs (need_dp = 1) go through it.
// Symbols compare exactly as in the DP: beyond the text end the sentinel 255, which a band-31
// (2-bit cached) column < 30 sees as 3 (alignment_base_inl.h:66-90).
// ---------------------------------------------------------------------------------------------
template <int BAND, int TYPE, int RBITS, int TBITS>
__global__ void __launch_bounds__(256)
ungapped_traceback_kernel(const BatchDev b, const SchemeDev sc, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                          uint2* __restrict__ sources, uint16_t* __restrict__ cigars, const uint32_t cigar_stride,
                          uint32_t* __restrict__ cigar_lens, uint8_t* __restrict__ need_dp,
                          uint8_t* __restrict__ band_off = nullptr, const int32_t gap_open_min = 0, const int32_t gap_ext_min = 0)
{
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= b.n) return;

    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = b.win_end[job] - tb;
    const uint2    sink  = sinks[job];
    const int32_t  best  = scores[job];

    need_dp[job] = 0;
    if (M > b.max_read_len) { sources[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); cigar_lens[job] = 0xFFFFFFFFu; return; }
    if (sink.x == 0xFFFFFFFFu || sink.y == 0xFFFFFFFFu)
    {
        sources[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); cigar_lens[job] = 0;
        return;
    }

    constexpr bool PACKED = !(BAND == 3 || BAND == 5 || BAND == 7 || BAND == 15);
    const uint32_t entry = sink.x - sink.y;
    const int32_t  H0 = (TYPE == NVBIO_GLOBAL && entry > 0u) ? sc.txt_go + (int32_t)(entry - 1u) * sc.txt_ge : 0;

    SymbolReader<TBITS> trd( b.text );
    SymbolReader<RBITS> prd( b.reads );

    int32_t  Q = 0;
    uint32_t k = 0;
    bool found = (TYPE == NVBIO_LOCAL) && (best == 0);
    bool counted = false;
    if (BAND == 31 && TYPE != NVBIO_LOCAL && TBITS == 2 && (RBITS == 2 || RBITS == 4))
    {
        // one mismatch penalty for every row and the diagonal inside the text: its score is a popcount over bit planes
        // (bitplanes.h) instead of a walk symbol by symbol -- the end-to-end case of a read batch
        const bool const_pen = (b.quals == nullptr) || (sc.mm_min == sc.mm_max);
        if (const_pen && M > 0u && M <= PLANE_MAX_READ && sink.y == M && (uint64_t)entry + M <= N)
        {
            constexpr int RB = (RBITS == 2 || RBITS == 4) ? RBITS : 4;
            uint64_t rlo[3], rhi[3], rn[3], tlo[4], thi[4];
            ReadWords<RB> rw; TextWords13 tw;
            load_read_words<RB>( b.reads, first, M, rw );
            load_text_words13( b.text, tb, N < 192u ? N : 192u, tw );
            read_planes192<RB>( rw, first, M, rev, comp, rlo, rhi, rn );
            text_planes208( tw, tb, tlo, thi );
            uint32_t cnt = 0;
            #pragma unroll
            for (int w = 0; w < 3; ++w)
            {
                const int32_t left = (int32_t)M - 64 * w;
                const uint64_t mask = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
                const uint64_t lo = entry ? ((tlo[w] >> entry) | (tlo[w + 1] << (64u - entry))) : tlo[w];
                const uint64_t hi = entry ? ((thi[w] >> entry) | (thi[w + 1] << (64u - entry))) : thi[w];
                cnt += (uint32_t)__popcll( ((rlo[w] ^ lo) | (rhi[w] ^ hi) | rn[w]) & mask );
            }
            Q = sc.match * (int32_t)(M - cnt) + s_mm[0] * (int32_t)cnt;          // s_mm[q] is the same for every q here
            k = M;
            counted = true;
        }
    }
    for (int32_t row = (int32_t)sink.y - 1; row >= 0 && !found && !counted; --row)
    {
        const uint32_t pidx = rev ? first + M - 1u - (uint32_t)row : first + (uint32_t)row;
        uint32_t q = prd.get( pidx );
        if (comp && q < 4u) q = 3u - q;
        const uint32_t qq = b.quals ? b.quals[pidx] : 0u;
        const uint32_t ti = (uint32_t)row + entry;
        const uint32_t g  = ti < N ? trd.get( tb + ti ) : 255u;
        const bool eq = (entry == (uint32_t)(BAND - 1) || !PACKED) ? (g == q) : (q < 4u && (g & 3u) == q);
        Q += eq ? sc.match : s_mm[qq < 63u ? qq : 63u];
        ++k;
        if (TYPE == NVBIO_LOCAL && Q == best) found = true;
    }
    if (TYPE != NVBIO_LOCAL) found = (H0 + Q == best);
    if (!found)
    {
        // Narrow-band route (band 31, end-to-end, match bonus 0; gap_open_min >= gap_ext_min > 0 are the cheapest open / extension
        // penalties): every step of a path scores <= 0, so a path that reaches the known optimum S* holds gaps of at most
        // G = (|S*| - open) / ext + 1 symbols in all (none if |S*| < open) and stays within G diagonals of the column it ends in.  With
        // G <= 7 all of them fit a band of 15 around the sink's diagonal: the DP over that band alone gives every cell ON such a path
        // its exact value (its best predecessor is on one too) and can only lower the alternatives a direction rule compares it with,
        // never one that ties -- a tie would be another optimal path, inside the band as well -- so the directions along the traced
        // path, hence the CIGAR, are those of the full band at half the cells.  (N >= M + 30: no sentinel column is involved.)
        uint8_t code = 1;
        if (BAND == 31 && TYPE == NVBIO_SEMI_GLOBAL && band_off && gap_ext_min > 0 && sink.y == M && N >= M + 30u && best <= 0)
        {
            const int32_t a = -best;
            const int32_t G = a < gap_open_min ? 0 : (a - gap_open_min) / gap_ext_min + 1;
            if (G <= 3)                                                    // a band of 7 (one short indel and at most one mismatch)
            {
                code = 3;
                band_off[job] = (uint8_t)(entry > 3u ? (entry - 3u < 24u ? entry - 3u : 24u) : 0u);
            }
            else if (G <= 7)
            {
                code = 2;
                band_off[job] = (uint8_t)(entry > 7u ? (entry - 7u < 16u ? entry - 7u : 16u) : 0u);
            }
        }
        need_dp[job] = code;
        return;
    }

    uint16_t* cig = cigars + (size_t)job * cigar_stride;
    uint32_t  clen = 0;
    auto emit = [&](const uint32_t type, const uint32_t len) {
        if (clen < cigar_stride) cig[clen] = (uint16_t)(type | (len << 2));
        ++clen;
    };
    if (M - sink.y) emit( 3u, M - sink.y );
    if (k)          emit( D_SUB, k );
    if (sink.y - k) emit( 3u, sink.y - k );
    sources[job]    = make_uint2( sink.x - k, sink.y - k );
    cigar_lens[job] = clen;
}

template <int BAND, int TYPE>
nvbio_status launch_ungapped(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, const int32_t* scores, const uint2* sinks,
                             uint2* sources, uint16_t* cigars, uint32_t stride, uint32_t* lens, uint8_t* need_dp, uint8_t* band_off, int32_t go_min, int32_t ge_min,
                             hipStream_t s)
{
    const dim3 grid( (b.n + 255u) / 256u ), block( 256 );
#define NVB_GO(RB, TB) hipLaunchKernelGGL( (ungapped_traceback_kernel<BAND,TYPE,RB,TB>), grid, block, 0, s, b, sc, scores, sinks, sources, cigars, stride, lens, need_dp, band_off, go_min, ge_min )
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

template <int BAND>
nvbio_status launch_ungapped_type(int type, const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, const int32_t* scores,
                                  const uint2* sinks, uint2* sources, uint16_t* cigars, uint32_t stride, uint32_t* lens, uint8_t* need_dp,
                                  uint8_t* band_off, int32_t go_min, int32_t ge_min, hipStream_t s)
{
    switch (type)
    {
    case NVBIO_GLOBAL:      return launch_ungapped<BAND,NVBIO_GLOBAL>     ( b, sc, rbits, tbits, scores, sinks, sources, cigars, stride, lens, need_dp, band_off, go_min, ge_min, s );
    case NVBIO_LOCAL:       return launch_ungapped<BAND,NVBIO_LOCAL>      ( b, sc, rbits, tbits, scores, sinks, sources, cigars, stride, lens, need_dp, band_off, go_min, ge_min, s );
    case NVBIO_SEMI_GLOBAL: return launch_ungapped<BAND,NVBIO_SEMI_GLOBAL>( b, sc, rbits, tbits, scores, sinks, sources, cigars, stride, lens, need_dp, band_off, go_min, ge_min, s );
    }
    set_error( "invalid alignment type %d", type );
    return NVBIO_ERR_INVALID;
}

template <int BAND, int TYPE>
nvbio_status launch_bits(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t job_begin, uint32_t jobs,
                         const uint32_t* job_list, const uint32_t* job_count, uint32_t* dirs, int32_t* scores, uint2* sources, uint2* sinks, uint16_t* cigars, uint32_t stride,
                         uint32_t* lens, hipStream_t s, const uint8_t* band_off = nullptr, const uint32_t full_ties = 0u)
{
    const dim3 grid( (jobs + 127u) / 128u ), block( 128 );
#define NVB_GO(RB, TB) hipLaunchKernelGGL( (banded_gotoh_traceback_kernel<BAND,TYPE,RB,TB>), grid, block, 0, s, b, sc, job_begin, jobs, job_list, job_count, dirs, scores, sources, sinks, cigars, stride, lens, band_off, full_ties )
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

template <int BAND>
nvbio_status launch_type(int type, const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t job_begin, uint32_t jobs,
                         const uint32_t* job_list, const uint32_t* job_count, uint32_t* dirs, int32_t* scores, uint2* sources, uint2* sinks, uint16_t* cigars, uint32_t stride,
                         uint32_t* lens, hipStream_t s, const uint8_t* band_off = nullptr, const uint32_t full_ties = 0u)
{
    switch (type)
    {
    case NVBIO_GLOBAL:      return launch_bits<BAND,NVBIO_GLOBAL>     ( b, sc, rbits, tbits, job_begin, jobs, job_list, job_count, dirs, scores, sources, sinks, cigars, stride, lens, s, band_off, full_ties );
    case NVBIO_LOCAL:       return launch_bits<BAND,NVBIO_LOCAL>      ( b, sc, rbits, tbits, job_begin, jobs, job_list, job_count, dirs, scores, sources, sinks, cigars, stride, lens, s, band_off, full_ties );
    case NVBIO_SEMI_GLOBAL: return launch_bits<BAND,NVBIO_SEMI_GLOBAL>( b, sc, rbits, tbits, job_begin, jobs, job_list, job_count, dirs, scores, sources, sinks, cigars, stride, lens, s, band_off, full_ties );
    }
    set_error( "invalid alignment type %d", type );
    return NVBIO_ERR_INVALID;
}

inline uint64_t row_bytes(const uint32_t band) { return (uint64_t)((band + 7u) / 8u) * sizeof(uint32_t); }
template <int CODE> struct IsCode { __host__ __device__ __forceinline__ uint8_t operator()(const uint8_t v) const { return v == (uint8_t)CODE ? 1u : 0u; } };

} // anonymous namespace

// The band-15 end-to-end traceback over a job list with the FULL matrix's tie rule: for the full-matrix traceback's jobs whose optimal
// paths are known to stay within 7 diagonals of their sink (gotoh_full_traceback.hip hands over a batch whose windows are those bands).
// `max_jobs` bounds the list's length, which stays on the device; the scratch is used in as many launches as it takes.
nvbio_status banded15_full_ties_traceback(const BatchDev& b, const SchemeDev& sc, const uint32_t rbits, const uint32_t tbits, const uint32_t max_jobs,
                                          const uint32_t* job_list, const uint32_t* job_count, uint32_t* dirs, const uint64_t dirs_bytes,
                                          int32_t* scores, uint2* sources, uint2* sinks, uint16_t* cigars, const uint32_t stride, uint32_t* lens, hipStream_t s)
{
    const uint64_t per_job = (uint64_t)b.max_read_len * row_bytes( 15 );
    const uint64_t cap = dirs_bytes / (per_job ? per_job : 1u);
    if (cap < 64u) { set_error( "full traceback: scratch too small for the band-15 route" ); return NVBIO_ERR_INVALID; }
    nvbio_status st = NVBIO_OK;
    for (uint64_t begin = 0; begin < max_jobs && st == NVBIO_OK; begin += cap)
    {
        const uint32_t jobs = (uint32_t)((max_jobs - begin) < cap ? (max_jobs - begin) : cap);
        st = launch_type<15>( NVBIO_SEMI_GLOBAL, b, sc, rbits, tbits, (uint32_t)begin, jobs, job_list, job_count, dirs, scores, sources, sinks, cigars, stride, lens, s,
                              nullptr, 1u );
    }
    return st;
}
} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" nvbio_status nvbio_banded_gotoh_traceback_temp_bytes(const nvbio_alignment_batch* batch, uint32_t band, uint64_t* bytes)
{
    NVB_REQUIRE( batch && bytes, "batch/bytes is NULL" );
    NVB_REQUIRE( band == 3 || band == 7 || band == 15 || band == 31, "band must be 3, 7, 15 or 31" );
    NVB_REQUIRE( batch->n == 0 || batch->max_read_len > 0, "batch.max_read_len must bound the pattern lengths" );
    *bytes = (uint64_t)batch->n * batch->max_read_len * row_bytes( band );
    return NVBIO_OK;
}

// the banded traceback of both aligner families: `gotoh` or `sw` is given (the other NULL)
static nvbio_status banded_traceback_impl(int device, uint32_t band, nvbio_alignment_type type,
                                          const nvbio_gotoh_scheme* gotoh, const nvbio_sw_scheme* sw, const nvbio_alignment_batch* batch,
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
    NVB_REQUIRE( band == 3 || band == 7 || band == 15 || band == 31, "band must be 3, 7, 15 or 31" );
    NVB_REQUIRE( scores_dev && sources_dev && sinks_dev && cigar_lens_dev, "NULL output pointer" );
    NVB_REQUIRE( cigars_dev != nullptr || cigar_stride == 0, "cigars_dev is NULL" );
    NVB_REQUIRE( b.max_read_len > 0, "batch.max_read_len must bound the pattern lengths (it sizes the direction-vector scratch)" );

    // the reference re-derives the direction vectors from int16 checkpoints (clamped at -32736,
    // gotoh_banded_inl.h:216-222); the single pass here equals that iff no score can leave that range
    {
        int64_t step = scheme->match < 0 ? -(int64_t)scheme->match : scheme->match;
        const int64_t c[] = { scheme->mm_min, scheme->mm_max, -(int64_t)scheme->pat_gap_open, -(int64_t)scheme->pat_gap_ext,
                              -(int64_t)scheme->txt_gap_open, -(int64_t)scheme->txt_gap_ext };
        for (int64_t v : c) { if (v < 0) v = -v; if (v > step) step = v; }
        if (((int64_t)b.max_read_len + band + 1) * step > 30000)
        {
            set_error( "banded traceback: scores of %u-symbol reads under this scheme can overflow the reference's int16 checkpoints", b.max_read_len );
            return NVBIO_ERR_UNSUPPORTED;
        }
    }

    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    SchemeDev sc = sw ? scheme_dev( sw, false ) : scheme_dev( gotoh );

    // ---- 1. scoring pass (the packed 16-bit kernel when the scheme allows) + 2. the ungapped shortcut ----------
    // (the diagonal shortcut of a LOCAL job rests on the walk stopping at the first SINK cell: not for the Smith-Waterman aligner)
    const bool shortcut = !(b.algo & NVBIO_ALN_NO_UNGAPPED_TRACEBACK) && !(sw && type == NVBIO_LOCAL);
    uint8_t*  need_dp   = nullptr;      // [n] flags: 0 settled, 1 the DP, 2 / 3 the DP over a band of 15 / 7 (band 31 only)
    uint32_t* job_list  = nullptr;      // [n] compacted job ids
    uint32_t* job_count = nullptr;      // [1]
    uint8_t*  band_off  = nullptr;      // [n] narrow-band route: first column of the job's band of 15 / 7
    uint32_t* job_list2 = nullptr;      // [n], [1]: the jobs of the band-15 route
    uint32_t* job_count2 = nullptr;
    uint32_t* job_list3 = nullptr;      // ... and of the band-7 route
    uint32_t* job_count3 = nullptr;
    // the narrow-band route applies to nvBowtie's end-to-end mode (see ungapped_traceback_kernel)
    const int32_t go_min = -(sc.pat_go > sc.txt_go ? sc.pat_go : sc.txt_go), ge_min = -(sc.pat_ge > sc.txt_ge ? sc.pat_ge : sc.txt_ge);
    const bool narrow = band == 31 && type == NVBIO_SEMI_GLOBAL && sc.match == 0 && sc.mm_min >= 0 && sc.mm_max >= 0 && plain_gotoh( sc ) &&
                        ge_min > 0 && go_min >= ge_min && !(b.algo & NVBIO_ALN_NO_NARROW_TRACEBACK);
    void*     sel_temp  = nullptr;
    void*     aux       = nullptr;
    if (shortcut)
    {
        if (!(flags & NVBIO_TRACEBACK_SINKS_GIVEN))
            NVB_CHECK( sw ? nvbio_banded_sw_score( device, band, type, sw, batch, scores_dev, sinks_dev, stream )
                          : nvbio_banded_gotoh_score( device, band, type, gotoh, batch, scores_dev, sinks_dev, stream ) );
        size_t sel_bytes = 0;
        hipcub::CountingInputIterator<uint32_t> ids( 0u );
        NVB_HIP( hipcub::DeviceSelect::Flagged( nullptr, sel_bytes, ids, (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)b.n, s ) );
        const uint64_t flags_bytes = ((uint64_t)b.n + 255u) & ~255ull;
        const uint64_t list_bytes  = ((uint64_t)b.n * 4u + 255u) & ~255ull;
        if (scratch_alloc( &aux, 2u * (flags_bytes + list_bytes + 256u) + list_bytes + 256u + sel_bytes, s ) != hipSuccess)
        {
            (void)hipGetLastError();
            set_error( "banded traceback: out of device memory for the job list" );
            return NVBIO_ERR_NOMEM;
        }
        need_dp   = (uint8_t*)aux;
        job_list  = (uint32_t*)((uint8_t*)aux + flags_bytes);
        job_count = (uint32_t*)((uint8_t*)aux + flags_bytes + list_bytes);
        band_off   = (uint8_t*)aux + flags_bytes + list_bytes + 256u;
        job_list2  = (uint32_t*)(band_off + flags_bytes);
        job_count2 = (uint32_t*)(band_off + flags_bytes + list_bytes);
        job_list3  = (uint32_t*)((uint8_t*)aux + 2u * (flags_bytes + list_bytes + 256u));
        job_count3 = (uint32_t*)((uint8_t*)job_list3 + list_bytes);
        sel_temp   = (uint8_t*)aux + 2u * (flags_bytes + list_bytes + 256u) + list_bytes + 256u;
        nvbio_status st1;
#define NVB_BAND(B) st1 = launch_ungapped_type<B>( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (const uint2*)sinks_dev, \
                                                   (uint2*)sources_dev, cigars_dev, cigar_stride, cigar_lens_dev, need_dp,             \
                                                   narrow ? band_off : nullptr, go_min, ge_min, s )
        switch (band)
        {
        case 3:  NVB_BAND(3);  break;
        case 7:  NVB_BAND(7);  break;
        case 15: NVB_BAND(15); break;
        default: NVB_BAND(31); break;
        }
#undef NVB_BAND
        if (st1 != NVBIO_OK) { scratch_free( aux, s ); return st1; }
        // ---- 3. the jobs that do need the DP, compacted (their number stays on the device) ----
        hipcub::TransformInputIterator<uint8_t, IsCode<1>, const uint8_t*> is_full( need_dp, IsCode<1>() );
        hipcub::TransformInputIterator<uint8_t, IsCode<2>, const uint8_t*> is_narrow( need_dp, IsCode<2>() );
        hipError_t e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, is_full, job_list, job_count, (int)b.n, s );
        if (e == hipSuccess && narrow) e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, is_narrow, job_list2, job_count2, (int)b.n, s );
        hipcub::TransformInputIterator<uint8_t, IsCode<3>, const uint8_t*> is_narrow7( need_dp, IsCode<3>() );
        if (e == hipSuccess && narrow) e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, is_narrow7, job_list3, job_count3, (int)b.n, s );
        if (e != hipSuccess) { scratch_free( aux, s ); set_error( "DeviceSelect failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    }

    // ---- 4. the DP with direction vectors + walk back, over the job list (or every job) ----
    const uint64_t per_job = (uint64_t)b.max_read_len * row_bytes( band );
    void*     owned = nullptr;
    uint32_t* dirs  = (uint32_t*)temp_dev;
    uint64_t  cap_jobs;
    if (dirs)
    {
        cap_jobs = temp_bytes / per_job;
        if (!(cap_jobs >= 64 || cap_jobs >= b.n))
        {
            if (aux) scratch_free( aux, s );
            set_error( "invalid argument: temp_bytes too small (see nvbio_banded_gotoh_traceback_temp_bytes)" );
            return NVBIO_ERR_INVALID;
        }
    }
    else
    {
        // with the shortcut only the gapped alignments reach the DP (a minority of a read batch): size the
        // scratch for a quarter of the jobs per launch -- launches over an exhausted job list exit at once
        cap_jobs = shortcut ? ((uint64_t)b.n + 3u) / 4u : b.n;
        if (cap_jobs < 65536u) cap_jobs = b.n < 65536u ? b.n : 65536u;
        const uint64_t budget = 16ull << 30;                     // at most 16 GiB of scratch per launch
        if (cap_jobs * per_job > budget) cap_jobs = budget / per_job;
        if (cap_jobs < 64) cap_jobs = 64;
        if (scratch_alloc( &owned, cap_jobs * per_job, s ) != hipSuccess)
        {
            (void)hipGetLastError();
            if (aux) scratch_free( aux, s );
            set_error( "banded traceback: out of device memory for %llu direction matrices", (unsigned long long)cap_jobs );
            return NVBIO_ERR_NOMEM;
        }
        dirs = (uint32_t*)owned;
    }
    nvbio_status st = NVBIO_OK;
    for (uint64_t begin = 0; begin < b.n && st == NVBIO_OK; begin += cap_jobs)
    {
        const uint32_t jobs = (uint32_t)((b.n - begin) < cap_jobs ? (b.n - begin) : cap_jobs);
#define NVB_BAND(B) st = launch_type<B>( type, b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, job_list, job_count, dirs, scores_dev, \
                                         (uint2*)sources_dev, (uint2*)sinks_dev, cigars_dev, cigar_stride, cigar_lens_dev, s )
        switch (band)
        {
        case 3:  NVB_BAND(3);  break;
        case 7:  NVB_BAND(7);  break;
        case 15: NVB_BAND(15); break;
        default: NVB_BAND(31); break;
        }
#undef NVB_BAND
    }
    if (narrow && shortcut)
    {
        // the jobs of the narrow-band route: the band-15 kernel over their list, twice as many per launch in the same scratch
        const uint64_t cap2 = cap_jobs * per_job / ((uint64_t)b.max_read_len * row_bytes( 15 ));
        for (uint64_t begin = 0; begin < b.n && st == NVBIO_OK; begin += cap2)
        {
            const uint32_t jobs = (uint32_t)((b.n - begin) < cap2 ? (b.n - begin) : cap2);
            st = launch_type<15>( type, b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, job_list2, job_count2, dirs, scores_dev,
                                  (uint2*)sources_dev, (uint2*)sinks_dev, cigars_dev, cigar_stride, cigar_lens_dev, s, band_off );
        }
        const uint64_t cap3 = cap_jobs * per_job / ((uint64_t)b.max_read_len * row_bytes( 7 ));
        for (uint64_t begin = 0; begin < b.n && st == NVBIO_OK; begin += cap3)
        {
            const uint32_t jobs = (uint32_t)((b.n - begin) < cap3 ? (b.n - begin) : cap3);
            st = launch_type<7>( type, b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, job_list3, job_count3, dirs, scores_dev,
                                 (uint2*)sources_dev, (uint2*)sinks_dev, cigars_dev, cigar_stride, cigar_lens_dev, s, band_off );
        }
    }
    if (owned) scratch_free( owned, s );
    if (aux)   scratch_free( aux, s );
    return st;
}

extern "C" nvbio_status nvbio_banded_gotoh_traceback(int device, uint32_t band, nvbio_alignment_type type,
                                                     const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                                     int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                                     uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                                     uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    return banded_traceback_impl( device, band, type, scheme, nullptr, batch, scores_dev, sources_dev, sinks_dev, cigars_dev, cigar_stride,
                                  cigar_lens_dev, flags, temp_dev, temp_bytes, stream );
}

extern "C" nvbio_status nvbio_banded_sw_traceback(int device, uint32_t band, nvbio_alignment_type type,
                                                  const nvbio_sw_scheme* scheme, const nvbio_alignment_batch* batch,
                                                  int32_t* scores_dev, nvbio_uint2* sources_dev, nvbio_uint2* sinks_dev,
                                                  uint16_t* cigars_dev, uint32_t cigar_stride, uint32_t* cigar_lens_dev,
                                                  uint32_t flags, void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    return banded_traceback_impl( device, band, type, nullptr, scheme, batch, scores_dev, sources_dev, sinks_dev, cigars_dev, cigar_stride,
                                  cigar_lens_dev, flags, temp_dev, temp_bytes, stream );
}
