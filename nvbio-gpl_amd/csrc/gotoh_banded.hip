// gotoh_banded.hip -- batched banded Gotoh (affine-gap Smith-Waterman) scoring for gfx950.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   gotoh_alignment_score_dispatch<BAND,TYPE>::run   nvbio/alignment/gotoh/gotoh_banded_inl.h:397-646
//   row-zero initialisation                          gotoh_banded_inl.h:37-68
//   Reference_cache<BAND> (band-31 2-bit text cache) nvbio/alignment/alignment_base_inl.h:66-90
//   BestSink<int32> (last maximum wins)              nvbio/alignment/sink_inl.h:31-49
//   batched driver (one job per work item)           nvbio/alignment/batched_banded_inl.h:34-157
//   nvBowtie read / window loading                   nvBowtie/bowtie2/cuda/alignment_utils.h:277-302, nvbio/io/utils.h:150-168
//
// MI355X design (integer VALU-bound; MFMA does not apply): one lane owns one alignment and keeps
// the whole band -- H[BAND], F[BAND] -- in VGPRs with every band index a compile-time constant, so
// a row is a straight line of v_add / v_max3 with no LDS traffic, no cross-lane exchange and no
// divergence between lanes of equal read length.  The band-31 text window lives in ONE 64-bit
// register pair (30 x 2 bits) and the per-row match flags for all 30 cached columns come from
// three 64-bit logic ops on it; the LOCAL sink is tracked with one packed (score<<5 | column)
// max per cell and one compare per row, which reproduces BestSink's row-major "last maximum
// wins" rule exactly.  Reads and windows are consumed straight from the packed HBM streams
// (one dword per 8 read symbols / 16 text symbols).
#include "gotoh_common.h"
#include "bitplanes.h"
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace nvbio_amd {

// BEST2: the cells are reported one by one, in the reference's order, into a Best2Sink<int32>( distinct_dist )
// (sink.h:96-116, sink_inl.h:55-83): best in scores / sinks, the second -- more than distinct_dist text positions away --
// in scores2 / sinks2.  A new best does not demote the old one, so the result depends on the order of the reports.
// STAGED: the result of the staged scheduler (BatchedBandedAlignmentScore<BAND,stream,DeviceStagedThreadScheduler>,
// batched_banded_inl.h:165-236): the windowed banded_alignment_score (gotoh_banded_inl.h:703-727) over 32-row windows
// (batched_stream.h:119,145-180).  The reference re-queues a job after every window to re-compact divergent waves; what that changes
// in the RESULT is (a) the exit at a window's end when no band cell can reach min_score any more (:610-622; LOCAL cells reported so far
// stand, GLOBAL / SEMI_GLOBAL report nothing) and (b) the band passing through int16 checkpoints clamped at int16_min + 32
// (:139-147,166-176).  Here a lane keeps its job for all rows -- lanes that exit idle until the wave ends -- and applies (a) and (b).
template <int BAND, int TYPE, int RBITS, int TBITS, bool BEST2 = false, bool STAGED = false>
__global__ void __launch_bounds__(128)
banded_gotoh_kernel(const BatchDev b, const SchemeDev sc, int32_t* __restrict__ scores, uint2* __restrict__ sinks,
                    const uint32_t distinct_dist = 0, int32_t* __restrict__ scores2 = nullptr, uint2* __restrict__ sinks2 = nullptr,
                    const int32_t* __restrict__ min_scores = nullptr, const int32_t min_score_all = 0)
{
    // mismatch score per quality value, computed once per workgroup
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
    // a pattern longer than the batch's declared max_read_len is rejected -- nothing reported, as for a text shorter than the
    // pattern -- in every kernel alike: the packed kernels rely on that bound for their 16-bit scores
    const uint32_t N     = (b.max_read_len && M > b.max_read_len) ? 0u : b.win_end[job] - tb;

    int32_t  best   = NVBIO_SCORE_MIN;
    uint32_t best_x = 0xFFFFFFFFu, best_y = 0xFFFFFFFFu;
    int32_t  sec    = NVBIO_SCORE_MIN;                           // BEST2
    uint32_t sec_x  = 0xFFFFFFFFu, sec_y = 0xFFFFFFFFu;
    auto report2 = [&](const int32_t h, const uint32_t x, const uint32_t y) {
        if (best <= h) { best = h; best_x = x; best_y = y; }
        else if (sec <= h && ((uint32_t)(x + distinct_dist) < best_x || x > (uint32_t)(best_x + distinct_dist))) { sec = h; sec_x = x; sec_y = y; }
    };

    if (N < M)                                                   // gotoh_banded_inl.h:422-423: nothing reported
    {
        scores[job] = best; sinks[job] = make_uint2( best_x, best_y );
        if (BEST2) { scores2[job] = sec; sinks2[job] = make_uint2( sec_x, sec_y ); }
        return;
    }

    constexpr bool PACKED = !(BAND == 3 || BAND == 5 || BAND == 7 || BAND == 15);
    static_assert( !PACKED || BAND <= 33, "packed text cache holds at most 32 symbols" );

    SymbolReader<TBITS> trd( b.text );
    SymbolReader<RBITS> prd( b.reads );

    // text cache: columns 0..BAND-2 of the current row
    uint64_t cache_bits = 0;                                     // PACKED: symbol j at bits [2j,2j+1]
    uint32_t cache_raw[PACKED ? 1 : BAND - 1];                   // !PACKED: whole symbols (255 stays 255)
    #pragma unroll
    for (int j = 0; j < BAND - 1; ++j)
    {
        const uint32_t g = ((uint32_t)j < N) ? trd.get( tb + j ) : 255u;
        if (PACKED) cache_bits |= (uint64_t)(g & 3u) << (2 * j);
        else        cache_raw[j] = g;
    }

    const int32_t G_o = sc.pat_go, G_e = sc.pat_ge;             // F: the text advances alone
    const int32_t I_o = sc.ins_go, I_e = sc.ins_ge;             // E: the pattern advances alone (= G for the Gotoh aligner)
    const int32_t infimum = -32768 - max2( max2( G_o, G_e ), max2( sc.txt_go, sc.txt_ge ) );
    const int32_t V = sc.match;

    int32_t H[BAND], F[BAND];
    H[0] = 0;
    #pragma unroll
    for (int j = 1; j < BAND; ++j) H[j] = (TYPE == NVBIO_GLOBAL) ? sc.txt_go + (j - 1) * sc.txt_ge : 0;
    #pragma unroll
    for (int j = 0; j < BAND; ++j) F[j] = infimum;

    bool stopped = false;                                        // STAGED: a window returned false
    const int32_t min_score = STAGED ? (min_scores ? min_scores[job] : min_score_all) : 0;
    for (uint32_t i = 0; i < M; ++i)
    {
        if (STAGED && i && (i & 31u) == 0u)
        {
            int32_t mx = H[0];
            #pragma unroll
            for (int j = 1; j < BAND; ++j) mx = max2( mx, H[j] );
            const int32_t thr = (int32_t)((uint32_t)min_score + (M - i) * (uint32_t)V);
            if (mx < thr) { stopped = true; break; }
            #pragma unroll
            for (int j = 0; j < BAND; ++j)
            {
                H[j] = (int32_t)(int16_t)max2( H[j], -32768 + 32 );
                F[j] = (int32_t)(int16_t)max2( F[j], -32768 + 32 );
            }
        }
        const uint32_t pidx = rev ? first + M - 1u - i : first + i;
        uint32_t q = prd.get( pidx );
        if (comp && q < 4u) q = 3u - q;
        const uint32_t qq = b.quals ? b.quals[pidx] : 0u;
        const int32_t  S  = s_mm[qq < 63u ? qq : 63u];          // qualities >= 40 all map to mm_max

        // new text symbol entering column BAND-1 (gotoh_banded_inl.h:569-570)
        const uint32_t g_new = (i + (uint32_t)(BAND - 1) < N) ? trd.get( tb + i + (BAND - 1) ) : 255u;

        // per-column match flags of the cached columns
        uint64_t eq_bits = 0;
        if (PACKED)
        {
            if (q < 4u)
            {
                const uint64_t t = cache_bits ^ ((uint64_t)q * 0x5555555555555555ull);
                eq_bits = ~(t | (t >> 1)) & 0x5555555555555555ull;
            }
        }

        int32_t E = 0;
        int32_t row_key = -1;                                    // LOCAL: max over j of (h << 5 | j)
        #pragma unroll
        for (int j = 0; j < BAND; ++j)
        {
            // F from the previous row's column j+1 (:476-479,513-516,575)
            const int32_t f = (j < BAND - 1) ? max2( F[j + 1] + G_e, H[j + 1] + G_o ) : infimum;
            F[j] = f;

            bool eq;
            if (j == BAND - 1)   eq = (g_new == q);
            else if (PACKED)     eq = ((eq_bits >> (2 * j)) & 1ull) != 0;
            else                 eq = (cache_raw[j] == q);
            const int32_t d = H[j] + (eq ? V : S);

            int32_t h;
            if (j == 0)             h = max2( f, d );
            else if (j == BAND - 1) h = max2( E, d );
            else                    h = max3( f, E, d );
            if (TYPE == NVBIO_LOCAL)
            {
                h = max2( h, 0 );
                if (BEST2) report2( h, i + (uint32_t)j + 1u, i + 1u );
                else       row_key = max2( row_key, (h << 5) | j );
            }
            H[j] = h;
            E = (j == 0) ? h + I_o : max2( h + I_o, E + I_e );   // :507,562-565
        }

        // shift the cache by one column and append the new symbol (:532,570)
        if (PACKED) cache_bits = (cache_bits >> 2) | ((uint64_t)(g_new & 3u) << (2 * (BAND - 2)));
        else
        {
            #pragma unroll
            for (int j = 0; j < BAND - 2; ++j) cache_raw[j] = cache_raw[j + 1];
            cache_raw[BAND - 2] = g_new;
        }

        if (TYPE == NVBIO_LOCAL && !BEST2)
        {
            // cells are reported row-major with j ascending and BestSink keeps the LAST maximum
            const int32_t h = row_key >> 5;
            if (h >= best) { best = h; best_x = i + (uint32_t)(row_key & 31) + 1u; best_y = i + 1u; }
        }
    }

    if (STAGED && stopped) { /* nothing more is reported */ }
    else if (TYPE == NVBIO_GLOBAL)                               // :629-630
    {
        if (BEST2) report2( H[BAND - 1], M + BAND - 1, M );
        else if (best <= H[BAND - 1]) { best = H[BAND - 1]; best_x = M + BAND - 1; best_y = M; }
    }
    else if (TYPE == NVBIO_SEMI_GLOBAL)                          // :631-643
    {
        const uint32_t mb = M + (uint32_t)(BAND - 1);
        const uint32_t m  = (mb < N ? mb : N) - (M - 1u);
        #pragma unroll
        for (int j = 0; j < BAND; ++j)
            if (j == 0 || (uint32_t)j < m)
            {
                if (BEST2) report2( H[j], M + j, M );
                else if (best <= H[j]) { best = H[j]; best_x = M + j; best_y = M; }
            }
    }
    scores[job] = best;
    sinks[job]  = make_uint2( best_x, best_y );
    if (BEST2) { scores2[job] = sec; sinks2[job] = make_uint2( sec_x, sec_y ); }
}

// ---------------------------------------------------------------------------------------------
// 16-bit packed variant for the production case (band 31, any alignment type): one lane owns TWO
// alignments, one in each half of every register, so H[31]/F[31] of both take the registers one
// alignment took before and every add / max is a v_pk_*_i16 doing two cells.  Exactness conditions,
// checked on the host (packed_ok; else the int32 kernel runs): LOCAL scores fit 10 bits
// (match * max_read_len <= 1000, so that (score << 5 | column) fits an int16), GLOBAL / SEMI_GLOBAL
// scores stay within +-8000 ((max_read_len + 32) * largest step), and penalties are < 4096, so that
// the -16384 stand-in for the reference's infimum can never win a max against a real score nor
// wrap.  The row-0 / column-30 infimum cells behave exactly as in the int32 kernel.  With a job
// list (the jobs the ungapped shortcut could not settle) lane p works on list entries 2p, 2p+1.
// ---------------------------------------------------------------------------------------------
typedef short    v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s pk(const int a, const int b) { v2s r; r.x = (short)a; r.y = (short)b; return r; }
__device__ __forceinline__ v2s pk_max(const v2s a, const v2s b) { return __builtin_elementwise_max( a, b ); }
__device__ __forceinline__ v2s pk_from_bits(const uint32_t u) { return __builtin_bit_cast( v2s, u ); }
// the same pair of lanes as two binary16 numbers: every integer of magnitude <= 2048 is exact there, and gfx950 has a packed three-operand
// maximum (v_pk_maximum3_f16) where the integer pipe needs two v_pk_max_i16
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2h pk_max(const v2h a, const v2h b) { return __builtin_elementwise_maximum( a, b ); }
__device__ __forceinline__ v2s pk_max3(const v2s a, const v2s b, const v2s c) { return pk_max( pk_max( a, b ), c ); }
__device__ __forceinline__ v2h pk_max3(const v2h a, const v2h b, const v2h c) { return __builtin_elementwise_maximum( __builtin_elementwise_maximum( a, b ), c ); }
template <typename V> __device__ __forceinline__ V pk_of(const int a, const int b);
template <> __device__ __forceinline__ v2s pk_of<v2s>(const int a, const int b) { return pk( a, b ); }
template <> __device__ __forceinline__ v2h pk_of<v2h>(const int a, const int b) { v2h r; r.x = (_Float16)a; r.y = (_Float16)b; return r; }
template <bool FP> struct PkLanes { typedef v2s type; };
template <> struct PkLanes<true> { typedef v2h type; };

// 32 bits of a big-endian packed stream starting at absolute bit position `bit`, assembled from the
// two words (a = word bit>>5, b = the next one) that were loaded one chunk earlier
__device__ __forceinline__ uint32_t funnel32(const uint32_t a, const uint32_t b, const uint32_t bit)
{
    const uint32_t sh = bit & 31u;
    return sh ? ((a << sh) | (b >> (32u - sh))) : a;
}
__device__ __forceinline__ uint32_t clamp_u32(const int64_t v, const uint32_t lo, const uint32_t hi)
{
    return v < (int64_t)lo ? lo : (v > (int64_t)hi ? hi : (uint32_t)v);
}

// bit k of the result = bit 2k of x
__device__ __forceinline__ uint32_t even_bits64(uint64_t x)
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16));
    return (uint32_t)x;
}

// Rows are processed in chunks of 8.  At the top of a chunk every lane assembles the chunk's 8 read
// symbols and 8 incoming text symbols (per alignment) from words loaded one chunk EARLIER, then
// issues the loads for the next chunk: one wave-uniform wait point per 8 rows with an 8-row
// (~5,000 instruction) head start, and no per-row branches.  Reads are packed RBITS (2 or 4) per
// symbol, the text 2 bits per symbol.
// MINW: the occupancy the register allocator must reach.  3 waves per SIMD leave 168 VGPRs: about 90 registers of prologue / epilogue
// state spill (the scratch accesses sit outside the row loop).  2 waves leave 256: nothing spills.  Which is faster is measured, not
// assumed: 3 waves won while the row loop still had idle issue slots to fill; with the hand-ordered loop 2 waves are 4 % ahead and
// are the default, NVBIO_ALN_PK_THREE_WAVES runs the other build.
// RAGGED (GLOBAL / SEMI_GLOBAL; NVBIO_ALN_RAGGED_READS): a lane whose two alignments differ in length still takes ONE pass: the shorter one
// starts late, at row pad = (rows of the longer) - (its own rows), so that both end in the same row and report from the band there.  Its
// streams are read at row - pad (before its start it computes on whatever they hold), and in the row it starts its half of the band and of
// the text cache is set to the initial state -- one extra wave-level branch per row while some lane of the wave is still waiting to start.
// FP (M0, not LOCAL; the host checks that every score stays inside +-2040): the two lanes of a register are binary16 numbers.  Integers of that
// size, their sums and maxima are exact there, the mismatch flag becomes 2^-7 (bit 13 of each half) against a penalty scaled by 128 in one
// v_pk_fma_f16, and h = max( f, d, E ) is ONE v_pk_maximum3_f16: 9 operations per cell instead of 10.  (The -16384 stand-in for the reference's
// infimum stays where it is under + GE: 16 is the spacing of binary16 there.)
template <int TYPE, int RBITS, int MINW = 3, bool M0 = false, bool RAGGED = false, bool FP = false>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(MINW, 3)))
banded_gotoh_band31_pk_kernel(const BatchDev b, const SchemeDev sc, int32_t* __restrict__ scores, uint2* __restrict__ sinks,
                              const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count)
{
    constexpr int BAND = 31;
    constexpr uint32_t RMASK = (1u << RBITS) - 1u;
    static_assert( !FP || (M0 && TYPE != NVBIO_LOCAL), "the binary16 build: match = 0, GLOBAL / SEMI_GLOBAL" );
    typedef typename PkLanes<FP>::type v2;
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    // with a job list (the jobs the ungapped pass could not settle) lane p works on entries 2p and 2p+1 of the list
    const uint32_t n_jobs = job_list ? *job_count : b.n;
    const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (2u * pair >= n_jobs) return;
    const uint32_t* __restrict__ rwords = (const uint32_t*)b.reads;
    const uint32_t* __restrict__ twords = (const uint32_t*)b.text;

    uint32_t first[2], M[2], tb[2], N[2], rows_all[2], out_id[2];
    bool     rev[2], comp[2], valid[2];
    #pragma unroll
    for (int u = 0; u < 2; ++u)
    {
        const uint32_t slot = 2u * pair + u;
        valid[u] = slot < n_jobs;
        const uint32_t ss  = valid[u] ? slot : 2u * pair;
        const uint32_t jj  = job_list ? job_list[ss] : ss;
        out_id[u] = jj;
        const uint32_t rid = b.read_id ? b.read_id[jj] : jj;
        first[u] = b.read_offsets[rid];
        M[u]     = b.read_offsets[rid + 1] - first[u];
        const uint32_t fl = b.flags ? b.flags[jj] : 0u;
        rev[u]  = (fl & NVBIO_READ_REVERSE) != 0;
        comp[u] = (fl & NVBIO_READ_COMPLEMENT) != 0;
        tb[u]   = b.win_begin[jj];
        N[u]    = (b.max_read_len && M[u] > b.max_read_len) ? 0u : b.win_end[jj] - tb[u];     // too long for the declared bound: rejected
        // rows this alignment really computes: none when the text is shorter than the pattern (nothing reported)
        rows_all[u] = (valid[u] && N[u] >= M[u]) ? M[u] : 0u;
    }

    // word ranges a stream may touch (loads are clamped into them; symbols outside are never used)
    uint32_t r_lo[2], r_hi[2], t_lo[2], t_hi[2];
    #pragma unroll
    for (int u = 0; u < 2; ++u)
    {
        r_lo[u] = (uint32_t)(((uint64_t)first[u] * RBITS) >> 5);
        r_hi[u] = (uint32_t)(((uint64_t)(first[u] + (M[u] ? M[u] - 1u : 0u)) * RBITS) >> 5);
        t_lo[u] = tb[u] >> 4;
        t_hi[u] = (tb[u] + (N[u] ? N[u] - 1u : 0u)) >> 4;
    }

    // storage position (symbol index) where the read chunk of rows [r0, r0+8) starts: forward reads start
    // at first+r0 and walk up; reversed reads cover [first+M-1-r0-7, first+M-1-r0] and walk down from its top
    auto read_chunk_start = [&](const int u, const uint32_t r0) -> int64_t {
        return rev[u] ? (int64_t)first[u] + (int64_t)M[u] - 1 - (int64_t)r0 - 7 : (int64_t)first[u] + r0;
    };
    auto read_chunk_start_at = [&](const int u, const int64_t r0) -> int64_t {      // RAGGED: rows before the alignment's start are negative
        return rev[u] ? (int64_t)first[u] + (int64_t)M[u] - 1 - r0 - 7 : (int64_t)first[u] + r0;
    };

    const uint32_t qadj = (uint32_t)((uintptr_t)b.quals & 3u);      // the quality stream need not be 4-byte aligned
    const uint32_t* __restrict__ qwords = (const uint32_t*)((uintptr_t)b.quals - qadj);
    const bool has_quals = (b.quals != nullptr);

    const v2 GO = pk_of<v2>( sc.pat_go, sc.pat_go ), GE = pk_of<v2>( sc.pat_ge, sc.pat_ge );
    const v2 INF = pk_of<v2>( -16384, -16384 ), ZERO = pk_of<v2>( 0, 0 );
    const v2s K32 = pk( 32, 32 );
    const int V = sc.match;
    const int S_noq = s_mm[0];                                       // without qualities every row scores a mismatch like this
    uint32_t cx[2], lim[2];
    #pragma unroll
    for (int u = 0; u < 2; ++u)
    {
        cx[u]  = comp[u] ? 3u : 0u;                                  // complement: A<->T, C<->G
        lim[u] = N[u] > (uint32_t)(BAND - 1) ? N[u] - (uint32_t)(BAND - 1) : 0u;     // row i's column 30 lies past the text end iff i >= lim
    }

    int32_t  best[2]   = { NVBIO_SCORE_MIN, NVBIO_SCORE_MIN };
    uint32_t best_x[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu }, best_y[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu };

    // GLOBAL / SEMI_GLOBAL report from the band after an alignment's LAST row (:624-645).  To keep that out
    // of the row loop, a lane whose two alignments have different lengths runs them one after the other
    // (two passes, one half active each); equal lengths -- the normal case for a read batch -- take one pass.
    // LOCAL reports every cell as it goes and always takes one pass.
    const bool want0 = valid[0] && N[0] >= M[0], want1 = valid[1] && N[1] >= M[1];
    const bool split = (TYPE != NVBIO_LOCAL) && !RAGGED && want0 && want1 && M[0] != M[1];
    // RAGGED: rows the shorter alignment waits before it starts
    const uint32_t rows_max = rows_all[0] > rows_all[1] ? rows_all[0] : rows_all[1];
    uint32_t pad[2] = { 0u, 0u };
    if (RAGGED && TYPE != NVBIO_LOCAL) { pad[0] = rows_max - rows_all[0]; pad[1] = rows_max - rows_all[1]; }

    for (int pass = 0; pass < (split ? 2 : 1); ++pass)
    {
        uint32_t rows_u[2];
        rows_u[0] = (split && pass != 0) ? 0u : rows_all[0];
        rows_u[1] = (split && pass != 1) ? 0u : rows_all[1];
        const uint32_t rows = rows_u[0] > rows_u[1] ? rows_u[0] : rows_u[1];

        // ---- text cache: two bit planes per alignment, c0 / c1 = low / high bits of the symbols.  Inside a row column j sits at bit
        // 30-j: the row's incoming symbol (column 30) is shifted in at bit 0 (one v_alignbit per plane) and every other column
        // thereby moves to the next row's position; between rows the planes hold columns 0..29 at bits 29-j ------------------------
        uint32_t c0[2], c1[2], c0i[2] = { 0, 0 }, c1i[2] = { 0, 0 };
        #pragma unroll
        for (int u = 0; u < 2; ++u)
        {
            const uint32_t w  = tb[u] >> 4;
            const uint32_t w0 = twords[clamp_u32( w,      t_lo[u], t_hi[u] )];
            const uint32_t w1 = twords[clamp_u32( w + 1u, t_lo[u], t_hi[u] )];
            const uint32_t w2 = twords[clamp_u32( w + 2u, t_lo[u], t_hi[u] )];
            const uint32_t bit = (tb[u] & 15u) * 2u;
            const uint64_t hi = ((uint64_t)funnel32( w0, w1, bit ) << 32) | funnel32( w1, w2, bit );
            uint64_t c = hi & ~0xFull;                               // 30 symbols = top 60 bits, column j at bits [62-2j, 63-2j]
            // symbols at or past the text end read as 3 (the 255 sentinel through a 2-bit cache)
            if (N[u] < 30u) c |= (~0ull >> (2u * N[u])) & ~0xFull;
            // even / odd bits of c, gathered: column j at bit 31-j, moved to bit 29-j (the row's incoming symbol is shifted in at bit 0)
            c0[u] = even_bits64( c ) >> 2;
            c1[u] = even_bits64( c >> 1 ) >> 2;
            if (RAGGED) { c0i[u] = c0[u]; c1i[u] = c1[u]; }
        }

        // ---- stream words for chunk 0 (loaded now, consumed at the top of the loop) ----------------------
        uint32_t ra[2], rb[2], ta[2], tbw[2];
        uint32_t qa[2] = { 0, 0 }, qb[2] = { 0, 0 }, qc[2] = { 0, 0 };  // quality bytes: 3 words cover 8 unaligned bytes
        auto issue_loads = [&](const uint32_t r0) {
            if (has_quals)
            {
                #pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const int64_t qw  = ((RAGGED ? read_chunk_start_at( u, (int64_t)r0 - pad[u] ) : read_chunk_start( u, r0 )) + qadj) >> 2;        // byte position / 4
                    const uint32_t lo = (first[u] + qadj) >> 2, hi = (first[u] + qadj + (M[u] ? M[u] - 1u : 0u)) >> 2;
                    qa[u] = qwords[clamp_u32( qw,     lo, hi )];
                    qb[u] = qwords[clamp_u32( qw + 1, lo, hi )];
                    qc[u] = qwords[clamp_u32( qw + 2, lo, hi )];
                }
            }
            #pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                const int64_t  rbit = (RAGGED ? read_chunk_start_at( u, (int64_t)r0 - pad[u] ) : read_chunk_start( u, r0 )) * RBITS;
                const int64_t  rw   = rbit >> 5;                          // arithmetic shift: floor for negatives
                ra[u]  = rwords[clamp_u32( rw,     r_lo[u], r_hi[u] )];
                rb[u]  = rwords[clamp_u32( rw + 1, r_lo[u], r_hi[u] )];
                // text symbol entering column 30 at row r0 (RAGGED: of the alignment's own row r0 - pad, possibly before the window)
                const int64_t  tw   = RAGGED ? (((int64_t)tb[u] + (int64_t)r0 - pad[u] + (BAND - 1)) >> 4)
                                             : (int64_t)(((uint64_t)tb[u] + r0 + (BAND - 1)) >> 4);
                ta[u]  = twords[clamp_u32( tw,     t_lo[u], t_hi[u] )];
                tbw[u] = twords[clamp_u32( tw + 1, t_lo[u], t_hi[u] )];
            }
        };
        issue_loads( 0 );

        v2 H[BAND], Hg[BAND], F[BAND];                               // Hg = H + GO, kept beside H: feeds the next row's F and this row's E
        #pragma unroll
        for (int j = 0; j < BAND; ++j)
        {
            const int h0 = (TYPE == NVBIO_GLOBAL && j > 0) ? sc.txt_go + (j - 1) * sc.txt_ge : 0;      // init_row_zero (:37-68)
            H[j] = pk_of<v2>( h0, h0 ); F[j] = INF; Hg[j] = H[j] + GO;
        }

        for (uint32_t r0 = 0; r0 < rows; r0 += 8u)
        {
            // ---- assemble this chunk from the words loaded a chunk ago, then request the next chunk ----
            uint32_t rchunk[2], tchunk[2], tchunk1[2]; int rsh[2], rstep[2];
            uint64_t qchunk[2] = { 0, 0 }; int qsh[2], qstep[2];
            #pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                if (has_quals)
                {
                    // 8 quality bytes of the chunk, byte k of the chunk at bits [8k, 8k+7] (memory order)
                    const uint32_t bs = ((uint32_t)(((RAGGED ? read_chunk_start_at( u, (int64_t)r0 - pad[u] ) : read_chunk_start( u, r0 )) + qadj) & 3)) * 8u;
                    const uint32_t lo = bs ? ((qa[u] >> bs) | (qb[u] << (32u - bs))) : qa[u];
                    const uint32_t hi = bs ? ((qb[u] >> bs) | (qc[u] << (32u - bs))) : qb[u];
                    qchunk[u] = ((uint64_t)hi << 32) | lo;
                }
                qsh[u] = rev[u] ? 56 : 0; qstep[u] = rev[u] ? -8 : 8;
                const int64_t rbit = (RAGGED ? read_chunk_start_at( u, (int64_t)r0 - pad[u] ) : read_chunk_start( u, r0 )) * RBITS;
                rchunk[u] = funnel32( ra[u], rb[u], (uint32_t)(rbit & 31) );
                rsh[u]    = rev[u] ? (32 - RBITS) - 7 * RBITS : (32 - RBITS);   // row 0 of the chunk: last / first symbol
                rstep[u]  = rev[u] ? RBITS : -RBITS;
                const uint32_t tsym = RAGGED ? (uint32_t)(((int64_t)tb[u] + (int64_t)r0 - pad[u] + (BAND - 1)) & 15)
                                             : (uint32_t)(((uint64_t)tb[u] + r0 + (BAND - 1)) & 15u);
                uint32_t tc = funnel32( ta[u], tbw[u], tsym * 2u );
                // symbols at or past the text end enter the cache as 3 (the 255 sentinel through a 2-bit cache): symbol k of the
                // chunk is text symbol r0 + 30 + k
                const int64_t kk = RAGGED ? (int64_t)N[u] - (int64_t)(BAND - 1) - ((int64_t)r0 - pad[u]) : (int64_t)N[u] - (int64_t)(BAND - 1) - (int64_t)r0;
                if (kk < 16) tc |= (kk <= 0) ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (2u * (uint32_t)kk));
                tchunk[u]  = tc;                                     // top bit = high bit of the next symbol
                tchunk1[u] = tc << 1;                                // top bit = its low bit
            }
            if (r0 + 8u < rows) issue_loads( r0 + 8u );

            const uint32_t r_end = (r0 + 8u < rows) ? 8u : rows - r0;
            for (uint32_t t = 0; t < r_end; ++t)
            {
                const uint32_t i = r0 + t;
                if (RAGGED && TYPE != NVBIO_LOCAL)
                {
                    // an alignment that starts in this row: its half of the band and of the text cache to the initial state
                    const bool s0 = pad[0] != 0u && i == pad[0], s1 = pad[1] != 0u && i == pad[1];
                    if (s0 || s1)
                    {
                        #pragma unroll
                        for (int j = 0; j < BAND; ++j)
                        {
                            const int h0 = (TYPE == NVBIO_GLOBAL && j > 0) ? sc.txt_go + (j - 1) * sc.txt_ge : 0;
                            const v2 hh = pk_of<v2>( h0, h0 ), hhg = pk_of<v2>( h0 + sc.pat_go, h0 + sc.pat_go );
                            if (s0) { H[j].x = hh.x; F[j].x = INF.x; Hg[j].x = hhg.x; }
                            if (s1) { H[j].y = hh.y; F[j].y = INF.y; Hg[j].y = hhg.y; }
                        }
                        if (s0) { c0[0] = c0i[0]; c1[0] = c1i[0]; }
                        if (s1) { c0[1] = c0i[1]; c1[1] = c1i[1]; }
                    }
                }
                // the row's pattern symbols / mismatch scores, the text symbols entering column 30, and the row's mismatch flags:
                // nq(u) bit 30-j = column j of alignment u does NOT match (31 columns from the two planes; an N in the read matches
                // nothing; a column-30 symbol past the text end is the 255 sentinel for this row and a 3 once it is in the cache).
                // (A half that has no rows left -- or never had any -- keeps computing on whatever its streams hold: nothing of it
                // is reported.)
                uint32_t nq[2]; int S[2];
                if (has_quals)
                {
                    #pragma unroll
                    for (int u = 0; u < 2; ++u)
                    {
                        const uint32_t ql = (uint32_t)(qchunk[u] >> qsh[u]) & 0xFFu; qsh[u] += qstep[u];
                        S[u] = s_mm[ql < 63u ? ql : 63u];
                    }
                }
                else S[0] = S[1] = S_noq;
                #pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const uint32_t q = ((rchunk[u] >> rsh[u]) & RMASK) ^ cx[u]; rsh[u] += rstep[u];     // (complementing keeps an N an N)
                    c1[u] = __builtin_amdgcn_alignbit( c1[u], tchunk[u],  31 );
                    c0[u] = __builtin_amdgcn_alignbit( c0[u], tchunk1[u], 31 );
                    tchunk[u] <<= 2; tchunk1[u] <<= 2;
                    uint32_t x = (c0[u] ^ (0u - (q & 1u))) | (c1[u] ^ (0u - ((q >> 1) & 1u)));
                    if (RBITS > 2 && q >= 4u) x = 0xFFFFFFFFu;
                    if (i >= (RAGGED ? lim[u] + pad[u] : lim[u])) x |= 1u;      // column 30 past the text end
                    nq[u] = x;
                }
                // both alignments side by side: column j >= 15 at bits 30-j / 46-j of NWa, column j < 15 at bits 14-j / 30-j of NWb
                const uint32_t NWa = __builtin_amdgcn_perm( nq[1], nq[0], 0x05040100u );
                const uint32_t NWb = __builtin_amdgcn_perm( nq[1], nq[0], 0x07060302u );

                // M0 (match = 0): d = H + mismatch * S; else d = (H + V) + mismatch * (S - V); FP: the flag is 2^-7, the penalty S * 128
                const v2 SS = FP ? pk_of<v2>( S[0] * 128, S[1] * 128 ) : pk_of<v2>( S[0], S[1] );
                const v2 SD = pk_of<v2>( S[0] - V, S[1] - V );
                const v2 VV = pk_of<v2>( V, V );
                // column j's pair of mismatch flags: bits 30-j / 46-j of NWa (j >= 15) or 14-j / 30-j of NWb (j < 15), moved to bit 0 (13 with FP)
                // of each half
                auto flags_at = [&](const int j) -> uint32_t {                // ... moved into place,
                    const uint32_t w = (j < 15) ? NWb : NWa;
                    const int bit = (j < 15) ? 14 - j : 30 - j;              // of the low half
                    const int to  = FP ? 13 : 0;
                    return bit >= to ? w >> (bit - to) : w << (to - bit);
                };
                auto flags_in = [&](const uint32_t q) -> v2 { return __builtin_bit_cast( v2, q & (FP ? 0x20002000u : 0x00010001u) ); };   // ... and masked
                auto flags_of = [&](const int j) -> v2 { return flags_in( flags_at( j ) ); };
                auto diag_of = [&](const v2 h, const v2 t) -> v2 {
                    if (FP) return __builtin_bit_cast( v2, __builtin_elementwise_fma( __builtin_bit_cast( v2h, t ), __builtin_bit_cast( v2h, SS ), __builtin_bit_cast( v2h, h ) ) );
                    return M0 ? h + t * SS : (h + VV) + t * SD;
                };

                // One cell is 10 operations: f = max(F[j+1] + GE, Hg[j+1]); d = H[j] + mismatch * S; t = max(f, d); h = max(t, E);
                // hg = h + GO; E' = max(hg, E + GE) -- 9 with FP, where h = max(f, d, E) is one operation.  Only E runs along the row; gfx950
                // needs one idle slot between a packed 16-bit operation and a consumer issued right behind it, so the row is software-pipelined by
                // hand: the E steps of cell j are interleaved with everything of cell j+1 that does not depend on E, in an order in which no
                // operation directly follows its producer (sched_barrier keeps the compiler from undoing it: measured 15 -> 10 issue slots per cell).
                v2 E = ZERO, a = ZERO, tt, ff = INF, dd = ZERO;              // (FP carries f and d of the next cell instead of their maximum)
                v2s key = pk( -1, -1 );
                {
                    const v2 x  = F[1] + GE;
                    const v2 t0 = flags_of( 0 );
                    const v2 d  = diag_of( H[0], t0 );
                    const v2 f  = pk_max( x, Hg[1] );
                    F[0] = f;
                    if (FP) { ff = f; dd = d; tt = ZERO; }
                    else
                    {
                        tt = pk_max( f, d );
                        if (TYPE == NVBIO_LOCAL) tt = pk_max( tt, ZERO );
                    }
                }
                #pragma unroll
                for (int j = 0; j < BAND; ++j)
                {
                    constexpr int Z = 0;
                    const int j1 = j + 1, j2 = j + 2;
                    v2 x = INF, d = ZERO, f = INF, t1 = ZERO, tn = ZERO, an = ZERO;
                    uint32_t q1 = 0;
                    if (j2 < BAND) x  = F[j2] + GE;
                    __builtin_amdgcn_sched_barrier( Z );
                    const v2 h = FP ? ((j == 0) ? pk_max( ff, dd ) : pk_max3( ff, dd, E )) : ((j == 0) ? tt : pk_max( tt, E ));
                    if (j1 < BAND) q1 = flags_at( j1 );
                    __builtin_amdgcn_sched_barrier( Z );
                    const v2 hg = h + GO;
                    if (!FP && j1 < BAND) t1 = flags_in( q1 );
                    if (j2 < BAND) f  = pk_max( x, Hg[j2] );
                    __builtin_amdgcn_sched_barrier( Z );
                    const v2 En = (j == 0) ? hg : pk_max( hg, a );
                    if (FP) __builtin_amdgcn_sched_barrier( Z );            // (one cell is an operation shorter: keep E' away from its consumer)
                    if (FP && j1 < BAND) t1 = flags_in( q1 );
                    if (j1 < BAND) d  = diag_of( H[j1], t1 );
                    __builtin_amdgcn_sched_barrier( Z );
                    if (j1 < BAND) an = En + GE;
                    if (TYPE == NVBIO_LOCAL) key = pk_max( key, __builtin_bit_cast( v2s, h ) * K32 + pk( j, j ) );
                    __builtin_amdgcn_sched_barrier( Z );
                    if (j1 < BAND)
                    {
                        if (FP) { ff = (j2 < BAND) ? f : d; dd = d; }
                        else
                        {
                            tn = (j2 < BAND) ? pk_max( f, d ) : d;
                            if (TYPE == NVBIO_LOCAL) tn = pk_max( tn, ZERO );
                        }
                        F[j1] = f;
                    }
                    H[j] = h; Hg[j] = hg;
                    E = En; a = an; tt = tn;
                    __builtin_amdgcn_sched_barrier( Z );
                }

                if (TYPE == NVBIO_LOCAL)
                {
                    // BestSink: row-major reports, the LAST maximum wins
                    const int k0 = key.x, k1 = key.y;
                    if (i < rows_u[0] && (k0 >> 5) >= best[0]) { best[0] = k0 >> 5; best_x[0] = i + (uint32_t)(k0 & 31) + 1u; best_y[0] = i + 1u; }
                    if (i < rows_u[1] && (k1 >> 5) >= best[1]) { best[1] = k1 >> 5; best_x[1] = i + (uint32_t)(k1 & 31) + 1u; best_y[1] = i + 1u; }
                }
            }
        }

        // ---- end-of-alignment reports (:624-645); every active half ended at row `rows` (or has no rows) ----
        if (TYPE != NVBIO_LOCAL)
        {
            #pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                const bool active = (u ? want1 : want0) && (!split || u == pass);
                if (!active) continue;
                if (TYPE == NVBIO_GLOBAL)
                {
                    const int v = u ? (int)H[BAND - 1].y : (int)H[BAND - 1].x;
                    if (best[u] <= v) { best[u] = v; best_x[u] = M[u] + BAND - 1; best_y[u] = M[u]; }
                }
                else
                {
                    const uint32_t mb = M[u] + (uint32_t)(BAND - 1);
                    const uint32_t m  = (mb < N[u] ? mb : N[u]) - (M[u] - 1u);
                    #pragma unroll
                    for (int j = 0; j < BAND; ++j)
                        if (j == 0 || (uint32_t)j < m)
                        {
                            const int v = u ? (int)H[j].y : (int)H[j].x;
                            if (best[u] <= v) { best[u] = v; best_x[u] = M[u] + j; best_y[u] = M[u]; }
                        }
                }
            }
        }
    }
    #pragma unroll
    for (int u = 0; u < 2; ++u)
        if (valid[u]) { scores[out_id[u]] = best[u]; sinks[out_id[u]] = make_uint2( best_x[u], best_y[u] ); }
}

// ---------------------------------------------------------------------------------------------
// Ungapped shortcut for end-to-end (SEMI_GLOBAL) scoring with match = 0 (nvBowtie's default mode).
// Every alignment with at least one gap scores at most G = max(pattern gap open, text gap open) < 0: nothing
// else in it can be positive.  The ungapped alignments are the 31 diagonals of the band, diagonal d scoring
// U_d = -P * (number of rows i with read[i] != text[i+d]) for a constant mismatch penalty P.  Hence, if
// U* = max over the reportable end columns d of U_d is > G, the DP's optimum is U*, the last row holds U* in
// exactly the columns with U_d = U*, and BestSink's "last maximum wins" picks the largest such d -- all known
// from 31 shifted XOR + popcounts over bit planes, about an eighth of the DP's instructions.  Jobs with
// U* <= G (two or more mismatches at -6/-8, or an indel) are flagged and go through the DP unchanged.
// Reportable end columns: d = 0 always, d >= 1 iff d < min(M+30, N) - (M-1) (gotoh_banded_inl.h:631-643); such
// diagonals lie inside the text, so the band-31 cache quirk for symbols past the text end never touches them.
// ---------------------------------------------------------------------------------------------
// THIRD = false: the pass over every job (first and second chance); a job that neither settles but whose U* a third chance could
//   still settle (see below) gets need_dp = 2.
// THIRD = true : the pass over the list of those jobs (job_list / job_count on the device), which resolves each to 0 or 1.
//   Kept apart because the third chance is ten times the work of the rest and only one job in eight needs it: inside the first
//   pass every wave paid for it (measured: 0.39 -> 2.1 ms per launch), on a dense list it costs what it saves several times over.
// MODE 0: the pass over every job: best diagonal, settled or not; a job the second (third) chance could still settle is flagged
//   need_dp = 3 (2) with its best diagonal stashed in scores / sinks -- those checks cost as much as everything else in this pass and
//   every wave paid for them whenever one lane needed one (the second chance, one job in four, was HALF of this kernel's 1.46 ms).
// MODE 1 / 2: the pass over the dense list of the need_dp = 3 / 2 jobs (job_list / job_count on the device): the second / third chance,
//   each job ends as 0 or 1.
// QUAL (MODE 0 only): the mismatch penalty depends on the row's base quality (nvBowtie's default ramp, scoring.h:73-92), P = the
//   scheme's SMALLEST penalty (> 0).  U_d = -(sum of the penalties of diagonal d's mismatching rows) lies between -pmax c_d and
//   -P c_d for c_d mismatches, so only diagonals with P c_d <= pmax min_d c_d can hold the maximum: those few (the read's own diagonal,
//   its partners across an indel) are summed exactly, row by row, from the quality bytes; every other diagonal is ruled out by its
//   count alone.  Everything after that -- which classes of gapped alignments could still reach U* -- is argued with P as the least a
//   mismatch can cost, which only makes the classes larger (more jobs to the DP, never a wrong answer).
// the gap chance's cost ladder (gap_chance_e2e31_kernel): the cheapest class of gapped alignments it neither evaluates nor rules out, with P the
// least a mismatch can cost and gaps of up to GAP_CHANCE_GA symbols evaluated
constexpr int GAP_CHANCE_GA = 5;
__device__ __forceinline__ int32_t gap_chance_unknown_cost(const int32_t P, const int32_t go, const int32_t ge)
{
    auto cg = [&](const int g) -> int32_t { return -(go + (g - 1) * ge); };
    int32_t c = cg( 1 ) + 3 * P;                                  // one gap and three mismatches
    const int32_t others[] = { cg( GAP_CHANCE_GA + 1 ), 3 * cg( 1 ), 2 * cg( 1 ) + P, cg( 1 ) + cg( 3 ), 2 * cg( 2 ) };
    #pragma unroll
    for (int k = 0; k < 5; ++k) c = others[k] < c ? others[k] : c;
    return c;
}

template <int RBITS, int MODE, bool QUAL = false>
__global__ void __launch_bounds__(256)
ungapped_e2e31_kernel(const BatchDev b, const int32_t P, const int32_t G, const int32_t gap_open, const int32_t gap_ext,
                      int32_t* __restrict__ scores, uint2* __restrict__ sinks, uint8_t* __restrict__ need_dp,
                      const uint32_t* __restrict__ job_list = nullptr, const uint32_t* __restrict__ job_count = nullptr,
                      const SchemeDev sc = SchemeDev{})
{
    __shared__ int32_t s_pen[QUAL ? 64 : 1];
    if (QUAL)
    {
        if (threadIdx.x < 64) s_pen[threadIdx.x] = -mismatch_score( sc, threadIdx.x );     // the DP kernels' table, negated
        __syncthreads();
    }
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr bool LIST = MODE != 0;
    if (LIST ? slot >= *job_count : slot >= b.n) return;
    const uint32_t job = LIST ? job_list[slot] : slot;
    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = (b.max_read_len && M > b.max_read_len) ? 0u : b.win_end[job] - tb;       // too long for the declared bound: rejected

    if (N < M)                                                   // nothing reported (gotoh_banded_inl.h:422-423)
    {
        scores[job] = NVBIO_SCORE_MIN; sinks[job] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu ); need_dp[job] = 0;
        return;
    }
    if (M == 0u || M > 161u) { need_dp[job] = 1; return; }      // planes below hold 192 text symbols

    // bit planes of the read and of the window (bitplanes.h); all loads first, then the bit work
    uint64_t rlo[3], rhi[3], rn[3], tlo[4], thi[4];
    {
        ReadWords<RBITS> rw; TextWords13 tw;
        load_read_words<RBITS>( b.reads, first, M, rw );
        load_text_words13( b.text, tb, N < 192u ? N : 192u, tw );
        read_planes192<RBITS>( rw, first, M, rev, comp, rlo, rhi, rn );
        text_planes208( tw, tb, tlo, thi );
    }
    uint64_t rmask[3];
    #pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const int32_t left = (int32_t)M - 64 * k;
        rmask[k] = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
    }

    const uint32_t mb = M + 30u;
    const uint32_t m  = (mb < N ? mb : N) - (M - 1u);
    // the diagonal loop on 32-bit words: the text planes move down one bit per diagonal (one v_alignbit per word),
    // a mismatch word is 5 logic ops, and v_bcnt accumulates the count
    uint32_t pl[6], ph[6], pn[6], pm[6], ql[7], qh[7];
    #pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        pl[2*k] = (uint32_t)rlo[k]; pl[2*k+1] = (uint32_t)(rlo[k] >> 32);
        ph[2*k] = (uint32_t)rhi[k]; ph[2*k+1] = (uint32_t)(rhi[k] >> 32);
        pm[2*k] = (uint32_t)rmask[k]; pm[2*k+1] = (uint32_t)(rmask[k] >> 32);
        pn[2*k] = (uint32_t)rn[k] & pm[2*k]; pn[2*k+1] = (uint32_t)(rn[k] >> 32) & pm[2*k+1];
    }
    #pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        ql[2*k] = (uint32_t)tlo[k]; ql[2*k+1] = (uint32_t)(tlo[k] >> 32);
        qh[2*k] = (uint32_t)thi[k]; qh[2*k+1] = (uint32_t)(thi[k] >> 32);
    }
    ql[6] = (uint32_t)tlo[3]; qh[6] = (uint32_t)thi[3];
    uint32_t ql0[7], qh0[7];
    #pragma unroll
    for (int k = 0; k < 7; ++k) { ql0[k] = ql[k]; qh0[k] = qh[k]; }

    uint32_t best_cnt = 0xFFFFFFFFu, best_d = 0;
    int64_t U = 0;
    if (LIST)
    {
        // the first pass left this job's best diagonal and its score in scores / sinks
        U      = scores[job];
        best_d = sinks[job].x - M;
    }
    else
    {
        // A diagonal matters only while its count can still change the outcome: no class of alignments the three chances know settles a job
        // whose best diagonal scores 3 G - P or less (the third chance's `beyond`), so counts above `cap` mean "DP" whatever they are, and
        // without qualities neither does a count above the best one found so far.  The first word of a diagonal (32 rows; about 24
        // mismatches on a diagonal that is not the read's own or its partner across an indel) decides that for the whole wave nearly
        // always: its other words are evaluated only if SOME lane is still interested (a wave-uniform branch) -- 24 of 31 diagonals cost one
        // word instead of six.  The text planes are taken d symbols on with one funnel shift per word (d is a constant of the unrolled loop).
        const int64_t floor_u = 3 * (int64_t)(G < gap_open ? G : gap_open) - P;          // U <= floor_u: DP whatever else holds
        const uint32_t cap = P > 0 ? (uint32_t)((-floor_u - 1) / (int64_t)P) : 0xFFFFFFFEu;   // the largest count with -P cnt > floor_u
        uint32_t cnt_d[QUAL ? 31 : 1];
        #pragma unroll
        for (uint32_t d = 0; d < 31u; ++d)
        {
            const bool on = (d == 0u || d < m);                      // reportable columns are a prefix of 0..30
            uint32_t cnt;
            {
                const uint32_t tl = d ? __builtin_amdgcn_alignbit( ql0[1], ql0[0], d ) : ql0[0], th = d ? __builtin_amdgcn_alignbit( qh0[1], qh0[0], d ) : qh0[0];
                cnt = (uint32_t)__popc( (((pl[0] ^ tl) | (ph[0] ^ th)) & pm[0]) | pn[0] );
            }
            const uint32_t thr = QUAL ? cap : (best_cnt < cap ? best_cnt : cap);
            if (__any( on && cnt <= thr ))
            {
                #pragma unroll
                for (int k = 1; k < 6; ++k)
                {
                    const uint32_t tl = d ? __builtin_amdgcn_alignbit( ql0[k + 1], ql0[k], d ) : ql0[k], th = d ? __builtin_amdgcn_alignbit( qh0[k + 1], qh0[k], d ) : qh0[k];
                    cnt += (uint32_t)__popc( (((pl[k] ^ tl) | (ph[k] ^ th)) & pm[k]) | pn[k] );
                }
            }
            else cnt = 0xFFFFFFFFu;                                  // (partial, and above every lane's threshold)
            if (QUAL) cnt_d[d] = on ? cnt : 0xFFFFFFFFu;
            if (on && cnt <= best_cnt) { best_cnt = cnt; best_d = d; }     // ties: the larger column, as BestSink's `<=`
        }
        if (best_cnt > cap)                                          // (includes: no diagonal evaluated)
        {
            // no diagonal within reach of the three chances: typically a read with an indel.  The gap chance (gap_chance_e2e31_kernel, its own
            // list pass) evaluates the one-gap alignments of such a job exactly; it needs every diagonal inside the text and plain gap terms
            // (not under a quality ramp: see the kernel's header)
            const bool gap_chance = !QUAL && P > 0 && N >= M + 30u && gap_ext < 0 && gap_open <= gap_ext && !(b.algo & NVBIO_ALN_NO_GAP_CHANCE);
            need_dp[job] = gap_chance ? 4 : 1;
            return;
        }
        U = -(int64_t)P * (int64_t)best_cnt;
        if (QUAL)
        {
            // no class of alignments the three chances know can be settled below 3 G - P (the third chance's `beyond`): such a job
            // needs the DP whatever its exact U*, so its qualities are not even read
            if (U <= 3 * (int64_t)G - P) { need_dp[job] = 1; return; }
            const uint32_t pmax  = (uint32_t)s_pen[63];
            const uint64_t bound = (uint64_t)pmax * best_cnt;                    // a diagonal with P c_d > pmax c_min cannot hold the maximum
            uint32_t cand = 0;
            #pragma unroll
            for (uint32_t d = 0; d < 31u; ++d)
                if (cnt_d[d] != 0xFFFFFFFFu && (uint64_t)(uint32_t)P * cnt_d[d] <= bound) cand |= 1u << d;
            uint32_t best_w = 0xFFFFFFFFu;
            while (cand)                                                          // ascending d: `<=` keeps the larger column on ties
            {
                const uint32_t d = (uint32_t)__builtin_ctz( cand );
                cand &= cand - 1u;
                uint32_t w = 0;
                #pragma unroll
                for (int k = 0; k < 6; ++k)
                {
                    // the text planes d symbols on (d <= 30 < 32: one funnel shift per word)
                    const uint32_t tl = __builtin_amdgcn_alignbit( ql0[k + 1], ql0[k], d ), th = __builtin_amdgcn_alignbit( qh0[k + 1], qh0[k], d );
                    uint32_t mm = (((pl[k] ^ tl) | (ph[k] ^ th)) & pm[k]) | pn[k];
                    while (mm)
                    {
                        const uint32_t row = 32u * k + (uint32_t)__builtin_ctz( mm );
                        mm &= mm - 1u;
                        const uint32_t qq = b.quals[rev ? first + M - 1u - row : first + row];
                        w += (uint32_t)s_pen[qq < 63u ? qq : 63u];
                    }
                }
                if (w <= best_w) { best_w = w; best_d = d; }
            }
            U = -(int64_t)best_w;
        }
    }
    bool settled = U > (int64_t)G;
    auto stash = [&]() { scores[job] = (int32_t)U; sinks[job] = make_uint2( M + best_d, M ); };
    const bool second_applies = !settled && (int64_t)G - P < U && 2 * (int64_t)G < U && N >= M + 30u && !(b.algo & NVBIO_ALN_NO_SECOND_CHANCE);
    if (MODE == 0 && second_applies)
    {
        int32_t gmax = 0;
        while (gmax < 5 && (int64_t)gap_open + (int64_t)gmax * gap_ext >= U) ++gmax;
        if (gmax >= 1 && gmax <= 4) { need_dp[job] = 3; stash(); return; }      // for the second-chance launch
    }
    if (MODE == 1 && second_applies)
    {
        // Second chance (two mismatches at nvBowtie's -6 / -8 / -3).  U* <= G, but one gap plus one mismatch and two gaps
        // both score below U*: the only gapped alignments that could reach U* have exactly ONE gap and NO mismatch, i.e. a
        // prefix of the read matching one diagonal exactly and the rest matching a diagonal g columns away, for the few gap
        // lengths g with open + (g-1) ext >= U*.  With lead_d / tail_d = the exactly matching prefix / suffix lengths of
        // diagonal d, a text gap (d-g -> d) exists iff lead_{d-g} + tail_d >= M, a pattern gap (d -> d-g) iff
        // lead_d + tail_{d-g} + g >= M.  If none exists the optimum is U* and only ungapped diagonals reach it.
        // (N >= M + 30: all 31 diagonals lie inside the text, so no sentinel cell is involved.)
        int32_t gmax = 0;
        while (gmax < 5 && (int64_t)gap_open + (int64_t)gmax * gap_ext >= U) ++gmax;      // lengths 1..gmax can reach U*
        if (gmax >= 1 && gmax <= 4)
        {
            #pragma unroll
            for (int k = 0; k < 7; ++k) { ql[k] = ql0[k]; qh[k] = qh0[k]; }
            uint32_t lead_prev[4] = { 0, 0, 0, 0 }, tail_prev[4] = { 0, 0, 0, 0 };        // diagonals d-1 .. d-4
            bool gapped = false;
            for (uint32_t d = 0; d < 31u && !gapped; ++d)
            {
                // first / last mismatching row of the diagonal, searched from the two ends and only as far as needed: on a diagonal that
                // is not (nearly) the read's own, word 0 and the top word already hold mismatches, and a word is evaluated by the wave
                // only while some lane is still looking -- 3 word evaluations per diagonal instead of 12 (the values are the same)
                uint32_t first = M, last = 0xFFFFFFFFu;
                #pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (first == M)
                    {
                        const uint32_t mm = (((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) & pm[k]) | pn[k];
                        if (mm) first = 32u * k + (uint32_t)__builtin_ctz( mm );
                    }
                #pragma unroll
                for (int k = 5; k >= 0; --k)
                    if (last == 0xFFFFFFFFu)
                    {
                        const uint32_t mm = (((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) & pm[k]) | pn[k];
                        if (mm) last = 32u * k + 31u - (uint32_t)__builtin_clz( mm );
                    }
                const uint32_t lead = first;                                               // rows 0..lead-1 match
                const uint32_t tail = (last == 0xFFFFFFFFu) ? M : M - 1u - last;           // the last `tail` rows match
                #pragma unroll
                for (int g = 1; g <= 4; ++g)
                    if (g <= gmax && d >= (uint32_t)g)
                    {
                        if (lead_prev[g - 1] + tail >= M) gapped = true;                   // text gap of g: diagonal d-g, then d
                        if (lead + tail_prev[g - 1] + (uint32_t)g >= M) gapped = true;     // pattern gap of g: diagonal d, then d-g
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
            settled = !gapped;
        }
    }
    // which classes of gapped alignments can reach U* by score alone (third chance; two or three mismatches at -6 / -8 / -3):
    //   A  one gap of g <= gmax0 symbols, no mismatch          (open + (g-1) ext >= U*)
    //   B  one gap of g <= gmax1 symbols and ONE mismatch      (open + (g-1) ext - P >= U*)
    //   C  two gaps of one symbol each, no mismatch            (2 open >= U*)
    // anything beyond (one gap + two mismatches, two gaps with a longer one or a mismatch, three gaps, gaps over 4) means DP.
    const int64_t go = gap_open, ge = gap_ext;
    int32_t gmax0 = 0, gmax1 = 0;
    while (gmax0 < 5 && go + (int64_t)gmax0 * ge >= U) ++gmax0;
    while (gmax1 < 5 && go + (int64_t)gmax1 * ge - P >= U) ++gmax1;
    const bool two11  = 2 * go >= U;
    const bool beyond = gmax0 > 4 || gmax1 > 4 || go - 2 * (int64_t)P >= U || 2 * go + ge >= U || 2 * go - P >= U || 3 * go >= U || ge < go;
    const bool third_applies = !settled && N >= M + 30u && !beyond && gmax0 >= 1 && (gmax1 >= 1 || two11);
    if (MODE == 0 && third_applies)
    {
        need_dp[job] = 2; stash();                               // for the third-chance launch
        return;
    }
    if (MODE != 0 && third_applies)                              // (MODE 1: a job the second chance did not settle)
    {
        // With lead0/lead1(d) = rows before the first / second mismatch of diagonal d and tail0/tail1(d) = rows after its last /
        // last-but-one mismatch:
        //   text gap g (prefix on d-g, suffix on d) with <= e mismatches  iff  lead_i(d-g) + tail_{e-i}(d) >= M for some i <= e
        //   pattern gap g (prefix on d, suffix on d-g)                    iff  lead_i(d) + tail_{e-i}(d-g) + g >= M
        //   C through a middle diagonal b with neighbours a, c = b +- 1: the prefix on a may run to row lo = lead0(a) (+1 after a
        //     pattern gap), the suffix on c may start at row hi = M - tail0(c) (-1 before a pattern gap): it exists iff lo >= hi
        //     or b has no mismatch in rows [lo, hi).
        // Existence is over-approximated (band limits and reportable columns are ignored), which only costs a DP.  If no class
        // has a member the optimum is U* and only ungapped diagonals reach it.  A gapped alignment that merely TIES U* also goes
        // to the DP (the sink rule decides there).  (N >= M + 30: all 31 diagonals lie inside the text, no sentinel cell.)
        #pragma unroll
        for (int k = 0; k < 7; ++k) { ql[k] = ql0[k]; qh[k] = qh0[k]; }
        uint32_t lead0_p[4] = { 0, 0, 0, 0 }, lead1_p[4] = { 0, 0, 0, 0 }, tail0_p[4] = { 0, 0, 0, 0 }, tail1_p[4] = { 0, 0, 0, 0 };   // d-1 .. d-4
        bool gapped = false;
        uint32_t mwp[6] = { 0, 0, 0, 0, 0, 0 };
        for (uint32_t d = 0; d <= 31u && !gapped; ++d)
        {
            const bool have = d < 31u;                                                  // d = 31 only closes class C for b = 30
            // the first two and the last two mismatching rows of the diagonal, searched from the two ends and only as far as needed: on a
            // diagonal that is not (nearly) the read's own the first and the top word already hold two mismatches each, and a word is
            // evaluated by the wave only while some lane is still looking (the values are the same as with all 12 word evaluations)
            uint32_t first = M, second = M, last = 0xFFFFFFFFu, last2 = 0xFFFFFFFFu;
            #pragma unroll
            for (int k = 0; k < 6; ++k)
                if (have && __any( second == M ))                       // a wave-uniform branch: really skipped when no lane is looking
                {
                    uint32_t w = (second == M) ? ((((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) & pm[k]) | pn[k]) : 0u;
                    if (first == M && w) { first = 32u * k + (uint32_t)__builtin_ctz( w ); w &= w - 1u; }
                    if (first != M && second == M && w) second = 32u * k + (uint32_t)__builtin_ctz( w );
                }
            #pragma unroll
            for (int k = 5; k >= 0; --k)
                if (have && __any( last2 == 0xFFFFFFFFu ))
                {
                    uint32_t w = (last2 == 0xFFFFFFFFu) ? ((((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) & pm[k]) | pn[k]) : 0u;
                    if (last == 0xFFFFFFFFu && w) { const uint32_t t = 31u - (uint32_t)__builtin_clz( w ); last = 32u * k + t; w &= ~(1u << t); }
                    if (last != 0xFFFFFFFFu && last2 == 0xFFFFFFFFu && w) last2 = 32u * k + 31u - (uint32_t)__builtin_clz( w );
                }
            const uint32_t lead0 = first, lead1 = second;                               // rows before the 1st / 2nd mismatch
            const uint32_t tail0 = (last  == 0xFFFFFFFFu) ? M : M - 1u - last;          // rows after the last / last-but-one
            const uint32_t tail1 = (last2 == 0xFFFFFFFFu) ? M : M - 1u - last2;
            if (have)
            {
                #pragma unroll
                for (int g = 1; g <= 4; ++g)
                    if (d >= (uint32_t)g)
                    {
                        if (g <= gmax0)
                        {
                            if (lead0_p[g - 1] + tail0 >= M) gapped = true;                                   // A, text gap
                            if (lead0 + tail0_p[g - 1] + (uint32_t)g >= M) gapped = true;                     // A, pattern gap
                        }
                        if (g <= gmax1)
                        {
                            if (lead0_p[g - 1] + tail1 >= M || lead1_p[g - 1] + tail0 >= M) gapped = true;    // B, text gap
                            if (lead0 + tail1_p[g - 1] + (uint32_t)g >= M || lead1 + tail0_p[g - 1] + (uint32_t)g >= M) gapped = true;
                        }
                    }
            }
            if (two11 && d >= 1u)
            {
                // middle diagonal b = d-1 (words mmp); neighbours d-2 (lead0_p[1] / tail0_p[1], if d >= 2) and d (if have)
                int32_t lo_c[2], hi_c[2]; int n_lo = 0, n_hi = 0;
                if (d >= 2u) { lo_c[n_lo++] = (int32_t)lead0_p[1];        hi_c[n_hi++] = (int32_t)M - (int32_t)tail0_p[1] - 1; }   // a = b-1: text gap in; c = b-1: pattern gap out
                if (have)    { lo_c[n_lo++] = (int32_t)lead0 + 1;         hi_c[n_hi++] = (int32_t)M - (int32_t)tail0; }           // a = b+1: pattern gap in; c = b+1: text gap out
                for (int x = 0; x < n_lo; ++x)
                    for (int y = 0; y < n_hi; ++y)
                    {
                        const int32_t lo = lo_c[x], hi = hi_c[y];
                        if (lo >= hi) { gapped = true; continue; }
                        // cheap first: b's first or last mismatch inside [lo, hi) settles it (nearly always, b being an unrelated
                        // diagonal); only otherwise look at the words
                        const int32_t bf = (int32_t)lead0_p[0], bl = (int32_t)M - 1 - (int32_t)tail0_p[0];
                        bool any = (bf >= lo && bf < hi) || (tail0_p[0] < M && bl >= lo && bl < hi);
                        bool full = !any;
                        if (__any( !any ))
                        {
                            // next: the 32 rows from lo on (two of b's words, picked by lo's word index) -- on an unrelated diagonal they hold a
                            // mismatch, which settles it; only an interval that is longer AND clean so far walks all the words
                            const uint32_t wi = (uint32_t)lo >> 5;
                            uint32_t w_lo = mwp[0], w_hi = mwp[1];
                            #pragma unroll
                            for (int k = 1; k < 6; ++k)
                            {
                                const uint32_t next = k < 5 ? mwp[k < 5 ? k + 1 : 5] : 0u;
                                w_lo = wi == (uint32_t)k ? mwp[k] : w_lo;
                                w_hi = wi == (uint32_t)k ? next   : w_hi;
                            }
                            uint32_t rows32 = __builtin_amdgcn_alignbit( w_hi, w_lo, (uint32_t)lo & 31u );      // b's mismatches in rows lo .. lo + 31
                            const int32_t span = hi - lo;                                                        // > 0 here
                            if (span < 32) rows32 &= (1u << span) - 1u;
                            if (!any && wi < 6u) { if (rows32) { any = true; full = false; } else if (span <= 32) full = false; }
                        }
                        if (full)
                        {
                            #pragma unroll
                            for (int k = 0; k < 6; ++k)
                            {
                                const int32_t a0 = lo - 32 * k > 0 ? lo - 32 * k : 0, a1 = hi - 32 * k < 32 ? hi - 32 * k : 32;
                                if (a0 < a1)
                                {
                                    const uint32_t hi_m = (a1 >= 32) ? 0xFFFFFFFFu : ((1u << a1) - 1u);
                                    const uint32_t lo_m = (1u << a0) - 1u;                                       // a0 < a1 <= 32 -> a0 <= 31
                                    // word k of the middle diagonal d-1: kept from the previous iteration
                                    any = any || ((mwp[k] & hi_m & ~lo_m) != 0u);
                                }
                            }
                        }
                        if (!any) gapped = true;
                    }
            }
            #pragma unroll
            for (int k = 3; k > 0; --k) { lead0_p[k] = lead0_p[k - 1]; lead1_p[k] = lead1_p[k - 1]; tail0_p[k] = tail0_p[k - 1]; tail1_p[k] = tail1_p[k - 1]; }
            lead0_p[0] = lead0; lead1_p[0] = lead1; tail0_p[0] = tail0; tail1_p[0] = tail1;
            if (two11 && have)                                   // this diagonal's mismatch words: the middle diagonal of the next iteration
            {
                #pragma unroll
                for (int k = 0; k < 6; ++k) mwp[k] = (((pl[k] ^ ql[k]) | (ph[k] ^ qh[k])) & pm[k]) | pn[k];
            }
            #pragma unroll
            for (int k = 0; k < 6; ++k)
            {
                ql[k] = __builtin_amdgcn_alignbit( ql[k + 1], ql[k], 1u );
                qh[k] = __builtin_amdgcn_alignbit( qh[k + 1], qh[k], 1u );
            }
            ql[6] >>= 1; qh[6] >>= 1;
        }
        settled = !gapped;
    }
    if (settled)
    {
        scores[job] = (int32_t)U; sinks[job] = make_uint2( M + best_d, M ); need_dp[job] = 0;
    }
    else need_dp[job] = 1;
}

// ---------------------------------------------------------------------------------------------
// The GAP chance (need_dp = 4: no diagonal of the window has few enough mismatches for the three chances above -- a read with an indel, its
// two candidate windows, four fifths of what used to reach the DP).  Same setting (SEMI_GLOBAL, match 0, one penalty P, plain gap terms,
// all 31 diagonals inside the text); with Cg(g) = -(open + (g-1) ext) the cost of a gap of g symbols, an alignment costs the sum of its gaps'
// Cg plus P per mismatch, and the classes of alignments are ordered by that cost.  EVALUATED exactly: ONE gap of g <= 5 symbols with e <= 2
// mismatches, by the lead / tail argument of the other chances taken to the third mismatch -- with lead_i(d) / tail_i(d) = the rows before
// the (i+1)-th / after the (i+1)-th-from-last mismatch of diagonal d, a text gap (prefix on d-g, suffix on d, ending in column d) with <= e
// mismatches exists iff lead_i(d-g) + tail_{e-i}(d) >= M for some i <= e, a pattern gap (prefix on d, suffix on d-g, ending in d-g) iff
// lead_i(d) + tail_{e-i}(d-g) + g >= M.  The minimum c* over all of them and the LARGEST end column among the members that reach it are the
// DP's optimum and BestSink's sink, PROVIDED no other alignment costs <= c*:
//   * ungapped: every diagonal has more than `cap` mismatches (why a need_dp = 4 job is here): >= (cap + 1) P.  A THIRD-CHANCE job (need_dp = 2)
//     brings its best diagonal instead (2 or 3 mismatches, U* and column stashed by the first pass): that class joins the evaluated ones with its
//     exact cost -P U* and its column, every other diagonal costs at least as much, and the term drops out of the bound;
//   * one gap with 3 mismatches, or of 6 symbols: Cg(1) + 3 P, Cg(6);  three gaps: 3 Cg(1);  two gaps and a mismatch: 2 Cg(1) + P;
//     two gaps of (1,3) / (2,2) or more: Cg(1) + Cg(3), 2 Cg(2)          -- c_unk = the least of these; c* < c_unk is required;
//   * two gaps of (1,1) or (1,2) / (2,1) symbols without a mismatch (costs 2 Cg(1), Cg(1) + Cg(2)): EXISTENCE is tested (over-approximated)
//     through the middle diagonal b as in the third chance: the prefix on a neighbour a may run to row lo, the suffix on a neighbour c may start
//     at row hi, a member exists iff lo >= hi or b has no mismatch in rows [lo, hi); one that costs <= c* sends the job to the DP.
// Anything else -- c* >= c_unk, no member at all, a two-gap member in reach -- is the DP's.  Over-approximation only ever costs a DP.
// ---------------------------------------------------------------------------------------------
// Not under a quality ramp.  Built and measured (members priced with the penalties their mismatching rows really carry, every split of them between
// prefix and suffix; parity green): with penalties 2..6 the cheapest class the kernel cannot see -- one gap and three mismatches at the SMALLEST
// penalty -- costs 14, so a job settles only while its optimum costs 13 or less; the third chance already takes U* >= -11 there, and on the robust
// batch the pass cost 1.1-1.7 ms for 0.5-1.0 ms of DP saved.  With the ladder taken to four priced mismatches per member (bound 18: two gaps and a
// mismatch; also built, also green) the DP went 3.9 -> 2.7 ms and the pass cost 2.1: a job costs the pass half a DP and only half of them settle.
template <int RBITS>
__global__ void __launch_bounds__(256)
gap_chance_e2e31_kernel(const BatchDev b, const int32_t P, const int32_t G, const int32_t gap_open, const int32_t gap_ext,
                        int32_t* __restrict__ scores, uint2* __restrict__ sinks, uint8_t* __restrict__ need_dp,
                        const uint32_t* __restrict__ job_list, const uint32_t* __restrict__ job_count)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= *job_count) return;
    const uint32_t job = job_list[slot];
    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0;
    const bool     comp  = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = b.win_end[job] - tb;
    // (flagged by the first pass: 1 <= M <= 161, N >= M + 30, P > 0, open <= ext < 0)
    // need_dp = 2: a third-chance job -- its best diagonal (2 or 3 mismatches) is in scores / sinks; that ungapped class joins the evaluated ones
    const bool     has_u = need_dp[job] == 2;
    const int32_t  cu    = has_u ? -scores[job] : 0x7FFFFFFF;
    const uint32_t cu_end = has_u ? sinks[job].x - M : 0u;

    uint32_t pl[6], ph[6], pn[6], pm[6], ql[7], qh[7];
    {
        uint64_t rlo[3], rhi[3], rn[3], tlo[4], thi[4];
        {
            ReadWords<RBITS> rw; TextWords13 tw;
            load_read_words<RBITS>( b.reads, first, M, rw );
            load_text_words13( b.text, tb, N < 192u ? N : 192u, tw );
            read_planes192<RBITS>( rw, first, M, rev, comp, rlo, rhi, rn );
            text_planes208( tw, tb, tlo, thi );
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
            ql[2*k] = (uint32_t)tlo[k]; ql[2*k+1] = (uint32_t)(tlo[k] >> 32);
            qh[2*k] = (uint32_t)thi[k]; qh[2*k+1] = (uint32_t)(thi[k] >> 32);
        }
        ql[6] = (uint32_t)tlo[3]; qh[6] = (uint32_t)thi[3];
    }

    // the cost ladder
    constexpr int GA = GAP_CHANCE_GA;
    const int32_t go = gap_open, ge = gap_ext;
    int32_t cg[GA + 2];                                          // cg[g] = cost of a gap of g symbols
    #pragma unroll
    for (int g = 1; g <= GA + 1; ++g) cg[g] = -(go + (g - 1) * ge);
    cg[0] = 0;
    const int64_t floor_u = 3 * (int64_t)(G < gap_open ? G : gap_open) - P;
    const int32_t cap = (int32_t)((-floor_u - 1) / (int64_t)P);
    int32_t c_unk = gap_chance_unknown_cost( P, go, ge );
    if (!has_u && (cap + 1) * P < c_unk) c_unk = (cap + 1) * P;  // (every OTHER diagonal of a job that brings its best one costs at least that much)
    const int32_t cost11 = 2 * cg[1], cost12 = cg[1] + cg[2];

    // mismatch word k (rows 32 k .. 32 k + 31) of diagonal x: the text planes x symbols on (x is wave-uniform: one funnel shift per plane word)
    auto mmw = [&](const int k, const uint32_t x) -> uint32_t {
        const uint32_t tl = __builtin_amdgcn_alignbit( ql[k + 1], ql[k], x ), th = __builtin_amdgcn_alignbit( qh[k + 1], qh[k], x );
        return (((pl[k] ^ tl) | (ph[k] ^ th)) & pm[k]) | pn[k];
    };
    // the read's LAST 32 rows (rows base .. base + 31; all of them if it has fewer) and the 62 text symbols they can meet, as words of their own:
    // bit j of (plT, phT, pnT, pmT) = row base + j, bit j of (qlT, qhT) = text symbol base + j
    const uint32_t base = M >= 32u ? M - 32u : 0u;
    uint32_t plT, phT, pnT, pmT, qlT[2], qhT[2];
    {
        const uint32_t bw = base >> 5, bs = base & 31u;
        uint32_t a[4] = { 0, 0, 0, 0 }, c[4] = { 0, 0, 0, 0 }, t[3] = { 0, 0, 0 }, u[3] = { 0, 0, 0 };
        #pragma unroll
        for (int k = 0; k < 6; ++k)
            if (bw == (uint32_t)k)
            {
                a[0] = pl[k]; a[1] = k + 1 < 6 ? pl[k + 1] : 0u; a[2] = ph[k]; a[3] = k + 1 < 6 ? ph[k + 1] : 0u;
                c[0] = pn[k]; c[1] = k + 1 < 6 ? pn[k + 1] : 0u; c[2] = pm[k]; c[3] = k + 1 < 6 ? pm[k + 1] : 0u;
                t[0] = ql[k]; t[1] = k + 1 < 7 ? ql[k + 1] : 0u; t[2] = k + 2 < 7 ? ql[k + 2] : 0u;
                u[0] = qh[k]; u[1] = k + 1 < 7 ? qh[k + 1] : 0u; u[2] = k + 2 < 7 ? qh[k + 2] : 0u;
            }
        plT = __builtin_amdgcn_alignbit( a[1], a[0], bs ); phT = __builtin_amdgcn_alignbit( a[3], a[2], bs );
        pnT = __builtin_amdgcn_alignbit( c[1], c[0], bs ); pmT = __builtin_amdgcn_alignbit( c[3], c[2], bs );
        qlT[0] = __builtin_amdgcn_alignbit( t[1], t[0], bs ); qlT[1] = __builtin_amdgcn_alignbit( t[2], t[1], bs );
        qhT[0] = __builtin_amdgcn_alignbit( u[1], u[0], bs ); qhT[1] = __builtin_amdgcn_alignbit( u[2], u[1], bs );
    }
    auto pen_of = [&](const int32_t) -> int32_t { return P; };   // what a mismatch costs, whatever its row (one penalty for every quality)
    // history of the last GA diagonals (slot k: diagonal d - 1 - k): (lead0, lead1, lead2) and (tail0, tail1, tail2) packed a byte each (<= 161)
    uint32_t Lp[GA], Tp[GA];
    #pragma unroll
    for (int k = 0; k < GA; ++k) { Lp[k] = 0; Tp[k] = 0; }
    uint32_t hot = 0;                                            // bit k: slot k's diagonal could be half of a one-gap alignment
    int32_t best_cost = 0x7FFFFFFF; uint32_t best_end = 0;
    bool ex11 = false, ex12 = false;
    const int32_t Mi = (int32_t)M;
    // lead_i(a) + tail_j(c) + g >= M with g <= GA needs one of the two at least (M - GA) / 2: only such diagonals are looked at pair by pair
    const int32_t hot_thr = (Mi - GA) / 2;

    // no mismatch of diagonal x in rows [lo, hi)?  (lo >= hi: an empty middle segment -- counted as a member.)  f2 / l2: its third mismatch from
    // either end (M / -1 if it has fewer): nearly always one of them lies inside and no word is looked at
    auto clean = [&](const uint32_t x, const int32_t bf, const int32_t bl, const int32_t lo, const int32_t hi) -> bool {
        if (lo >= hi) return true;
        if ((bf >= lo && bf < hi) || (bl >= lo && bl < hi)) return false;
        bool any = false;
        #pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            const int32_t a0 = lo - 32 * k > 0 ? lo - 32 * k : 0, a1 = hi - 32 * k < 32 ? hi - 32 * k : 32;
            if (a0 < a1)
            {
                const uint32_t hi_m = (a1 >= 32) ? 0xFFFFFFFFu : ((1u << a1) - 1u);
                const uint32_t lo_m = (1u << a0) - 1u;
                any = any || ((mmw( k, x ) & hi_m & ~lo_m) != 0u);
            }
        }
        return !any;
    };

    for (uint32_t d = 0; d <= 32u; ++d)
    {
        const bool have = d < 31u;
        // the first three and the last three mismatching rows of diagonal d.  A diagonal that is not the read's own (or its partner across the
        // indel) holds three mismatches in its first 32 and in its last 32 rows: those two words decide, branch-free; only where some lane of
        // the wave found fewer are the words walked from both ends, as far as some lane still needs
        uint32_t f0 = M, f1 = M, f2 = M, l0 = 0xFFFFFFFFu, l1 = 0xFFFFFFFFu, l2 = 0xFFFFFFFFu;
        if (have)
        {
            uint32_t w = mmw( 0, d );
            if (w) { f0 = (uint32_t)__builtin_ctz( w ); w &= w - 1u; }
            if (w) { f1 = (uint32_t)__builtin_ctz( w ); w &= w - 1u; }
            if (w) f2 = (uint32_t)__builtin_ctz( w );
            const uint32_t tl = __builtin_amdgcn_alignbit( qlT[1], qlT[0], d ), th = __builtin_amdgcn_alignbit( qhT[1], qhT[0], d );
            w = (((plT ^ tl) | (phT ^ th)) & pmT) | pnT;          // rows base .. base + 31
            if (w) { const uint32_t t = 31u - (uint32_t)__builtin_clz( w ); l0 = base + t; w &= ~(1u << t); }
            if (w) { const uint32_t t = 31u - (uint32_t)__builtin_clz( w ); l1 = base + t; w &= ~(1u << t); }
            if (w) l2 = base + 31u - (uint32_t)__builtin_clz( w );
        }
        if (have && __any( M > 32u && (f2 == M || l2 == 0xFFFFFFFFu) ))
        {
            f0 = f1 = f2 = M; l0 = l1 = l2 = 0xFFFFFFFFu;
            #pragma unroll
            for (int k = 0; k < 6; ++k)
                if (__any( f2 == M && pm[k] != 0u ))
                {
                    uint32_t w = (f2 == M) ? mmw( k, d ) : 0u;
                    if (f0 == M && w) { f0 = 32u * k + (uint32_t)__builtin_ctz( w ); w &= w - 1u; }
                    if (f0 != M && f1 == M && w) { f1 = 32u * k + (uint32_t)__builtin_ctz( w ); w &= w - 1u; }
                    if (f1 != M && f2 == M && w) f2 = 32u * k + (uint32_t)__builtin_ctz( w );
                }
            #pragma unroll
            for (int k = 5; k >= 0; --k)
                if (__any( l2 == 0xFFFFFFFFu && pm[k] != 0u ))
                {
                    uint32_t w = (l2 == 0xFFFFFFFFu) ? mmw( k, d ) : 0u;
                    if (l0 == 0xFFFFFFFFu && w) { const uint32_t t = 31u - (uint32_t)__builtin_clz( w ); l0 = 32u * k + t; w &= ~(1u << t); }
                    if (l0 != 0xFFFFFFFFu && l1 == 0xFFFFFFFFu && w) { const uint32_t t = 31u - (uint32_t)__builtin_clz( w ); l1 = 32u * k + t; w &= ~(1u << t); }
                    if (l1 != 0xFFFFFFFFu && l2 == 0xFFFFFFFFu && w) l2 = 32u * k + 31u - (uint32_t)__builtin_clz( w );
                }
        }
        const int32_t L0 = (int32_t)f0, L1 = (int32_t)f1, L2 = (int32_t)f2;          // rows before the 1st / 2nd / 3rd mismatch
        const int32_t T0 = l0 == 0xFFFFFFFFu ? Mi : Mi - 1 - (int32_t)l0;             // rows after the last / last-but-one / last-but-two
        const int32_t T1 = l1 == 0xFFFFFFFFu ? Mi : Mi - 1 - (int32_t)l1;
        const int32_t T2 = l2 == 0xFFFFFFFFu ? Mi : Mi - 1 - (int32_t)l2;
        const bool hot_d = have && (L2 >= hot_thr || T2 >= hot_thr);

        if (have && __any( hot_d || (hot & 31u) != 0u ))
        {
            #pragma unroll
            for (int g = 1; g <= GA; ++g)
                if (d >= (uint32_t)g && (hot_d || ((hot >> (g - 1)) & 1u)))
                {
                    const int32_t a0 = (int32_t)(Lp[g - 1] & 255u), a1 = (int32_t)((Lp[g - 1] >> 8) & 255u), a2 = (int32_t)(Lp[g - 1] >> 16);
                    const int32_t t0 = (int32_t)(Tp[g - 1] & 255u), t1 = (int32_t)((Tp[g - 1] >> 8) & 255u), t2 = (int32_t)(Tp[g - 1] >> 16);
                    // text gap of g: diagonal d - g (prefix), then d (suffix); ends in column d
                    if (a2 + T2 >= Mi)
                    {
                        const int32_t pa[2] = { pen_of( a0 ), pen_of( a1 ) };                     // the prefix diagonal's first / second mismatch
                        const int32_t pt[2] = { pen_of( Mi - 1 - T0 ), pen_of( Mi - 1 - T1 ) };   // the suffix diagonal's last / last-but-one
                        const int32_t lead[3] = { a0, a1, a2 }, tail[3] = { T0, T1, T2 };
                        int32_t c = 0x7FFFFFFF;
                        #pragma unroll
                        for (int i = 0; i <= 2; ++i)
                            #pragma unroll
                            for (int j = 0; i + j <= 2; ++j)
                                if (lead[i] + tail[j] >= Mi)
                                {
                                    const int32_t x = cg[g] + (i > 0 ? pa[0] : 0) + (i > 1 ? pa[1] : 0) + (j > 0 ? pt[0] : 0) + (j > 1 ? pt[1] : 0);
                                    c = x < c ? x : c;
                                }
                        if (c != 0x7FFFFFFF && (c < best_cost || (c == best_cost && d > best_end))) { best_cost = c; best_end = d; }
                    }
                    // pattern gap of g: diagonal d (prefix), then d - g (suffix); ends in column d - g
                    if (L2 + t2 + g >= Mi)
                    {
                        const int32_t pa[2] = { pen_of( L0 ), pen_of( L1 ) };
                        const int32_t pt[2] = { pen_of( Mi - 1 - t0 ), pen_of( Mi - 1 - t1 ) };
                        const int32_t lead[3] = { L0, L1, L2 }, tail[3] = { t0, t1, t2 };
                        int32_t c = 0x7FFFFFFF;
                        #pragma unroll
                        for (int i = 0; i <= 2; ++i)
                            #pragma unroll
                            for (int j = 0; i + j <= 2; ++j)
                                if (lead[i] + tail[j] + g >= Mi)
                                {
                                    const int32_t x = cg[g] + (i > 0 ? pa[0] : 0) + (i > 1 ? pa[1] : 0) + (j > 0 ? pt[0] : 0) + (j > 1 ? pt[1] : 0);
                                    c = x < c ? x : c;
                                }
                        const uint32_t end = d - (uint32_t)g;
                        if (c != 0x7FFFFFFF && (c < best_cost || (c == best_cost && end > best_end))) { best_cost = c; best_end = end; }
                    }
                }
        }
        // two gaps around the middle diagonal bm = d - 2 (history slot 1): neighbours bm - 2 (slot 3), bm - 1 (slot 2), bm + 1 (slot 0),
        // bm + 2 (this diagonal)
        if (d >= 2u)
        {
            const uint32_t bm = d - 2u;                            // <= 30
            const int32_t NEG = -(1 << 20), POS = 1 << 20;
            int32_t lo1 = NEG, lo2 = NEG, hi1 = POS, hi2 = POS;
            if (bm >= 1u)       { lo1 = (int32_t)(Lp[2] & 255u);    hi1 = Mi - (int32_t)(Tp[2] & 255u) - 1; }      // a = bm - 1: text gap in; c = bm - 1: pattern gap out
            if (bm + 1u <= 30u) { const int32_t x = (int32_t)(Lp[0] & 255u) + 1, y = Mi - (int32_t)(Tp[0] & 255u); lo1 = x > lo1 ? x : lo1; hi1 = y < hi1 ? y : hi1; }
            if (bm >= 2u)       { lo2 = (int32_t)(Lp[3] & 255u);    hi2 = Mi - (int32_t)(Tp[3] & 255u) - 2; }      // a / c = bm - 2
            if (have)           { const int32_t x = L0 + 2, y = Mi - T0; lo2 = x > lo2 ? x : lo2; hi2 = y < hi2 ? y : hi2; }   // a / c = bm + 2
            // the middle diagonal's third mismatch from either end (first / last if it has fewer): a row that is NOT clean
            const int32_t b2 = (int32_t)(Lp[1] >> 16), b1 = (int32_t)((Lp[1] >> 8) & 255u), b0 = (int32_t)(Lp[1] & 255u);
            const int32_t bf = b2 < Mi ? b2 : (b1 < Mi ? b1 : b0);
            const int32_t e2 = (int32_t)(Tp[1] >> 16), e1 = (int32_t)((Tp[1] >> 8) & 255u), e0 = (int32_t)(Tp[1] & 255u);
            const int32_t et = e2 < Mi ? e2 : (e1 < Mi ? e1 : e0);
            const int32_t bl = et < Mi ? Mi - 1 - et : NEG;
            if (lo1 > NEG && hi1 < POS) ex11 = ex11 || clean( bm, bf, bl, lo1, hi1 );
            if (lo1 > NEG && hi2 < POS) ex12 = ex12 || clean( bm, bf, bl, lo1, hi2 );
            if (lo2 > NEG && hi1 < POS) ex12 = ex12 || clean( bm, bf, bl, lo2, hi1 );
        }
        // shift the history
        #pragma unroll
        for (int k = GA - 1; k > 0; --k) { Lp[k] = Lp[k - 1]; Tp[k] = Tp[k - 1]; }
        Lp[0] = (uint32_t)L0 | ((uint32_t)L1 << 8) | ((uint32_t)L2 << 16);
        Tp[0] = (uint32_t)T0 | ((uint32_t)T1 << 8) | ((uint32_t)T2 << 16);
        hot = (hot << 1) | (hot_d ? 1u : 0u);
    }
    if (cu < best_cost || (cu == best_cost && cu_end > best_end)) { best_cost = cu; best_end = cu_end; }
    const bool settled = best_cost < c_unk && !(ex11 && cost11 <= best_cost) && !(ex12 && cost12 <= best_cost);
    if (settled) { scores[job] = -best_cost; sinks[job] = make_uint2( M + best_end, M ); need_dp[job] = 0; }
    else need_dp[job] = 1;
}

// host-side conditions of the shortcut: SEMI_GLOBAL, match = 0, one mismatch penalty for every quality,
// non-positive gap terms
static bool ungapped_ok(const SchemeDev& sc, const BatchDev& b, int32_t* P, bool* by_quality)
{
    if (sc.match != 0) return false;
    if (sc.mm_min < 0 || sc.mm_max < 0) return false;
    // the penalty depends on the row (nvBowtie's default: 2..6 by base quality): the first pass sums the candidates' rows exactly
    // (QUAL); it needs a ramp that does not decrease with the quality and a smallest penalty > 0
    *by_quality = b.quals != nullptr && sc.mm_min != sc.mm_max;
    if (*by_quality && (sc.mm_min <= 0 || sc.mm_max < sc.mm_min || (b.algo & NVBIO_ALN_NO_QUALITY_SHORTCUT))) return false;
    if (sc.pat_go >= 0 || sc.txt_go >= 0 || sc.pat_ge > 0 || sc.txt_ge > 0) return false;
    *P = sc.mm_min;                                               // quality 0 / constant ramp: mismatch = -mm_min
    return true;
}

// the packed kernel is exact iff no intermediate value can leave the int16 range or meet the -16384 stand-in
// for the reference's infimum: LOCAL additionally packs (score << 5 | column), so scores must fit 10 bits
static bool packed_ok(const int type, const SchemeDev& sc, const uint32_t max_read_len)
{
    if (max_read_len == 0) return false;
    const int lim = 4096;
    if (sc.mm_min < 0 || sc.mm_max < 0 || sc.mm_min > lim || sc.mm_max > lim) return false;
    if (sc.pat_go > 0 || sc.pat_ge > 0 || sc.pat_go < -lim || sc.pat_ge < -lim) return false;
    if (sc.match < 0) return false;
    if (type == NVBIO_LOCAL) return (uint64_t)sc.match * max_read_len <= 1000u;
    if (sc.txt_go > 0 || sc.txt_ge > 0 || sc.txt_go < -lim || sc.txt_ge < -lim) return false;
    // |score| <= (rows + band) * (largest single step) must stay far from -16384
    int64_t step = sc.match;
    const int c[] = { sc.mm_min, sc.mm_max, -sc.pat_go, -sc.pat_ge, -sc.txt_go, -sc.txt_ge };
    for (int v : c) if (v > step) step = v;
    return ((int64_t)max_read_len + 32) * step <= 8000;
}

struct IsTwo { __host__ __device__ __forceinline__ uint8_t operator()(const uint8_t v) const { return v == 2u ? 1u : 0u; } };
struct FlagIs { const uint8_t* flags; uint8_t code; __host__ __device__ __forceinline__ bool operator()(const uint32_t i) const { return flags[i] == code; } };
struct FlagIn { const uint8_t* flags; uint32_t mask; __host__ __device__ __forceinline__ bool operator()(const uint32_t i) const { return ((mask >> flags[i]) & 1u) != 0u; } };


// RAGGED batches: the DP's job list in ascending order of read length, so that the two alignments of a lane -- and the lanes of a wave --
// (nearly always) run the same number of rows: sort key = the job's read length, 0xFFFF behind the list's end (jobs = NULL: every job of
// the batch).  One or two radix passes over (uint16 key, uint32 job).
__global__ void __launch_bounds__(256)
job_length_keys_kernel(const BatchDev b, const uint32_t* __restrict__ jobs, const uint32_t* __restrict__ job_count, uint16_t* __restrict__ keys,
                       uint32_t* __restrict__ jobs_all, uint32_t* __restrict__ count_all)
{
    const uint32_t n = jobs ? *job_count : b.n;
    if (!jobs && blockIdx.x == 0 && threadIdx.x == 0) *count_all = b.n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += gridDim.x * blockDim.x)
    {
        uint32_t key = 0xFFFFu;
        if (i < n)
        {
            const uint32_t job = jobs ? jobs[i] : i;
            const uint32_t rid = b.read_id ? b.read_id[job] : job;
            const uint32_t M   = b.read_offsets[rid + 1] - b.read_offsets[rid];
            key = M < 0xFFFEu ? M : 0xFFFEu;
        }
        keys[i] = (uint16_t)key;
        if (!jobs) jobs_all[i] = i;
    }
}

// sorted list and its length (device) in *list_out / *count_out; the scratch is freed stream-ordered by the caller through *aux_out
static nvbio_status sort_jobs_by_length(const BatchDev& b, const uint32_t* job_list, const uint32_t* job_count, const uint32_t** list_out,
                                        const uint32_t** count_out, void** aux_out, hipStream_t s)
{
    const uint64_t kb = ((uint64_t)b.n * 2u + 255u) & ~255ull, lb = ((uint64_t)b.n * 4u + 255u) & ~255ull;
    size_t sort_bytes = 0;
    int bits = 1; while ((1u << bits) <= (b.max_read_len ? b.max_read_len : 0xFFFEu) && bits < 16) ++bits;
    bits = 16;                                   // (the 0xFFFF keys behind the list's end must sort last: all 16 bits)
    NVB_HIP( hipcub::DeviceRadixSort::SortPairs( nullptr, sort_bytes, (const uint16_t*)nullptr, (uint16_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                                 (int)b.n, 0, bits, s ) );
    uint8_t* aux = nullptr;
    if (scratch_alloc( (void**)&aux, 2u * kb + 2u * lb + 256u + sort_bytes, s ) != hipSuccess)
    {
        (void)hipGetLastError();
        set_error( "banded score: out of device memory for the length-sorted job list" );
        return NVBIO_ERR_NOMEM;
    }
    uint16_t* k_in = (uint16_t*)aux; uint16_t* k_out = (uint16_t*)(aux + kb);
    uint32_t* l_all = (uint32_t*)(aux + 2u * kb); uint32_t* l_out = (uint32_t*)(aux + 2u * kb + lb);
    uint32_t* c_all = (uint32_t*)(aux + 2u * kb + 2u * lb);
    void* tmp = aux + 2u * kb + 2u * lb + 256u;
    hipLaunchKernelGGL( job_length_keys_kernel, dim3( (b.n + 255u) / 256u < 65536u ? (b.n + 255u) / 256u : 65536u ), dim3( 256 ), 0, s, b, job_list, job_count, k_in, l_all, c_all );
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs( tmp, sort_bytes, (const uint16_t*)k_in, k_out, job_list ? job_list : (const uint32_t*)l_all, l_out, (int)b.n, 0, bits, s );
    if (e != hipSuccess) { scratch_free( aux, s ); set_error( "job sort failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    *list_out = l_out; *count_out = job_list ? job_count : c_all; *aux_out = aux;
    return NVBIO_OK;
}

// the packed kernel's instantiation for this scheme: match = 0 (every end-to-end scheme of nvBowtie) drops one operation per cell
template <int TYPE, int RB>
static void launch_pk_kernel(const BatchDev& b, const SchemeDev& sc, const uint32_t pairs, int32_t* scores, uint2* sinks,
                             const uint32_t* job_list, const uint32_t* job_count, hipStream_t s)
{
    const dim3 grid( (pairs + 127u) / 128u ), block( 128 );
    const bool two = (b.algo & NVBIO_ALN_PK_THREE_WAVES) == 0;
    // the binary16 build is exact while every score stays an integer of magnitude <= 2040 (and the scaled penalties finite)
    int64_t step = 0;
    { const int c[] = { sc.mm_min, sc.mm_max, -sc.pat_go, -sc.pat_ge, -sc.txt_go, -sc.txt_ge }; for (int v : c) if (v > step) step = v; }
    const bool fp = two && sc.match == 0 && TYPE == NVBIO_SEMI_GLOBAL && !(b.algo & NVBIO_ALN_NO_F16_DP) && ((int64_t)b.max_read_len + 32) * step <= 2040 && sc.mm_max <= 400;
    if (TYPE == NVBIO_SEMI_GLOBAL && sc.match == 0 && (b.algo & NVBIO_ALN_RAGGED_READS))
    {
        // reads of different lengths (the caller says so): both alignments of a lane in one pass
        constexpr int T = (TYPE == NVBIO_SEMI_GLOBAL) ? TYPE : NVBIO_SEMI_GLOBAL;
        if (fp) hipLaunchKernelGGL( (banded_gotoh_band31_pk_kernel<T,RB,2,true,true,true>), grid, block, 0, s, b, sc, scores, sinks, job_list, job_count );
        else    hipLaunchKernelGGL( (banded_gotoh_band31_pk_kernel<T,RB,2,true,true>), grid, block, 0, s, b, sc, scores, sinks, job_list, job_count );
        return;
    }
    if (TYPE != NVBIO_LOCAL && sc.match == 0)
    {
        constexpr int T = (TYPE == NVBIO_LOCAL) ? NVBIO_SEMI_GLOBAL : TYPE;      // (never LOCAL here: keeps that instantiation out)
        // (the binary16 build of the GLOBAL kernel spills: SEMI_GLOBAL only)
        if (fp)  hipLaunchKernelGGL( (banded_gotoh_band31_pk_kernel<NVBIO_SEMI_GLOBAL,RB,2,true,false,true>), grid, block, 0, s, b, sc, scores, sinks, job_list, job_count );
        else if (two) hipLaunchKernelGGL( (banded_gotoh_band31_pk_kernel<T,RB,2,true>), grid, block, 0, s, b, sc, scores, sinks, job_list, job_count );
        else     hipLaunchKernelGGL( (banded_gotoh_band31_pk_kernel<T,RB,3,true>), grid, block, 0, s, b, sc, scores, sinks, job_list, job_count );
    }
    else
    {
        if (two) hipLaunchKernelGGL( (banded_gotoh_band31_pk_kernel<TYPE,RB,2,false>), grid, block, 0, s, b, sc, scores, sinks, job_list, job_count );
        else     hipLaunchKernelGGL( (banded_gotoh_band31_pk_kernel<TYPE,RB,3,false>), grid, block, 0, s, b, sc, scores, sinks, job_list, job_count );
    }
}

template <int TYPE, int RB>
static nvbio_status launch_pk(const BatchDev& b, const SchemeDev& sc, int32_t* scores, uint2* sinks, hipStream_t s)
{
    const uint32_t pairs = (b.n + 1u) / 2u;
    int32_t P = 0; bool by_quality = false;
    // (txt_go / txt_ge never enter the banded recurrences -- both gap recurrences take the pattern-gap terms, gotoh_common.h:23-30 -- which is
    // why the chance kernels below are handed pat_go / pat_ge only; plain_gotoh() at the call site guarantees it.  A scheme that separates
    // the two must disable the shortcut, as gotoh_full.hip does with `second_chance`.)
    if (TYPE == NVBIO_SEMI_GLOBAL && ungapped_ok( sc, b, &P, &by_quality ) && !(b.algo & NVBIO_ALN_NO_UNGAPPED_SCORE))
    {
        // 1. settle the jobs whose best diagonal beats every gapped alignment; 2. compact the rest; 3. DP over the list
        const int32_t G = sc.pat_go > sc.txt_go ? sc.pat_go : sc.txt_go;
        size_t sel_bytes = 0;
        hipcub::CountingInputIterator<uint32_t> ids( 0u );
        NVB_HIP( hipcub::DeviceSelect::Flagged( nullptr, sel_bytes, ids, (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)b.n, s ) );
        const uint64_t flags_bytes = ((uint64_t)b.n + 255u) & ~255ull;
        const uint64_t list_bytes  = ((uint64_t)b.n * 4u + 255u) & ~255ull;
        const bool third = !(b.algo & NVBIO_ALN_NO_THIRD_CHANCE);
        void* aux = nullptr;
        // three-way partition of the job ids by flag (3: second chance, 2: third chance), the rest discarded
        size_t part_bytes = 0;
        hipcub::DiscardOutputIterator<uint32_t> nowhere;
        const FlagIs is3 = { nullptr, 3 }; const FlagIn is2 = { nullptr, 0u };
        NVB_HIP( hipcub::DevicePartition::If( nullptr, part_bytes, ids, (uint32_t*)nullptr, (uint32_t*)nullptr, nowhere, (uint32_t*)nullptr, (int)b.n, is3, is2, s ) );
        if (part_bytes > sel_bytes) sel_bytes = part_bytes;
        if (scratch_alloc( &aux, flags_bytes + 3u * list_bytes + 256u + sel_bytes, s ) != hipSuccess)
        {
            (void)hipGetLastError();
            set_error( "banded score: out of device memory for the job list" );
            return NVBIO_ERR_NOMEM;
        }
        uint8_t*  need_dp   = (uint8_t*)aux;
        uint32_t* job_list  = (uint32_t*)((uint8_t*)aux + flags_bytes);
        uint32_t* list_s    = (uint32_t*)((uint8_t*)aux + flags_bytes + list_bytes);          // second-chance jobs
        uint32_t* list_t    = (uint32_t*)((uint8_t*)aux + flags_bytes + 2u * list_bytes);     // third-chance jobs
        uint32_t* job_count = (uint32_t*)((uint8_t*)aux + flags_bytes + 3u * list_bytes);
        uint32_t* count_st  = job_count + 2;                                                  // [2]: second, third
        void*     sel_temp  = (uint8_t*)aux + flags_bytes + 3u * list_bytes + 256u;
        if (by_quality)
            hipLaunchKernelGGL( (ungapped_e2e31_kernel<RB,0,true>), dim3( (b.n + 255u) / 256u ), dim3( 256 ), 0, s, b, P, G, sc.pat_go, sc.pat_ge, scores, sinks, need_dp,
                                (const uint32_t*)nullptr, (const uint32_t*)nullptr, sc );
        else
            hipLaunchKernelGGL( (ungapped_e2e31_kernel<RB,0>), dim3( (b.n + 255u) / 256u ), dim3( 256 ), 0, s, b, P, G, sc.pat_go, sc.pat_ge, scores, sinks, need_dp,
                                (const uint32_t*)nullptr, (const uint32_t*)nullptr, sc );
        hipError_t e = hipSuccess;
        {
            // the jobs a chance can still settle, compacted by ONE three-way partition, each list through its own launch; every job ends as 0 or 1.
            //   list 1: need_dp == 3, the second chance.
            //   list 2: the gap chance's jobs -- need_dp == 4 (no diagonal in reach of the other chances: reads with an indel, mostly) and
            //           need_dp == 2 (third-chance jobs: the gap chance evaluates what the third chance only rules out, with the job's best diagonal as
            //           one more class) -- or, without the gap chance (qualities, NVBIO_ALN_NO_GAP_CHANCE), need_dp == 2 for the third chance.
            // (With NVBIO_ALN_NO_THIRD_CHANCE the third-chance jobs are simply handed to the DP: a non-zero flag selects;
            // NVBIO_ALN_NO_SECOND_CHANCE keeps the first pass from flagging any.)
            const bool gapc = !by_quality && !(b.algo & NVBIO_ALN_NO_GAP_CHANCE);
            const FlagIs f3 = { need_dp, 3 };
            const FlagIn f2 = { need_dp, gapc ? ((third ? 4u : 0u) | 16u) : 4u };
            size_t pb = sel_bytes;
            e = hipcub::DevicePartition::If( sel_temp, pb, ids, list_s, list_t, nowhere, count_st, (int)b.n, f3, f2, s );
            if (e == hipSuccess)
                hipLaunchKernelGGL( (ungapped_e2e31_kernel<RB,1>), dim3( (b.n + 255u) / 256u ), dim3( 256 ), 0, s, b, P, G, sc.pat_go, sc.pat_ge, scores, sinks, need_dp,
                                    (const uint32_t*)list_s, (const uint32_t*)count_st );
            if (e == hipSuccess && gapc)
                hipLaunchKernelGGL( (gap_chance_e2e31_kernel<RB>), dim3( (b.n + 255u) / 256u ), dim3( 256 ), 0, s, b, P, G, sc.pat_go, sc.pat_ge, scores, sinks, need_dp,
                                    (const uint32_t*)list_t, (const uint32_t*)(count_st + 1) );
            else if (e == hipSuccess && third)
                hipLaunchKernelGGL( (ungapped_e2e31_kernel<RB,2>), dim3( (b.n + 255u) / 256u ), dim3( 256 ), 0, s, b, P, G, sc.pat_go, sc.pat_ge, scores, sinks, need_dp,
                                    (const uint32_t*)list_t, (const uint32_t*)(count_st + 1) );
        }
        if (e == hipSuccess) e = hipcub::DeviceSelect::Flagged( sel_temp, sel_bytes, ids, need_dp, job_list, job_count, (int)b.n, s );
        if (e == hipSuccess)
        {
            const uint32_t* jl = job_list; const uint32_t* jc = job_count; void* aux2 = nullptr;
            if (TYPE == NVBIO_SEMI_GLOBAL && sc.match == 0 && (b.algo & NVBIO_ALN_RAGGED_READS) && !(b.algo & NVBIO_ALN_NO_LENGTH_SORT))
            {
                const nvbio_status st = sort_jobs_by_length( b, job_list, job_count, &jl, &jc, &aux2, s );
                if (st != NVBIO_OK) { scratch_free( aux, s ); return st; }
            }
            launch_pk_kernel<TYPE,RB>( b, sc, pairs, scores, sinks, jl, jc, s );
            if (aux2) scratch_free( aux2, s );
        }
        scratch_free( aux, s );
        if (e != hipSuccess) { set_error( "DeviceSelect failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
        NVB_HIP( hipGetLastError() );
        return NVBIO_OK;
    }
    if (TYPE == NVBIO_SEMI_GLOBAL && sc.match == 0 && (b.algo & NVBIO_ALN_RAGGED_READS) && !(b.algo & NVBIO_ALN_NO_LENGTH_SORT) && b.n > 1u)
    {
        const uint32_t* jl = nullptr; const uint32_t* jc = nullptr; void* aux2 = nullptr;
        NVB_CHECK( sort_jobs_by_length( b, nullptr, nullptr, &jl, &jc, &aux2, s ) );
        launch_pk_kernel<TYPE,RB>( b, sc, pairs, scores, sinks, jl, jc, s );
        scratch_free( aux2, s );
        NVB_HIP( hipGetLastError() );
        return NVBIO_OK;
    }
    launch_pk_kernel<TYPE,RB>( b, sc, pairs, scores, sinks, nullptr, nullptr, s );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

template <int BAND, int TYPE>
static nvbio_status launch_bits(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits,
                                int32_t* scores, uint2* sinks, hipStream_t s)
{
    if (BAND == 31 && plain_gotoh( sc ) && packed_ok( TYPE, sc, b.max_read_len ) && !(b.algo & NVBIO_ALN_NO_PACKED_DP))
    {
        if      (rbits == 4 && tbits == 2) return launch_pk<TYPE,4>( b, sc, scores, sinks, s );
        else if (rbits == 2 && tbits == 2) return launch_pk<TYPE,2>( b, sc, scores, sinks, s );
    }
    const dim3 grid( (b.n + 127u) / 128u ), block( 128 );
#define NVB_GO(RB, TB) hipLaunchKernelGGL( (banded_gotoh_kernel<BAND,TYPE,RB,TB>), grid, block, 0, s, b, sc, scores, sinks )
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

template <int BAND, int TYPE>
static nvbio_status launch_staged_bits(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits,
                                       const int32_t* min_scores, int32_t min_score, int32_t* scores, uint2* sinks, hipStream_t s)
{
    const dim3 grid( (b.n + 127u) / 128u ), block( 128 );
#define NVB_GO(RB, TB) hipLaunchKernelGGL( (banded_gotoh_kernel<BAND,TYPE,RB,TB,false,true>), grid, block, 0, s, b, sc, scores, sinks, \
                                           0u, (int32_t*)nullptr, (uint2*)nullptr, min_scores, min_score )
    if      (rbits == 4 && tbits == 2) NVB_GO(4, 2);
    else if (rbits == 2 && tbits == 2) NVB_GO(2, 2);
    else if (rbits == 8 && tbits == 8) NVB_GO(8, 8);
    else { set_error( "staged scoring: read_bits/text_bits %u/%u not instantiated (4/2, 2/2, 8/8)", rbits, tbits ); return NVBIO_ERR_UNSUPPORTED; }
#undef NVB_GO
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}
template <int BAND>
static nvbio_status launch_staged_type(int type, const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits,
                                       const int32_t* min_scores, int32_t min_score, int32_t* scores, uint2* sinks, hipStream_t s)
{
    switch (type)
    {
    case NVBIO_GLOBAL:      return launch_staged_bits<BAND,NVBIO_GLOBAL>     ( b, sc, rbits, tbits, min_scores, min_score, scores, sinks, s );
    case NVBIO_LOCAL:       return launch_staged_bits<BAND,NVBIO_LOCAL>      ( b, sc, rbits, tbits, min_scores, min_score, scores, sinks, s );
    case NVBIO_SEMI_GLOBAL: return launch_staged_bits<BAND,NVBIO_SEMI_GLOBAL>( b, sc, rbits, tbits, min_scores, min_score, scores, sinks, s );
    }
    set_error( "invalid alignment type %d", type );
    return NVBIO_ERR_INVALID;
}

template <int BAND>
static nvbio_status launch_type(int type, const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits,
                                int32_t* scores, uint2* sinks, hipStream_t s)
{
    switch (type)
    {
    case NVBIO_GLOBAL:      return launch_bits<BAND,NVBIO_GLOBAL>     ( b, sc, rbits, tbits, scores, sinks, s );
    case NVBIO_LOCAL:       return launch_bits<BAND,NVBIO_LOCAL>      ( b, sc, rbits, tbits, scores, sinks, s );
    case NVBIO_SEMI_GLOBAL: return launch_bits<BAND,NVBIO_SEMI_GLOBAL>( b, sc, rbits, tbits, scores, sinks, s );
    }
    set_error( "invalid alignment type %d", type );
    return NVBIO_ERR_INVALID;
}

// the packed band-31 end-to-end kernel over a job list, for the full-matrix scorer's narrow route (gotoh_full.hip): 4- or 2-bit reads in a
// 2-bit text, SEMI_GLOBAL; `max_jobs` bounds the list's length (which stays on the device)
bool banded31_packed_ok(const SchemeDev& sc, const uint32_t max_read_len)
{
    return plain_gotoh( sc ) && packed_ok( NVBIO_SEMI_GLOBAL, sc, max_read_len );
}
void banded31_packed_launch(const BatchDev& b, const SchemeDev& sc, const uint32_t read_bits, const uint32_t max_jobs, int32_t* scores, uint2* sinks,
                            const uint32_t* job_list, const uint32_t* job_count, hipStream_t s)
{
    const uint32_t pairs = (max_jobs + 1u) / 2u;
    if (read_bits == 4) launch_pk_kernel<NVBIO_SEMI_GLOBAL,4>( b, sc, pairs, scores, sinks, job_list, job_count, s );
    else                launch_pk_kernel<NVBIO_SEMI_GLOBAL,2>( b, sc, pairs, scores, sinks, job_list, job_count, s );
}

nvbio_status make_batch(const nvbio_alignment_batch* in, BatchDev* b)
{
    NVB_REQUIRE( in != nullptr, "batch is NULL" );
    NVB_REQUIRE( in->read_bits == 2 || in->read_bits == 4 || in->read_bits == 8, "read_bits must be 2, 4 or 8" );
    NVB_REQUIRE( in->text_bits == 2 || in->text_bits == 8, "text_bits must be 2 or 8" );
    NVB_REQUIRE( in->n < (1u << 31), "at most 2^31 - 1 jobs per batch (the job lists are compacted with 32-bit signed counts)" );
    if (in->n)
    {
        NVB_REQUIRE( in->reads_dev && in->read_offsets_dev && in->text_dev && in->win_begin_dev && in->win_end_dev,
                     "NULL device pointer in batch" );
    }
    b->reads = in->reads_dev; b->read_offsets = in->read_offsets_dev; b->quals = in->quals_dev;
    b->read_id = in->read_id_dev; b->flags = in->flags_dev; b->text = in->text_dev;
    b->win_begin = in->win_begin_dev; b->win_end = in->win_end_dev; b->n = in->n; b->max_read_len = in->max_read_len; b->algo = in->algo_flags;
    return NVBIO_OK;
}

} // namespace nvbio_amd

using namespace nvbio_amd;

template <int BAND, int TYPE>
static nvbio_status launch_best2_bits(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t dist,
                                      int32_t* scores, uint2* sinks, int32_t* scores2, uint2* sinks2, hipStream_t s)
{
    const dim3 grid( (b.n + 127u) / 128u ), block( 128 );
#define NVB_GO2(RB, TB) hipLaunchKernelGGL( (banded_gotoh_kernel<BAND,TYPE,RB,TB,true>), grid, block, 0, s, b, sc, scores, sinks, dist, scores2, sinks2 )
    if      (rbits == 4 && tbits == 2) NVB_GO2(4, 2);
    else if (rbits == 2 && tbits == 2) NVB_GO2(2, 2);
    else if (rbits == 8 && tbits == 2) NVB_GO2(8, 2);
    else if (rbits == 8 && tbits == 8) NVB_GO2(8, 8);
    else { set_error( "Best2Sink scoring: read_bits/text_bits %u/%u not instantiated (4/2, 2/2, 8/2, 8/8)", rbits, tbits ); return NVBIO_ERR_UNSUPPORTED; }
#undef NVB_GO2
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}
template <int BAND>
static nvbio_status launch_best2_type(int type, const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t dist,
                                      int32_t* scores, uint2* sinks, int32_t* scores2, uint2* sinks2, hipStream_t s)
{
    switch (type)
    {
    case NVBIO_GLOBAL:      return launch_best2_bits<BAND,NVBIO_GLOBAL>     ( b, sc, rbits, tbits, dist, scores, sinks, scores2, sinks2, s );
    case NVBIO_LOCAL:       return launch_best2_bits<BAND,NVBIO_LOCAL>      ( b, sc, rbits, tbits, dist, scores, sinks, scores2, sinks2, s );
    case NVBIO_SEMI_GLOBAL: return launch_best2_bits<BAND,NVBIO_SEMI_GLOBAL>( b, sc, rbits, tbits, dist, scores, sinks, scores2, sinks2, s );
    }
    set_error( "invalid alignment type %d", type );
    return NVBIO_ERR_INVALID;
}

extern "C" nvbio_status nvbio_banded_gotoh_score_best2(int device, uint32_t band, nvbio_alignment_type type,
                                                       const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch, uint32_t distinct_dist,
                                                       int32_t* scores_dev, nvbio_uint2* sinks_dev, int32_t* scores2_dev, nvbio_uint2* sinks2_dev,
                                                       void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (band != 3 && band != 7 && band != 15 && band != 31)
    {
        set_error( "band %u is not instantiated (3, 7, 15, 31)", band );
        return NVBIO_ERR_UNSUPPORTED;
    }
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev && scores2_dev && sinks2_dev, "NULL output pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    const SchemeDev sc = scheme_dev( scheme );
    hipStream_t s = (hipStream_t)stream;
    switch (band)
    {
    case 3:  return launch_best2_type<3> ( type, b, sc, batch->read_bits, batch->text_bits, distinct_dist, scores_dev, (uint2*)sinks_dev, scores2_dev, (uint2*)sinks2_dev, s );
    case 7:  return launch_best2_type<7> ( type, b, sc, batch->read_bits, batch->text_bits, distinct_dist, scores_dev, (uint2*)sinks_dev, scores2_dev, (uint2*)sinks2_dev, s );
    case 15: return launch_best2_type<15>( type, b, sc, batch->read_bits, batch->text_bits, distinct_dist, scores_dev, (uint2*)sinks_dev, scores2_dev, (uint2*)sinks2_dev, s );
    default: return launch_best2_type<31>( type, b, sc, batch->read_bits, batch->text_bits, distinct_dist, scores_dev, (uint2*)sinks_dev, scores2_dev, (uint2*)sinks2_dev, s );
    }
}

static nvbio_status banded_score(int device, uint32_t band, int type, const SchemeDev sc, const BatchDev& b, const nvbio_alignment_batch* batch,
                                 int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream)
{
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    switch (band)
    {
    case 3:  return launch_type<3> ( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    case 7:  return launch_type<7> ( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    case 15: return launch_type<15>( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    default: return launch_type<31>( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    }
}

extern "C" nvbio_status nvbio_banded_gotoh_score(int device, uint32_t band, nvbio_alignment_type type,
                                                 const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                                 int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (band != 3 && band != 7 && band != 15 && band != 31)
    {
        set_error( "band %u is not instantiated (3, 7, 15, 31)", band );
        return NVBIO_ERR_UNSUPPORTED;
    }
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev, "NULL output pointer" );
    return banded_score( device, band, type, scheme_dev( scheme ), b, batch, scores_dev, sinks_dev, stream );
}

extern "C" nvbio_status nvbio_banded_gotoh_score_staged(int device, uint32_t band, nvbio_alignment_type type,
                                                        const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                                        const int32_t* min_scores_dev, int32_t min_score,
                                                        int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (band != 3 && band != 7 && band != 15 && band != 31)
    {
        set_error( "band %u is not instantiated (3, 7, 15, 31)", band );
        return NVBIO_ERR_UNSUPPORTED;
    }
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev, "NULL output pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const SchemeDev sc = scheme_dev( scheme );
    switch (band)
    {
    case 3:  return launch_staged_type<3> ( type, b, sc, batch->read_bits, batch->text_bits, min_scores_dev, min_score, scores_dev, (uint2*)sinks_dev, s );
    case 7:  return launch_staged_type<7> ( type, b, sc, batch->read_bits, batch->text_bits, min_scores_dev, min_score, scores_dev, (uint2*)sinks_dev, s );
    case 15: return launch_staged_type<15>( type, b, sc, batch->read_bits, batch->text_bits, min_scores_dev, min_score, scores_dev, (uint2*)sinks_dev, s );
    default: return launch_staged_type<31>( type, b, sc, batch->read_bits, batch->text_bits, min_scores_dev, min_score, scores_dev, (uint2*)sinks_dev, s );
    }
}

extern "C" nvbio_status nvbio_banded_sw_score(int device, uint32_t band, nvbio_alignment_type type,
                                              const nvbio_sw_scheme* scheme, const nvbio_alignment_batch* batch,
                                              int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (band != 3 && band != 7 && band != 15 && band != 31)
    {
        set_error( "band %u is not instantiated (3, 7, 15, 31)", band );
        return NVBIO_ERR_UNSUPPORTED;
    }
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev, "NULL output pointer" );
    // in the band the boundary row runs over the text; with deletion == insertion this is Gotoh(open = extension) and takes
    // the packed kernel and the ungapped shortcut like any other Gotoh scheme
    SchemeDev sc = scheme_dev( scheme, false );
    sc.wide = 0;
    return banded_score( device, band, type, sc, b, batch, scores_dev, sinks_dev, stream );
}
