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
#include <stdlib.h>

namespace nvbio_amd {

template <int BAND, int TYPE, int RBITS, int TBITS>
__global__ void __launch_bounds__(128)
banded_gotoh_kernel(const BatchDev b, const SchemeDev sc, int32_t* __restrict__ scores, uint2* __restrict__ sinks)
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
    const uint32_t N     = b.win_end[job] - tb;

    int32_t  best   = NVBIO_SCORE_MIN;
    uint32_t best_x = 0xFFFFFFFFu, best_y = 0xFFFFFFFFu;

    if (N < M)                                                   // gotoh_banded_inl.h:422-423: nothing reported
    {
        scores[job] = best; sinks[job] = make_uint2( best_x, best_y );
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

    const int32_t G_o = sc.pat_go, G_e = sc.pat_ge;
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
                row_key = max2( row_key, (h << 5) | j );
            }
            H[j] = h;
            E = (j == 0) ? h + G_o : max2( h + G_o, E + G_e );   // :507,562-565
        }

        // shift the cache by one column and append the new symbol (:532,570)
        if (PACKED) cache_bits = (cache_bits >> 2) | ((uint64_t)(g_new & 3u) << (2 * (BAND - 2)));
        else
        {
            #pragma unroll
            for (int j = 0; j < BAND - 2; ++j) cache_raw[j] = cache_raw[j + 1];
            cache_raw[BAND - 2] = g_new;
        }

        if (TYPE == NVBIO_LOCAL)
        {
            // cells are reported row-major with j ascending and BestSink keeps the LAST maximum
            const int32_t h = row_key >> 5;
            if (h >= best) { best = h; best_x = i + (uint32_t)(row_key & 31) + 1u; best_y = i + 1u; }
        }
    }

    if (TYPE == NVBIO_GLOBAL)                                    // :629-630
    {
        if (best <= H[BAND - 1]) { best = H[BAND - 1]; best_x = M + BAND - 1; best_y = M; }
    }
    else if (TYPE == NVBIO_SEMI_GLOBAL)                          // :631-643
    {
        const uint32_t mb = M + (uint32_t)(BAND - 1);
        const uint32_t m  = (mb < N ? mb : N) - (M - 1u);
        #pragma unroll
        for (int j = 0; j < BAND; ++j)
            if (j == 0 || (uint32_t)j < m)
                if (best <= H[j]) { best = H[j]; best_x = M + j; best_y = M; }
    }
    scores[job] = best;
    sinks[job]  = make_uint2( best_x, best_y );
}

// ---------------------------------------------------------------------------------------------
// 16-bit packed variant for the production case (band 31, LOCAL): one lane owns TWO alignments,
// one in each half of every register, so H[31]/F[31] of both take the registers one alignment
// took before and every add / max is a v_pk_*_i16 doing two cells.  Exactness conditions,
// checked on the host (else the int32 kernel runs): all LOCAL scores fit 10 bits
// (match * max_read_len <= 1000, so that (score << 5 | column) fits an int16) and penalties are
// < 4096, so that the -16384 stand-in for the reference's infimum can never win a max against a
// real score nor wrap.  The row-0 / column-30 infimum cells behave exactly as in the int32 kernel.
// ---------------------------------------------------------------------------------------------
typedef short    v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s pk(const int a, const int b) { v2s r; r.x = (short)a; r.y = (short)b; return r; }
__device__ __forceinline__ v2s pk_max(const v2s a, const v2s b) { return __builtin_elementwise_max( a, b ); }
__device__ __forceinline__ v2s pk_from_bits(const uint32_t u) { return __builtin_bit_cast( v2s, u ); }

template <int RBITS, int TBITS>
__global__ void __launch_bounds__(128)
banded_gotoh_local31_pk_kernel(const BatchDev b, const SchemeDev sc, int32_t* __restrict__ scores, uint2* __restrict__ sinks)
{
    constexpr int BAND = 31;
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (2u * pair >= b.n) return;

    uint32_t first[2], M[2], tb[2], N[2];
    bool     rev[2], comp[2], valid[2];
    #pragma unroll
    for (int u = 0; u < 2; ++u)
    {
        const uint32_t job = 2u * pair + u;
        valid[u] = job < b.n;
        const uint32_t jj  = valid[u] ? job : 2u * pair;
        const uint32_t rid = b.read_id ? b.read_id[jj] : jj;
        first[u] = b.read_offsets[rid];
        M[u]     = b.read_offsets[rid + 1] - first[u];
        const uint32_t fl = b.flags ? b.flags[jj] : 0u;
        rev[u]  = (fl & NVBIO_READ_REVERSE) != 0;
        comp[u] = (fl & NVBIO_READ_COMPLEMENT) != 0;
        tb[u]   = b.win_begin[jj];
        N[u]    = b.win_end[jj] - tb[u];
    }
    // rows each half really computes: none when the text is shorter than the pattern (nothing reported)
    const uint32_t rows0 = (valid[0] && N[0] >= M[0]) ? M[0] : 0u;
    const uint32_t rows1 = (valid[1] && N[1] >= M[1]) ? M[1] : 0u;
    const uint32_t rows  = rows0 > rows1 ? rows0 : rows1;

    SymbolReader<TBITS> trd0( b.text ), trd1( b.text );
    SymbolReader<RBITS> prd0( b.reads ), prd1( b.reads );

    uint64_t cache0 = 0, cache1 = 0;                             // 30 cached text symbols per alignment
    #pragma unroll
    for (int j = 0; j < BAND - 1; ++j)
    {
        const uint32_t g0 = ((uint32_t)j < N[0]) ? trd0.get( tb[0] + j ) : 255u;
        const uint32_t g1 = ((uint32_t)j < N[1]) ? trd1.get( tb[1] + j ) : 255u;
        cache0 |= (uint64_t)(g0 & 3u) << (2 * j);
        cache1 |= (uint64_t)(g1 & 3u) << (2 * j);
    }

    const v2s GO = pk( sc.pat_go, sc.pat_go ), GE = pk( sc.pat_ge, sc.pat_ge );
    const v2s INF = pk( -16384, -16384 ), ZERO = pk( 0, 0 ), K32 = pk( 32, 32 );
    const int V = sc.match;

    v2s H[BAND], F[BAND];
    #pragma unroll
    for (int j = 0; j < BAND; ++j) { H[j] = ZERO; F[j] = INF; }

    int32_t  best[2]   = { NVBIO_SCORE_MIN, NVBIO_SCORE_MIN };
    uint32_t best_x[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu }, best_y[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu };

    for (uint32_t i = 0; i < rows; ++i)
    {
        // the row's pattern symbols and mismatch scores
        uint32_t q0 = 255u, q1 = 255u; int S0 = 0, S1 = 0;
        if (i < rows0)
        {
            const uint32_t pidx = rev[0] ? first[0] + M[0] - 1u - i : first[0] + i;
            q0 = prd0.get( pidx ); if (comp[0] && q0 < 4u) q0 = 3u - q0;
            const uint32_t qq = b.quals ? b.quals[pidx] : 0u; S0 = s_mm[qq < 63u ? qq : 63u];
        }
        if (i < rows1)
        {
            const uint32_t pidx = rev[1] ? first[1] + M[1] - 1u - i : first[1] + i;
            q1 = prd1.get( pidx ); if (comp[1] && q1 < 4u) q1 = 3u - q1;
            const uint32_t qq = b.quals ? b.quals[pidx] : 0u; S1 = s_mm[qq < 63u ? qq : 63u];
        }
        const uint32_t gn0 = (i + (uint32_t)(BAND - 1) < N[0]) ? trd0.get( tb[0] + i + (BAND - 1) ) : 255u;
        const uint32_t gn1 = (i + (uint32_t)(BAND - 1) < N[1]) ? trd1.get( tb[1] + i + (BAND - 1) ) : 255u;

        // match flags of the 30 cached columns: alignment 0 at bit 2j, alignment 1 at bit 2j+1
        uint64_t e0 = 0, e1 = 0;
        if (q0 < 4u) { const uint64_t t = cache0 ^ ((uint64_t)q0 * 0x5555555555555555ull); e0 = ~(t | (t >> 1)) & 0x5555555555555555ull; }
        if (q1 < 4u) { const uint64_t t = cache1 ^ ((uint64_t)q1 * 0x5555555555555555ull); e1 = ~(t | (t >> 1)) & 0x5555555555555555ull; }
        const uint64_t EQ = e0 | (e1 << 1);

        const v2s SS = pk( S0, S1 );                              // mismatch scores of the two rows
        const v2s DV = pk( V - S0, V - S1 );                      // match - mismatch

        v2s E = ZERO;
        v2s key = pk( -1, -1 );
        #pragma unroll
        for (int j = 0; j < BAND; ++j)
        {
            const v2s f = (j < BAND - 1) ? pk_max( F[j + 1] + GE, H[j + 1] + GO ) : INF;
            F[j] = f;

            uint32_t eq01;
            if (j == BAND - 1) eq01 = (gn0 == q0 ? 1u : 0u) | (gn1 == q1 ? 0x10000u : 0u);
            else               eq01 = (uint32_t)((EQ >> (2 * j)) & 1ull) | ((uint32_t)((EQ >> (2 * j + 1)) & 1ull) << 16);
            const v2s d = H[j] + SS + pk_from_bits( eq01 ) * DV;

            v2s h;
            if (j == 0)             h = pk_max( f, d );
            else if (j == BAND - 1) h = pk_max( E, d );
            else                    h = pk_max( pk_max( f, E ), d );
            h = pk_max( h, ZERO );
            key = pk_max( key, h * K32 + pk( j, j ) );
            H[j] = h;
            E = (j == 0) ? h + GO : pk_max( h + GO, E + GE );
        }

        cache0 = (cache0 >> 2) | ((uint64_t)(gn0 & 3u) << (2 * (BAND - 2)));
        cache1 = (cache1 >> 2) | ((uint64_t)(gn1 & 3u) << (2 * (BAND - 2)));

        // BestSink: row-major reports, the LAST maximum wins
        const int k0 = key.x, k1 = key.y;
        if (i < rows0 && (k0 >> 5) >= best[0]) { best[0] = k0 >> 5; best_x[0] = i + (uint32_t)(k0 & 31) + 1u; best_y[0] = i + 1u; }
        if (i < rows1 && (k1 >> 5) >= best[1]) { best[1] = k1 >> 5; best_x[1] = i + (uint32_t)(k1 & 31) + 1u; best_y[1] = i + 1u; }
    }
    #pragma unroll
    for (int u = 0; u < 2; ++u)
        if (valid[u]) { scores[2u * pair + u] = best[u]; sinks[2u * pair + u] = make_uint2( best_x[u], best_y[u] ); }
}

// the packed kernel is exact iff every LOCAL score fits 10 bits and penalties are small
static bool packed_local_ok(const SchemeDev& sc, const uint32_t max_read_len)
{
    if (max_read_len == 0) return false;
    if (sc.match < 0 || (uint64_t)sc.match * max_read_len > 1000u) return false;
    const int lim = 4096;
    if (sc.mm_min < 0 || sc.mm_max < 0 || sc.mm_min > lim || sc.mm_max > lim) return false;
    if (sc.pat_go > 0 || sc.pat_ge > 0 || sc.pat_go < -lim || sc.pat_ge < -lim) return false;
    return true;
}

template <int RB, int TBITS_>
static void launch_pk(const BatchDev& b, const SchemeDev& sc, int32_t* scores, uint2* sinks, hipStream_t s)
{
    const uint32_t pairs = (b.n + 1u) / 2u;
    hipLaunchKernelGGL( (banded_gotoh_local31_pk_kernel<RB,TBITS_>), dim3( (pairs + 127u) / 128u ), dim3( 128 ), 0, s, b, sc, scores, sinks );
}

template <int BAND, int TYPE>
static nvbio_status launch_bits(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits,
                                int32_t* scores, uint2* sinks, hipStream_t s)
{
    if (BAND == 31 && TYPE == NVBIO_LOCAL && packed_local_ok( sc, b.max_read_len ) && !getenv( "NVBIO_AMD_NO_PACKED_DP" ))
    {
        if      (rbits == 4 && tbits == 2) { launch_pk<4,2>( b, sc, scores, sinks, s ); NVB_HIP( hipGetLastError() ); return NVBIO_OK; }
        else if (rbits == 2 && tbits == 2) { launch_pk<2,2>( b, sc, scores, sinks, s ); NVB_HIP( hipGetLastError() ); return NVBIO_OK; }
        else if (rbits == 8 && tbits == 8) { launch_pk<8,8>( b, sc, scores, sinks, s ); NVB_HIP( hipGetLastError() ); return NVBIO_OK; }
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

nvbio_status make_batch(const nvbio_alignment_batch* in, BatchDev* b)
{
    NVB_REQUIRE( in != nullptr, "batch is NULL" );
    NVB_REQUIRE( in->read_bits == 2 || in->read_bits == 4 || in->read_bits == 8, "read_bits must be 2, 4 or 8" );
    NVB_REQUIRE( in->text_bits == 2 || in->text_bits == 8, "text_bits must be 2 or 8" );
    if (in->n)
    {
        NVB_REQUIRE( in->reads_dev && in->read_offsets_dev && in->text_dev && in->win_begin_dev && in->win_end_dev,
                     "NULL device pointer in batch" );
    }
    b->reads = in->reads_dev; b->read_offsets = in->read_offsets_dev; b->quals = in->quals_dev;
    b->read_id = in->read_id_dev; b->flags = in->flags_dev; b->text = in->text_dev;
    b->win_begin = in->win_begin_dev; b->win_end = in->win_end_dev; b->n = in->n; b->max_read_len = in->max_read_len;
    return NVBIO_OK;
}

} // namespace nvbio_amd

using namespace nvbio_amd;

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
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    SchemeDev sc = { scheme->match, scheme->mm_min, scheme->mm_max, scheme->pat_gap_open, scheme->pat_gap_ext,
                     scheme->txt_gap_open, scheme->txt_gap_ext };
    hipStream_t s = (hipStream_t)stream;
    switch (band)
    {
    case 3:  return launch_type<3> ( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    case 7:  return launch_type<7> ( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    case 15: return launch_type<15>( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    default: return launch_type<31>( type, b, sc, batch->read_bits, batch->text_bits, scores_dev, (uint2*)sinks_dev, s );
    }
}
