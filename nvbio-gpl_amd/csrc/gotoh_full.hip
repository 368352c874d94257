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

namespace nvbio_amd {

namespace {

struct Sink
{
    int32_t score; uint32_t x, y;
    __device__ __forceinline__ void init() { score = NVBIO_SCORE_MIN; x = y = 0xFFFFFFFFu; }
    __device__ __forceinline__ void report(const int32_t s, const uint32_t sx, const uint32_t sy)
    {
        if (score <= s) { score = s; x = sx; y = sy; }          // last maximum wins
    }
};

// (H,E) boundary cell: two int16, truncating like make_vector<short>(H,E) (gotoh_inl.h:942)
__device__ __forceinline__ uint32_t pack_cell(const int32_t h, const int32_t e) { return ((uint32_t)h & 0xFFFFu) | ((uint32_t)e << 16); }
__device__ __forceinline__ int32_t  cell_h(const uint32_t c) { return (int32_t)(int16_t)(c & 0xFFFFu); }
__device__ __forceinline__ int32_t  cell_e(const uint32_t c) { return (int32_t)(int16_t)(c >> 16); }

constexpr int STRIPE = 8;

// TEXT_BLOCKING: stripes run over the text, the boundary column over the M pattern rows
// otherwise     : stripes run over the pattern, the boundary column over the N text rows
template <int TYPE, bool TEXT_BLOCKING, int RBITS, int TBITS>
__global__ void __launch_bounds__(128)
full_gotoh_kernel(const BatchDev b, const SchemeDev sc, const uint32_t job_begin, const uint32_t jobs, const int32_t* __restrict__ min_scores,
                  uint32_t* __restrict__ column, int32_t* __restrict__ scores, uint2* __restrict__ sinks)
{
    __shared__ int32_t s_mm[64];
    if (threadIdx.x < 64) s_mm[threadIdx.x] = mismatch_score( sc, threadIdx.x );
    __syncthreads();

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // slot inside this launch
    if (t >= jobs) return;
    const uint32_t job = job_begin + t;

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

    const int32_t G_o = sc.pat_go, G_e = sc.pat_ge;
    const int32_t infimum = -32768 - (G_o < G_e ? G_o : G_e);   // gotoh_inl.h:634,1038
    const int32_t V = sc.match;

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

    Sink sink; sink.init();
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
            H[j] = penal ? ((block + j > 0) ? G_o + G_e * (int32_t)(block + j - 1u) : 0) : 0;
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
                F[j] = max2( F[j] + G_e, H[j] + G_o );
                E    = max2( E + G_e, H[j - 1] + G_o );
                const int32_t S = (c_sym[j - 1] == r_sym) ? V : (TEXT_BLOCKING ? r_mm : c_mm[j - 1]);
                int32_t hi = max3( E, F[j], H_diag + S );
                if (TYPE == NVBIO_LOCAL) hi = max2( hi, 0 );
                H_diag = H[j];
                H[j]   = hi;
            }
            col[(size_t)i * jobs] = pack_cell( H[STRIPE], E );
            max_score = max2( max_score, H[STRIPE] );

            if (TYPE == NVBIO_LOCAL)
            {
                #pragma unroll
                for (int j = 1; j <= STRIPE; ++j)
                    if (!last || block + j <= cols)
                    {
                        if (TEXT_BLOCKING) sink.report( H[j], block + j, i + 1u );
                        else               sink.report( H[j], i + 1u, block + j );
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
            if (max_score + missing * V < min_score) ok = false;                     // stripe early exit
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
}

template <int TYPE, bool TB>
nvbio_status launch_bits(const BatchDev& b, const SchemeDev& sc, uint32_t rbits, uint32_t tbits, uint32_t job_begin, uint32_t jobs,
                         const int32_t* min_scores, uint32_t* column, int32_t* scores, uint2* sinks, hipStream_t s)
{
    const dim3 grid( (jobs + 127u) / 128u ), block( 128 );
#define NVB_GO(RB, TBITS) hipLaunchKernelGGL( (full_gotoh_kernel<TYPE,TB,RB,TBITS>), grid, block, 0, s, b, sc, job_begin, jobs, min_scores, column, scores, sinks )
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
                         const int32_t* min_scores, uint32_t* column, int32_t* scores, uint2* sinks, hipStream_t s)
{
    switch (type)
    {
    case NVBIO_GLOBAL:      return launch_bits<NVBIO_GLOBAL,TB>     ( b, sc, rbits, tbits, job_begin, jobs, min_scores, column, scores, sinks, s );
    case NVBIO_LOCAL:       return launch_bits<NVBIO_LOCAL,TB>      ( b, sc, rbits, tbits, job_begin, jobs, min_scores, column, scores, sinks, s );
    case NVBIO_SEMI_GLOBAL: return launch_bits<NVBIO_SEMI_GLOBAL,TB>( b, sc, rbits, tbits, job_begin, jobs, min_scores, column, scores, sinks, s );
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

extern "C" nvbio_status nvbio_full_gotoh_score(int device, nvbio_alignment_type type, int text_blocking,
                                               const nvbio_gotoh_scheme* scheme, const nvbio_alignment_batch* batch,
                                               uint32_t max_pattern_len, uint32_t max_text_len,
                                               const int32_t* min_scores_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev,
                                               void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( scheme != nullptr, "scheme is NULL" );
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev, "NULL output pointer" );
    const uint64_t rows = text_blocking ? max_pattern_len : max_text_len;
    NVB_REQUIRE( rows > 0, "max_pattern_len / max_text_len must bound the boundary column" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    SchemeDev sc = { scheme->match, scheme->mm_min, scheme->mm_max, scheme->pat_gap_open, scheme->pat_gap_ext,
                     scheme->txt_gap_open, scheme->txt_gap_ext };

    // boundary columns: caller scratch if given, else stream-ordered scratch; jobs are processed in
    // as many launches as the scratch allows (at least one wave of jobs per launch)
    void*    owned = nullptr;
    uint32_t* column = (uint32_t*)temp_dev;
    uint64_t  cap_jobs;
    if (column)
    {
        cap_jobs = temp_bytes / (rows * sizeof(uint32_t));
        NVB_REQUIRE( cap_jobs >= 64 || cap_jobs >= b.n, "temp_bytes too small (see nvbio_full_gotoh_temp_bytes)" );
    }
    else
    {
        cap_jobs = b.n;
        const uint64_t budget = 8ull << 30;                      // at most 8 GiB of scratch per launch
        if (cap_jobs * rows * sizeof(uint32_t) > budget) cap_jobs = budget / (rows * sizeof(uint32_t));
        if (cap_jobs < 64) cap_jobs = 64;
        if (hipMallocAsync( &owned, cap_jobs * rows * sizeof(uint32_t), s ) != hipSuccess)
        {
            set_error( "full Gotoh: out of device memory for %llu boundary columns", (unsigned long long)cap_jobs );
            return NVBIO_ERR_NOMEM;
        }
        column = (uint32_t*)owned;
    }
    nvbio_status st = NVBIO_OK;
    for (uint64_t begin = 0; begin < b.n && st == NVBIO_OK; begin += cap_jobs)
    {
        const uint32_t jobs = (uint32_t)((b.n - begin) < cap_jobs ? (b.n - begin) : cap_jobs);
        st = text_blocking ?
            launch_type<true> ( type, b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, min_scores_dev, column, scores_dev, (uint2*)sinks_dev, s ) :
            launch_type<false>( type, b, sc, batch->read_bits, batch->text_bits, (uint32_t)begin, jobs, min_scores_dev, column, scores_dev, (uint2*)sinks_dev, s );
    }
    if (owned) (void)hipFreeAsync( owned, s );
    return st;
}
