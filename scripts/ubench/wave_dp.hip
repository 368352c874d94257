// wave_dp.hip -- a measured prototype of the layout the north star prescribes for the extension DP: one wavefront per alignment
// (here two: a band of 31 columns fills half a wave64), lanes = band columns, the row recurrences through cross-lane moves.
// DESIGN.md 4.3 argues that this layout issues several times more wave instructions per cell than lane-per-alignment; this
// program measures it instead: band-31 LOCAL Gotoh scores (match 2, mismatch -6, gap open -8, extension -3) of n pairs
// (150 x 181, one symbol per byte, no packing, no sinks -- everything that is not the recurrence is left out, in the prototype's
// favour), checked against a plain per-thread DP of the same recurrence, and its rate in GCUPS (cells = 31 x 150 per pair, the
// reference's definition, alignment_test.cu:509).
//
// Row i, column j (text index i + j), as gotoh_banded_inl.h:473-602 evaluates it:
//   F[j] = max( F_prev[j+1] + ge, H_prev[j+1] + go )            <- one cross-lane move (lane j+1 -> j) of a packed (H, F) pair
//   D[j] = H_prev[j] + s( text[i+j], pattern[i] )               <- the text window slides one lane per row (another move)
//   G[j] = max( F[j], D[j], 0 )
//   E[j] = max over j' < j of ( G[j'] + go + (j - j' - 1) ge )  <- max-plus prefix over the band: log2(32) = 5 move + max steps
//   H[j] = max( G[j], E[j] )
// (E taken from G instead of H is exact when go <= ge: an E chain never improves by restarting from a cell it produced.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf( stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString( e_ ) ); exit( 1 ); } } while (0)

constexpr int BAND = 31, M = 150, N = 181;
constexpr int MATCH = 2, MISMATCH = -6, GO = -8, GE = -3, NEG = -100000;

__global__ void __launch_bounds__(256)
wave_dp_kernel(const uint8_t* __restrict__ pats, const uint8_t* __restrict__ txts, const uint32_t n, int32_t* __restrict__ scores)
{
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, col = lane & 31u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t job = 2u * wave + half;
    const bool live = job < n && col < (uint32_t)BAND;
    const uint8_t* P = pats + (size_t)(job < n ? job : 0u) * M;
    const uint8_t* T = txts + (size_t)(job < n ? job : 0u) * N;
    int32_t H = 0, F = NEG, best = 0;
    uint32_t t = T[col < (uint32_t)N ? col : 0];                      // text[0 + j]
    for (int i = 0; i < M; ++i)
    {
        const uint32_t p = P[i];                                      // the same address across the half-wave: one request
        // (H, F) of lane j+1, previous row
        const int32_t Hn = __shfl_down( H, 1, 32 ), Fn = __shfl_down( F, 1, 32 );
        const int32_t f  = col == (uint32_t)(BAND - 1) ? NEG : max( Fn + GE, Hn + GO );
        const int32_t d  = H + (t == p ? MATCH : MISMATCH);
        const int32_t g  = max( max( f, d ), 0 );
        // exclusive max-plus prefix: E[j] = max_{j'<j} ( g[j'] + GO + (j - j' - 1) GE ) = j GE + max_{j'<j} w[j'],  w[j'] = g[j'] + GO - (j'+1) GE
        int32_t w = g + GO - (int32_t)(col + 1u) * GE;
        #pragma unroll
        for (int s = 1; s < 32; s <<= 1)
        {
            const int32_t o = __shfl_up( w, s, 32 );
            if (col >= (uint32_t)s) w = max( w, o );
        }
        const int32_t wx = __shfl_up( w, 1, 32 );
        const int32_t e  = col == 0u ? NEG : wx + (int32_t)col * GE;
        H = max( g, e ); F = f;
        best = max( best, H );
        // slide the text window: lane j takes lane j+1's symbol, the last column loads the incoming one
        const uint32_t tn = __shfl_down( t, 1, 32 );
        t = col == (uint32_t)(BAND - 1) ? (i + BAND < N ? T[i + BAND] : 255u) : tn;
    }
    if (!live) best = 0;
    #pragma unroll
    for (int s = 16; s > 0; s >>= 1) best = max( best, __shfl_xor( best, s, 32 ) );
    if (col == 0u && job < n) scores[job] = best;
}

// the same recurrence, one thread per pair, straight loops: the check
__global__ void ref_dp_kernel(const uint8_t* __restrict__ pats, const uint8_t* __restrict__ txts, const uint32_t n, int32_t* __restrict__ scores)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= n) return;
    const uint8_t* P = pats + (size_t)job * M; const uint8_t* T = txts + (size_t)job * N;
    int32_t H[BAND], F[BAND], best = 0;
    for (int j = 0; j < BAND; ++j) { H[j] = 0; F[j] = NEG; }
    for (int i = 0; i < M; ++i)
    {
        int32_t e = NEG, hn[BAND];
        for (int j = 0; j < BAND; ++j)
        {
            const int32_t f = j == BAND - 1 ? NEG : max( F[j + 1] + GE, H[j + 1] + GO );
            const uint32_t t = i + j < N ? T[i + j] : 255u;
            const int32_t d = H[j] + (t == P[i] ? MATCH : MISMATCH);
            const int32_t h = max( max( max( f, d ), e ), 0 );
            hn[j] = h; F[j] = f;
            e = max( e + GE, h + GO );
            best = max( best, h );
        }
        for (int j = 0; j < BAND; ++j) H[j] = hn[j];
    }
    scores[job] = best;
}

int main(int argc, char** argv)
{
    const uint32_t n = 1u << 21;                                       // 2 M pairs
    std::vector<uint8_t> hp( (size_t)n * M ), ht( (size_t)n * N );
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (uint32_t k = 0; k < n; ++k)
    {
        for (int j = 0; j < N; ++j) ht[(size_t)k * N + j] = rnd() & 3;
        for (int i = 0; i < M; ++i) hp[(size_t)k * M + i] = (rnd() % 50) ? ht[(size_t)k * N + 15 + i] : (uint8_t)(rnd() & 3);   // true diagonal 15, 2 % errors
        if (k % 7 == 0) { const int at = 40 + (int)(rnd() % 60); for (int i = M - 1; i > at; --i) hp[(size_t)k * M + i] = hp[(size_t)k * M + i - 1]; }   // an insertion
    }
    uint8_t *dp, *dt; int32_t *d1, *d2;
    CHECK( hipMalloc( (void**)&dp, hp.size() ) ); CHECK( hipMalloc( (void**)&dt, ht.size() ) );
    CHECK( hipMalloc( (void**)&d1, n * 4 ) ); CHECK( hipMalloc( (void**)&d2, n * 4 ) );
    CHECK( hipMemcpy( dp, hp.data(), hp.size(), hipMemcpyHostToDevice ) ); CHECK( hipMemcpy( dt, ht.data(), ht.size(), hipMemcpyHostToDevice ) );
    hipEvent_t a, b; CHECK( hipEventCreate( &a ) ); CHECK( hipEventCreate( &b ) );
    const uint32_t waves = (n + 1) / 2, blocks = (waves * 64 + 255) / 256;
    float best_ms = 1e30f;
    for (int rep = 0; rep < 4; ++rep)
    {
        CHECK( hipEventRecord( a, 0 ) );
        hipLaunchKernelGGL( wave_dp_kernel, dim3( blocks ), dim3( 256 ), 0, 0, dp, dt, n, d1 );
        CHECK( hipEventRecord( b, 0 ) ); CHECK( hipEventSynchronize( b ) );
        float ms; CHECK( hipEventElapsedTime( &ms, a, b ) );
        if (rep && ms < best_ms) best_ms = ms;
    }
    hipLaunchKernelGGL( ref_dp_kernel, dim3( (n + 127) / 128 ), dim3( 128 ), 0, 0, dp, dt, n, d2 );
    CHECK( hipDeviceSynchronize() );
    std::vector<int32_t> s1( n ), s2( n );
    CHECK( hipMemcpy( s1.data(), d1, n * 4, hipMemcpyDeviceToHost ) ); CHECK( hipMemcpy( s2.data(), d2, n * 4, hipMemcpyDeviceToHost ) );
    uint32_t bad = 0; for (uint32_t k = 0; k < n; ++k) bad += s1[k] != s2[k];
    const double gcups = (double)n * BAND * M / (best_ms * 1e-3) / 1e9;
    FILE* out = argc > 1 ? fopen( argv[1], "w" ) : stdout;
    fprintf( out, "{\"prototype\": \"wave-per-alignment band-31 LOCAL Gotoh, lanes = band columns, 2 alignments per wave64, cross-lane recurrences\", "
                  "\"pairs\": %u, \"pattern_len\": %d, \"text_len\": %d, \"ms\": %.4f, \"gcups\": %.1f, \"mismatching_scores_vs_per_thread_dp\": %u}\n",
             n, M, N, best_ms, gcups, bad );
    if (out != stdout) fclose( out );
    fprintf( stderr, "wave-per-alignment prototype: %.3f ms for %u pairs = %.1f GCUPS; %u scores differ from the per-thread DP\n", best_ms, n, gcups, bad );
    return bad ? 1 : 0;
}
