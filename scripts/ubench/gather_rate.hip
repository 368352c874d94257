// gather_rate.hip -- what the MI355X memory system sustains for random gathers, by footprint and access shape.
//
// The seed pass (csrc/fm_index.hip) is a short chain of dependent random gathers: one 8-byte k-mer table entry
// (128 GiB table at k = 17), then a text word pair (0.75 GB) and sometimes an SA word (12 GB) / bwt_occ records (1.5 GB).
// Every gather moves one 64-byte sector over the fabric whatever its width, so the honest ceiling of such a pass is the
// number of random SECTORS per second the chip delivers -- a function of the footprint (address-translation reach) and of
// how many requests are in flight.  This program measures it, so that "% of achievable" in DESIGN.md is a measured ratio.
//
//   ./gather_rate [out.jsonl]
//
// One JSON line per configuration: footprint, element bytes, window (locality of consecutive lanes' addresses: the whole
// table = uniform random; W bytes = what a partition of the requests by the high address bits would give), gathers in
// flight per lane, temporal hint, chain length (dependent gathers), and the rate in G gathers/s (= G sectors/s, every
// gather touching its own sector) with the sector traffic it implies (x 64 B).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define Q_LOG 27
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf( stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString( e_ ) ); exit( 1 ); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// element index of request i: the table is cut into n_win windows of win_elems elements; request i goes to window
// floor(i * n_win / Q) (requests in window order, as after a partition by high address bits), at a random place inside
template <typename T, int INFLIGHT, bool NT, int CHAIN>
__global__ void __launch_bounds__(256)
gather_kernel(const T* __restrict__ table, const uint64_t n_elems, const uint64_t win_elems, const uint64_t n_win, const uint64_t Q,
              const uint64_t salt, uint64_t* __restrict__ sink)
{
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256u * INFLIGHT;
    for (uint64_t base = (uint64_t)blockIdx.x * 256u * INFLIGHT; base < Q; base += stride)
    {
        uint64_t idx[INFLIGHT];
        #pragma unroll
        for (int g = 0; g < INFLIGHT; ++g)
        {
            const uint64_t i = base + (uint64_t)g * 256u + threadIdx.x;
            const uint64_t w = __umul64hi( i << (64 - Q_LOG), n_win );              // floor(i * n_win / Q), Q = 2^Q_LOG
            idx[g] = w * win_elems + __umul64hi( mix( i ^ salt ), win_elems );
            if (idx[g] >= n_elems) idx[g] = n_elems - 1u;
        }
        #pragma unroll
        for (int c = 0; c < CHAIN; ++c)
        {
            uint64_t v[INFLIGHT];
            #pragma unroll
            for (int g = 0; g < INFLIGHT; ++g)
            {
                const T* p = table + idx[g];
                if (sizeof(T) == 4)       v[g] = NT ? __builtin_nontemporal_load( (const uint32_t*)p ) : *(const uint32_t*)p;
                else if (sizeof(T) == 8)  v[g] = NT ? __builtin_nontemporal_load( (const uint64_t*)p ) : *(const uint64_t*)p;
                else
                {
                    // 32-byte record (two 16-byte vectors, as a bwt_occ record)
                    const uint4 a = ((const uint4*)p)[0], b = ((const uint4*)p)[1];
                    v[g] = (uint64_t)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
                }
            }
            #pragma unroll
            for (int g = 0; g < INFLIGHT; ++g)
            {
                acc += v[g];
                // next link of the chain: an address that depends on the loaded value (the table holds zeros)
                const uint64_t i = base + (uint64_t)g * 256u + threadIdx.x;
                idx[g] = __umul64hi( mix( (i + v[g]) ^ (salt + 77u * (c + 1)) ), n_elems );
            }
        }
    }
    if (acc == 0x123456789ull) sink[0] = acc;              // never true: keeps the loads alive
}

struct Rec32 { uint4 a, b; };

template <typename T, int INFLIGHT, bool NT, int CHAIN>
static double run(const void* table, uint64_t bytes, uint64_t win_bytes, uint64_t Q, uint64_t* sink, int blocks)
{
    const uint64_t n_elems = bytes / sizeof(T);
    uint64_t win_elems = win_bytes / sizeof(T);
    if (win_elems == 0 || win_elems > n_elems) win_elems = n_elems;
    const uint64_t n_win = n_elems / win_elems;
    hipEvent_t a, b; CHECK( hipEventCreate( &a ) ); CHECK( hipEventCreate( &b ) );
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep)
    {
        CHECK( hipEventRecord( a, 0 ) );
        hipLaunchKernelGGL( (gather_kernel<T,INFLIGHT,NT,CHAIN>), dim3( blocks ), dim3( 256 ), 0, 0, (const T*)table, n_elems, win_elems, n_win, Q,
                            (uint64_t)(rep * 1315423911u + 12345u), sink );
        CHECK( hipEventRecord( b, 0 ) );
        CHECK( hipEventSynchronize( b ) );
        float ms = 0; CHECK( hipEventElapsedTime( &ms, a, b ) );
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv)
{
    FILE* out = argc > 1 ? fopen( argv[1], "w" ) : stdout;
    if (!out) { perror( "out" ); return 1; }
    const uint64_t GiB = 1ull << 30;
    const uint64_t max_bytes = 128 * GiB;
    void* table = nullptr;
    CHECK( hipMalloc( &table, max_bytes ) );
    CHECK( hipMemset( table, 0, max_bytes ) );
    uint64_t* sink = nullptr; CHECK( hipMalloc( (void**)&sink, 8 ) );
    CHECK( hipDeviceSynchronize() );
    const uint64_t Q = 1ull << Q_LOG;                             // 134 M gathers per launch (the seed pass: 90 M seeds)
    const int blocks = 256 * 256;                                 // the seed pass's grid

    struct Foot { const char* name; uint64_t bytes; };
    const Foot feet[] = { { "0.25GiB", GiB / 4 }, { "0.75GB(text)", 750000000ull }, { "1.5GB(bwt_occ)", 1500000000ull }, { "12GB(SA)", 12000000000ull },
                          { "32GiB(k=16)", 32 * GiB }, { "64GiB", 64 * GiB }, { "128GiB(k=17)", 128 * GiB } };
#define EMIT(shape, T, INF, NTF, CH, fbytes, fname, wbytes)                                                                                 \
    do {                                                                                                                                    \
        const double ms = run<T,INF,NTF,CH>( table, fbytes, wbytes, Q, sink, blocks );                                                      \
        const double g  = (double)Q * CH / (ms * 1e-3) / 1e9;                                                                              \
        fprintf( out, "{\"footprint\": \"%s\", \"footprint_bytes\": %llu, \"elem_bytes\": %d, \"window_bytes\": %llu, \"inflight_per_lane\": %d, "    \
                      "\"nontemporal\": %s, \"chain\": %d, \"gathers\": %llu, \"ms\": %.4f, \"G_gathers_per_s\": %.2f, \"sector_GBps\": %.1f, "       \
                      "\"frac_of_8TBps\": %.3f}\n", fname, (unsigned long long)(fbytes), (int)sizeof(T), (unsigned long long)(wbytes), INF,          \
                 NTF ? "true" : "false", CH, (unsigned long long)(Q * CH), ms, g, g * 64.0, g * 64.0 / 8000.0 );                                    \
        fflush( out );                                                                                                                      \
        fprintf( stderr, "%-16s elem %2d win %12llu inflight %d nt %d chain %d: %8.3f ms  %6.2f G/s\n", fname, (int)sizeof(T),                \
                 (unsigned long long)(wbytes), INF, (int)NTF, CH, ms, g );                                                                     \
    } while (0)

    // 1. uniform random over the footprint: 8-byte elements, one gather in flight per lane (the seed pass's shape)
    for (const Foot& f : feet) { EMIT( "u8", uint64_t, 1, false, 1, f.bytes, f.name, 0ull ); }
    // 2. the same with a non-temporal hint, 2 and 4 gathers in flight per lane
    for (const Foot& f : feet) { EMIT( "u8nt", uint64_t, 1, true, 1, f.bytes, f.name, 0ull ); }
    for (const Foot& f : feet) { EMIT( "u8x2", uint64_t, 2, false, 1, f.bytes, f.name, 0ull ); }
    for (const Foot& f : feet) { EMIT( "u8x4", uint64_t, 4, false, 1, f.bytes, f.name, 0ull ); }
    // 3. 4-byte elements and 32-byte records
    for (const Foot& f : feet) { EMIT( "u4", uint32_t, 1, false, 1, f.bytes, f.name, 0ull ); }
    for (const Foot& f : feet) { EMIT( "r32", Rec32, 1, false, 1, f.bytes, f.name, 0ull ); }
    // 4. locality: requests in window order over the 128 GiB / 64 GiB table (what a partition by the high key bits gives)
    const uint64_t wins[] = { 2ull << 20, 16ull << 20, 64ull << 20, 256ull << 20, 1ull << 30, 8ull << 30 };
    for (uint64_t w : wins) { EMIT( "w8", uint64_t, 1, false, 1, 128 * GiB, "128GiB(k=17)", w ); }
    for (uint64_t w : wins) { EMIT( "w8x2", uint64_t, 2, false, 1, 128 * GiB, "128GiB(k=17)", w ); }
    for (uint64_t w : wins) { EMIT( "w4", uint32_t, 1, false, 1, 64 * GiB, "64GiB", w ); }
    // 5. dependent chains of 2 and 3 uniform gathers (table -> text -> ...), per-lane
    for (const Foot& f : feet) { EMIT( "c2", uint64_t, 1, false, 2, f.bytes, f.name, 0ull ); }
    EMIT( "c3", uint64_t, 1, false, 3, 128 * GiB, "128GiB(k=17)", 0ull );
    if (out != stdout) fclose( out );
    CHECK( hipFree( table ) );
    return 0;
}
