// gather2.hip -- second random-gather micro-benchmark: what caps the MI355X's rate of random 64-byte sectors?
//
// gather_rate.hip (round 2) read 55 G gathers/s at 0.25 GiB and at 1.5 GB alike, 48.5 G/s over 128 GiB, the same with 2 / 4 gathers in
// flight per lane -- and 63 / 74 G/s for chains of 2 / 3 DEPENDENT gathers, i.e. marginal hops at 90-110 G/s.  Three things in that program
// blur the reading: (1) every address costs ~20 quarter-rate 32-bit multiplies (two 64 x 64 -> 128 products and a 64-bit mixer), the same
// for every footprint and every in-flight count; (2) 0.25 GiB is exactly the Infinity Cache's size, so no footprint there was cache
// resident; (3) occupancy was never varied.  This program separates them:
//   * addresses from two 32-bit multiplies (or from a precomputed, coalesced index stream: `pre`), power-of-two footprints (a mask);
//   * footprints from 32 MiB (L2 / Infinity-Cache resident) to 128 GiB;
//   * waves per SIMD limited through dynamic LDS; independent gathers per lane 1..8; dependent chains 1..3; 4- / 8- / 16-byte elements;
//   * one configuration per process invocation, so that `rocprofv3 --pmc ... -- ./gather2 one ...` attributes counters to it.
//
//   ./gather2 sweep out.jsonl            every configuration of the tables in DESIGN.md
//   ./gather2 one <log2 footprint bytes> <elem bytes> <inflight> <chain> <waves per SIMD> <pre 0|1> <memory 0 default|1 uncached|2 fine-grained>
//                 [launches [load 0 plain|1 nt|2 sc1|3 sc0 sc1 (forces 8-byte elements, 4 in flight, chain 1)]]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define Q_LOG 27
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf( stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString( e_ ) ); exit( 1 ); } } while (0)

__device__ __forceinline__ uint32_t fmix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// 37 random bits from a 32-bit counter: two multiplies for the low word, a third for the high bits
__device__ __forceinline__ uint64_t address_bits(const uint32_t i, const uint32_t salt)
{
    const uint32_t lo = fmix( i ^ salt );
    const uint32_t hi = (lo ^ 0x9E3779B9u) * 0x27D4EB2Fu;
    return ((uint64_t)(hi >> 16) << 32) | lo;
}

// LOAD: 0 plain, 1 non-temporal (nt), 2 agent-scope relaxed atomic load (sc1), 3 system-scope (sc0 sc1)
template <int LOAD>
__device__ __forceinline__ uint64_t load8(const uint8_t* p)
{
    if (LOAD == 1) return __builtin_nontemporal_load( (const uint64_t*)p );
    if (LOAD == 2) return __hip_atomic_load( (const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    if (LOAD == 3) return __hip_atomic_load( (const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
    return *(const uint64_t*)p;
}

template <int EB, int INF, int CHAIN, bool PRE, int LOAD = 0>
__global__ void __launch_bounds__(256)
gather2_kernel(const uint8_t* __restrict__ table, const uint64_t elem_mask, const uint64_t* __restrict__ pre, const uint32_t Q, const uint32_t salt,
               uint64_t* __restrict__ sink)
{
    extern __shared__ uint32_t lds_pad[];                     // only there to limit the waves per SIMD
    uint64_t acc = 0;
    const uint32_t stride = gridDim.x * 256u * INF;
    for (uint32_t base = blockIdx.x * 256u * INF; base < Q; base += stride)
    {
        uint64_t idx[INF];
        #pragma unroll
        for (int g = 0; g < INF; ++g)
        {
            const uint32_t i = base + (uint32_t)g * 256u + threadIdx.x;
            idx[g] = (PRE ? pre[i] : address_bits( i, salt )) & elem_mask;
        }
        #pragma unroll
        for (int c = 0; c < CHAIN; ++c)
        {
            uint64_t v[INF];
            #pragma unroll
            for (int g = 0; g < INF; ++g)
            {
                const uint8_t* p = table + idx[g] * EB;
                if (EB == 4)      v[g] = *(const uint32_t*)p;
                else if (EB == 8) v[g] = load8<LOAD>( p );
                else { const uint4 a = *(const uint4*)p; v[g] = (uint64_t)a.x + a.y + a.z + a.w; }
            }
            #pragma unroll
            for (int g = 0; g < INF; ++g)
            {
                acc += v[g];
                // the next link depends on the loaded value (the table holds zeros; the compiler cannot know)
                idx[g] = address_bits( (uint32_t)idx[g] + (uint32_t)v[g], salt + 77u * (uint32_t)(c + 1) ) & elem_mask;
            }
        }
    }
    if (acc == 0x123456789ull) sink[0] = acc + lds_pad[0];     // never true: keeps the loads alive
}

__global__ void fill_pre_kernel(uint64_t* pre, const uint32_t Q, const uint32_t salt)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < Q) pre[i] = address_bits( i, salt );
}

struct Config { int flog, eb, inf, chain, wps, pre, uncached; int load = 0; };

typedef void (*kern_t)(const uint8_t*, uint64_t, const uint64_t*, uint32_t, uint32_t, uint64_t*);

template <int EB, int INF, bool PRE>
static kern_t pick_chain(int chain)
{
    switch (chain) { case 1: return gather2_kernel<EB,INF,1,PRE>; case 2: return gather2_kernel<EB,INF,2,PRE>; default: return gather2_kernel<EB,INF,3,PRE>; }
}
template <int EB, bool PRE>
static kern_t pick_inf(int inf, int chain)
{
    switch (inf) { case 1: return pick_chain<EB,1,PRE>( chain ); case 2: return pick_chain<EB,2,PRE>( chain ); case 4: return pick_chain<EB,4,PRE>( chain );
                   default: return pick_chain<EB,8,PRE>( chain ); }
}
static kern_t pick(const Config& c)
{
    if (c.load == 1) return gather2_kernel<8,4,1,false,1>;
    if (c.load == 2) return gather2_kernel<8,4,1,false,2>;
    if (c.load == 3) return gather2_kernel<8,4,1,false,3>;
    if (c.pre) return c.eb == 4 ? pick_inf<4,true>( c.inf, c.chain ) : c.eb == 8 ? pick_inf<8,true>( c.inf, c.chain ) : pick_inf<16,true>( c.inf, c.chain );
    return c.eb == 4 ? pick_inf<4,false>( c.inf, c.chain ) : c.eb == 8 ? pick_inf<8,false>( c.inf, c.chain ) : pick_inf<16,false>( c.inf, c.chain );
}

static double run(const Config& c, const void* table, const uint64_t* pre, uint64_t* sink, int launches, double* all_ms)
{
    const uint64_t bytes = 1ull << c.flog;
    const uint64_t elem_mask = bytes / (uint64_t)c.eb - 1u;
    const uint32_t Q = 1u << Q_LOG;
    kern_t k = pick( c );
    // waves per SIMD = 256-thread blocks per CU: limited by dynamic LDS (160 KiB per CU)
    const size_t lds = c.wps >= 8 ? 0 : (size_t)(160 * 1024 / c.wps) - 1024u * (c.wps > 1 ? 1u : 0u);
    CHECK( hipFuncSetAttribute( (const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds ) );
    const int blocks = 256 * c.wps * 8;                       // eight rounds of resident blocks: a wave makes Q / (blocks * 256 * inf) iterations
    hipEvent_t a, b; CHECK( hipEventCreate( &a ) ); CHECK( hipEventCreate( &b ) );
    double best = 1e30;
    for (int rep = 0; rep < launches; ++rep)
    {
        CHECK( hipEventRecord( a, 0 ) );
        hipLaunchKernelGGL( k, dim3( blocks ), dim3( 256 ), lds, 0, (const uint8_t*)table, elem_mask, pre, Q, (uint32_t)(rep * 1315423911u + 12345u), sink );
        CHECK( hipEventRecord( b, 0 ) );
        CHECK( hipEventSynchronize( b ) );
        float ms = 0; CHECK( hipEventElapsedTime( &ms, a, b ) );
        if (all_ms) all_ms[rep] = ms;
        if (rep > 0 && ms < best) best = ms;
    }
    CHECK( hipGetLastError() );
    return best;
}

static void emit(FILE* out, const Config& c, double ms)
{
    const double g = (double)(1u << Q_LOG) * c.chain / (ms * 1e-3) / 1e9;
    fprintf( out, "{\"footprint_log2\": %d, \"footprint_GiB\": %.4f, \"elem_bytes\": %d, \"inflight_per_lane\": %d, \"chain\": %d, \"waves_per_simd\": %d, "
                  "\"precomputed_addresses\": %s, \"memory\": \"%s\", \"load\": \"%s\", \"gathers\": %llu, \"ms\": %.4f, \"G_gathers_per_s\": %.2f, \"sector_TBps\": %.3f, \"frac_of_8TBps\": %.3f}\n",
             c.flog, (double)(1ull << c.flog) / (double)(1ull << 30), c.eb, c.inf, c.chain, c.wps, c.pre ? "true" : "false", c.uncached == 1 ? "uncached" : c.uncached == 2 ? "finegrained" : "default",
             c.load == 1 ? "nt" : c.load == 2 ? "sc1" : c.load == 3 ? "sc0 sc1" : "plain",
             (unsigned long long)(1u << Q_LOG) * c.chain, ms, g, g * 64e-3, g * 64.0 / 8000.0 );
    fflush( out );
    fprintf( stderr, "2^%d B elem %2d inflight %d chain %d wps %d pre %d mem %d load %d: %8.3f ms  %6.2f G/s\n", c.flog, c.eb, c.inf, c.chain, c.wps, c.pre, c.uncached, c.load, ms, g );
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf( stderr, "usage: gather2 sweep out.jsonl | gather2 one flog eb inf chain wps pre uncached [launches]\n" ); return 2; }
    const bool one = strcmp( argv[1], "one" ) == 0;
    Config c1 = { 37, 8, 1, 1, 8, 0, 0 };
    int launches = 4;
    if (one)
    {
        if (argc < 9) { fprintf( stderr, "one: 7 numbers\n" ); return 2; }
        c1 = Config{ atoi( argv[2] ), atoi( argv[3] ), atoi( argv[4] ), atoi( argv[5] ), atoi( argv[6] ), atoi( argv[7] ), atoi( argv[8] ) };
        if (argc > 9) launches = atoi( argv[9] );
        if (argc > 10) { c1.load = atoi( argv[10] ); c1.eb = 8; c1.inf = 4; c1.chain = 1; c1.pre = 0; }
    }
    const int max_log = one ? c1.flog : 37;
    const uint64_t max_bytes = 1ull << max_log;
    void* table = nullptr;
    if (one && c1.uncached) CHECK( hipExtMallocWithFlags( &table, max_bytes, c1.uncached == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached ) );
    else CHECK( hipMalloc( &table, max_bytes ) );
    CHECK( hipMemset( table, 0, max_bytes ) );
    uint64_t* sink = nullptr; CHECK( hipMalloc( (void**)&sink, 8 ) );
    uint64_t* pre = nullptr; CHECK( hipMalloc( (void**)&pre, 8ull << Q_LOG ) );
    hipLaunchKernelGGL( fill_pre_kernel, dim3( (1u << Q_LOG) / 256u ), dim3( 256 ), 0, 0, pre, 1u << Q_LOG, 0xABCDEFu );
    CHECK( hipDeviceSynchronize() );
    if (one)
    {
        double ms[64];
        if (launches > 64) launches = 64;
        const double best = run( c1, table, pre, sink, launches, ms );
        emit( stdout, c1, best );
        return 0;
    }
    FILE* out = fopen( argv[2], "w" );
    if (!out) { perror( "out" ); return 1; }
    const int feet[] = { 25, 27, 30, 32, 34, 37 };
    for (int f : feet)
    {
        // 1. the round-2 shape with cheap addresses; more gathers in flight; fewer waves
        for (int inf : { 1, 2, 4, 8 })
            for (int wps : { 8, 4, 2, 1 }) { const Config c = { f, 8, inf, 1, wps, 0, 0 }; emit( out, c, run( c, table, pre, sink, 4, nullptr ) ); }
        // 2. dependent chains
        for (int ch : { 2, 3 })
            for (int inf : { 1, 4 }) { const Config c = { f, 8, inf, ch, 8, 0, 0 }; emit( out, c, run( c, table, pre, sink, 4, nullptr ) ); }
        // 3. element width; precomputed addresses
        for (int eb : { 4, 16 }) { const Config c = { f, eb, 4, 1, 8, 0, 0 }; emit( out, c, run( c, table, pre, sink, 4, nullptr ) ); }
        for (int inf : { 1, 4 }) { const Config c = { f, 8, inf, 1, 8, 1, 0 }; emit( out, c, run( c, table, pre, sink, 4, nullptr ) ); }
    }
    fclose( out );
    CHECK( hipFree( table ) );
    return 0;
}
