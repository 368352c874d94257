// seed_hits.hip -- nvBowtie's seed-hit bookkeeping around the seed-and-extend path, for gfx950:
//   * the per-read deque of seed hits filled by the exact seed mapper, capped at max_hits, and the reseeding decision
//         seed_mapper<EXACT_MAPPING>::enact + map_kernel            nvBowtie/bowtie2/cuda/mapping_inl.h:193-282,485-556
//         SeedHit, hit_compare                                      nvBowtie/bowtie2/cuda/seed_hit.h:45-229
//         priority_deque over an interval heap                      nvbio/basic/priority_deque.h, interval_heap.h
//   * select_kernel: the next SA row of every active read's top hit                 select_inl.h:62-130
//   * score_reduce_kernel with ReduceBestApproxContext: best / second best in arrival order, the effort counter, the stop
//                                                                   reduce_inl.h:65-140, reduce.h:55-99
// One lane owns one read, as in the reference: these are small state machines per read (a deque of at most 2 x seeds entries,
// two alignments, a counter), streamed once per extension pass; what costs time in this mode is the number of passes, not
// these kernels.  The deque is an interval heap: pairs (2k, 2k+1) hold the low and the high end of node k, "low" meaning
// first under hit_compare (the LARGEST range), so that element 0 is what a full deque drops and element 1 (or 0 when alone) is
// the smallest range, the one select takes rows from.  Hits of equal size are ordered by the heap's moves alone, so the
// moves below are the reference container's, one for one (checked against it through the oracle: tests/test_gpu_seed_hits.py).
#include "seed_hits_device.h"
#include <hipcub/hipcub.hpp>

namespace nvbio_amd {

// ---- map: the deques of a batch of reads from the match ranges of their seeds ----
// Read r (uniform length read_len) has seeds j = 0 .. spr-1 at stored offsets first_off + j * interval; fw / rc hold what
// match_range returned for the forward scan of the stored read and for its reverse scan complemented (inclusive, empty iff x > y).
__global__ void __launch_bounds__(256)
seed_hits_map_kernel(const uint2* __restrict__ fw, const uint2* __restrict__ rc, const uint32_t* __restrict__ queue, const uint32_t n_reads, const uint32_t spr,
                     const uint32_t first_off, const uint32_t interval, const uint32_t seed_len, const uint32_t read_len, const uint32_t max_hits,
                     const uint32_t rep_seeds, const uint32_t cap, uint2* __restrict__ deques, uint32_t* __restrict__ sizes,
                     uint8_t* __restrict__ reseed)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_reads; t += gridDim.x * blockDim.x)
    {
        const uint32_t r = queue ? queue[t] : t;
        HitHeap heap; heap.a = deques + (uint64_t)r * cap; heap.n = 0;
        uint32_t range_sum = 0, range_count = 0;
        for (uint32_t j = 0; j < spr; ++j)
        {
            const uint32_t off = first_off + j * interval;
            #pragma unroll
            for (uint32_t strand = 0; strand < 2u; ++strand)
            {
                const uint2 g = strand ? rc[(uint64_t)t * spr + j] : fw[(uint64_t)t * spr + j];
                if (g.x > g.y) continue;
                const uint32_t pos = strand ? off : read_len - off - seed_len;       // SeedHit::build_flags (mapping_inl.h:241,275)
                if (heap.n == max_hits) heap.pop_bottom();
                heap.push( make_uint2( g.x, ((g.y + 1u - g.x) & 0xFFFFFu) | ((pos & 0x3FFu) << 20) | (strand << 30) ) );
                range_sum += g.y - g.x + 1u; ++range_count;
            }
        }
        sizes[r] = heap.n;
        if (reseed) reseed[r] = (range_count == 0u || range_sum >= rep_seeds * range_count) ? 1 : 0;
    }
}

// ---- select: one SA row per active read ----
// active_in[t] = read id | top_flag << 31 (packed_read, defs.h).  A read that still has a hit takes a slot of the output queue:
// active_out[slot], hit_read_id[slot], hit_loc[slot] = the SA row, hit_seed[slot] = packed_seed.  counts[0] = slots written.
__global__ void __launch_bounds__(256)
seed_hits_select_kernel(const uint32_t* __restrict__ active_in, const uint32_t n_active, const uint32_t* __restrict__ trys, const uint32_t cap,
                        uint2* __restrict__ deques, uint32_t* __restrict__ sizes, uint32_t* __restrict__ active_out, uint32_t* __restrict__ hit_read_id,
                        uint32_t* __restrict__ hit_loc, uint32_t* __restrict__ hit_seed, unsigned int* __restrict__ counts)
{
    __shared__ uint32_t s_cnt[4], s_base[2];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n_active; base += gridDim.x * blockDim.x)
    {
        const uint32_t t = base + threadIdx.x;
        bool     take = false;
        uint32_t read = 0, top_flag = 0, row = 0, seed = 0;
        if (t < n_active)
        {
            read = active_in[t] & 0x7FFFFFFFu; top_flag = active_in[t] >> 31;
            if (!(trys && trys[read] == 0u))                             // context.stop( read_id )
            {
                HitHeap heap; heap.a = deques + (uint64_t)read * cap; heap.n = sizes[read];
                if (heap.n)
                {
                    uint32_t k = heap.top();
                    if (HitHeap::size_of( heap.a[k] ) == 0u)             // the top range is used up
                    {
                        heap.pop_top();
                        top_flag = 0u;
                        k = heap.top();
                    }
                    if (heap.n)
                    {
                        uint2 h = heap.a[k];
                        row = h.x;                                       // SeedHit::pop_front
                        h.x += 1u; h.y = (h.y & ~0xFFFFFu) | ((HitHeap::size_of( h ) - 1u) & 0xFFFFFu);
                        heap.a[k] = h;
                        seed = ((h.y >> 20) & 0x3FFu) | (((h.y >> 31) & 1u) << 12) | (((h.y >> 30) & 1u) << 13) | (top_flag << 14);
                        take = true;
                    }
                    sizes[read] = heap.n;
                }
            }
        }
        // slots: ONE returning atomic per workgroup (every wave adding to the same counter serialises in L2: ~10 ns each, 3 ms per 10 M reads)
        const uint64_t m = __ballot( take );
        const uint32_t wave = threadIdx.x >> 6;
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll( m );
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            s_base[0] = tot ? atomicAdd( &counts[0], (unsigned int)tot ) : 0u;
        }
        __syncthreads();
        if (take)
        {
            uint32_t slot = s_base[0] + (uint32_t)__popcll( m & ((1ull << lane) - 1ull) );
            for (uint32_t w = 0; w < wave; ++w) slot += s_cnt[w];
            active_out[slot] = read | (top_flag << 31); hit_read_id[slot] = read; hit_loc[slot] = row; hit_seed[slot] = seed;
        }
        __syncthreads();
    }
}

// ---- locate helper: hit.loc = locate(row) - pos_in_read (locate_inl.h:127-136), uint32 arithmetic ----
__global__ void __launch_bounds__(256)
seed_hits_loc_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ hit_seed, const uint32_t n, uint32_t* __restrict__ hit_loc)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) hit_loc[i] = pos[i] - (hit_seed[i] & 0xFFFu);
}

// ---- reduce: best / second best in arrival order, the effort counter ----
// best[r] = { a1 score, a1 pos, a2 score, a2 pos } with the strands in bits 0 / 1 of best_rc[r]; positions 0xFFFFFFFF = unaligned.
// Work item i is the hit in slot i of the scoring queue (one hit per active read and pass).
__device__ __forceinline__ bool distinct_loci(const uint32_t pos1, const uint32_t rc1, const uint32_t pos2, const uint32_t rc2, const uint32_t dist)
{
    if (rc1 != rc2) return true;
    return !(pos1 >= pos2 - (pos2 < dist ? pos2 : dist) && pos1 <= pos2 + dist);          // io::distinct_alignments, uint32 arithmetic
}

__global__ void __launch_bounds__(256)
score_reduce_effort_kernel(const uint32_t* __restrict__ active, const uint32_t n, const int32_t* __restrict__ hit_score, const uint32_t* __restrict__ hit_loc,
                           const uint32_t* __restrict__ hit_seed, const uint32_t read_len, const uint32_t ext, const uint32_t max_effort,
                           const uint32_t min_ext, const uint32_t max_ext, int4* __restrict__ best, uint8_t* __restrict__ best_rc,
                           uint32_t* __restrict__ trys, uint32_t* __restrict__ sizes)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const uint32_t read = active[i] & 0x7FFFFFFFu;
        const uint32_t rc = (hit_seed[i] >> 13) & 1u, top_flag = (hit_seed[i] >> 14) & 1u;
        const int32_t  score = hit_score[i];
        const uint32_t g = hit_loc[i];
        int4 b = best[read];
        uint32_t rcs = best_rc[read];
        const uint32_t rc1 = rcs & 1u, rc2 = (rcs >> 1) & 1u;
        if ((rc == rc1 && g == (uint32_t)b.y) || (rc == rc2 && g == (uint32_t)b.w)) continue;     // a locus already held: free
        if (score > b.x)
        {
            trys[read] = max_effort;
            b.z = b.x; b.w = b.y; b.x = score; b.y = (int32_t)g;
            rcs = rc | (rc1 << 1);
        }
        else if (score > b.z && distinct_loci( (uint32_t)b.y, rc1, g, rc, read_len / 2u ))
        {
            trys[read] = max_effort;
            b.z = score; b.w = (int32_t)g;
            rcs = rc1 | (rc << 1);
        }
        else
        {
            uint32_t t = trys[read];
            if (t > 0u)
            {
                bool stop = false;
                if (ext >= min_ext && top_flag == 0u) { --t; trys[read] = t; stop = (t == 0u); }
                if (stop || ext >= max_ext) sizes[read] = 0u;            // pipeline.hits.erase( read_id )
            }
        }
        best[read] = b; best_rc[read] = (uint8_t)rcs;
    }
}


// ---- select, several hits per read (select_multi_kernel, select_inl.h:268-437): once fewer than half a batch of reads are active the
// reference takes up to n_multi SA rows per read and pass (aligner_best_approx.h:487-510) so that its launches stay large.  Per read: the
// loop of the reference -- top range used up -> pop_top, top_flag off; no hit left -> stop; else pop_front -- run n_multi times.  A read's
// hits take CONSECUTIVE slots of the hit queue, in selection order (the reference hands out slots one by one through an atomic and keeps
// a per-read index; what the reduction needs is the order within a read): first = hits_first[out slot], count = hits_count[out slot].
// The count is known before the pops: one row per iteration until the deque is empty, i.e. min( n_multi, rows left in the deque ).
__global__ void __launch_bounds__(256)
seed_hits_select_multi_kernel(const uint32_t* __restrict__ active_in, const uint32_t n_active, const uint32_t* __restrict__ trys, const uint32_t cap,
                              const uint32_t n_multi, uint2* __restrict__ deques, uint32_t* __restrict__ sizes, uint32_t* __restrict__ active_out,
                              uint32_t* __restrict__ hits_first, uint32_t* __restrict__ hits_count, uint32_t* __restrict__ hit_read_id,
                              uint32_t* __restrict__ hit_loc, uint32_t* __restrict__ hit_seed, unsigned int* __restrict__ counts)
{
    __shared__ uint32_t s_cnt[4], s_hits[4], s_base[2];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n_active; base += gridDim.x * blockDim.x)
    {
        const uint32_t t = base + threadIdx.x;
        uint32_t read = 0, top_flag = 0, want = 0;
        HitHeap heap; heap.a = deques; heap.n = 0;
        if (t < n_active)
        {
            read = active_in[t] & 0x7FFFFFFFu; top_flag = active_in[t] >> 31;
            if (!(trys && trys[read] == 0u))                             // context.stop( read_id )
            {
                heap.a = deques + (uint64_t)read * cap; heap.n = sizes[read];
                uint64_t rows = 0;
                for (uint32_t k = 0; k < heap.n; ++k) rows += HitHeap::size_of( heap.a[k] );
                want = (uint32_t)(rows < n_multi ? rows : n_multi);
            }
        }
        // slots: one of the read queue per read with a hit, `want` consecutive ones of the hit queue
        const uint64_t m = __ballot( want != 0u );
        uint32_t incl = want;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t a = (uint32_t)__shfl_up( (int)incl, d ); if (lane >= (uint32_t)d) incl += a; }
        const uint32_t wave_hits = (uint32_t)__shfl( (int)incl, 63 );
        // one returning atomic per workgroup and counter (every wave adding to the same two counters serialises in L2)
        const uint32_t wave = threadIdx.x >> 6;
        if (lane == 0) { s_cnt[wave] = (uint32_t)__popcll( m ); s_hits[wave] = wave_hits; }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], toth = s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3];
            s_base[0] = tot ? atomicAdd( &counts[0], (unsigned int)tot ) : 0u;
            s_base[1] = toth ? atomicAdd( &counts[1], (unsigned int)toth ) : 0u;
        }
        __syncthreads();
        uint32_t slot0 = s_base[0], hit0 = s_base[1];
        for (uint32_t w = 0; w < wave; ++w) { slot0 += s_cnt[w]; hit0 += s_hits[w]; }
        __syncthreads();
        if (want)
        {
            const uint32_t slot = slot0 + (uint32_t)__popcll( m & ((1ull << lane) - 1ull) );
            const uint32_t first = hit0 + incl - want;
            for (uint32_t i = 0; i < want; ++i)
            {
                uint32_t k = heap.top();
                if (HitHeap::size_of( heap.a[k] ) == 0u) { heap.pop_top(); top_flag = 0u; k = heap.top(); }     // (rows are left: the deque is not empty)
                uint2 h = heap.a[k];
                const uint32_t row = h.x;
                h.x += 1u; h.y = (h.y & ~0xFFFFFu) | ((HitHeap::size_of( h ) - 1u) & 0xFFFFFu);
                heap.a[k] = h;
                hit_read_id[first + i] = read; hit_loc[first + i] = row;
                hit_seed[first + i] = ((h.y >> 20) & 0x3FFu) | (((h.y >> 31) & 1u) << 12) | (((h.y >> 30) & 1u) << 13) | (top_flag << 14);
            }
            sizes[read] = heap.n;
            active_out[slot] = read | (top_flag << 31); hits_first[slot] = first; hits_count[slot] = want;
        }
        else if (t < n_active && heap.n)
        {
            // a deque that holds nothing but used-up ranges: the reference pops it empty and drops the read (select_inl.h:373-383)
            sizes[read] = 0u;
        }
    }
}

// ---- reduce over the hits of every active read, in selection order (score_reduce_kernel's loop, reduce_inl.h:94-134) ----
__global__ void __launch_bounds__(256)
score_reduce_effort_multi_kernel(const uint32_t* __restrict__ active, const uint32_t n, const uint32_t* __restrict__ hits_first,
                                 const uint32_t* __restrict__ hits_count, const int32_t* __restrict__ hit_score, const uint32_t* __restrict__ hit_loc,
                                 const uint32_t* __restrict__ hit_seed, const uint32_t read_len, const uint32_t ext, const uint32_t max_effort,
                                 const uint32_t min_ext, const uint32_t max_ext, int4* __restrict__ best, uint8_t* __restrict__ best_rc,
                                 uint32_t* __restrict__ trys, uint32_t* __restrict__ sizes)
{
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x)
    {
        const uint32_t read = active[a] & 0x7FFFFFFFu;
        const uint32_t first = hits_first[a], count = hits_count[a];
        int4 b = best[read];
        uint32_t rcs = best_rc[read];
        uint32_t t = trys[read];
        for (uint32_t idx = 0; idx < count; ++idx)
        {
            const uint32_t i = first + idx;
            const uint32_t rc = (hit_seed[i] >> 13) & 1u, top_flag = (hit_seed[i] >> 14) & 1u;
            const int32_t  score = hit_score[i];
            const uint32_t g = hit_loc[i];
            const uint32_t rc1 = rcs & 1u, rc2 = (rcs >> 1) & 1u;
            if ((rc == rc1 && g == (uint32_t)b.y) || (rc == rc2 && g == (uint32_t)b.w)) continue;     // a locus already held: free
            if (score > b.x)
            {
                t = max_effort;
                b.z = b.x; b.w = b.y; b.x = score; b.y = (int32_t)g;
                rcs = rc | (rc1 << 1);
            }
            else if (score > b.z && distinct_loci( (uint32_t)b.y, rc1, g, rc, read_len / 2u ))
            {
                t = max_effort;
                b.z = score; b.w = (int32_t)g;
                rcs = rc1 | (rc << 1);
            }
            else if (t > 0u)                                             // ReduceBestApproxContext::failure( idx, ... ) (reduce.h:82-92)
            {
                bool stop = false;
                if (ext + idx >= min_ext && top_flag == 0u) { --t; stop = (t == 0u); }
                if (stop || ext + idx >= max_ext) sizes[read] = 0u;      // pipeline.hits.erase( read_id ); the remaining scores are still used
            }
        }
        best[read] = b; best_rc[read] = (uint8_t)rcs; trys[read] = t;
    }
}

// ---- the read queues of the best-approx loop (aligner_best_approx.h:77,148-207,363-450), so that a host loop over the C ABI needs no
// device code of its own ----
__global__ void __launch_bounds__(256)
best_approx_init_kernel(const uint32_t n, const int32_t worst, int4* __restrict__ best, uint8_t* __restrict__ best_rc)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
    {
        best[r] = make_int4( worst, -1, worst, -1 ); best_rc[r] = 0;      // init_alignments( reads, threshold_score, .. ) (aligner.h:279-301)
    }
}
__global__ void __launch_bounds__(256)
read_queue_begin_kernel(const uint32_t* __restrict__ queue, const uint32_t n, const uint32_t read_len, const uint32_t first_off, const uint32_t top_seed,
                        const uint32_t max_effort_init, uint32_t* __restrict__ seed_offsets, uint32_t* __restrict__ active, uint32_t* __restrict__ trys)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
    {
        const uint32_t r = queue ? queue[t] : t;
        if (seed_offsets) seed_offsets[t] = r * read_len + first_off;     // where the read's first seed of this seeding pass begins
        if (active) active[t] = r | (top_seed << 31);                     // packed_read( read_id, top_flag )
        if (trys) trys[r] = max_effort_init;                              // select_init
    }
}
struct ReadFlagSet { const uint8_t* flags; __host__ __device__ __forceinline__ bool operator()(const uint32_t r) const { return flags[r] != 0; } };

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" {

nvbio_status nvbio_seed_hits_capacity(uint32_t seeds_per_read, uint32_t max_hits, uint32_t* capacity)
{
    NVB_REQUIRE( capacity != nullptr, "capacity is NULL" );
    const uint64_t c = 2ull * seeds_per_read;
    *capacity = (uint32_t)((c < max_hits ? c : max_hits) + 1u);
    return NVBIO_OK;
}

nvbio_status nvbio_seed_hits_map(int device, const nvbio_uint2* fw_ranges_dev, const nvbio_uint2* rc_ranges_dev, const uint32_t* read_queue_dev,
                                 uint32_t n_reads, const nvbio_seed_hits_params* p, nvbio_uint2* deques_dev, uint32_t* sizes_dev, uint8_t* reseed_dev,
                                 void* stream)
{
    NVB_REQUIRE( p != nullptr, "params is NULL" );
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( fw_ranges_dev && rc_ranges_dev && deques_dev && sizes_dev, "NULL device pointer" );
    NVB_REQUIRE( p->max_hits > 0 && p->seeds_per_read > 0, "max_hits and seeds_per_read must be positive" );
    NVB_REQUIRE( (uint64_t)p->first_offset + (uint64_t)(p->seeds_per_read - 1u) * p->seed_interval + p->seed_len <= p->read_len, "seeds do not fit the read" );
    NVB_REQUIRE( p->read_len < 1024u, "SeedHit keeps the seed position in 10 bits (seed_hit.h:217)" );
    uint32_t cap = 0; NVB_CHECK( nvbio_seed_hits_capacity( p->seeds_per_read, p->max_hits, &cap ) );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( seed_hits_map_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream, (const uint2*)fw_ranges_dev, (const uint2*)rc_ranges_dev,
                        read_queue_dev, n_reads, p->seeds_per_read, p->first_offset, p->seed_interval, p->seed_len, p->read_len, p->max_hits, p->rep_seeds, cap,
                        (uint2*)deques_dev, sizes_dev, reseed_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_seed_hits_select(int device, const uint32_t* active_in_dev, uint32_t n_active, const uint32_t* trys_dev, uint32_t capacity,
                                    nvbio_uint2* deques_dev, uint32_t* sizes_dev, uint32_t* active_out_dev, const nvbio_hit_queues* hits,
                                    uint32_t* count_dev, void* stream)
{
    NVB_REQUIRE( count_dev != nullptr && hits != nullptr, "NULL argument" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    NVB_HIP( hipMemsetAsync( count_dev, 0, sizeof(uint32_t), (hipStream_t)stream ) );
    if (n_active == 0) return NVBIO_OK;
    NVB_REQUIRE( active_in_dev && deques_dev && sizes_dev && active_out_dev && hits->hit_read_id_dev && hits->hit_loc_dev && hits->hit_seed_dev, "NULL device pointer" );
    hipLaunchKernelGGL( seed_hits_select_kernel, dim3( grid_for( n_active ) ), dim3(256), 0, (hipStream_t)stream, active_in_dev, n_active, trys_dev, capacity,
                        (uint2*)deques_dev, sizes_dev, active_out_dev, (uint32_t*)hits->hit_read_id_dev, (uint32_t*)hits->hit_loc_dev, (uint32_t*)hits->hit_seed_dev,
                        (unsigned int*)count_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_seed_hits_loc(int device, const uint32_t* positions_dev, const nvbio_hit_queues* hits, void* stream)
{
    NVB_REQUIRE( hits != nullptr, "hits is NULL" );
    if (hits->n == 0) return NVBIO_OK;
    NVB_REQUIRE( positions_dev && hits->hit_seed_dev && hits->hit_loc_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( seed_hits_loc_kernel, dim3( grid_for( hits->n ) ), dim3(256), 0, (hipStream_t)stream, positions_dev, hits->hit_seed_dev, hits->n, (uint32_t*)hits->hit_loc_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_score_reduce_effort(int device, const uint32_t* active_dev, const nvbio_hit_queues* hits, uint32_t read_len, uint32_t n_ext,
                                       const nvbio_seed_hits_params* p, int32_t* best_dev, uint8_t* best_rc_dev, uint32_t* trys_dev, uint32_t* sizes_dev,
                                       void* stream)
{
    NVB_REQUIRE( hits != nullptr && p != nullptr, "NULL argument" );
    if (hits->n == 0) return NVBIO_OK;
    NVB_REQUIRE( active_dev && hits->hit_score_dev && hits->hit_loc_dev && hits->hit_seed_dev && best_dev && best_rc_dev && trys_dev && sizes_dev, "NULL device pointer" );
    NVB_REQUIRE( ((uintptr_t)best_dev & 15u) == 0, "best_dev must be 16-byte aligned" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( score_reduce_effort_kernel, dim3( grid_for( hits->n ) ), dim3(256), 0, (hipStream_t)stream, active_dev, hits->n, hits->hit_score_dev,
                        hits->hit_loc_dev, hits->hit_seed_dev, read_len, n_ext, p->max_effort, p->min_ext, p->max_ext, (int4*)best_dev, best_rc_dev, trys_dev, sizes_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_seed_hits_select_multi(int device, const uint32_t* active_in_dev, uint32_t n_active, const uint32_t* trys_dev, uint32_t capacity,
                                          uint32_t n_multi, nvbio_uint2* deques_dev, uint32_t* sizes_dev, uint32_t* active_out_dev,
                                          uint32_t* hits_first_dev, uint32_t* hits_count_dev, const nvbio_hit_queues* hits, uint32_t* counts_dev,
                                          void* stream)
{
    NVB_REQUIRE( counts_dev != nullptr && hits != nullptr, "NULL argument" );
    NVB_REQUIRE( n_multi >= 1u, "n_multi must be positive" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    NVB_HIP( hipMemsetAsync( counts_dev, 0, 2 * sizeof(uint32_t), (hipStream_t)stream ) );
    if (n_active == 0) return NVBIO_OK;
    NVB_REQUIRE( active_in_dev && deques_dev && sizes_dev && active_out_dev && hits_first_dev && hits_count_dev && hits->hit_read_id_dev && hits->hit_loc_dev &&
                 hits->hit_seed_dev, "NULL device pointer" );
    hipLaunchKernelGGL( seed_hits_select_multi_kernel, dim3( grid_for( n_active ) ), dim3(256), 0, (hipStream_t)stream, active_in_dev, n_active, trys_dev, capacity,
                        n_multi, (uint2*)deques_dev, sizes_dev, active_out_dev, hits_first_dev, hits_count_dev, (uint32_t*)hits->hit_read_id_dev,
                        (uint32_t*)hits->hit_loc_dev, (uint32_t*)hits->hit_seed_dev, (unsigned int*)counts_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_score_reduce_effort_multi(int device, const uint32_t* active_dev, uint32_t n_active, const uint32_t* hits_first_dev,
                                             const uint32_t* hits_count_dev, const nvbio_hit_queues* hits, uint32_t read_len, uint32_t n_ext,
                                             const nvbio_seed_hits_params* p, int32_t* best_dev, uint8_t* best_rc_dev, uint32_t* trys_dev,
                                             uint32_t* sizes_dev, void* stream)
{
    NVB_REQUIRE( hits != nullptr && p != nullptr, "NULL argument" );
    if (n_active == 0) return NVBIO_OK;
    NVB_REQUIRE( active_dev && hits_first_dev && hits_count_dev && hits->hit_score_dev && hits->hit_loc_dev && hits->hit_seed_dev && best_dev && best_rc_dev &&
                 trys_dev && sizes_dev, "NULL device pointer" );
    NVB_REQUIRE( ((uintptr_t)best_dev & 15u) == 0, "best_dev must be 16-byte aligned" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( score_reduce_effort_multi_kernel, dim3( grid_for( n_active ) ), dim3(256), 0, (hipStream_t)stream, active_dev, n_active, hits_first_dev,
                        hits_count_dev, hits->hit_score_dev, hits->hit_loc_dev, hits->hit_seed_dev, read_len, n_ext, p->max_effort, p->min_ext, p->max_ext,
                        (int4*)best_dev, best_rc_dev, trys_dev, sizes_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_best_approx_init(int device, uint32_t n_reads, int32_t worst_score, int32_t* best_dev, uint8_t* best_rc_dev, void* stream)
{
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( best_dev && best_rc_dev, "NULL device pointer" );
    NVB_REQUIRE( ((uintptr_t)best_dev & 15u) == 0, "best_dev must be 16-byte aligned" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( best_approx_init_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream, n_reads, worst_score, (int4*)best_dev, best_rc_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_read_queue_begin(int device, const uint32_t* queue_dev, uint32_t n, uint32_t read_len, uint32_t first_offset, uint32_t top_seed,
                                    uint32_t max_effort_init, uint32_t* seed_offsets_dev, uint32_t* active_dev, uint32_t* trys_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( read_queue_begin_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, queue_dev, n, read_len, first_offset, top_seed & 1u,
                        max_effort_init, seed_offsets_dev, active_dev, trys_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_read_queue_filter(int device, const uint32_t* queue_dev, uint32_t n, const uint8_t* read_flags_dev, uint32_t* queue_out_dev,
                                     uint32_t* count_dev, void* stream)
{
    NVB_REQUIRE( count_dev != nullptr, "count_dev is NULL" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    NVB_HIP( hipMemsetAsync( count_dev, 0, sizeof(uint32_t), s ) );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( read_flags_dev && queue_out_dev, "NULL device pointer" );
    NVB_REQUIRE( n < (1u << 31), "n too large" );
    const ReadFlagSet pred = { read_flags_dev };
    size_t bytes = 0; void* tmp = nullptr;
    hipcub::CountingInputIterator<uint32_t> ids( 0u );
    if (queue_dev) NVB_HIP( hipcub::DeviceSelect::If( nullptr, bytes, queue_dev, queue_out_dev, count_dev, (int)n, pred, s ) );
    else           NVB_HIP( hipcub::DeviceSelect::If( nullptr, bytes, ids, queue_out_dev, count_dev, (int)n, pred, s ) );
    if (scratch_alloc( &tmp, bytes ? bytes : 16, s ) != hipSuccess) { (void)hipGetLastError(); set_error( "read_queue_filter: out of device memory" ); return NVBIO_ERR_NOMEM; }
    const hipError_t e = queue_dev ? hipcub::DeviceSelect::If( tmp, bytes, queue_dev, queue_out_dev, count_dev, (int)n, pred, s )
                                   : hipcub::DeviceSelect::If( tmp, bytes, ids, queue_out_dev, count_dev, (int)n, pred, s );
    scratch_free( tmp, s );
    if (e != hipSuccess) { set_error( "read_queue_filter failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    return NVBIO_OK;
}

nvbio_status nvbio_select_flagged_indices(int device, const uint8_t* flags_dev, uint32_t n, uint32_t* queue_out_dev, uint32_t* count_dev, void* stream)
{
    NVB_REQUIRE( count_dev != nullptr, "count_dev is NULL" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    NVB_HIP( hipMemsetAsync( count_dev, 0, sizeof(uint32_t), s ) );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( flags_dev && queue_out_dev, "NULL device pointer" );
    NVB_REQUIRE( n < (1u << 31), "n too large" );
    size_t bytes = 0; void* tmp = nullptr;
    hipcub::CountingInputIterator<uint32_t> ids( 0u );
    NVB_HIP( hipcub::DeviceSelect::Flagged( nullptr, bytes, ids, flags_dev, queue_out_dev, count_dev, (int)n, s ) );
    if (scratch_alloc( &tmp, bytes ? bytes : 16, s ) != hipSuccess) { (void)hipGetLastError(); set_error( "select_flagged_indices: out of device memory" ); return NVBIO_ERR_NOMEM; }
    const hipError_t e = hipcub::DeviceSelect::Flagged( tmp, bytes, ids, flags_dev, queue_out_dev, count_dev, (int)n, s );
    scratch_free( tmp, s );
    if (e != hipSuccess) { set_error( "select_flagged_indices failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    return NVBIO_OK;
}

} // extern "C"
