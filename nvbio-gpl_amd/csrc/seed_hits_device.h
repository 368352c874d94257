// seed_hits_device.h -- nvBowtie's per-read seed-hit deque on the device: the interval heap under priority_deque<SeedHit, ., hit_compare>
// (nvBowtie/bowtie2/cuda/seed_hit.h:45-229, nvbio/basic/priority_deque.h, interval_heap.h).  See seed_hits.hip for the layout and for why
// the moves are the reference container's, one for one.
#pragma once
#include "common.h"

namespace nvbio_amd {

struct HitHeap
{
    uint2*   a;          // word x = range_begin, word y = range_delta:20 | pos:10 | rc:1 | indexdir:1
    uint32_t n;

    __device__ __forceinline__ static uint32_t size_of(const uint2 h) { return h.y & 0xFFFFFu; }
    // hit_compare: f goes before s iff f's range is larger
    __device__ __forceinline__ bool before(const uint32_t i, const uint32_t j) const { return size_of( a[i] ) > size_of( a[j] ); }
    __device__ __forceinline__ void exchange(const uint32_t i, const uint32_t j) { const uint2 t = a[i]; a[i] = a[j]; a[j] = t; }

    // climb from element i along the low ends (towards element 0) or the high ends (towards element 1)
    __device__ void climb(uint32_t i, const bool low)
    {
        while (i >= 2u)
        {
            const uint32_t up = ((i / 2u - 1u) | 1u) ^ (low ? 1u : 0u);
            const bool move = low ? before( i, up ) : before( up, i );
            if (!move) break;
            exchange( i, up ); i = up;
        }
    }
    // a leaf settles against the other end of its interval (or of its parent's, for a lone low end), then climbs
    __device__ void settle_high(const uint32_t i)
    {
        const uint32_t other = (2u * i < n) ? 2u * i : (i ^ 1u);
        if (before( i, other )) { exchange( i, other ); climb( other, true ); }
        else climb( i, false );
    }
    __device__ void settle_low(const uint32_t i)
    {
        uint32_t other = i | 1u;
        if (other >= n)
        {
            if (other == 1u) return;
            other = (other / 2u - 1u) | 1u;
        }
        if (before( other, i )) { exchange( i, other ); climb( other, false ); }
        else climb( i, true );
    }
    // the element at i sinks to a leaf -- along the first-ordered low children, or the last-ordered high ones -- and settles there
    __device__ void sink(uint32_t i, const bool low)
    {
        const int32_t two_children_end = (int32_t)(n / 2u) - ((low && (n & 3u) == 0u) ? 2 : 1);
        while ((int32_t)i < two_children_end)
        {
            uint32_t c = 2u * i + (low ? 2u : 1u);
            if (low ? before( c + 2u, c ) : before( c, c + 2u )) c += 2u;
            exchange( i, c ); i = c;
        }
        if ((int32_t)i <= two_children_end + (low ? 0 : 1))
        {
            uint32_t c = 2u * i + (low ? 2u : 1u);
            if (c < n)
            {
                if (!low && c + 1u < n && before( c, c + 1u ))
                {
                    ++c; exchange( i, c );
                    settle_low( c );
                    return;
                }
                exchange( i, c ); i = c;
            }
        }
        if (low) settle_low( i ); else settle_high( i );
    }

    __device__ void push(const uint2 h)
    {
        a[n] = h; ++n;
        if ((n - 1u) & 1u) settle_high( n - 1u ); else settle_low( n - 1u );
    }
    __device__ void pop_bottom()                     // the largest range goes
    {
        --n;
        exchange( 0u, n );
        sink( 0u, true );
    }
    __device__ void pop_top()                        // the smallest range goes
    {
        if (n > 2u)
        {
            exchange( 1u, n - 1u );
            --n;
            sink( 1u, false );
        }
        else --n;
    }
    __device__ __forceinline__ uint32_t top() const { return n > 1u ? 1u : 0u; }
};

// one more hit under the mapper's rule: a full deque drops its largest range first (mapping_inl.h:242-244)
__device__ __forceinline__ void push_seed_hit(HitHeap& heap, const uint32_t max_hits, const uint32_t x, const uint32_t y, const uint32_t flags,
                                              uint32_t& range_sum, uint32_t& range_count)
{
    if (heap.n == max_hits) heap.pop_bottom();
    heap.push( make_uint2( x, ((y + 1u - x) & 0xFFFFFu) | flags ) );               // SeedHit( flags, inclusive_to_exclusive( range ) )
    range_sum += y - x + 1u; ++range_count;
}

} // namespace nvbio_amd
