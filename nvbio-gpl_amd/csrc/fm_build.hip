// fm_build.hip -- FM-index construction on the GPU (gfx950): suffix sort -> BWT -> occ -> SSA.
//
// What is built is exactly the index the reference loads / builds on the host:
//   suffix array convention (row 0 = empty suffix)     nvbio/fmindex/bwt.h:28-37
//   BWT with the primary row squeezed out               nvbio/fmindex/bwt.h:41-53
//   occ table, K = 64                                   nvbio/fmindex/rank_dictionary_inl.h:33-66
//   interleaved 32-byte records                         nvbio/io/fmindex/fmindex_impl.cu:300-313
//   SSA_index_multiple<16>, entry 0 = -1                nvbio/fmindex/ssa_inl.h:254-301
// The reference does this offline with a CPU/GPU BWT builder (nvBWT, nvbio/sufsort/, 9.8k LoC,
// out of scope) plus a serial host pass; this file exists so that a 3 Gbp synthetic reference
// can be indexed inside the benchmark in seconds, sized for 288 GB of HBM:
//
//   1. every suffix gets a 64-bit key = its first 32 symbols (zero padded); suffixes are dealt
//      into 4^b buckets by their first b symbols and each bucket is radix-sorted (rocPRIM) as
//      (key, position) pairs -- for an i.i.d. 3 Gbp text this already orders all but a handful
//      of suffixes (expected number of pairs sharing 32 symbols: n^2 / 2 / 4^32 < 1);
//   2. suffixes whose keys tie are finished by prefix doubling restricted to the tied segments
//      (Manber-Myers on the unresolved set only, sort key = (segment, rank[i+h])), with the
//      end-of-text rule "a proper prefix sorts first";
//   3. BWT words, per-block symbol counts, their exclusive scan (occ), L2 and the SSA are one
//      pass each over the sorted positions.
#include "fm_device.h"
#include <rocprim/rocprim.hpp>
#include <vector>
#include <stdlib.h>
#include <stdio.h>

namespace nvbio_amd {

nvbio_status fm_index_adopt(const nvbio_fm_index_view* view, int device, uint32_t kmer_len, bool owns, hipStream_t stream, nvbio_fm_index_t* out,
                            uint32_t* isa, uint32_t* text, uint32_t table_flags);

namespace {

// ---- packed text access ---------------------------------------------------------------------
struct TextView
{
    const uint32_t* words;
    uint32_t        n;
    uint32_t        n_words;

    __device__ __forceinline__ uint32_t word(const uint32_t w) const { return w < n_words ? words[w] : 0u; }
    __device__ __forceinline__ uint32_t symbol(const uint32_t i) const { return (words[i >> 4] >> (30u - 2u * (i & 15u))) & 3u; }

    // first 32 symbols of suffix i as a big-endian 64-bit integer, zero padded past the end
    __device__ __forceinline__ uint64_t key(const uint32_t i) const
    {
        const uint32_t w = i >> 4, sh = 2u * (i & 15u);
        const uint64_t a = ((uint64_t)word( w ) << 32) | word( w + 1u );
        const uint64_t b = ((uint64_t)word( w + 2u ) << 32);
        uint64_t k = sh ? ((a << sh) | (b >> (64u - sh))) : a;
        const uint32_t valid = n - i;                           // symbols available (i < n)
        if (valid < 32u) k &= ~0ull << (64u - 2u * valid);
        return k;
    }
};

struct KeyOf
{
    TextView t;
    __device__ __forceinline__ uint64_t operator()(const uint32_t i) const { return t.key( i ); }
};
struct InBucket
{
    TextView t; uint32_t shift; uint32_t bucket;
    __device__ __forceinline__ bool operator()(const uint32_t i) const { return (uint32_t)(t.key( i ) >> shift) == bucket; }
};

// ---- kernels --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bucket_histogram_kernel(const TextView t, const uint32_t shift, uint32_t* __restrict__ hist /* per block partials avoided: atomics on 4^b counters */)
{
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd( &s_hist[(uint32_t)(t.key( (uint32_t)i ) >> shift)], 1u );
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd( &hist[threadIdx.x], s_hist[threadIdx.x] );
}

// head[s] = s if element s starts a new key segment else 0 (s = 0 is a head with value 0)
__global__ void __launch_bounds__(256)
head_from_keys_kernel(const uint64_t* __restrict__ keys, const uint64_t n, uint32_t* __restrict__ head)
{
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x)
        head[s] = (s == 0 || keys[s] != keys[s - 1]) ? (uint32_t)s : 0u;
}
__global__ void patch_first_kernel(uint32_t* p, const uint32_t* carry) { if (*carry > *p) *p = *carry; }

// unresolved[s] = 1 iff the segment of slot s has more than one element (seg = head slot of s)
struct Unresolved
{
    const uint32_t* seg; uint64_t n;
    __device__ __forceinline__ bool operator()(const uint32_t s) const
    {
        const bool head = (seg[s] == s);
        const bool next_head = ((uint64_t)s + 1 >= n) || (seg[s + 1] == s + 1u);
        return !(head && next_head);
    }
};

__global__ void __launch_bounds__(256)
scatter_rank_kernel(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ seg, const uint64_t n, uint32_t* __restrict__ rank)
{
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x)
        rank[sa[s]] = seg[s];
}

// doubling round: key2 = (segment << 32) | ext_rank(i + h) for every unresolved slot
__global__ void __launch_bounds__(256)
doubling_keys_kernel(const uint32_t* __restrict__ U, const uint32_t m, const uint32_t* __restrict__ sa, const uint32_t* __restrict__ seg,
                     const uint32_t* __restrict__ rank, const uint32_t n, const uint32_t h,
                     uint64_t* __restrict__ key2, uint32_t* __restrict__ val)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x)
    {
        const uint32_t s = U[k], i = sa[s];
        const uint64_t p = (uint64_t)i + h;
        // past the end: a shorter suffix (larger p) is smaller, and all of them sort before any real rank
        const uint32_t ext = (p < n) ? rank[p] + h : (h - 1u) - (uint32_t)(p - n);
        key2[k] = ((uint64_t)seg[s] << 32) | ext;
        val[k]  = i;
    }
}
// after sorting (key2,val): place values back into the slots, flag the new segment heads
__global__ void __launch_bounds__(256)
doubling_place_kernel(const uint32_t* __restrict__ U, const uint32_t m, const uint64_t* __restrict__ key2, const uint32_t* __restrict__ val,
                      uint32_t* __restrict__ sa, uint32_t* __restrict__ newhead /* m entries */)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x)
    {
        sa[U[k]]   = val[k];
        newhead[k] = (k == 0 || key2[k] != key2[k - 1]) ? U[k] : 0u;
    }
}
// newhead has been max-scanned: it now holds the head SLOT of every unresolved element
__global__ void __launch_bounds__(256)
doubling_update_kernel(const uint32_t* __restrict__ U, const uint32_t m, const uint32_t* __restrict__ newseg, const uint32_t* __restrict__ val,
                       uint32_t* __restrict__ seg, uint32_t* __restrict__ rank)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x)
    {
        seg[U[k]]    = newseg[k];
        rank[val[k]] = newseg[k];
    }
}
struct StillUnresolvedU
{
    const uint32_t* U; const uint32_t* newseg; uint32_t m;
    __device__ __forceinline__ bool operator()(const uint32_t k) const
    {
        const bool head = (newseg[k] == U[k]);
        const bool next_head = (k + 1u >= m) || (newseg[k + 1] == U[k + 1]);
        return !(head && next_head);
    }
};
__global__ void __launch_bounds__(256)
gather_u_kernel(const uint32_t* __restrict__ U, const uint32_t* __restrict__ sel, const uint32_t m2, uint32_t* __restrict__ U2)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m2; k += gridDim.x * blockDim.x) U2[k] = U[sel[k]];
}

// inverse suffix array over the rows of the full BWT matrix: isa[sa[s]] = s + 1, isa[n] = 0
__global__ void __launch_bounds__(256)
isa_kernel(const uint32_t* __restrict__ sa, const uint64_t n, uint32_t* __restrict__ isa)
{
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= n; s += (uint64_t)gridDim.x * blockDim.x)
    {
        if (s == n) isa[n] = 0u;
        else        isa[sa[s]] = (uint32_t)s + 1u;
    }
}

__global__ void __launch_bounds__(256)
find_primary_kernel(const uint32_t* __restrict__ sa, const uint64_t n, uint32_t* __restrict__ primary)
{
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x)
        if (sa[s] == 0) *primary = (uint32_t)s + 1u;
}

// one thread per BWT word (16 symbols): bwt[k] = T[SA_full[row]-1], row = k < primary ? k : k+1
__global__ void __launch_bounds__(256)
bwt_words_kernel(const TextView t, const uint32_t* __restrict__ sa, const uint32_t* __restrict__ primary_p,
                 const uint32_t n_words_padded, uint32_t* __restrict__ bwt_occ, uint4* __restrict__ block_cnt)
{
    const uint32_t primary = *primary_p;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words_padded; w += gridDim.x * blockDim.x)
    {
        uint32_t word = 0;
        #pragma unroll 4
        for (uint32_t r = 0; r < 16; ++r)
        {
            const uint64_t k = (uint64_t)w * 16u + r;
            if (k < t.n)
            {
                const uint64_t row = (k < primary) ? k : k + 1u;
                const uint32_t i   = (row == 0) ? t.n : sa[row - 1u];
                word |= t.symbol( i - 1u ) << (30u - 2u * r);
            }
        }
        bwt_occ[(size_t)(w >> 2) * 8u + (w & 3u)] = word;
    }
}
// per 64-symbol block: number of A,C,G,T among its valid symbols
__global__ void __launch_bounds__(256)
block_counts_kernel(const uint32_t* __restrict__ bwt_occ, const uint32_t n, const uint32_t n_blocks, uint4* __restrict__ cnt)
{
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += gridDim.x * blockDim.x)
    {
        const uint4 w = ((const uint4*)bwt_occ)[2u * b];
        const uint64_t rem = (uint64_t)n - (uint64_t)b * 64u;                 // valid symbols from this block on
        uint4 c = make_uint4( 0, 0, 0, 0 );
        if (rem > 0)
        {
            const uint32_t p = rem >= 64u ? 63u : (uint32_t)rem - 1u;
            c = count4_in_block( w, p );
        }
        cnt[b] = c;
    }
}
struct AddU4
{
    __device__ __host__ __forceinline__ uint4 operator()(const uint4 a, const uint4 b) const
    { return make_uint4( a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w ); }
};
struct MaxU32
{
    __device__ __host__ __forceinline__ uint32_t operator()(const uint32_t a, const uint32_t b) const { return a > b ? a : b; }
};
__global__ void __launch_bounds__(256)
write_occ_kernel(const uint4* __restrict__ occ, const uint32_t n_blocks, uint32_t* __restrict__ bwt_occ)
{
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += gridDim.x * blockDim.x)
        ((uint4*)bwt_occ)[2u * b + 1u] = occ[b];
}
__global__ void __launch_bounds__(256)
ssa_kernel(const uint32_t* __restrict__ sa, const uint32_t sa_int, const uint64_t n_ssa, uint32_t* __restrict__ ssa)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_ssa; j += (uint64_t)gridDim.x * blockDim.x)
        ssa[j] = (j == 0) ? 0xFFFFFFFFu : sa[(size_t)sa_int * j - 1u];  // row sa_int*j of the full SA
}

// ---- small RAII for device scratch ----------------------------------------------------------
struct Scratch
{
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree( p ); }
    template <typename T> T* alloc(size_t count)
    {
        void* p = nullptr;
        if (hipMalloc( &p, (count ? count : 1) * sizeof(T) ) != hipSuccess) return nullptr;
        ptrs.push_back( p );
        return (T*)p;
    }
    void release(void* p)
    {
        for (size_t i = 0; i < ptrs.size(); ++i) if (ptrs[i] == p) { (void)hipFree( p ); ptrs.erase( ptrs.begin() + i ); return; }
    }
    void forget(void* p)
    {
        for (size_t i = 0; i < ptrs.size(); ++i) if (ptrs[i] == p) { ptrs.erase( ptrs.begin() + i ); return; }
    }
};

#define NVB_ALLOC(var, T, count)                                                                   \
    T* var = scratch.alloc<T>( count );                                                            \
    if (!var) { set_error( "index build: out of device memory (%s, %zu bytes)", #var, (size_t)(count) * sizeof(T) ); return NVBIO_ERR_NOMEM; }

// inclusive max-scan in place, in chunks small enough for 32-bit-sized device primitives
static nvbio_status scan_max_inplace(uint32_t* buf, uint64_t n, Scratch& scratch, hipStream_t s)
{
    const uint64_t CHUNK = 1ull << 30;
    size_t temp_bytes = 0;
    NVB_HIP( rocprim::inclusive_scan( nullptr, temp_bytes, buf, buf, (size_t)(n < CHUNK ? n : CHUNK), MaxU32(), s ) );
    NVB_ALLOC( temp, uint8_t, temp_bytes );
    for (uint64_t b = 0; b < n; b += CHUNK)
    {
        const size_t len = (size_t)((n - b) < CHUNK ? (n - b) : CHUNK);
        if (b) hipLaunchKernelGGL( patch_first_kernel, dim3(1), dim3(1), 0, s, buf + b, buf + b - 1 );
        NVB_HIP( rocprim::inclusive_scan( temp, temp_bytes, buf + b, buf + b, len, MaxU32(), s ) );
    }
    scratch.release( temp );
    return NVBIO_OK;
}

// out[0..count) = { i in [0,n) : pred(i) } in increasing order, chunked; *count on the host
template <typename Pred>
static nvbio_status select_indices(const uint64_t n, Pred pred, uint32_t* out, uint64_t* count, Scratch& scratch, hipStream_t s)
{
    const uint64_t CHUNK = 1ull << 30;
    NVB_ALLOC( d_cnt, size_t, 1 );
    size_t temp_bytes = 0;
    NVB_HIP( rocprim::select( nullptr, temp_bytes, rocprim::counting_iterator<uint32_t>( 0 ), out, d_cnt,
                              (size_t)(n < CHUNK ? n : CHUNK), pred, s ) );
    NVB_ALLOC( temp, uint8_t, temp_bytes );
    uint64_t total = 0;
    for (uint64_t b = 0; b < n; b += CHUNK)
    {
        const size_t len = (size_t)((n - b) < CHUNK ? (n - b) : CHUNK);
        NVB_HIP( rocprim::select( temp, temp_bytes, rocprim::counting_iterator<uint32_t>( (uint32_t)b ), out + total, d_cnt, len, pred, s ) );
        size_t c = 0;
        NVB_HIP( hipMemcpyAsync( &c, d_cnt, sizeof(size_t), hipMemcpyDeviceToHost, s ) );
        NVB_HIP( hipStreamSynchronize( s ) );
        total += c;
    }
    *count = total;
    scratch.release( temp ); scratch.release( d_cnt );
    return NVBIO_OK;
}

static nvbio_status sort_pairs(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n,
                               unsigned begin_bit, unsigned end_bit, Scratch& scratch, hipStream_t s)
{
    size_t temp_bytes = 0;
    NVB_HIP( rocprim::radix_sort_pairs( nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, s ) );
    NVB_ALLOC( temp, uint8_t, temp_bytes );
    NVB_HIP( rocprim::radix_sort_pairs( temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, s ) );
    NVB_HIP( hipStreamSynchronize( s ) );
    scratch.release( temp );
    return NVBIO_OK;
}

static nvbio_status build_impl(const uint32_t* text2_dev, const uint32_t n, const int device, const uint32_t kmer_len,
                               const uint32_t sa_int, uint32_t max_lcp, const bool verify, const uint32_t table_flags, const uint32_t bucket_symbols,
                               hipStream_t s, nvbio_fm_index_t* out)
{
    Scratch scratch;
    if (max_lcp == 0) max_lcp = 4096;
    if (max_lcp > (1u << 20)) max_lcp = 1u << 20;
    if (max_lcp > 0xFFFFFFFFu - n - 64u) max_lcp = 0xFFFFFFFFu - n - 64u;       // rank + h must not wrap

    TextView t; t.words = text2_dev; t.n = n; t.n_words = (n + 15u) / 16u;

    // ---- 1. bucketed sort of (32-mer key, position) ------------------------------------------
    uint32_t bsym = 0;                                          // symbols used for bucketing (<= 4 -> <= 256 buckets)
    while (bsym < 4 && ((uint64_t)n >> (2 * bsym)) > (1ull << 28)) ++bsym;
    if (bucket_symbols) bsym = bucket_symbols - 1u;              // nvbio_fm_build_options::bucket_symbols: force the bucketed path on small texts (tests)
    const uint32_t n_buckets = 1u << (2 * bsym);
    const uint32_t bshift    = 64u - 2u * bsym;

    std::vector<uint32_t> hist( 256, 0 );
    if (bsym)
    {
        NVB_ALLOC( d_hist, uint32_t, 256 );
        NVB_HIP( hipMemsetAsync( d_hist, 0, 256 * sizeof(uint32_t), s ) );
        hipLaunchKernelGGL( bucket_histogram_kernel, dim3( grid_for( n ) ), dim3(256), 0, s, t, bshift, d_hist );
        NVB_HIP( hipMemcpyAsync( hist.data(), d_hist, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, s ) );
        NVB_HIP( hipStreamSynchronize( s ) );
        scratch.release( d_hist );
    }
    else hist[0] = n;
    uint32_t max_bucket = 0;
    for (uint32_t b = 0; b < n_buckets; ++b) if (hist[b] > max_bucket) max_bucket = hist[b];
    if (max_bucket > (1u << 31))
    {
        set_error( "index build: a %u-symbol prefix bucket holds %u suffixes (text too skewed)", bsym, max_bucket );
        return NVBIO_ERR_UNSUPPORTED;
    }

    NVB_ALLOC( sa,   uint32_t, n );          // sorted suffix positions (rows 1..n of the full SA)
    NVB_ALLOC( keys, uint64_t, n );          // their sorted keys
    {
        NVB_ALLOC( b_idx,  uint32_t, max_bucket );
        NVB_ALLOC( b_keys, uint64_t, max_bucket );
        uint64_t offset = 0;
        for (uint32_t b = 0; b < n_buckets; ++b)
        {
            const uint32_t nb = hist[b];
            if (nb == 0) continue;
            uint32_t* idx_in = b_idx;
            if (bsym)
            {
                uint64_t cnt = 0;
                InBucket pred; pred.t = t; pred.shift = bshift; pred.bucket = b;
                NVB_CHECK( select_indices( n, pred, b_idx, &cnt, scratch, s ) );
                if (cnt != nb) { set_error( "index build: bucket %u size mismatch (%llu vs %u)", b, (unsigned long long)cnt, nb ); return NVBIO_ERR_HIP; }
                KeyOf kf; kf.t = t;
                NVB_HIP( rocprim::transform( b_idx, b_keys, (size_t)nb, kf, s ) );
                NVB_CHECK( sort_pairs( b_keys, keys + offset, idx_in, sa + offset, nb, 0, bshift, scratch, s ) );
            }
            else
            {
                // single bucket: keys and positions straight from iterators, no staging copy
                KeyOf kf; kf.t = t;
                auto vals_in = rocprim::counting_iterator<uint32_t>( 0 );
                auto keys_in = rocprim::make_transform_iterator( vals_in, kf );
                size_t temp_bytes = 0;
                NVB_HIP( rocprim::radix_sort_pairs( nullptr, temp_bytes, keys_in, keys, vals_in, sa, (size_t)n, 0u, 64u, s ) );
                NVB_ALLOC( temp, uint8_t, temp_bytes );
                NVB_HIP( rocprim::radix_sort_pairs( temp, temp_bytes, keys_in, keys, vals_in, sa, (size_t)n, 0u, 64u, s ) );
                NVB_HIP( hipStreamSynchronize( s ) );
                scratch.release( temp );
            }
            offset += nb;
        }
        scratch.release( b_idx ); scratch.release( b_keys );
    }

    // ---- 2. finish tied suffixes by prefix doubling on the unresolved set --------------------
    NVB_ALLOC( seg, uint32_t, n );           // head slot of the segment each slot belongs to
    hipLaunchKernelGGL( head_from_keys_kernel, dim3( grid_for( n ) ), dim3(256), 0, s, (const uint64_t*)keys, (uint64_t)n, seg );
    NVB_HIP( hipGetLastError() );
    scratch.release( keys );
    NVB_CHECK( scan_max_inplace( seg, n, scratch, s ) );

    uint64_t m = 0;
    NVB_ALLOC( U, uint32_t, n );             // unresolved slots (upper bound n; usually tiny)
    {
        Unresolved pred; pred.seg = seg; pred.n = n;
        NVB_CHECK( select_indices( n, pred, U, &m, scratch, s ) );
    }
    if (m > 0)
    {
        NVB_ALLOC( rank, uint32_t, n );
        hipLaunchKernelGGL( scatter_rank_kernel, dim3( grid_for( n ) ), dim3(256), 0, s, (const uint32_t*)sa, (const uint32_t*)seg, (uint64_t)n, rank );
        NVB_ALLOC( key2,   uint64_t, m );
        NVB_ALLOC( key2s,  uint64_t, m );
        NVB_ALLOC( val,    uint32_t, m );
        NVB_ALLOC( vals,   uint32_t, m );
        NVB_ALLOC( newseg, uint32_t, m );
        NVB_ALLOC( sel,    uint32_t, m );
        NVB_ALLOC( U2,     uint32_t, m );
        uint32_t h = 32;
        while (m > 0)
        {
            if (h > max_lcp)
            {
                set_error( "index build: %llu suffixes still tie after %u symbols (max_lcp = %u)", (unsigned long long)m, h, max_lcp );
                return NVBIO_ERR_UNSUPPORTED;
            }
            const uint32_t mm = (uint32_t)m;
            hipLaunchKernelGGL( doubling_keys_kernel, dim3( grid_for( mm ) ), dim3(256), 0, s,
                                (const uint32_t*)U, mm, (const uint32_t*)sa, (const uint32_t*)seg, (const uint32_t*)rank, n, h, key2, val );
            NVB_CHECK( sort_pairs( key2, key2s, val, vals, mm, 0, 64, scratch, s ) );
            hipLaunchKernelGGL( doubling_place_kernel, dim3( grid_for( mm ) ), dim3(256), 0, s,
                                (const uint32_t*)U, mm, (const uint64_t*)key2s, (const uint32_t*)vals, sa, newseg );
            NVB_CHECK( scan_max_inplace( newseg, mm, scratch, s ) );
            hipLaunchKernelGGL( doubling_update_kernel, dim3( grid_for( mm ) ), dim3(256), 0, s,
                                (const uint32_t*)U, mm, (const uint32_t*)newseg, (const uint32_t*)vals, seg, rank );
            NVB_HIP( hipGetLastError() );
            uint64_t m2 = 0;
            StillUnresolvedU pred; pred.U = U; pred.newseg = newseg; pred.m = mm;
            NVB_CHECK( select_indices( mm, pred, sel, &m2, scratch, s ) );
            if (m2)
            {
                hipLaunchKernelGGL( gather_u_kernel, dim3( grid_for( m2 ) ), dim3(256), 0, s, (const uint32_t*)U, (const uint32_t*)sel, (uint32_t)m2, U2 );
                NVB_HIP( hipMemcpyAsync( U, U2, m2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s ) );
            }
            m = m2;
            if (h > (1u << 30)) break;
            h *= 2;
        }
        scratch.release( rank ); scratch.release( key2 ); scratch.release( key2s ); scratch.release( val );
        scratch.release( vals ); scratch.release( newseg ); scratch.release( sel ); scratch.release( U2 );
    }
    scratch.release( U ); scratch.release( seg );

    // ---- 3. BWT, occ, L2, SSA ---------------------------------------------------------------
    const uint32_t words    = ((t.n_words + 3u) & ~3u);          // padded to whole 64-symbol blocks
    const uint32_t n_blocks = words / 4u;
    const uint64_t n_ssa    = (uint64_t)n / sa_int + 1u;

    NVB_ALLOC( d_primary, uint32_t, 1 );
    NVB_HIP( hipMemsetAsync( d_primary, 0, sizeof(uint32_t), s ) );
    hipLaunchKernelGGL( find_primary_kernel, dim3( grid_for( n ) ), dim3(256), 0, s, (const uint32_t*)sa, (uint64_t)n, d_primary );

    NVB_ALLOC( bwt_occ, uint32_t, (size_t)words * 2u );
    NVB_ALLOC( ssa,     uint32_t, n_ssa );
    hipLaunchKernelGGL( bwt_words_kernel, dim3( grid_for( words ) ), dim3(256), 0, s, t, (const uint32_t*)sa, (const uint32_t*)d_primary, words, bwt_occ, (uint4*)nullptr );
    hipLaunchKernelGGL( ssa_kernel, dim3( grid_for( n_ssa ) ), dim3(256), 0, s, (const uint32_t*)sa, sa_int, n_ssa, ssa );
    uint32_t* isa = nullptr; uint32_t* text_copy = nullptr;
    if (sa_int == 1)                                             // the full SA + the text: nvbio_fm_match_direct can finish on the text
    {
        NVB_ALLOC( txt_, uint32_t, (size_t)t.n_words + 4u );
        NVB_HIP( hipMemsetAsync( txt_ + t.n_words, 0, 4u * sizeof(uint32_t), s ) );
        NVB_HIP( hipMemcpyAsync( txt_, text2_dev, (size_t)t.n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, s ) );
        text_copy = txt_;
    }
    if (verify)
    {
        NVB_ALLOC( isa_, uint32_t, (size_t)n + 1u );
        hipLaunchKernelGGL( isa_kernel, dim3( grid_for( (uint64_t)n + 1u ) ), dim3(256), 0, s, (const uint32_t*)sa, (uint64_t)n, isa_ );
        isa = isa_;
    }
    NVB_HIP( hipGetLastError() );
    NVB_HIP( hipStreamSynchronize( s ) );
    scratch.release( sa );

    NVB_ALLOC( cnt, uint4, (size_t)n_blocks + 1u );
    NVB_ALLOC( occ, uint4, (size_t)n_blocks + 1u );
    hipLaunchKernelGGL( block_counts_kernel, dim3( grid_for( n_blocks ) ), dim3(256), 0, s, (const uint32_t*)bwt_occ, n, n_blocks, cnt );
    NVB_HIP( hipMemsetAsync( cnt + n_blocks, 0, sizeof(uint4), s ) );                  // extra entry: its scan value is the total
    {
        size_t temp_bytes = 0;
        NVB_HIP( rocprim::exclusive_scan( nullptr, temp_bytes, cnt, occ, make_uint4( 0, 0, 0, 0 ), (size_t)n_blocks + 1u, AddU4(), s ) );
        NVB_ALLOC( temp, uint8_t, temp_bytes );
        NVB_HIP( rocprim::exclusive_scan( temp, temp_bytes, cnt, occ, make_uint4( 0, 0, 0, 0 ), (size_t)n_blocks + 1u, AddU4(), s ) );
        NVB_HIP( hipStreamSynchronize( s ) );
        scratch.release( temp );
    }
    hipLaunchKernelGGL( write_occ_kernel, dim3( grid_for( n_blocks ) ), dim3(256), 0, s, (const uint4*)occ, n_blocks, bwt_occ );
    uint4    totals;
    uint32_t primary = 0;
    NVB_HIP( hipMemcpyAsync( &totals, occ + n_blocks, sizeof(uint4), hipMemcpyDeviceToHost, s ) );
    NVB_HIP( hipMemcpyAsync( &primary, d_primary, sizeof(uint32_t), hipMemcpyDeviceToHost, s ) );
    NVB_HIP( hipStreamSynchronize( s ) );

    nvbio_fm_index_view view;
    view.length  = n;
    view.primary = primary;
    view.L2[0] = 0; view.L2[1] = totals.x; view.L2[2] = view.L2[1] + totals.y; view.L2[3] = view.L2[2] + totals.z; view.L2[4] = view.L2[3] + totals.w;
    view.bwt_occ_dev = bwt_occ; view.bwt_occ_words = (uint64_t)words * 2u;
    view.ssa_dev = ssa;         view.ssa_words = n_ssa;   view.sa_int = sa_int;
    if (view.L2[4] != n || primary == 0)
    {
        set_error( "index build: inconsistent result (sum of counts %u, n %u, primary %u)", view.L2[4], n, primary );
        return NVBIO_ERR_HIP;
    }
    scratch.forget( bwt_occ ); scratch.forget( ssa );           // ownership moves to the handle
    if (isa) scratch.forget( isa );
    if (text_copy) scratch.forget( text_copy );
    return fm_index_adopt( &view, device, kmer_len, true, s, out, isa, text_copy, table_flags );
}

// BWT words (plain, 16 symbols per word) <-> the BWT half of the interleaved 32-byte records
__global__ void __launch_bounds__(256)
interleave_bwt_kernel(const uint32_t* __restrict__ bwt, const uint32_t n_words, const uint32_t n_words_padded, uint32_t* __restrict__ bwt_occ)
{
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words_padded; w += gridDim.x * blockDim.x)
        bwt_occ[(size_t)(w >> 2) * 8u + (w & 3u)] = w < n_words ? bwt[w] : 0u;
}
__global__ void __launch_bounds__(256)
deinterleave_bwt_kernel(const uint32_t* __restrict__ bwt_occ, const uint32_t n_words, uint32_t* __restrict__ bwt)
{
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += gridDim.x * blockDim.x)
        bwt[w] = bwt_occ[(size_t)(w >> 2) * 8u + (w & 3u)];
}

struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose( f ); } };

static nvbio_status load_impl(const char* bwt_path, const char* sa_path, const int device, const uint32_t kmer_len,
                              hipStream_t s, nvbio_fm_index_t* out)
{
    Scratch scratch;
    // ---- .bwt: primary, cumulative counts (the last is the length), packed words (fmindex_impl.cu:111-170) ----
    FileCloser bf = { fopen( bwt_path, "rb" ) };
    if (!bf.f) { set_error( "unable to open bwt \"%s\"", bwt_path ); return NVBIO_ERR_INVALID; }
    uint32_t hdr[5];
    if (fread( hdr, sizeof(uint32_t), 5, bf.f ) != 5) { set_error( "failed reading bwt header \"%s\"", bwt_path ); return NVBIO_ERR_INVALID; }
    const uint32_t primary = hdr[0], n = hdr[4];
    if (n == 0 || primary > n) { set_error( "bad bwt header in \"%s\" (length %u, primary %u)", bwt_path, n, primary ); return NVBIO_ERR_INVALID; }
    const uint32_t n_words = (n + 15u) / 16u;
    const uint32_t words   = (n_words + 3u) & ~3u;
    const uint32_t n_blocks = words / 4u;
    std::vector<uint32_t> h_bwt( words, 0u );
    const size_t got = fread( h_bwt.data(), sizeof(uint32_t), words, bf.f );
    if (((got + 3u) & ~(size_t)3u) != words) { set_error( "failed reading bwt \"%s\" (%zu of %u words)", bwt_path, got, n_words ); return NVBIO_ERR_INVALID; }
    // clear whatever follows the last symbol (the reference's builder leaves a stale (n+1)-th symbol there)
    if (n & 15u) h_bwt[n_words - 1u] &= ~0u << (32u - 2u * (n & 15u));
    for (uint32_t w = n_words; w < words; ++w) h_bwt[w] = 0u;

    NVB_ALLOC( d_bwt, uint32_t, words );
    NVB_ALLOC( bwt_occ, uint32_t, (size_t)words * 2u );
    NVB_HIP( hipMemcpyAsync( d_bwt, h_bwt.data(), (size_t)words * sizeof(uint32_t), hipMemcpyHostToDevice, s ) );
    hipLaunchKernelGGL( interleave_bwt_kernel, dim3( grid_for( words ) ), dim3(256), 0, s, (const uint32_t*)d_bwt, n_words, words, bwt_occ );

    // ---- occurrence table (K = 64) on the GPU: per-record counts, exclusive scan, interleave (fmindex_impl.cu:254-331) ----
    NVB_ALLOC( cnt, uint4, (size_t)n_blocks + 1u );
    NVB_ALLOC( occ, uint4, (size_t)n_blocks + 1u );
    hipLaunchKernelGGL( block_counts_kernel, dim3( grid_for( n_blocks ) ), dim3(256), 0, s, (const uint32_t*)bwt_occ, n, n_blocks, cnt );
    NVB_HIP( hipMemsetAsync( cnt + n_blocks, 0, sizeof(uint4), s ) );
    {
        size_t temp_bytes = 0;
        NVB_HIP( rocprim::exclusive_scan( nullptr, temp_bytes, cnt, occ, make_uint4( 0, 0, 0, 0 ), (size_t)n_blocks + 1u, AddU4(), s ) );
        NVB_ALLOC( temp, uint8_t, temp_bytes );
        NVB_HIP( rocprim::exclusive_scan( temp, temp_bytes, cnt, occ, make_uint4( 0, 0, 0, 0 ), (size_t)n_blocks + 1u, AddU4(), s ) );
        NVB_HIP( hipStreamSynchronize( s ) );
        scratch.release( temp );
    }
    hipLaunchKernelGGL( write_occ_kernel, dim3( grid_for( n_blocks ) ), dim3(256), 0, s, (const uint4*)occ, n_blocks, bwt_occ );
    uint4 totals;
    NVB_HIP( hipMemcpyAsync( &totals, occ + n_blocks, sizeof(uint4), hipMemcpyDeviceToHost, s ) );
    NVB_HIP( hipStreamSynchronize( s ) );

    nvbio_fm_index_view view;
    view.length = n; view.primary = primary;
    view.L2[0] = 0; view.L2[1] = totals.x; view.L2[2] = view.L2[1] + totals.y; view.L2[3] = view.L2[2] + totals.z; view.L2[4] = view.L2[3] + totals.w;
    if (view.L2[4] != n) { set_error( "bwt \"%s\": symbol counts do not add up to the length", bwt_path ); return NVBIO_ERR_INVALID; }
    view.bwt_occ_dev = bwt_occ; view.bwt_occ_words = (uint64_t)words * 2u;
    view.ssa_dev = nullptr; view.ssa_words = 0; view.sa_int = 16;

    // ---- .sa: primary, counts, SA_INT, length, ssa[1..] (fmindex_impl.cu:172-252) ----
    uint32_t* ssa = nullptr;
    if (sa_path)
    {
        FileCloser sf = { fopen( sa_path, "rb" ) };
        if (!sf.f) { set_error( "unable to open sa \"%s\"", sa_path ); return NVBIO_ERR_INVALID; }
        uint32_t sh[7];
        if (fread( sh, sizeof(uint32_t), 7, sf.f ) != 7) { set_error( "failed reading sa header \"%s\"", sa_path ); return NVBIO_ERR_INVALID; }
        if (sh[0] != primary || sh[6] != n) { set_error( "SA file mismatch \"%s\" (primary %u/%u, length %u/%u)", sa_path, sh[0], primary, sh[6], n ); return NVBIO_ERR_INVALID; }
        const uint32_t K = sh[5];
        if (K == 0 || K > 64 || (K & (K - 1u))) { set_error( "unsupported SA interval %u in \"%s\"", K, sa_path ); return NVBIO_ERR_UNSUPPORTED; }
        const uint64_t sa_size = (uint64_t)n / K + 1u;
        std::vector<uint32_t> h_ssa( sa_size );
        h_ssa[0] = 0xFFFFFFFFu;
        if (sa_size > 1 && fread( &h_ssa[1], sizeof(uint32_t), sa_size - 1u, sf.f ) != sa_size - 1u) { set_error( "failed reading sa \"%s\"", sa_path ); return NVBIO_ERR_INVALID; }
        NVB_ALLOC( d_ssa, uint32_t, sa_size );
        NVB_HIP( hipMemcpyAsync( d_ssa, h_ssa.data(), sa_size * sizeof(uint32_t), hipMemcpyHostToDevice, s ) );
        NVB_HIP( hipStreamSynchronize( s ) );
        ssa = d_ssa;
        view.ssa_dev = ssa; view.ssa_words = sa_size; view.sa_int = K;
    }
    scratch.forget( bwt_occ ); if (ssa) scratch.forget( ssa );
    return fm_index_adopt( &view, device, kmer_len, true, s, out, nullptr, nullptr, 0u );
}

static nvbio_status save_impl(const nvbio_fm_index_view& v, const char* bwt_path, const char* sa_path, hipStream_t s)
{
    Scratch scratch;
    const uint32_t n = v.length, n_words = (n + 15u) / 16u;
    NVB_ALLOC( d_bwt, uint32_t, n_words );
    hipLaunchKernelGGL( deinterleave_bwt_kernel, dim3( grid_for( n_words ) ), dim3(256), 0, s, v.bwt_occ_dev, n_words, d_bwt );
    std::vector<uint32_t> h_bwt( n_words );
    NVB_HIP( hipMemcpyAsync( h_bwt.data(), d_bwt, (size_t)n_words * sizeof(uint32_t), hipMemcpyDeviceToHost, s ) );
    NVB_HIP( hipStreamSynchronize( s ) );
    const uint32_t hdr[5] = { v.primary, v.L2[1], v.L2[2], v.L2[3], v.L2[4] };
    {
        FileCloser bf = { fopen( bwt_path, "wb" ) };
        if (!bf.f) { set_error( "unable to create \"%s\"", bwt_path ); return NVBIO_ERR_INVALID; }
        if (fwrite( hdr, sizeof(uint32_t), 5, bf.f ) != 5 || fwrite( h_bwt.data(), sizeof(uint32_t), n_words, bf.f ) != n_words)
        { set_error( "failed writing \"%s\"", bwt_path ); return NVBIO_ERR_INVALID; }
    }
    if (sa_path)
    {
        if (!v.ssa_dev) { set_error( "index has no sampled suffix array to save" ); return NVBIO_ERR_INVALID; }
        std::vector<uint32_t> h_ssa( v.ssa_words );
        NVB_HIP( hipMemcpyAsync( h_ssa.data(), v.ssa_dev, v.ssa_words * sizeof(uint32_t), hipMemcpyDeviceToHost, s ) );
        NVB_HIP( hipStreamSynchronize( s ) );
        FileCloser sf = { fopen( sa_path, "wb" ) };
        if (!sf.f) { set_error( "unable to create \"%s\"", sa_path ); return NVBIO_ERR_INVALID; }
        const uint32_t sh[7] = { v.primary, v.L2[1], v.L2[2], v.L2[3], v.L2[4], v.sa_int ? v.sa_int : 16u, n };
        if (fwrite( sh, sizeof(uint32_t), 7, sf.f ) != 7 ||
            (v.ssa_words > 1 && fwrite( &h_ssa[1], sizeof(uint32_t), v.ssa_words - 1u, sf.f ) != v.ssa_words - 1u))
        { set_error( "failed writing \"%s\"", sa_path ); return NVBIO_ERR_INVALID; }
    }
    return NVBIO_OK;
}

} // anonymous namespace
} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" nvbio_status nvbio_fm_index_load(const char* bwt_path, const char* sa_path, int device, uint32_t kmer_len,
                                            void* stream, nvbio_fm_index_t* out)
{
    NVB_REQUIRE( bwt_path && out, "bwt_path/out is NULL" );
    NVB_REQUIRE( kmer_len <= 17, "kmer_len must be <= 17" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    return load_impl( bwt_path, sa_path, device, kmer_len, (hipStream_t)stream, out );
}

extern "C" nvbio_status nvbio_fm_index_save(nvbio_fm_index_t index, const char* bwt_path, const char* sa_path, void* stream)
{
    NVB_REQUIRE( index && bwt_path, "index/bwt_path is NULL" );
    nvbio_fm_index_view v;
    NVB_CHECK( nvbio_fm_index_get_view( index, &v ) );
    int dev = 0;
    if (hipGetDevice( &dev ) != hipSuccess) dev = 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes( &attr, v.bwt_occ_dev ) == hipSuccess) dev = attr.device;
    DeviceGuard g( dev ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    return save_impl( v, bwt_path, sa_path, (hipStream_t)stream );
}

extern "C" nvbio_status nvbio_fm_index_build(const uint32_t* text2_dev, uint32_t length, int device,
                                             const nvbio_fm_build_options* options, void* stream, nvbio_fm_index_t* out)
{
    NVB_REQUIRE( text2_dev && out, "text2_dev/out is NULL" );
    NVB_REQUIRE( length > 0, "empty text" );
    NVB_REQUIRE( length <= 0xFFFFFFFFu - 8192u, "text too long for 32-bit coordinates" );
    const uint32_t kmer_len = options ? options->kmer_len : 0u;
    const uint32_t sa_int   = (options && options->sa_int) ? options->sa_int : 16u;
    const uint32_t max_lcp  = options ? options->max_lcp : 0u;
    NVB_REQUIRE( kmer_len <= 17, "kmer_len must be <= 17" );
    NVB_REQUIRE( sa_int <= 64 && (sa_int & (sa_int - 1u)) == 0, "sa_int must be a power of two in [1,64]" );
    const bool verify = options && options->verify;
    NVB_REQUIRE( !verify || sa_int == 1, "verify needs the full suffix array (sa_int = 1)" );
    NVB_REQUIRE( !options || options->bucket_symbols <= 5u, "bucket_symbols must be 0 (automatic) or 1 + a value in 0..4" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    return build_impl( text2_dev, length, device, kmer_len, sa_int, max_lcp, verify, options ? options->table_flags : 0u, options ? options->bucket_symbols : 0u, (hipStream_t)stream, out );
}
