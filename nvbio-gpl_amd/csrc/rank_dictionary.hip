// rank_dictionary.hip -- the GENERIC rank dictionary of the reference for gfx950: 2-bit big-endian text in plain 32- or 64-bit words, a
// separate occurrence table, any block size K, 32- or 64-bit indices.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   build_occurrence_table<K>                         nvbio/fmindex/rank_dictionary_inl.h:33-66
//   dispatch_rank<2,K,PackedStream<..>,..,word,index>  nvbio/fmindex/rank_dictionary_inl.h:206-336 (run, run4)
//   rank / rank4 wrappers                             nvbio/fmindex/rank_dictionary_inl.h:482-539
//   the configurations its test runs                  nvbio-test/rank_test.cu:83-227 (uint32 / K 64, uint64 / K 128 with 64-bit indices)
// The production layout (uint4 records of BWT + occ, 32-bit) has its own fused path in fm_device.h; this one serves callers that
// keep the reference's plain layouts, and texts beyond 2^32 symbols.
//
// MI355X design: a rank is one gather of the block's counters (16 or 32 bytes) plus the block's words up to the index (at most K / 4
// bytes, contiguous): HBM-gather-bound, one lane per query; popcounts are eq-mask & prefix-mask on whole words.  The build is three
// streaming passes: per-block symbol counts, a scan per symbol (hipCUB), the interleave.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace nvbio_amd {

// occurrences of symbol c among the first p (1 .. W/2) symbols of a big-endian 2-bit word
template <typename W>
__device__ __forceinline__ uint32_t popc_symbol_prefix(const W w, const uint32_t c, const uint32_t p)
{
    constexpr uint32_t BITS = sizeof(W) * 8u;
    const W fives = (W)0x5555555555555555ull;
    const W x = w ^ (W)((W)c * fives);                        // 00 where the symbol equals c
    W e = (W)(~(x | (x >> 1))) & fives;                       // bit 2s set iff symbol s == c
    e &= (p >= BITS / 2u) ? (W)~(W)0 : (W)((W)~(W)0 << (BITS - 2u * p));
    return sizeof(W) == 8 ? (uint32_t)__popcll( (unsigned long long)e ) : (uint32_t)__popc( (unsigned int)e );
}

template <typename W, typename I, bool ALL4>
__global__ void __launch_bounds__(256)
rank_generic_kernel(const W* __restrict__ text, const I* __restrict__ occ, const uint32_t K, const I* __restrict__ idx, const uint8_t* __restrict__ sym,
                    const uint32_t n, I* __restrict__ out)
{
    constexpr uint32_t SPW = sizeof(W) * 4u;                  // symbols per word
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x)
    {
        const I i = idx[q];
        if (i == (I)~(I)0)                                    // rank( dict, -1, c ) = 0 (rank_dictionary_inl.h:278-279)
        {
            if (ALL4) { out[4u * q] = 0; out[4u * q + 1u] = 0; out[4u * q + 2u] = 0; out[4u * q + 3u] = 0; }
            else out[q] = 0;
            continue;
        }
        const uint64_t k = (uint64_t)i / K;
        const uint32_t in_block = (uint32_t)((uint64_t)i - k * K);
        const uint32_t full = in_block / SPW, part = in_block % SPW + 1u;
        const W* words = text + k * (K / SPW);
        uint32_t cnt[4] = { 0, 0, 0, 0 };
        const uint32_t c0 = ALL4 ? 0u : (sym[q] & 3u);
        for (uint32_t j = 0; j <= full; ++j)
        {
            const W w = words[j];
            const uint32_t p = j < full ? SPW : part;
            if (ALL4) { cnt[0] += popc_symbol_prefix( w, 0u, p ); cnt[1] += popc_symbol_prefix( w, 1u, p ); cnt[2] += popc_symbol_prefix( w, 2u, p ); cnt[3] += popc_symbol_prefix( w, 3u, p ); }
            else cnt[0] += popc_symbol_prefix( w, c0, p );
        }
        if (ALL4) { out[4u * q] = occ[k * 4u] + cnt[0]; out[4u * q + 1u] = occ[k * 4u + 1u] + cnt[1]; out[4u * q + 2u] = occ[k * 4u + 2u] + cnt[2]; out[4u * q + 3u] = occ[k * 4u + 3u] + cnt[3]; }
        else out[q] = occ[k * 4u + c0] + cnt[0];
    }
}

// symbols of each kind in block b = text[b K, min((b+1) K, length))
template <typename W>
__global__ void __launch_bounds__(256)
block_symbol_counts_kernel(const W* __restrict__ text, const uint64_t length, const uint32_t K, const uint64_t n_blocks, uint32_t* __restrict__ cnt)
{
    constexpr uint32_t SPW = sizeof(W) * 4u;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t begin = b * K, end = begin + K < length ? begin + K : length;
        uint32_t c[4] = { 0, 0, 0, 0 };
        for (uint64_t s = begin; s < end; s += SPW)
        {
            const W w = text[s / SPW];
            const uint32_t p = end - s < SPW ? (uint32_t)(end - s) : SPW;
            c[0] += popc_symbol_prefix( w, 0u, p ); c[1] += popc_symbol_prefix( w, 1u, p ); c[2] += popc_symbol_prefix( w, 2u, p ); c[3] += popc_symbol_prefix( w, 3u, p );
        }
        cnt[b] = c[0]; cnt[n_blocks + b] = c[1]; cnt[2u * n_blocks + b] = c[2]; cnt[3u * n_blocks + b] = c[3];
    }
}
template <typename I>
__global__ void __launch_bounds__(256)
interleave_occ_kernel(const uint64_t* __restrict__ sums, const uint64_t n_blocks, I* __restrict__ occ)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x)
        for (uint32_t c = 0; c < 4u; ++c) occ[b * 4u + c] = (I)sums[c * n_blocks + b];
}
struct U32toU64 { __host__ __device__ __forceinline__ uint64_t operator()(const uint32_t v) const { return v; } };

static nvbio_status check_dict(const nvbio_rank_dictionary* d)
{
    NVB_REQUIRE( d != nullptr, "dictionary is NULL" );
    NVB_REQUIRE( d->word_bits == 32 || d->word_bits == 64, "word_bits must be 32 or 64" );
    NVB_REQUIRE( d->index_bits == 32 || d->index_bits == 64, "index_bits must be 32 or 64" );
    const uint32_t spw = d->word_bits / 2u;
    NVB_REQUIRE( d->K >= spw && d->K % spw == 0 && (d->K & (d->K - 1u)) == 0 && d->K <= 1024u, "K must be a power of two, a multiple of the symbols per word, at most 1024" );
    NVB_REQUIRE( d->index_bits == 64 || d->length <= 0xFFFFFFFFull, "a text beyond 2^32 symbols needs 64-bit indices" );
    return NVBIO_OK;
}

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" {

nvbio_status nvbio_rank_dictionary_occ_entries(const nvbio_rank_dictionary* dict, uint64_t* entries)
{
    NVB_CHECK( check_dict( dict ) );
    NVB_REQUIRE( entries != nullptr, "entries is NULL" );
    *entries = 4ull * ((dict->length + dict->K - 1u) / dict->K);
    return NVBIO_OK;
}

nvbio_status nvbio_rank_dictionary_build(int device, const nvbio_rank_dictionary* dict, void* occ_out_dev, uint64_t counts[4], void* stream)
{
    NVB_CHECK( check_dict( dict ) );
    NVB_REQUIRE( counts != nullptr, "counts is NULL" );
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (dict->length == 0) return NVBIO_OK;
    NVB_REQUIRE( dict->text_dev && occ_out_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t nb = (dict->length + dict->K - 1u) / dict->K;
    NVB_REQUIRE( nb < (1ull << 31), "too many blocks" );
    uint32_t* cnt = nullptr; uint64_t* sums = nullptr; void* temp = nullptr; size_t temp_bytes = 0;
    hipcub::TransformInputIterator<uint64_t, U32toU64, const uint32_t*> in0( (const uint32_t*)nullptr, U32toU64() );
    hipError_t e = hipcub::DeviceScan::ExclusiveSum( nullptr, temp_bytes, in0, (uint64_t*)nullptr, (int)nb, s );
    if (e == hipSuccess) e = scratch_alloc( (void**)&cnt, 4ull * nb * sizeof(uint32_t), s );
    if (e == hipSuccess) e = scratch_alloc( (void**)&sums, 4ull * nb * sizeof(uint64_t), s );
    if (e == hipSuccess) e = scratch_alloc( &temp, temp_bytes ? temp_bytes : 16, s );
    if (e == hipSuccess)
    {
        if (dict->word_bits == 32) hipLaunchKernelGGL( block_symbol_counts_kernel<uint32_t>, dim3( grid_for( nb ) ), dim3(256), 0, s, (const uint32_t*)dict->text_dev, dict->length, dict->K, nb, cnt );
        else                       hipLaunchKernelGGL( block_symbol_counts_kernel<uint64_t>, dim3( grid_for( nb ) ), dim3(256), 0, s, (const uint64_t*)dict->text_dev, dict->length, dict->K, nb, cnt );
        e = hipGetLastError();
    }
    for (uint32_t c = 0; c < 4u && e == hipSuccess; ++c)
    {
        hipcub::TransformInputIterator<uint64_t, U32toU64, const uint32_t*> in( (const uint32_t*)cnt + c * nb, U32toU64() );
        e = hipcub::DeviceScan::ExclusiveSum( temp, temp_bytes, in, sums + c * nb, (int)nb, s );
    }
    if (e == hipSuccess)
    {
        if (dict->index_bits == 32) hipLaunchKernelGGL( interleave_occ_kernel<uint32_t>, dim3( grid_for( nb ) ), dim3(256), 0, s, (const uint64_t*)sums, nb, (uint32_t*)occ_out_dev );
        else                        hipLaunchKernelGGL( interleave_occ_kernel<uint64_t>, dim3( grid_for( nb ) ), dim3(256), 0, s, (const uint64_t*)sums, nb, (uint64_t*)occ_out_dev );
        e = hipGetLastError();
    }
    // totals = the last block's exclusive sum + its own counts
    uint64_t last_sum[4] = { 0, 0, 0, 0 }; uint32_t last_cnt[4] = { 0, 0, 0, 0 };
    for (uint32_t c = 0; c < 4u && e == hipSuccess; ++c)
    {
        e = hipMemcpyAsync( &last_sum[c], sums + c * nb + nb - 1u, sizeof(uint64_t), hipMemcpyDeviceToHost, s );
        if (e == hipSuccess) e = hipMemcpyAsync( &last_cnt[c], cnt + c * nb + nb - 1u, sizeof(uint32_t), hipMemcpyDeviceToHost, s );
    }
    if (e == hipSuccess) e = hipStreamSynchronize( s );
    if (cnt)  scratch_free( cnt, s );
    if (sums) scratch_free( sums, s );
    if (temp) scratch_free( temp, s );
    if (e != hipSuccess) { (void)hipGetLastError(); set_error( "rank_dictionary_build failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    for (uint32_t c = 0; c < 4u; ++c) counts[c] = last_sum[c] + last_cnt[c];
    return NVBIO_OK;
}

static nvbio_status rank_common(int device, const nvbio_rank_dictionary* d, const void* idx, const uint8_t* sym, uint32_t n, void* out, bool all4, void* stream)
{
    NVB_CHECK( check_dict( d ) );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( d->text_dev && d->occ_dev && idx && out && (all4 || sym), "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    const dim3 grid( grid_for( n ) ), block( 256 );
    hipStream_t s = (hipStream_t)stream;
#define NVB_RK(W, I) do { if (all4) hipLaunchKernelGGL( (rank_generic_kernel<W,I,true>),  grid, block, 0, s, (const W*)d->text_dev, (const I*)d->occ_dev, d->K, (const I*)idx, sym, n, (I*)out ); \
                          else      hipLaunchKernelGGL( (rank_generic_kernel<W,I,false>), grid, block, 0, s, (const W*)d->text_dev, (const I*)d->occ_dev, d->K, (const I*)idx, sym, n, (I*)out ); } while (0)
    if      (d->word_bits == 32 && d->index_bits == 32) NVB_RK( uint32_t, uint32_t );
    else if (d->word_bits == 32)                        NVB_RK( uint32_t, uint64_t );
    else if (d->index_bits == 32)                       NVB_RK( uint64_t, uint32_t );
    else                                                NVB_RK( uint64_t, uint64_t );
#undef NVB_RK
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_rank_dictionary_rank(int device, const nvbio_rank_dictionary* dict, const void* idx_dev, const uint8_t* syms_dev, uint32_t n, void* out_dev, void* stream)
{ return rank_common( device, dict, idx_dev, syms_dev, n, out_dev, false, stream ); }

nvbio_status nvbio_rank_dictionary_rank4(int device, const nvbio_rank_dictionary* dict, const void* idx_dev, uint32_t n, void* out_dev, void* stream)
{ return rank_common( device, dict, idx_dev, nullptr, n, out_dev, true, stream ); }

} // extern "C"
