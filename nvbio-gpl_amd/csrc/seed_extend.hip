// seed_extend.hip -- the index arithmetic between FMIndexFilter::locate and the banded aligner.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   hit_to_diagonal functor          examples/fmmap/fmmap.cu:92-117
//   genome_infixes functor           examples/fmmap/fmmap.cu:169-196
//   nvBowtie scoring window          nvBowtie/bowtie2/cuda/score_inl.h:100-106 (BestScoreStream::init_context)
//   read orientation flags           nvBowtie/bowtie2/cuda/alignment_utils.h:291-296
// Both kernels are pure streaming (coalesced 8-16 B per element in, 8-13 B out): HBM-bound.
#include "common.h"

namespace nvbio_amd {

__global__ void __launch_bounds__(256)
hits_to_diagonals_kernel(const uint2* __restrict__ hits, const uint64_t n, const uint32_t spr, const uint32_t interval,
                         const uint32_t seed_len, const uint32_t read_len, const uint32_t strand, uint64_t* __restrict__ keys)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < n; h += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint2    hit = hits[h];
        const uint32_t rid = hit.y / spr;
        uint32_t       p   = (hit.y - rid * spr) * interval;
        if (strand) p = read_len - p - seed_len;                 // offset of the seed in the reverse-complemented read
        const uint64_t diag = (uint64_t)hit.x + 1024u - p;
        keys[h] = ((uint64_t)rid << 34) | ((uint64_t)(strand & 1u) << 33) | diag;
    }
}

__global__ void __launch_bounds__(256)
diagonals_to_windows_kernel(const uint64_t* __restrict__ keys, const uint64_t n, const uint32_t band, const uint32_t read_len,
                            const uint32_t genome_len, uint32_t* __restrict__ read_id, uint8_t* __restrict__ flags,
                            uint32_t* __restrict__ wb, uint32_t* __restrict__ we)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t k    = keys[i];
        const uint64_t d    = k & ((1ull << 33) - 1ull);
        const uint32_t g    = d > 1024u ? (uint32_t)(d - 1024u) : 0u;          // clamp the diagonal at the genome start
        const uint32_t half = band / 2u;
        const uint32_t b    = g > half ? g - half : 0u;
        const uint64_t e    = (uint64_t)b + band + read_len;
        read_id[i] = (uint32_t)(k >> 34);
        flags[i]   = ((k >> 33) & 1ull) ? (uint8_t)(NVBIO_READ_REVERSE | NVBIO_READ_COMPLEMENT) : (uint8_t)0;
        wb[i]      = b;
        we[i]      = e < genome_len ? (uint32_t)e : genome_len;
    }
}

// best candidate per read: selection key = (score + 2^20, clamped at 0) << 34 | strand << 33 | end position, end position
// = window begin + sink.x (hit.sink, score_inl.h:127-129); one 64-bit atomic max per candidate into best[read]
// (fmmap reduces the score per read the same way, examples/fmmap/fmmap.cu:367-376; the key makes the choice unique
// and independent of the order candidates arrive in)
__global__ void __launch_bounds__(256)
best_candidate_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                      const uint32_t* __restrict__ wb, const uint64_t n, unsigned long long* __restrict__ best)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t k   = keys[i];
        const int64_t  s   = (int64_t)scores[i] + (1ll << 20);
        const uint64_t pos = (uint64_t)wb[i] + (uint64_t)sinks[i].x;
        const uint64_t sel = ((uint64_t)(s > 0 ? s : 0) << 34) | (k & (1ull << 33)) | (pos & ((1ull << 33) - 1ull));
        atomicMax( &best[k >> 34], (unsigned long long)sel );
    }
}

// the per-read selection keys of best_candidate_kernel back into (score, end position, strand); 0 = no candidate
__global__ void __launch_bounds__(256)
best_unpack_kernel(const unsigned long long* __restrict__ best, const uint32_t n, int32_t* __restrict__ score, int64_t* __restrict__ pos,
                   uint8_t* __restrict__ rc)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const unsigned long long k = best[i];
        const int64_t sv = (int64_t)(k >> 34);
        score[i] = (k && sv > 0) ? (int32_t)(sv - (1ll << 20)) : NVBIO_SCORE_MIN;
        pos[i]   = k ? (int64_t)(k & ((1ull << 33) - 1ull)) : -1ll;
        rc[i]    = k ? (uint8_t)((k >> 33) & 1ull) : (uint8_t)0;
    }
}

// opposite-mate window of a paired-end alignment: BestOppositeScoreStream::init_context
// (nvBowtie/bowtie2/cuda/score_inl.h:389-425) with frame_opposite_mate (alignment_utils.h:52-88)
__global__ void __launch_bounds__(256)
opposite_mate_windows_kernel(const uint32_t* __restrict__ g_pos, const uint8_t* __restrict__ anchor_rc, const uint32_t n,
                             const uint32_t a_len, const uint32_t o_gapped_len, const uint32_t anchor, const uint32_t policy,
                             const uint32_t min_frag, const uint32_t max_frag, const uint32_t overlap, const uint32_t genome_len,
                             uint32_t* __restrict__ wb, uint32_t* __restrict__ we, uint8_t* __restrict__ flags, uint8_t* __restrict__ valid)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const uint64_t g = g_pos[i];
        const bool anchor_fw = anchor_rc[i] == 0;
        const bool anchor_1  = (anchor == 0u);
        bool left, fw;
        switch (policy)
        {
        case NVBIO_PE_POLICY_FF: left = (anchor_1 != anchor_fw); fw =  anchor_fw; break;
        case NVBIO_PE_POLICY_RR: left = (anchor_1 == anchor_fw); fw =  anchor_fw; break;
        case NVBIO_PE_POLICY_FR: left = !anchor_fw;              fw = !anchor_fw; break;
        default:                 left =  anchor_fw;              fw = !anchor_fw; break;     // RF
        }
        uint64_t begin, end;
        if (left)
        {
            const uint64_t max_end = g + a_len + o_gapped_len > min_frag ? g + a_len + o_gapped_len - min_frag : 0ull;
            begin = g + a_len > max_frag ? g + a_len - max_frag : 0ull;
            end   = overlap ? g + a_len : g;
            end   = end < max_end ? end : max_end;
        }
        else
        {
            const uint64_t min_begin = g + min_frag > o_gapped_len ? g + min_frag - o_gapped_len : 0ull;
            end   = g + max_frag;
            begin = overlap ? g : g + a_len;
            begin = begin > min_begin ? begin : min_begin;
        }
        end = end < genome_len ? end : genome_len;
        const bool ok = begin < genome_len && begin < end;
        wb[i] = ok ? (uint32_t)begin : 0u;
        we[i] = ok ? (uint32_t)end : 0u;
        flags[i] = fw ? (uint8_t)0 : (uint8_t)(NVBIO_READ_REVERSE | NVBIO_READ_COMPLEMENT);
        valid[i] = ok ? 1 : 0;
    }
}

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" nvbio_status nvbio_opposite_mate_windows(int device, const uint32_t* g_pos_dev, const uint8_t* anchor_rc_dev, uint32_t n,
                                                    uint32_t anchor_len, uint32_t opposite_gapped_len, uint32_t anchor, uint32_t policy,
                                                    uint32_t min_frag_len, uint32_t max_frag_len, uint32_t overlap, uint32_t genome_len,
                                                    uint32_t* win_begin_dev, uint32_t* win_end_dev, uint8_t* flags_dev, uint8_t* valid_dev,
                                                    void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( g_pos_dev && anchor_rc_dev && win_begin_dev && win_end_dev && flags_dev && valid_dev, "NULL device pointer" );
    NVB_REQUIRE( policy <= NVBIO_PE_POLICY_RR, "invalid paired-end policy" );
    NVB_REQUIRE( anchor <= 1u, "anchor must be 0 or 1" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( opposite_mate_windows_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream,
                        g_pos_dev, anchor_rc_dev, n, anchor_len, opposite_gapped_len, anchor, policy, min_frag_len, max_frag_len, overlap,
                        genome_len, win_begin_dev, win_end_dev, flags_dev, valid_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_hits_to_diagonals(int device, const nvbio_uint2* hits_dev, uint64_t n_hits, uint32_t seeds_per_read,
                                                uint32_t seed_interval, uint32_t seed_len, uint32_t read_len, uint32_t strand,
                                                uint64_t* keys_dev, void* stream)
{
    if (n_hits == 0) return NVBIO_OK;
    NVB_REQUIRE( hits_dev && keys_dev, "NULL device pointer" );
    NVB_REQUIRE( seeds_per_read > 0, "seeds_per_read must be positive" );
    NVB_REQUIRE( (uint64_t)(seeds_per_read - 1u) * seed_interval + seed_len <= read_len, "seeds do not fit the read" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( hits_to_diagonals_kernel, dim3( grid_for( n_hits ) ), dim3(256), 0, (hipStream_t)stream,
                        (const uint2*)hits_dev, n_hits, seeds_per_read, seed_interval, seed_len, read_len, strand, keys_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_diagonals_to_windows(int device, const uint64_t* keys_dev, uint64_t n, uint32_t band, uint32_t read_len,
                                                   uint32_t genome_len, uint32_t* read_id_dev, uint8_t* flags_dev,
                                                   uint32_t* win_begin_dev, uint32_t* win_end_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && read_id_dev && flags_dev && win_begin_dev && win_end_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( diagonals_to_windows_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream,
                        keys_dev, n, band, read_len, genome_len, read_id_dev, flags_dev, win_begin_dev, win_end_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_best_candidate_reduce(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                                    const uint32_t* win_begin_dev, uint64_t n, uint64_t* best_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && scores_dev && sinks_dev && win_begin_dev && best_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( best_candidate_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream,
                        keys_dev, scores_dev, (const uint2*)sinks_dev, win_begin_dev, n, (unsigned long long*)best_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_best_candidate_unpack(int device, const uint64_t* best_dev, uint32_t n_reads, int32_t* scores_dev,
                                                    int64_t* end_pos_dev, uint8_t* rc_dev, void* stream)
{
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( best_dev && scores_dev && end_pos_dev && rc_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( best_unpack_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream,
                        (const unsigned long long*)best_dev, n_reads, scores_dev, end_pos_dev, rc_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}
