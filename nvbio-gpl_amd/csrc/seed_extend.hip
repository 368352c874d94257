// seed_extend.hip -- the index arithmetic between FMIndexFilter::locate and the banded aligner.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   hit_to_diagonal functor          examples/fmmap/fmmap.cu:92-117
//   genome_infixes functor           examples/fmmap/fmmap.cu:169-196
//   nvBowtie scoring window          nvBowtie/bowtie2/cuda/score_inl.h:100-106 (BestScoreStream::init_context)
//   read orientation flags           nvBowtie/bowtie2/cuda/alignment_utils.h:291-296
// Both kernels are pure streaming (coalesced 8-16 B per element in, 8-13 B out): HBM-bound.
#include "common.h"
#include <hipcub/hipcub.hpp>

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
diagonals_to_windows_kernel(const uint64_t* __restrict__ keys, const uint64_t n, const uint32_t band, const uint32_t read_len_all,
                            const uint32_t genome_len, uint32_t* __restrict__ read_id, uint8_t* __restrict__ flags,
                            uint32_t* __restrict__ wb, uint32_t* __restrict__ we, const uint32_t* __restrict__ read_offsets = nullptr)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t k    = keys[i];
        const uint64_t d    = k & ((1ull << 33) - 1ull);
        const uint32_t g    = d > 1024u ? (uint32_t)(d - 1024u) : 0u;          // clamp the diagonal at the genome start
        const uint32_t half = band / 2u;
        const uint32_t b    = g > half ? g - half : 0u;
        const uint32_t rid  = (uint32_t)(k >> 34);
        const uint32_t read_len = read_offsets ? read_offsets[rid + 1] - read_offsets[rid] : read_len_all;     // ragged reads: each its own length
        const uint64_t e    = (uint64_t)b + band + read_len;
        read_id[i] = rid;
        flags[i]   = ((k >> 33) & 1ull) ? (uint8_t)(NVBIO_READ_REVERSE | NVBIO_READ_COMPLEMENT) : (uint8_t)0;
        wb[i]      = b;
        we[i]      = e < genome_len ? (uint32_t)e : genome_len;
    }
}

// best candidate per read: selection key = (score + 2^20, clamped at 0) << 34 | strand << 33 | end position, end position
// = window begin + sink.x (hit.sink, score_inl.h:127-129); one 64-bit atomic max per candidate into best[read]
// (fmmap reduces the score per read the same way, examples/fmmap/fmmap.cu:367-376; the key makes the choice unique
// and independent of the order candidates arrive in)
// One atomic per RUN of consecutive candidates of the same read instead of one per candidate: candidates arrive grouped by read (seed
// order; the residual ones sorted by seed id), and a read inside a repeat brings 16 or more of them -- all hitting one address, which
// the atomic unit serialises (0.5 ms per 19 M candidates with a 5 % repeat share against 0.11 ms per 11 M without).  Doubling max over
// the lanes of a wave, restricted to equal `seg` (runs are contiguous); the run's first lane gets the result.
__device__ __forceinline__ bool run_max(const uint32_t seg, unsigned long long& v)
{
    const uint32_t lane = threadIdx.x & 63u;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const uint32_t os = (uint32_t)__shfl_down( (int)seg, d );
        const uint32_t lo = (uint32_t)__shfl_down( (int)(uint32_t)v, d ), hi = (uint32_t)__shfl_down( (int)(uint32_t)(v >> 32), d );
        const unsigned long long ov = ((unsigned long long)hi << 32) | lo;
        if (lane + (uint32_t)d < 64u && os == seg && ov > v) v = ov;
    }
    const uint32_t ps = (uint32_t)__shfl_up( (int)seg, 1 );
    return lane == 0u || ps != seg;
}

__global__ void __launch_bounds__(256)
best_candidate_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                      const uint32_t* __restrict__ wb, const uint64_t n, unsigned long long* __restrict__ best)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride)       // wave-uniform bound
    {
        const uint64_t i = i0 + (threadIdx.x & 63u);
        uint32_t seg = 0xFFFFFF00u + (threadIdx.x & 63u);                                                    // beyond the list: a run of its own
        unsigned long long sel = 0ull;
        if (i < n)
        {
            const uint64_t k   = keys[i];
            const int64_t  s   = (int64_t)scores[i] + (1ll << 20);
            const uint64_t pos = (uint64_t)wb[i] + (uint64_t)sinks[i].x;
            sel = ((uint64_t)(s > 0 ? s : 0) << 34) | (k & (1ull << 33)) | (pos & ((1ull << 33) - 1ull));
            seg = (uint32_t)(k >> 34);
        }
        if (run_max( seg, sel ) && i < n) atomicMax( &best[seg], sel );
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

// the window (and locus) of every read's best candidate: see nvbio_best_candidate_windows
__global__ void __launch_bounds__(256)
best_window_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                   const uint32_t* __restrict__ wb, const uint64_t n, const unsigned long long* __restrict__ best,
                   long long* __restrict__ best_wb, long long* __restrict__ best_g)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t k   = keys[i];
        const int64_t  s   = (int64_t)scores[i] + (1ll << 20);
        const uint64_t pos = (uint64_t)wb[i] + (uint64_t)sinks[i].x;
        const uint64_t sel = ((uint64_t)(s > 0 ? s : 0) << 34) | (k & (1ull << 33)) | (pos & ((1ull << 33) - 1ull));
        if (sel != best[k >> 34]) continue;
        atomicMax( &best_wb[k >> 34], (long long)wb[i] );
        if (best_g)
        {
            const int64_t d = (int64_t)(k & ((1ull << 33) - 1ull)) - 1024;
            atomicMax( &best_g[k >> 34], (long long)(d > 0 ? d : 0) );
        }
    }
}

// the traceback batch of every read's best alignment: see nvbio_traceback_best_batch
__global__ void __launch_bounds__(256)
traceback_best_batch_kernel(const unsigned long long* __restrict__ best, const long long* __restrict__ best_wb, const uint32_t n,
                            const uint32_t read_len_all, const uint32_t band, const uint32_t genome_len, const int32_t min_score_all,
                            uint8_t* __restrict__ flags, uint32_t* __restrict__ wb, uint32_t* __restrict__ we, int32_t* __restrict__ scores,
                            uint2* __restrict__ sinks, const uint32_t* __restrict__ read_offsets = nullptr, const int32_t* __restrict__ min_scores = nullptr)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
    {
        const uint32_t read_len  = read_offsets ? read_offsets[r + 1] - read_offsets[r] : read_len_all;
        const int32_t  min_score = min_scores ? min_scores[r] : min_score_all;
        const unsigned long long k = best[r];
        const long long w = best_wb[r];
        const int64_t sv = (int64_t)(k >> 34);
        const int32_t sc = (k && sv > 0) ? (int32_t)(sv - (1ll << 20)) : NVBIO_SCORE_MIN;
        const bool aligned = k != 0ull && w >= 0 && sc >= min_score;
        if (aligned)
        {
            const uint64_t end = (uint64_t)w + band + read_len;
            flags[r]  = ((k >> 33) & 1ull) ? (uint8_t)(NVBIO_READ_REVERSE | NVBIO_READ_COMPLEMENT) : (uint8_t)0;
            wb[r]     = (uint32_t)w;
            we[r]     = end < genome_len ? (uint32_t)end : genome_len;
            scores[r] = sc;
            sinks[r]  = make_uint2( (uint32_t)((k & ((1ull << 33) - 1ull)) - (uint64_t)w), read_len );
        }
        else
        {
            flags[r] = 0; wb[r] = 0; we[r] = 0; scores[r] = NVBIO_SCORE_MIN; sinks[r] = make_uint2( 0xFFFFFFFFu, 0xFFFFFFFFu );
        }
    }
}

// second-best candidate per read: nvBowtie's score_reduce_kernel (nvBowtie/bowtie2/cuda/reduce_inl.h:65-140) keeps, beside the
// best alignment a1, a second one a2 that must be `distinct` from a1 (io::distinct_alignments, nvbio/io/alignments_inl.h:26-38:
// other strand, or more than read_len/2 away) and score above the read's threshold; it skips candidates at a location already
// held (:104-107).  That loop depends on the order candidates arrive in; fed in descending order of the selection key it ends
// with a1 = the largest key and a2 = the largest key among the candidates distinct from a1 -- which is what this pass computes
// with one atomic max per candidate, given the final a1 of every read (best_candidate_kernel over ALL candidates first).
__device__ __forceinline__ bool distinct_alignments(const uint64_t pos1, const uint32_t rc1, const uint64_t pos2, const uint32_t rc2, const uint64_t dist)
{
    if (rc1 != rc2) return true;
    return !(pos1 >= pos2 - (pos2 < dist ? pos2 : dist) && pos1 <= pos2 + dist);
}

__global__ void __launch_bounds__(256)
second_candidate_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                        const uint32_t* __restrict__ wb, const uint64_t n, const unsigned long long* __restrict__ best,
                        const uint32_t dist_all, const int32_t worst_score_all, unsigned long long* __restrict__ second,
                        const uint32_t* __restrict__ read_offsets = nullptr, const int32_t* __restrict__ min_scores = nullptr)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride)       // wave-uniform bound
    {
        const uint64_t i = i0 + (threadIdx.x & 63u);
        uint32_t seg = 0xFFFFFF00u + (threadIdx.x & 63u);
        unsigned long long sel = 0ull;                            // 0: this candidate does not compete
        if (i < n)
        {
            const int32_t sc = scores[i];
            const uint64_t k   = keys[i];
            seg = (uint32_t)(k >> 34);
            // ragged reads: distinct_dist = read_len / 2 and the threshold min_score - 1 of the candidate's own read
            const uint32_t dist        = read_offsets ? (read_offsets[(k >> 34) + 1] - read_offsets[k >> 34]) / 2u : dist_all;
            const int32_t  worst_score = min_scores ? min_scores[k >> 34] - 1 : worst_score_all;
            if (sc > worst_score)                                  // `score > best.m_a2.score()` with a2 initialised to the threshold
            {
                const int64_t  s   = (int64_t)sc + (1ll << 20);
                const uint64_t pos = ((uint64_t)wb[i] + (uint64_t)sinks[i].x) & ((1ull << 33) - 1ull);
                const uint64_t cand = ((uint64_t)(s > 0 ? s : 0) << 34) | (k & (1ull << 33)) | pos;
                const unsigned long long b = best[k >> 34];
                // not the best itself (or a copy of it: location already held), and distinct from it
                if (cand != b && distinct_alignments( b & ((1ull << 33) - 1ull), (uint32_t)((b >> 33) & 1ull), pos, (uint32_t)((k >> 33) & 1ull), dist ))
                    sel = cand;
            }
        }
        if (run_max( seg, sel ) && sel != 0ull) atomicMax( &second[seg], sel );
    }
}

// Bowtie2's mapping quality from (best, second best, read length): BowtieMapq2 (the one nvBowtie instantiates,
// bowtie2_cuda_driver.cu:277) and BowtieMapq3, nvBowtie/bowtie2/cuda/mapq.h:32-297, single-end form.  float arithmetic as
// the reference's (no contraction); perfect = scheme.perfect_score(len), minimum = scheme.min_score(len).
__device__ __forceinline__ int mapq_v3(const int32_t best_score, const bool has_second, const int32_t second_score, const float max_score, const float min_score)
{
    const int unpaired_one[11]         = { 43, 42, 41, 36, 32, 27, 20, 11, 4, 1, 0 };
    const int unpaired_two_perfect[11] = { 2, 16, 23, 30, 31, 32, 34, 36, 38, 40, 42 };
    const int unpaired_two[11][11] = {
        {  2,  2,  2,  1,  1, 0, 0, 0, 0, 0, 0 }, { 20, 14,  7,  3,  2, 1, 0, 0, 0, 0, 0 }, { 20, 16, 10,  6,  3, 1, 0, 0, 0, 0, 0 },
        { 20, 17, 13,  9,  3, 1, 1, 0, 0, 0, 0 }, { 21, 19, 15,  9,  5, 2, 2, 0, 0, 0, 0 }, { 22, 21, 16, 11, 10, 5, 0, 0, 0, 0, 0 },
        { 23, 22, 19, 16, 11, 0, 0, 0, 0, 0, 0 }, { 24, 25, 21, 30,  0, 0, 0, 0, 0, 0, 0 }, { 30, 26, 29,  0,  0, 0, 0, 0, 0, 0, 0 },
        { 30, 27,  0,  0,  0, 0, 0, 0, 0, 0, 0 }, { 30,  0,  0,  0,  0, 0, 0, 0, 0, 0, 0 } };
    const float norm_factor = 10.0f / (max_score - min_score);
    if ((float)best_score < min_score) return 0;
    const int best     = ((int)max_score - best_score) > 0 ? ((int)max_score - best_score) : 0;     // negated best score
    int       best_bin = (int)((float)best * norm_factor + 0.5f);
    best_bin = best_bin < 0 ? 0 : (best_bin > 10 ? 10 : best_bin);
    if (has_second)
    {
        const int diff     = best_score - second_score;
        int       diff_bin = (int)((float)diff * norm_factor + 0.5f);
        diff_bin = diff_bin < 0 ? 0 : (diff_bin > 10 ? 10 : diff_bin);
        return ((float)best == max_score) ? unpaired_two_perfect[best_bin] : unpaired_two[diff_bin][best_bin];
    }
    return ((float)best == max_score) ? 44 : unpaired_one[best_bin];
}

__device__ __forceinline__ int mapq_v2(const int32_t best_score, const bool has_second, const int32_t second_score, const float max_score, const float min_score,
                                       const bool monotone)
{
    const float diff = max_score - min_score;
    const float best = (float)best_score;
    if (best < min_score) return 0;
    const float best_over = best - min_score;
    if (monotone)
    {
        if (!has_second)
        {
            if      (best_over >= diff * 0.8f) return 42;
            else if (best_over >= diff * 0.7f) return 40;
            else if (best_over >= diff * 0.6f) return 24;
            else if (best_over >= diff * 0.5f) return 23;
            else if (best_over >= diff * 0.4f) return 8;
            else if (best_over >= diff * 0.3f) return 3;
            else                               return 0;
        }
        const float best_diff = fabsf( fabsf( best ) - fabsf( (float)second_score ) );
        if      (best_diff >= diff * 0.9f) return (best_over == diff) ? 39 : 33;
        else if (best_diff >= diff * 0.8f) return (best_over == diff) ? 38 : 27;
        else if (best_diff >= diff * 0.7f) return (best_over == diff) ? 37 : 26;
        else if (best_diff >= diff * 0.6f) return (best_over == diff) ? 36 : 22;
        else if (best_diff >= diff * 0.5f) return (best_over == diff) ? 35 : (best_over >= diff * 0.84f) ? 25 : (best_over >= diff * 0.68f) ? 16 : 5;
        else if (best_diff >= diff * 0.4f) return (best_over == diff) ? 34 : (best_over >= diff * 0.84f) ? 21 : (best_over >= diff * 0.68f) ? 14 : 4;
        else if (best_diff >= diff * 0.3f) return (best_over == diff) ? 32 : (best_over >= diff * 0.88f) ? 18 : (best_over >= diff * 0.67f) ? 15 : 3;
        else if (best_diff >= diff * 0.2f) return (best_over == diff) ? 31 : (best_over >= diff * 0.88f) ? 17 : (best_over >= diff * 0.67f) ? 11 : 0;
        else if (best_diff >= diff * 0.1f) return (best_over == diff) ? 30 : (best_over >= diff * 0.88f) ? 12 : (best_over >= diff * 0.67f) ? 7 : 0;
        else if (best_diff > 0)            return (best_over >= diff * 0.67f) ? 6 : 2;
        else                               return (best_over >= diff * 0.67f) ? 1 : 0;
    }
    if (!has_second)
    {
        if      (best_over >= diff * 0.8f) return 44;
        else if (best_over >= diff * 0.7f) return 42;
        else if (best_over >= diff * 0.6f) return 41;
        else if (best_over >= diff * 0.5f) return 36;
        else if (best_over >= diff * 0.4f) return 28;
        else if (best_over >= diff * 0.3f) return 24;
        else                               return 22;
    }
    const float best_diff = fabsf( fabsf( best ) - fabsf( (float)second_score ) );
    if      (best_diff >= diff * 0.9f) return 40;
    else if (best_diff >= diff * 0.8f) return 39;
    else if (best_diff >= diff * 0.7f) return 38;
    else if (best_diff >= diff * 0.6f) return 37;
    else if (best_diff >= diff * 0.5f) return (best_over == diff) ? 35 : (best_over >= diff * 0.50f) ? 25 : 20;
    else if (best_diff >= diff * 0.4f) return (best_over == diff) ? 34 : (best_over >= diff * 0.50f) ? 21 : 19;
    else if (best_diff >= diff * 0.3f) return (best_over == diff) ? 33 : (best_over >= diff * 0.5f) ? 18 : 16;
    else if (best_diff >= diff * 0.2f) return (best_over == diff) ? 32 : (best_over >= diff * 0.5f) ? 17 : 12;
    else if (best_diff >= diff * 0.1f) return (best_over == diff) ? 31 : (best_over >= diff * 0.5f) ? 14 : 9;
    else if (best_diff > 0)            return (best_over >= diff * 0.5f) ? 11 : 2;
    else                               return (best_over >= diff * 0.5f) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
mapq_kernel(const unsigned long long* __restrict__ best, const unsigned long long* __restrict__ second, const uint32_t n,
            const int version, const bool monotone, const float max_score_all, const float min_score_all,
            int32_t* __restrict__ second_score, uint8_t* __restrict__ mapq,
            const uint32_t* __restrict__ read_offsets = nullptr, const int32_t* __restrict__ min_scores = nullptr, const int32_t match = 0)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        // ragged reads: perfect_score = match x read_len (scoring.h:274) and min_score( read_len ) of the read itself
        const float max_score = read_offsets ? (float)(match * (int32_t)(read_offsets[i + 1] - read_offsets[i])) : max_score_all;
        const float min_score = min_scores ? (float)min_scores[i] : min_score_all;
        const unsigned long long b = best[i], s2 = second ? second[i] : 0ull;
        const int32_t bs = b  ? (int32_t)((int64_t)(b  >> 34) - (1ll << 20)) : NVBIO_SCORE_MIN;
        const int32_t ss = s2 ? (int32_t)((int64_t)(s2 >> 34) - (1ll << 20)) : NVBIO_SCORE_MIN;
        int q = 0;
        if (b) q = (version == 3) ? mapq_v3( bs, s2 != 0ull, ss, max_score, min_score ) : mapq_v2( bs, s2 != 0ull, ss, max_score, min_score, monotone );
        if (second_score) second_score[i] = ss;
        mapq[i] = (uint8_t)q;
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

// nvBowtie's scoring stream, flattened: BestScoreStream::init_context for every work item of the stream
// (nvBowtie/bowtie2/cuda/score_inl.h:85-115) -- the hit to score is hits[ idx_queue[i] ]; its read, strand (packed_seed::rc, bit 13
// of the word: defs.h:162-172 `pos_in_read:12, index_dir:1, rc:1, top_flag:1`) and locus give the banded window
// [ loc > band/2 ? loc - band/2 : 0,  min( begin + band + read_len, genome_length ) ) -- and the orientation load_strings asks the
// read loader for (alignment_utils.h:291-296: nvBowtie stores its reads REVERSED, so a forward hit reads the stream backwards
// and a reverse-complemented one reads it forwards, complemented).
__global__ void __launch_bounds__(256)
score_stream_flatten_kernel(const uint32_t* __restrict__ idx_queue, const uint32_t n, const uint32_t* __restrict__ hit_read_id,
                            const uint32_t* __restrict__ hit_seed, const uint32_t* __restrict__ hit_loc, const uint32_t* __restrict__ read_index,
                            const uint32_t band, const uint32_t genome_len, const uint32_t reads_reversed,
                            uint32_t* __restrict__ read_id, uint8_t* __restrict__ flags, uint32_t* __restrict__ wb, uint32_t* __restrict__ we)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const uint32_t idx = idx_queue ? idx_queue[i] : i;
        const uint32_t rid = hit_read_id[idx];
        const uint32_t rc  = (hit_seed[idx] >> 13) & 1u;
        const uint32_t g   = hit_loc[idx];
        const uint32_t len = read_index[rid + 1u] - read_index[rid];
        const uint32_t b   = g > band / 2u ? g - band / 2u : 0u;
        const uint32_t e   = b + band + len;                                   // uint32 arithmetic, as the reference's
        read_id[i] = rid;
        flags[i]   = reads_reversed ? (rc ? (uint8_t)NVBIO_READ_COMPLEMENT : (uint8_t)NVBIO_READ_REVERSE)
                                    : (rc ? (uint8_t)(NVBIO_READ_REVERSE | NVBIO_READ_COMPLEMENT) : (uint8_t)0);
        // a locus that wrapped below zero (a seed hanging over the genome start: locate_inl.h:133 subtracts pos_in_read in uint32) gives the
        // reference a window that begins past the genome's end, which it then reads out of bounds; here such a job gets the empty
        // window [0, 0) -- no kernel ever forms an address from it -- and reports nothing
        const uint32_t end = e < genome_len ? e : genome_len;
        const bool     bad = b >= genome_len || end < b;
        wb[i] = bad ? 0u : b;
        we[i] = bad ? 0u : end;
    }
}

// BestScoreStream::output (score_inl.h:119-133): hit.score = max( sink.score, worst_score ); hit.sink = genome_begin + sink.sink.x
__global__ void __launch_bounds__(256)
score_stream_output_kernel(const uint32_t* __restrict__ idx_queue, const uint32_t n, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                           const uint32_t* __restrict__ wb, const int32_t worst, int32_t* __restrict__ hit_score, uint32_t* __restrict__ hit_sink)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const uint32_t idx = idx_queue ? idx_queue[i] : i;
        const int32_t  s   = scores[i];
        hit_score[idx] = s > worst ? s : worst;
        hit_sink[idx]  = wb[i] + sinks[i].x;
    }
}

// sw-benchmark keeps its reference text 2-bit LITTLE-endian (sw-benchmark/sw-benchmark.cu:64-65: symbol i at bits [2(i&15), +2)
// of word i >> 4) and its scores as int16 (:197); the kernels here read big-endian text (io::SequenceData<DNA>) and write int32
__global__ void __launch_bounds__(256)
text_le_to_be_kernel(const uint32_t* __restrict__ in, const uint32_t n_words, uint32_t* __restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x)
    {
        uint32_t v = __brev( in[i] );                                      // symbol order reversed, the two bits of each swapped
        out[i] = ((v & 0x55555555u) << 1) | ((v >> 1) & 0x55555555u);
    }
}
__global__ void __launch_bounds__(256)
scores_to_int16_kernel(const int32_t* __restrict__ in, const uint32_t n, int16_t* __restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (int16_t)in[i];   // as `m_scores[i] = sink.score`
}

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" nvbio_status nvbio_text_2bit_le_to_be(int device, const uint32_t* in_dev, uint32_t n_words, uint32_t* out_dev, void* stream)
{
    if (n_words == 0) return NVBIO_OK;
    NVB_REQUIRE( in_dev && out_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( text_le_to_be_kernel, dim3( grid_for( n_words ) ), dim3(256), 0, (hipStream_t)stream, in_dev, n_words, out_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_scores_to_int16(int device, const int32_t* scores_dev, uint32_t n, int16_t* out_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && out_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( scores_to_int16_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, scores_dev, n, out_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_score_stream_flatten(int device, const nvbio_hit_queues* hits, const uint32_t* read_index_dev, uint32_t band_len,
                                                   uint32_t genome_len, uint32_t reads_reversed, uint32_t* read_id_dev, uint8_t* flags_dev,
                                                   uint32_t* win_begin_dev, uint32_t* win_end_dev, void* stream)
{
    NVB_REQUIRE( hits != nullptr, "hits is NULL" );
    if (hits->n == 0) return NVBIO_OK;
    NVB_REQUIRE( hits->hit_read_id_dev && hits->hit_seed_dev && hits->hit_loc_dev && read_index_dev, "NULL device pointer in the hit queues" );
    NVB_REQUIRE( read_id_dev && flags_dev && win_begin_dev && win_end_dev, "NULL output pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( score_stream_flatten_kernel, dim3( grid_for( hits->n ) ), dim3(256), 0, (hipStream_t)stream, hits->idx_queue_dev, hits->n,
                        hits->hit_read_id_dev, hits->hit_seed_dev, hits->hit_loc_dev, read_index_dev, band_len, genome_len, reads_reversed,
                        read_id_dev, flags_dev, win_begin_dev, win_end_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_score_stream_output(int device, const nvbio_hit_queues* hits, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                                  const uint32_t* win_begin_dev, int32_t worst_score, void* stream)
{
    NVB_REQUIRE( hits != nullptr, "hits is NULL" );
    if (hits->n == 0) return NVBIO_OK;
    NVB_REQUIRE( hits->hit_score_dev && hits->hit_sink_dev && scores_dev && sinks_dev && win_begin_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( score_stream_output_kernel, dim3( grid_for( hits->n ) ), dim3(256), 0, (hipStream_t)stream, hits->idx_queue_dev, hits->n,
                        scores_dev, (const uint2*)sinks_dev, win_begin_dev, worst_score, hits->hit_score_dev, hits->hit_sink_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

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

extern "C" nvbio_status nvbio_diagonals_to_windows_ragged(int device, const uint64_t* keys_dev, uint64_t n, uint32_t band, const uint32_t* read_offsets_dev,
                                                          uint32_t genome_len, uint32_t* read_id_dev, uint8_t* flags_dev,
                                                          uint32_t* win_begin_dev, uint32_t* win_end_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && read_offsets_dev && read_id_dev && flags_dev && win_begin_dev && win_end_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( diagonals_to_windows_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream,
                        keys_dev, n, band, 0u, genome_len, read_id_dev, flags_dev, win_begin_dev, win_end_dev, read_offsets_dev );
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

extern "C" nvbio_status nvbio_best_candidate_windows(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                                     const uint32_t* win_begin_dev, uint64_t n, const uint64_t* best_dev, int64_t* best_wb_dev,
                                                     int64_t* best_locus_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && scores_dev && sinks_dev && win_begin_dev && best_dev && best_wb_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( best_window_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, keys_dev, scores_dev, (const uint2*)sinks_dev,
                        win_begin_dev, n, (const unsigned long long*)best_dev, (long long*)best_wb_dev, (long long*)best_locus_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_traceback_best_batch(int device, const uint64_t* best_dev, const int64_t* best_wb_dev, uint32_t n_reads, uint32_t read_len,
                                                   uint32_t band, uint32_t genome_len, int32_t min_score, uint8_t* flags_dev, uint32_t* win_begin_dev,
                                                   uint32_t* win_end_dev, int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream)
{
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( best_dev && best_wb_dev && flags_dev && win_begin_dev && win_end_dev && scores_dev && sinks_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( traceback_best_batch_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)best_dev,
                        (const long long*)best_wb_dev, n_reads, read_len, band, genome_len, min_score, flags_dev, win_begin_dev, win_end_dev,
                        scores_dev, (uint2*)sinks_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_traceback_best_batch_ragged(int device, const uint64_t* best_dev, const int64_t* best_wb_dev, uint32_t n_reads,
                                                          const uint32_t* read_offsets_dev, uint32_t band, uint32_t genome_len, const int32_t* min_scores_dev,
                                                          uint8_t* flags_dev, uint32_t* win_begin_dev, uint32_t* win_end_dev, int32_t* scores_dev,
                                                          nvbio_uint2* sinks_dev, void* stream)
{
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( best_dev && best_wb_dev && read_offsets_dev && min_scores_dev && flags_dev && win_begin_dev && win_end_dev && scores_dev && sinks_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( traceback_best_batch_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)best_dev,
                        (const long long*)best_wb_dev, n_reads, 0u, band, genome_len, 0, flags_dev, win_begin_dev, win_end_dev,
                        scores_dev, (uint2*)sinks_dev, read_offsets_dev, min_scores_dev );
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

extern "C" nvbio_status nvbio_second_candidate_reduce(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                                      const uint32_t* win_begin_dev, uint64_t n, const uint64_t* best_dev,
                                                      uint32_t distinct_dist, int32_t worst_score, uint64_t* second_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && scores_dev && sinks_dev && win_begin_dev && best_dev && second_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( second_candidate_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream,
                        keys_dev, scores_dev, (const uint2*)sinks_dev, win_begin_dev, n, (const unsigned long long*)best_dev,
                        distinct_dist, worst_score, (unsigned long long*)second_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_second_candidate_reduce_ragged(int device, const uint64_t* keys_dev, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                                             const uint32_t* win_begin_dev, uint64_t n, const uint64_t* best_dev,
                                                             const uint32_t* read_offsets_dev, const int32_t* min_scores_dev, uint64_t* second_dev, void* stream)
{
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && scores_dev && sinks_dev && win_begin_dev && best_dev && second_dev && read_offsets_dev && min_scores_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( second_candidate_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream,
                        keys_dev, scores_dev, (const uint2*)sinks_dev, win_begin_dev, n, (const unsigned long long*)best_dev,
                        0u, 0, (unsigned long long*)second_dev, read_offsets_dev, min_scores_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_mapq_ragged(int device, const uint64_t* best_dev, const uint64_t* second_dev, uint32_t n_reads, int32_t version, int32_t match,
                                          const uint32_t* read_offsets_dev, const int32_t* min_scores_dev, int32_t* second_scores_dev, uint8_t* mapq_dev,
                                          void* stream)
{
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( best_dev && mapq_dev && read_offsets_dev && min_scores_dev, "NULL pointer" );
    NVB_REQUIRE( version == 2 || version == 3, "mapq version must be 2 or 3" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( mapq_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream,
                        (const unsigned long long*)best_dev, (const unsigned long long*)second_dev, n_reads, (int)version,
                        match == 0, 0.0f, 0.0f, second_scores_dev, mapq_dev, read_offsets_dev, min_scores_dev, match );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_mapq(int device, const uint64_t* best_dev, const uint64_t* second_dev, uint32_t n_reads,
                                   const nvbio_mapq_params* params, int32_t* second_scores_dev, uint8_t* mapq_dev, void* stream)
{
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( best_dev && mapq_dev && params, "NULL pointer" );
    NVB_REQUIRE( params->version == 2 || params->version == 3, "mapq version must be 2 or 3" );
    NVB_REQUIRE( params->perfect_score > params->min_score, "perfect_score must exceed min_score" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( mapq_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream,
                        (const unsigned long long*)best_dev, (const unsigned long long*)second_dev, n_reads, (int)params->version,
                        params->monotone != 0, (float)params->perfect_score, (float)params->min_score, second_scores_dev, mapq_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

// sort + unique of candidate keys: what fmmap does with its diagonals before extending them (examples/fmmap/fmmap.cu:320-344:
// sort_by_key on the diagonal, unique) for callers that have no device sort of their own (a C++ host composition over this ABI)
extern "C" nvbio_status nvbio_sort_unique_keys_temp_bytes(uint64_t n, uint64_t* bytes)
{
    NVB_REQUIRE( bytes != nullptr, "bytes is NULL" );
    NVB_REQUIRE( n < (1ull << 31), "n too large" );
    size_t a = 0, b = 0;
    NVB_HIP( hipcub::DeviceRadixSort::SortKeys( nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n, 0, 64, (hipStream_t)0 ) );
    NVB_HIP( hipcub::DeviceSelect::Unique( nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (int)n, (hipStream_t)0 ) );
    *bytes = (((a > b ? a : b) + 255u) & ~(uint64_t)255u) + n * sizeof(uint64_t) + 512u;
    return NVBIO_OK;
}

extern "C" nvbio_status nvbio_sort_unique_keys(int device, uint64_t* keys_dev, uint64_t n, uint32_t* n_out_dev, void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( n_out_dev != nullptr, "n_out_dev is NULL" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { NVB_HIP( hipMemsetAsync( n_out_dev, 0, sizeof(uint32_t), s ) ); return NVBIO_OK; }
    NVB_REQUIRE( keys_dev != nullptr, "keys_dev is NULL" );
    uint64_t need = 0; NVB_CHECK( nvbio_sort_unique_keys_temp_bytes( n, &need ) );
    uint8_t* temp = (uint8_t*)temp_dev; bool own = false;
    if (!temp)
    {
        if (scratch_alloc( (void**)&temp, need, s ) != hipSuccess) { (void)hipGetLastError(); set_error( "sort_unique_keys: out of device memory" ); return NVBIO_ERR_NOMEM; }
        own = true;
    }
    else NVB_REQUIRE( temp_bytes >= need, "temp_bytes too small (nvbio_sort_unique_keys_temp_bytes)" );
    uint8_t*  base   = (uint8_t*)(((uintptr_t)temp + 255u) & ~(uintptr_t)255u);
    uint64_t* sorted = (uint64_t*)base;
    void*     work   = base + ((n * sizeof(uint64_t) + 255u) & ~(uint64_t)255u);
    size_t    work_bytes = (size_t)(need - 512u - n * sizeof(uint64_t));
    hipError_t e = hipcub::DeviceRadixSort::SortKeys( work, work_bytes, (const uint64_t*)keys_dev, sorted, (int)n, 0, 64, s );
    if (e == hipSuccess) e = hipcub::DeviceSelect::Unique( work, work_bytes, (const uint64_t*)sorted, keys_dev, n_out_dev, (int)n, s );
    if (own) scratch_free( temp, s );
    if (e != hipSuccess) { set_error( "sort_unique_keys failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    return NVBIO_OK;
}
