// myers.hip -- the Myers bit-vector aligner of the reference, batched, for gfx950.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   aln::banded_alignment_score<BAND>( EditDistanceAligner<TYPE, MyersTag<A>>, ... )   nvbio/alignment/myers/myers_banded_inl.h:296-315
//   banded_myers<BAND, 0, TYPE, A>, diagonal_column, horizontal_column                :172-294
//   its caller: examples/fmmap/fmmap.cu:346-359 (SEMI_GLOBAL, MyersTag<5>, band 31)
// A band of BAND bits slides down the main diagonal of the (text x pattern) matrix: while pattern symbols are still entering the
// band every text symbol is a diagonal step, afterwards a horizontal one; the running value is MINUS the edit distance, reported
// for every column of the horizontal phase (SEMI_GLOBAL; BestSink keeps the last best) or for the final one (GLOBAL).  As the
// code behaves: min_score is an int16 there (:258), so the caller's int32 is truncated (Field_traits<int32>::min() becomes 0).
//
// MI355X design: one lane per alignment, the whole state in five 32-bit match vectors + VP / VN: ~25 integer ops per text symbol,
// no LDS, no cross-lane traffic; reads 2 / 4 / 8 bits per symbol (optionally reversed / complemented), text 2 or 8 bits.  HBM traffic is
// the packed inputs once: the kernel is bound by instruction issue like the DP kernels, at a fiftieth of their work per alignment.
#include "gotoh_common.h"

namespace nvbio_amd {

template <int RBITS, int TBITS>
__global__ void __launch_bounds__(256)
banded_myers_kernel(const BatchDev b, const uint32_t band, const int type, const int32_t min_score32, int32_t* __restrict__ scores, uint2* __restrict__ sinks)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= b.n) return;
    const uint32_t rid   = b.read_id ? b.read_id[job] : job;
    const uint32_t first = b.read_offsets[rid];
    const uint32_t M     = b.read_offsets[rid + 1] - first;
    const uint32_t fl    = b.flags ? b.flags[job] : 0u;
    const bool     rev   = (fl & NVBIO_READ_REVERSE) != 0, comp = (fl & NVBIO_READ_COMPLEMENT) != 0;
    const uint32_t tb    = b.win_begin[job];
    const uint32_t N     = (b.max_read_len && M > b.max_read_len) ? 0u : b.win_end[job] - tb;

    int32_t  best = NVBIO_SCORE_MIN;
    uint32_t bx = 0xFFFFFFFFu, by = 0xFFFFFFFFu;
    if (N >= M)
    {
        const int32_t min_score = (int32_t)(int16_t)min_score32;            // `const int16 min_score`
        SymbolReader<RBITS> pr( b.reads );
        SymbolReader<TBITS> tr( b.text );
        auto pattern = [&](const uint32_t i) -> uint32_t {
            const uint32_t c = pr.get( rev ? first + M - 1u - i : first + i );
            return (comp && c < 4u) ? 3u - c : c;
        };
        uint32_t B0 = 0, B1 = 0, B2 = 0, B3 = 0, B4 = 0;                     // MyersBitVectors<5>
        uint32_t VP = 0xFFFFFFFFu, VN = 0u;
        int32_t  dist = 0;
        const uint32_t last = (N - 1u < M) ? N - 1u : M;
        const uint32_t top  = 1u << (band - 1u);
        auto column = [&](const uint32_t eq, uint32_t& D0, uint32_t& HP, uint32_t& HN) {
            uint32_t X = eq | VN;
            D0 = ((VP + (X & VP)) ^ VP) | X;
            HN = VP & D0;
            HP = VN | ~(VP | D0);
            X  = D0 >> 1;
            VN = X & HP;
            VP = HN | ~(X | HP);
        };
        auto match_vector = [&](const uint32_t c) -> uint32_t {
            return c <= 1u ? (c == 0u ? B0 : B1) : c <= 3u ? (c == 2u ? B2 : B3) : B4;
        };
        for (uint32_t i = 0; i < last; ++i)
        {
            B0 >>= 1; B1 >>= 1; B2 >>= 1; B3 >>= 1; B4 >>= 1;
            const uint32_t p = pattern( i );
            B0 |= p == 0u ? top : 0u; B1 |= p == 1u ? top : 0u; B2 |= p == 2u ? top : 0u; B3 |= p == 3u ? top : 0u; B4 |= p == 4u ? top : 0u;
            uint32_t D0, HP, HN;
            column( match_vector( tr.get( tb + i ) ), D0, HP, HN );
            dist -= 1 - (int32_t)((D0 >> (band - 1u)) & 1u);
        }
        int32_t s = (int32_t)band - 1 + (int32_t)M - (int32_t)last;
        for (uint32_t i = last; i < N && s >= 0; ++i, --s)
        {
            B0 >>= 1; B1 >>= 1; B2 >>= 1; B3 >>= 1; B4 >>= 1;
            uint32_t D0, HP, HN;
            column( match_vector( tr.get( tb + i ) ), D0, HP, HN );
            dist -= (int32_t)((HP >> s) & 1u) - (int32_t)((HN >> s) & 1u);
            if (type == NVBIO_SEMI_GLOBAL && dist >= min_score && best <= dist) { best = dist; bx = i + 1u; by = M; }
        }
        if (type == NVBIO_GLOBAL && dist >= min_score && best <= dist) { best = dist; bx = N; by = M; }
    }
    scores[job] = best; sinks[job] = make_uint2( bx, by );
}

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" nvbio_status nvbio_banded_myers_score(int device, uint32_t band, nvbio_alignment_type type, const nvbio_alignment_batch* batch, int32_t min_score,
                                                 int32_t* scores_dev, nvbio_uint2* sinks_dev, void* stream)
{
    BatchDev b; NVB_CHECK( make_batch( batch, &b ) );
    NVB_REQUIRE( band >= 1 && band <= 32, "band must be 1..32 (one 32-bit vector)" );
    NVB_REQUIRE( type == NVBIO_GLOBAL || type == NVBIO_SEMI_GLOBAL, "the Myers aligner reports GLOBAL and SEMI_GLOBAL alignments only" );
    if (b.n == 0) return NVBIO_OK;
    NVB_REQUIRE( scores_dev && sinks_dev, "NULL output pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    const dim3 grid( (b.n + 255u) / 256u ), block( 256 );
    hipStream_t s = (hipStream_t)stream;
#define NVB_MY(RB, TB) hipLaunchKernelGGL( (banded_myers_kernel<RB,TB>), grid, block, 0, s, b, band, (int)type, min_score, scores_dev, (uint2*)sinks_dev )
    const uint32_t rb = batch->read_bits, tbits = batch->text_bits;
    if      (rb == 4 && tbits == 2) NVB_MY( 4, 2 );
    else if (rb == 2 && tbits == 2) NVB_MY( 2, 2 );
    else if (rb == 8 && tbits == 2) NVB_MY( 8, 2 );
    else if (rb == 8 && tbits == 8) NVB_MY( 8, 8 );
    else { set_error( "unsupported read_bits / text_bits combination %u / %u", rb, tbits ); return NVBIO_ERR_UNSUPPORTED; }
#undef NVB_MY
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}
