// pe_best_approx.hip -- the data-parallel steps of nvBowtie's PAIRED-END best-approx loop, for gfx950:
//   Aligner::best_approx, paired form                       nvBowtie/bowtie2/cuda/aligner_best_approx_paired.h:84-200,590-1000
//   BestAnchorScoreStream::init_context / output            nvBowtie/bowtie2/cuda/score_inl.h:143-274
//   BestOppositeScoreStream::init_context / output          nvBowtie/bowtie2/cuda/score_inl.h:283-456
//   compute_target_score, frame_opposite_mate               nvBowtie/bowtie2/cuda/alignment_utils.h:52-102
//   score_reduce_paired_kernel + ReduceBestApproxContext     nvBowtie/bowtie2/cuda/reduce_inl.h:157-290, reduce.h:55-99
//   io::Alignment / BestPairedAlignments / distinct_alignments   nvbio/io/alignments.h:71-330, alignments_inl.h:26-110
// For every anchor mate in turn the loop walks the anchor's seed hits as the single-end loop does (seed_hits.hip: deques, select); what
// differs is the scoring of a selected hit -- the anchor is band-aligned against a threshold derived from the best PAIRS found so far
// (`min_score` tightens as pairs are found: compute_target_score, Bowtie2's `tighten = 3`), hits whose anchor passes get the opposite
// mate aligned by full-matrix DP in the window the fragment constraints allow, again against a pair-derived threshold -- and the
// reduction, which keeps the best two PAIRS (falling back to per-mate unpaired bests while no pair has been found).
// One lane owns one hit (flatten / output) or one read (reduce): small state machines streamed once per extension pass.
//
// Alignment here = 4 x int32 { score, position (0xFFFFFFFF = not aligned), sink offset (the reference keeps it in its 10-bit `ed` field),
// flags = rc | mate << 1 | paired << 2 }; per read pair two arrays of two: best_a = { a1, a2 } (pipeline.best_alignments),
// best_o = { o1, o2 } (pipeline.best_alignments_o).
// One term of the reference is not reproduced because it reads an uninitialised variable: BestAnchorScoreStream::init_context tests
// `context->min_score > a_optimal_score` BEFORE it assigns context->min_score (score_inl.h:246 vs :249); it is taken as false.
#include "common.h"

namespace nvbio_amd {

struct PeAln { int32_t score; uint32_t pos; uint32_t sink; uint32_t flags; };
__device__ __forceinline__ bool     pe_aligned(const PeAln& a) { return a.pos != 0xFFFFFFFFu; }
__device__ __forceinline__ uint32_t pe_rc(const PeAln& a)      { return a.flags & 1u; }
__device__ __forceinline__ uint32_t pe_mate(const PeAln& a)    { return (a.flags >> 1) & 1u; }
__device__ __forceinline__ bool     pe_paired(const PeAln& a)  { return ((a.flags >> 2) & 1u) != 0u && pe_aligned( a ); }
__device__ __forceinline__ PeAln    pe_make(const uint32_t pos, const uint32_t sink, const int32_t score, const uint32_t rc, const uint32_t mate, const bool paired)
{
    PeAln a; a.score = score; a.pos = pos; a.sink = sink; a.flags = (rc & 1u) | ((mate & 1u) << 1) | (paired ? 4u : 0u);
    return a;
}

struct PeParams
{
    uint32_t anchor;                    // 0: mate 1 is the anchor
    uint32_t a_len, o_len;              // read lengths of the anchor / opposite mates (uniform)
    int32_t  a_opt, o_opt;              // perfect_score( len )
    int32_t  a_worst, o_worst;          // min_score( len )
    int32_t  score_limit;               // scheme.score_limit(): Field_traits<int32>::min() for the Smith-Waterman scheme (scoring.h:269)
    int32_t  worst_score;               // scheme_type::worst_score = -65536
    int32_t  match, txt_gap_open, txt_gap_ext;      // aln::max_text_gaps( GotohAligner ) (utils_inl.h:145-167)
    uint32_t band, genome_len;
    uint32_t policy, min_frag, max_frag, overlap, unpaired;
    uint32_t max_effort, min_ext, max_ext;
};

struct PeBest { PeAln a1, a2, o1, o2; };
__device__ __forceinline__ PeBest pe_load(const PeAln* __restrict__ best_a, const PeAln* __restrict__ best_o, const uint32_t r)
{
    PeBest b; b.a1 = best_a[2u * r]; b.a2 = best_a[2u * r + 1u]; b.o1 = best_o[2u * r]; b.o2 = best_o[2u * r + 1u];
    return b;
}
__device__ __forceinline__ int32_t pe_best_score(const PeBest& b)   { return b.a1.score + (pe_paired( b.a1 ) ? b.o1.score : 0); }
__device__ __forceinline__ int32_t pe_second_score(const PeBest& b) { return b.a2.score + (pe_paired( b.a2 ) ? b.o2.score : 0); }
// compute_target_score (alignment_utils.h:93-102)
__device__ __forceinline__ int32_t pe_target_score(const PeBest& b, const int32_t a_worst, const int32_t o_worst)
{
    if (!pe_paired( b.a2 )) return a_worst + o_worst;
    const int32_t delta = pe_best_score( b ) - pe_second_score( b );
    return pe_second_score( b ) + delta * 3 / 4;
}
__device__ __forceinline__ bool pe_visited(const PeBest& b, const uint32_t mate, const uint32_t rc, const uint32_t g)
{
    return (mate == pe_mate( b.a1 ) && rc == pe_rc( b.a1 ) && g == b.a1.pos) || (mate == pe_mate( b.o1 ) && rc == pe_rc( b.o1 ) && g == b.o1.pos) ||
           (mate == pe_mate( b.a2 ) && rc == pe_rc( b.a2 ) && g == b.a2.pos) || (mate == pe_mate( b.o2 ) && rc == pe_rc( b.o2 ) && g == b.o2.pos);
}
__device__ __forceinline__ void pe_frame(const uint32_t policy, const uint32_t anchor, const bool anchor_fw, bool& left, bool& fw)
{
    const bool anchor_1 = (anchor == 0u);
    switch (policy)
    {
    case NVBIO_PE_POLICY_FF: left = (anchor_1 != anchor_fw); fw =  anchor_fw; break;
    case NVBIO_PE_POLICY_RR: left = (anchor_1 == anchor_fw); fw =  anchor_fw; break;
    case NVBIO_PE_POLICY_FR: left = !anchor_fw;              fw = !anchor_fw; break;
    default:                 left =  anchor_fw;              fw = !anchor_fw; break;
    }
}

// init_alignments( reads1, threshold, best_data, 0 ) / ( reads2, threshold, best_data_o, 1 ) (aligner_best_approx_paired.h:80-81): the
// fourth constructor argument of io::Alignment is `rc`, so the `mate` the reference passes there lands in the strand bit
// (aligner_inl.h: io::Alignment( uint32(-1), max_ed(), worst_score, mate )) -- reproduced as it stands
__global__ void __launch_bounds__(256)
pe_init_kernel(const uint32_t n, const int32_t worst1, const int32_t worst2, PeAln* __restrict__ best_a, PeAln* __restrict__ best_o)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
    {
        best_a[2u * r] = best_a[2u * r + 1u] = pe_make( 0xFFFFFFFFu, 255u, worst1, 0u, 0u, false );
        best_o[2u * r] = best_o[2u * r + 1u] = pe_make( 0xFFFFFFFFu, 255u, worst2, 1u, 0u, false );
    }
}

// BestAnchorScoreStream::init_context for every selected hit: the banded window, the orientation (reads are stored reversed), the hit's
// min_score (INT32_MAX for a locus already held: the hit then scores as worst_score)
__global__ void __launch_bounds__(256)
pe_anchor_flatten_kernel(const PeParams p, const uint32_t n, const uint32_t* __restrict__ hit_read_id, const uint32_t* __restrict__ hit_seed,
                         const uint32_t* __restrict__ hit_loc, const PeAln* __restrict__ best_a, const PeAln* __restrict__ best_o,
                         uint32_t* __restrict__ read_id, uint8_t* __restrict__ flags, uint32_t* __restrict__ wb, uint32_t* __restrict__ we,
                         int32_t* __restrict__ min_score)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const uint32_t rid = hit_read_id[i], rc = (hit_seed[i] >> 13) & 1u, g = hit_loc[i];
        const PeBest b = pe_load( best_a, best_o, rid );
        const int32_t target_pair = pe_target_score( b, p.a_worst, p.o_worst );
        int32_t target_mate = target_pair - p.o_opt;
        target_mate = target_mate > p.a_worst ? target_mate : p.a_worst;
        const bool skip = pe_visited( b, p.anchor, rc, g );
        const uint32_t begin = g > p.band / 2u ? g - p.band / 2u : 0u;
        const uint32_t e = begin + p.band + p.a_len;
        const uint32_t end = e < p.genome_len ? e : p.genome_len;
        const bool bad = skip || begin >= p.genome_len || end < begin;          // (a locus that wrapped below zero: the empty window, as seed_extend.hip)
        read_id[i] = rid;
        flags[i]   = rc ? (uint8_t)NVBIO_READ_COMPLEMENT : (uint8_t)NVBIO_READ_REVERSE;
        wb[i] = bad ? 0u : begin; we[i] = bad ? 0u : end;
        const int32_t ms = target_mate + 1 > p.score_limit ? target_mate + 1 : p.score_limit;
        min_score[i] = skip ? 0x7FFFFFFF : ms;
    }
}

// BestAnchorScoreStream::output: hit.score = sink.score >= min_score ? sink.score : worst_score; hit.sink = genome_begin + sink.x;
// valid[i] = 1 where the anchor passed (the hits whose opposite mate is aligned next)
__global__ void __launch_bounds__(256)
pe_anchor_output_kernel(const uint32_t n, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks, const uint32_t* __restrict__ wb,
                        const int32_t* __restrict__ min_score, const int32_t worst, int32_t* __restrict__ hit_score, uint32_t* __restrict__ hit_sink,
                        int32_t* __restrict__ hit_oscore, uint8_t* __restrict__ valid)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const int32_t s = scores[i];
        const bool ok = s >= min_score[i];
        hit_score[i]  = ok ? s : worst;
        hit_sink[i]   = wb[i] + sinks[i].x;
        hit_oscore[i] = worst;                                                  // thrust::fill( opposite_score_queue, worst_score )
        valid[i] = ok && (ok ? s : worst) != worst ? 1 : 0;
    }
}

// BestOppositeScoreStream::init_context for the hits of `queue` (those whose anchor passed): threshold from the pairs found so far and this
// hit's anchor score, the window of the opposite mate, its orientation; a job that cannot run gets the empty window
__global__ void __launch_bounds__(256)
pe_opposite_flatten_kernel(const PeParams p, const uint32_t* __restrict__ queue, const uint32_t n, const uint32_t* __restrict__ hit_read_id,
                           const uint32_t* __restrict__ hit_seed, const uint32_t* __restrict__ hit_loc, const int32_t* __restrict__ hit_score,
                           const PeAln* __restrict__ best_a, const PeAln* __restrict__ best_o, uint32_t* __restrict__ read_id, uint8_t* __restrict__ flags,
                           uint32_t* __restrict__ wb, uint32_t* __restrict__ we, int32_t* __restrict__ min_score)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    {
        const uint32_t i = queue[j];
        const uint32_t rid = hit_read_id[i], rc = (hit_seed[i] >> 13) & 1u, g = hit_loc[i];
        const PeBest b = pe_load( best_a, best_o, rid );
        const int32_t target_pair = pe_target_score( b, p.a_worst, p.o_worst );
        int32_t target_mate = target_pair - hit_score[i];
        target_mate = target_mate > p.o_worst ? target_mate : p.o_worst;
        const int32_t ms = target_mate + 1 > p.score_limit ? target_mate + 1 : p.score_limit;
        bool run = !(ms > p.o_opt);
        bool o_left, o_fw;
        pe_frame( p.policy, p.anchor, rc == 0u, o_left, o_fw );
        // aln::max_text_gaps( aligner, min_score, o_len ) (utils_inl.h:145-167), int32 as the caller stores it
        int32_t gaps = 0;
        {
            int32_t score = (int32_t)p.o_len * p.match;
            if (score >= ms)
            {
                score += p.txt_gap_open;
                uint32_t k = 0;
                while (score >= ms && k < p.o_len) { score += p.txt_gap_ext; ++k; }
                gaps = (int32_t)(k - 1u);
            }
        }
        const uint32_t o_gapped = p.o_len + (uint32_t)gaps;
        uint32_t begin, end;
        if (o_left)
        {
            const uint32_t max_end = g + p.a_len + o_gapped > p.min_frag ? g + p.a_len + o_gapped - p.min_frag : 0u;
            begin = g + p.a_len > p.max_frag ? (g + p.a_len) - p.max_frag : 0u;
            end   = p.overlap ? g + p.a_len : g;
            end   = end < max_end ? end : max_end;
        }
        else
        {
            const uint32_t min_begin = g + p.min_frag > o_gapped ? g + p.min_frag - o_gapped : 0u;
            end   = g + p.max_frag;
            begin = p.overlap ? g : g + p.a_len;
            begin = begin > min_begin ? begin : min_begin;
        }
        end = end < p.genome_len ? end : p.genome_len;
        if (begin >= p.genome_len) run = false;
        const uint32_t o_rc = o_fw ? 0u : 1u;
        if (pe_visited( b, p.anchor ? 0u : 1u, o_rc, g ) || begin == end) run = false;
        if (end < begin) run = false;                                           // (uint32 wrap-around near the genome start: nothing to align)
        read_id[j] = rid;
        flags[j]   = o_rc ? (uint8_t)NVBIO_READ_COMPLEMENT : (uint8_t)NVBIO_READ_REVERSE;
        wb[j] = run ? begin : 0u; we[j] = run ? end : 0u;
        min_score[j] = ms;
    }
}

// BestOppositeScoreStream::output
__global__ void __launch_bounds__(256)
pe_opposite_output_kernel(const uint32_t* __restrict__ queue, const uint32_t n, const int32_t* __restrict__ scores, const uint2* __restrict__ sinks,
                          const uint32_t* __restrict__ wb, const uint32_t* __restrict__ we, const int32_t* __restrict__ min_score, const int32_t worst,
                          int32_t* __restrict__ hit_oscore, uint32_t* __restrict__ hit_oloc, uint32_t* __restrict__ hit_osink)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    {
        const uint32_t i = queue[j];
        const bool ran = we[j] > wb[j];
        const int32_t s = scores[j];
        hit_oscore[i] = (ran && s >= min_score[j]) ? s : worst;
        hit_oloc[i]   = wb[j];
        hit_osink[i]  = wb[j] + (sinks[j].x != 0xFFFFFFFFu ? sinks[j].x : 0u);
    }
}

// io::distinct_alignments for pairs (alignments_inl.h:44-110): compared on the END positions (alignment + sink) of mate 0 and mate 1
__device__ __forceinline__ const PeAln& pe_pair_mate(const PeAln& a, const PeAln& o, const uint32_t m) { return m == pe_mate( a ) ? a : o; }
__device__ __forceinline__ bool pe_pairs_distinct(const PeAln& a1, const PeAln& o1, const PeAln& a2, const PeAln& o2, const uint32_t dist, const bool with_dist)
{
    const PeAln &p10 = pe_pair_mate( a1, o1, 0u ), &p11 = pe_pair_mate( a1, o1, 1u ), &p20 = pe_pair_mate( a2, o2, 0u ), &p21 = pe_pair_mate( a2, o2, 1u );
    const uint32_t ap1 = p10.pos + p10.sink, op1 = p11.pos + p11.sink, ap2 = p20.pos + p20.sink, op2 = p21.pos + p21.sink;
    const bool arc1 = pe_rc( p10 ) != 0u, orc1 = pe_rc( p11 ) != 0u, arc2 = pe_rc( p20 ) != 0u, orc2 = pe_rc( p21 ) != 0u;
    if (!with_dist) return (arc1 != arc2) || (orc1 != orc2) || (ap1 != ap2) || (op1 != op2);
    if (arc1 != arc2 || orc1 != orc2) return true;
    const bool near_a = ap1 >= ap2 - (ap2 < dist ? ap2 : dist) && ap1 <= ap2 + dist;
    const bool near_o = op1 >= op2 - (op2 < dist ? op2 : dist) && op1 <= op2 + dist;
    return !(near_a && near_o);
}
__device__ __forceinline__ bool pe_distinct(const uint32_t pos1, const uint32_t rc1, const uint32_t pos2, const uint32_t rc2, const uint32_t dist)
{
    if (rc1 != rc2) return true;
    return !(pos1 >= pos2 - (pos2 < dist ? pos2 : dist) && pos1 <= pos2 + dist);
}

// score_reduce_paired_kernel over the hits of every active read, in selection order
__global__ void __launch_bounds__(256)
pe_reduce_kernel(const PeParams p, const uint32_t* __restrict__ active, const uint32_t n, const uint32_t* __restrict__ hits_first,
                 const uint32_t* __restrict__ hits_count, const uint32_t* __restrict__ hit_seed, const uint32_t* __restrict__ hit_loc,
                 const uint32_t* __restrict__ hit_sink, const int32_t* __restrict__ hit_score, const int32_t* __restrict__ hit_oscore,
                 const uint32_t* __restrict__ hit_oloc, const uint32_t* __restrict__ hit_osink, const uint32_t ext,
                 PeAln* __restrict__ best_a, PeAln* __restrict__ best_o, uint32_t* __restrict__ trys, uint32_t* __restrict__ sizes)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
    {
        const uint32_t read = active[t] & 0x7FFFFFFFu;
        const uint32_t first = hits_first[t], count = hits_count[t];
        // b = the kernel's local best_pairs; m = what it has written to memory.  The reference updates both on a paired update but only
        // memory on an unpaired one (reduce_inl.h:262-273 write `best`, not `best_pairs`), so with several hits per read a later hit sees the
        // local copy without the unpaired updates made before it, and a later paired update writes that copy over them: reproduced
        PeBest b = pe_load( best_a, best_o, read ), m = b;
        uint32_t tr = trys[read];
        auto failure = [&](const uint32_t idx, const uint32_t top_flag) -> bool {       // ReduceBestApproxContext::failure (reduce.h:82-92)
            if (tr > 0u)
            {
                if ((ext + idx >= p.min_ext && top_flag == 0u && --tr == 0u) || (ext + idx >= p.max_ext)) return true;
            }
            return false;
        };
        for (uint32_t idx = 0; idx < count; ++idx)
        {
            const uint32_t i = first + idx;
            const uint32_t rc = (hit_seed[i] >> 13) & 1u, top_flag = (hit_seed[i] >> 14) & 1u;
            const uint32_t gx = hit_loc[i], gy = hit_sink[i];
            const int32_t  s1 = hit_score[i], s2 = hit_oscore[i], score = s1 + s2;
            bool o_left, o_fw;
            pe_frame( p.policy, p.anchor, rc == 0u, o_left, o_fw );
            const uint32_t ox = hit_oloc[i], oy = hit_osink[i], o_rc = o_fw ? 0u : 1u;
            const PeAln pa = pe_make( gx, gy - gx, s1, rc, p.anchor, true ), po = pe_make( ox, oy - ox, s2, o_rc, p.anchor ? 0u : 1u, true );
            // a pair already held: free
            if (!pe_pairs_distinct( b.a1, b.o1, pa, po, 0u, false ) || !pe_pairs_distinct( b.a2, b.o2, pa, po, 0u, false )) continue;
            if (score > pe_best_score( b ))
            {
                tr = p.max_effort;
                b.a2 = b.a1; b.o2 = b.o1; b.a1 = pa; b.o1 = po;
                m = b;
            }
            else if (score > pe_second_score( b ) && pe_pairs_distinct( b.a1, b.o1, pa, po, p.a_len / 2u, true ))
            {
                tr = p.max_effort;
                b.a2 = pa; b.o2 = po;
                m = b;
            }
            else if (p.unpaired && !pe_paired( b.a1 ))
            {
                // no pair yet: the best two alignments of each mate on its own, mate 1's in best_a, mate 2's in best_o
                PeAln m1 = p.anchor == pe_mate( b.a1 ) ? b.a1 : b.o1, m2 = p.anchor == pe_mate( b.a2 ) ? b.a2 : b.o2;     // best_pairs.mate( anchor )
                bool wrote = false;
                if (s1 > m1.score)      { m2 = m1; m1 = pe_make( gx, gy - gx, s1, rc, p.anchor, false ); wrote = true; }
                else if (s1 > m2.score && pe_distinct( m1.pos, pe_rc( m1 ), gx, rc, p.a_len / 2u )) { m2 = pe_make( gx, gy - gx, s1, rc, p.anchor, false ); wrote = true; }
                else if (failure( idx, top_flag )) sizes[read] = 0u;
                if (wrote) { if (p.anchor) { m.o1 = m1; m.o2 = m2; } else { m.a1 = m1; m.a2 = m2; } }
            }
            else if (failure( idx, top_flag )) sizes[read] = 0u;
        }
        best_a[2u * read] = m.a1; best_a[2u * read + 1u] = m.a2; best_o[2u * read] = m.o1; best_o[2u * read + 1u] = m.o2;
        trys[read] = tr;
    }
}

} // namespace nvbio_amd

using namespace nvbio_amd;

static PeParams pe_params(const nvbio_pe_params* q)
{
    PeParams p;
    p.anchor = q->anchor; p.a_len = q->anchor_len; p.o_len = q->opposite_len; p.a_opt = q->anchor_perfect_score; p.o_opt = q->opposite_perfect_score;
    p.a_worst = q->anchor_min_score; p.o_worst = q->opposite_min_score; p.score_limit = q->score_limit; p.worst_score = q->worst_score;
    p.match = q->match; p.txt_gap_open = q->txt_gap_open; p.txt_gap_ext = q->txt_gap_ext; p.band = q->band; p.genome_len = q->genome_len;
    p.policy = q->policy; p.min_frag = q->min_frag_len; p.max_frag = q->max_frag_len; p.overlap = q->overlap; p.unpaired = q->unpaired;
    p.max_effort = q->max_effort; p.min_ext = q->min_ext; p.max_ext = q->max_ext;
    return p;
}

extern "C" {

nvbio_status nvbio_pe_init(int device, uint32_t n_reads, int32_t worst_score_mate1, int32_t worst_score_mate2, int32_t* best_a_dev, int32_t* best_o_dev, void* stream)
{
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( best_a_dev && best_o_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( pe_init_kernel, dim3( grid_for( n_reads ) ), dim3(256), 0, (hipStream_t)stream, n_reads, worst_score_mate1, worst_score_mate2,
                        (PeAln*)best_a_dev, (PeAln*)best_o_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_pe_anchor_flatten(int device, const nvbio_pe_params* params, const nvbio_hit_queues* hits, const int32_t* best_a_dev, const int32_t* best_o_dev,
                                     uint32_t* read_id_dev, uint8_t* flags_dev, uint32_t* win_begin_dev, uint32_t* win_end_dev, int32_t* min_scores_dev,
                                     void* stream)
{
    NVB_REQUIRE( params && hits, "NULL argument" );
    if (hits->n == 0) return NVBIO_OK;
    NVB_REQUIRE( hits->hit_read_id_dev && hits->hit_seed_dev && hits->hit_loc_dev && best_a_dev && best_o_dev && read_id_dev && flags_dev && win_begin_dev &&
                 win_end_dev && min_scores_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( pe_anchor_flatten_kernel, dim3( grid_for( hits->n ) ), dim3(256), 0, (hipStream_t)stream, pe_params( params ), hits->n, hits->hit_read_id_dev,
                        hits->hit_seed_dev, hits->hit_loc_dev, (const PeAln*)best_a_dev, (const PeAln*)best_o_dev, read_id_dev, flags_dev, win_begin_dev,
                        win_end_dev, min_scores_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_pe_anchor_output(int device, const nvbio_pe_params* params, const nvbio_hit_queues* hits, const int32_t* scores_dev, const nvbio_uint2* sinks_dev,
                                    const uint32_t* win_begin_dev, const int32_t* min_scores_dev, int32_t* hit_opposite_score_dev, uint8_t* valid_dev,
                                    void* stream)
{
    NVB_REQUIRE( params && hits, "NULL argument" );
    if (hits->n == 0) return NVBIO_OK;
    NVB_REQUIRE( hits->hit_score_dev && hits->hit_sink_dev && scores_dev && sinks_dev && win_begin_dev && min_scores_dev && hit_opposite_score_dev && valid_dev,
                 "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( pe_anchor_output_kernel, dim3( grid_for( hits->n ) ), dim3(256), 0, (hipStream_t)stream, hits->n, scores_dev, (const uint2*)sinks_dev,
                        win_begin_dev, min_scores_dev, params->worst_score, hits->hit_score_dev, hits->hit_sink_dev, hit_opposite_score_dev, valid_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_pe_opposite_flatten(int device, const nvbio_pe_params* params, const uint32_t* queue_dev, uint32_t n, const nvbio_hit_queues* hits,
                                       const int32_t* best_a_dev, const int32_t* best_o_dev, uint32_t* read_id_dev, uint8_t* flags_dev,
                                       uint32_t* win_begin_dev, uint32_t* win_end_dev, int32_t* min_scores_dev, void* stream)
{
    NVB_REQUIRE( params && hits, "NULL argument" );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( queue_dev && hits->hit_read_id_dev && hits->hit_seed_dev && hits->hit_loc_dev && hits->hit_score_dev && best_a_dev && best_o_dev && read_id_dev &&
                 flags_dev && win_begin_dev && win_end_dev && min_scores_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( pe_opposite_flatten_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, pe_params( params ), queue_dev, n, hits->hit_read_id_dev,
                        hits->hit_seed_dev, hits->hit_loc_dev, hits->hit_score_dev, (const PeAln*)best_a_dev, (const PeAln*)best_o_dev, read_id_dev, flags_dev,
                        win_begin_dev, win_end_dev, min_scores_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_pe_opposite_output(int device, const nvbio_pe_params* params, const uint32_t* queue_dev, uint32_t n, const int32_t* scores_dev,
                                      const nvbio_uint2* sinks_dev, const uint32_t* win_begin_dev, const uint32_t* win_end_dev, const int32_t* min_scores_dev,
                                      int32_t* hit_opposite_score_dev, uint32_t* hit_opposite_loc_dev, uint32_t* hit_opposite_sink_dev, void* stream)
{
    NVB_REQUIRE( params != nullptr, "params is NULL" );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( queue_dev && scores_dev && sinks_dev && win_begin_dev && win_end_dev && min_scores_dev && hit_opposite_score_dev && hit_opposite_loc_dev &&
                 hit_opposite_sink_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( pe_opposite_output_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, queue_dev, n, scores_dev, (const uint2*)sinks_dev,
                        win_begin_dev, win_end_dev, min_scores_dev, params->worst_score, hit_opposite_score_dev, hit_opposite_loc_dev, hit_opposite_sink_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_pe_score_reduce(int device, const nvbio_pe_params* params, const uint32_t* active_dev, uint32_t n_active, const uint32_t* hits_first_dev,
                                   const uint32_t* hits_count_dev, const nvbio_hit_queues* hits, const int32_t* hit_opposite_score_dev,
                                   const uint32_t* hit_opposite_loc_dev, const uint32_t* hit_opposite_sink_dev, uint32_t n_ext, int32_t* best_a_dev,
                                   int32_t* best_o_dev, uint32_t* trys_dev, uint32_t* sizes_dev, void* stream)
{
    NVB_REQUIRE( params && hits, "NULL argument" );
    if (n_active == 0) return NVBIO_OK;
    NVB_REQUIRE( active_dev && hits_first_dev && hits_count_dev && hits->hit_seed_dev && hits->hit_loc_dev && hits->hit_sink_dev && hits->hit_score_dev &&
                 hit_opposite_score_dev && hit_opposite_loc_dev && hit_opposite_sink_dev && best_a_dev && best_o_dev && trys_dev && sizes_dev, "NULL device pointer" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( pe_reduce_kernel, dim3( grid_for( n_active ) ), dim3(256), 0, (hipStream_t)stream, pe_params( params ), active_dev, n_active, hits_first_dev,
                        hits_count_dev, hits->hit_seed_dev, hits->hit_loc_dev, hits->hit_sink_dev, hits->hit_score_dev, hit_opposite_score_dev,
                        hit_opposite_loc_dev, hit_opposite_sink_dev, n_ext, (PeAln*)best_a_dev, (PeAln*)best_o_dev, trys_dev, sizes_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

} // extern "C"
