// fm_device.h -- device-side FM-index arithmetic for gfx950.
//
// What is computed (and must agree bit for bit with the reference's CPU path):
//   rank(fmi,k,c)         nvbio/fmindex/fmindex_inl.h:27-47   over the production record layout
//   rank(fmi,(l,r),c)     fmindex_inl.h:56-88                 (nvbio/io/fmindex/fmindex.h:151-177)
//   one backward-search step of match()                       fmindex_inl.h:224-237
//   one LF step of locate()                                   fmindex_inl.h:380-392
//
// How it is computed here: a 64-symbol block is two 64-bit lanes of 2-bit symbols; the
// occurrences of c among its first p+1 symbols are popcount( eq_mask & prefix_mask ) on the
// two halves (no per-word loop and no c==0 correction term, unlike popc_2bit(mask,c,i) in
// nvbio/basic/popcount_inl.h:334-341), and rank over a range is the same pure function applied
// to both ends, sharing the 32-byte record when both ends fall in one block.
#pragma once
#include "common.h"

namespace nvbio_amd {

struct DevIndex
{
    uint32_t        length;
    uint32_t        primary;
    uint32_t        L2_0, L2_1, L2_2, L2_3, L2_4;
    const uint4*    rec;      // record k: rec[2k] = 64 BWT symbols, rec[2k+1] = occ{A,C,G,T}
    const uint32_t* ssa;      // ssa[j] = SA[j << sa_log], ssa[0] = 0xFFFFFFFF
    uint32_t        sa_log;   // log2 of the SA sampling interval (4 in the reference's layout)
    const uint2*    ktab;     // optional: SA range of every kmer-mer (in scan order), or NULL
    uint32_t        kmer;
    const uint2*    dtab;     // optional (full SA + text): the table of the direct seed pass, one symbol longer than ktab (dkmer = kmer + 1):
                              // SA range of every dkmer-mer, entries of k-mers with few occurrences rewritten to text positions
    uint32_t        dkmer;
    const uint2*    side;     // optional: the direct table's groups for k-mers with 2..7 occurrences (fm_seed_device.h)
    uint32_t        dmark;    // entries of dtab with hi >= dmark hold positions, not SA ranges (0xFFFFFFFF: format 1, no context)
    uint32_t        dctx;     // text symbols to the left of the occurrence that such an entry carries (15 or 0)
    const uint32_t* isa;      // optional (with a full SA): isa[p] = row of suffix p, isa[length] = 0
    const uint32_t* text;     // optional: the 2-bit packed text the index was built from
    const uint2*    ctab;     // optional (instead of dtab): the canonical two-strand table of ckmer-mers (fm_canon_inl.h)
    const uint2*    cside;    // its groups
    uint32_t        ckmer;    // odd
};

__device__ __forceinline__ uint32_t pick4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t i)
{
    const uint32_t lo = (i & 1u) ? b : a;
    const uint32_t hi = (i & 1u) ? d : c;
    return (i & 2u) ? hi : lo;
}
__device__ __forceinline__ uint32_t L2_of(const DevIndex& f, uint32_t c)    { return pick4(f.L2_0, f.L2_1, f.L2_2, f.L2_3, c); }
__device__ __forceinline__ uint32_t count_of(const DevIndex& f, uint32_t c) { return pick4(f.L2_1 - f.L2_0, f.L2_2 - f.L2_1, f.L2_3 - f.L2_2, f.L2_4 - f.L2_3, c); }

// occurrences of c among symbols [0, p] (p in 0..63) of a 64-symbol block
__device__ __forceinline__ uint32_t count_in_block(const uint4 b, const uint32_t p, const uint32_t c)
{
    const uint64_t hi = ((uint64_t)b.x << 32) | b.y;
    const uint64_t lo = ((uint64_t)b.z << 32) | b.w;
    const uint64_t cc = (uint64_t)c * 0x5555555555555555ull;
    const uint64_t th = hi ^ cc, tl = lo ^ cc;
    const uint64_t eh = ~(th | (th >> 1)) & 0x5555555555555555ull;
    const uint64_t el = ~(tl | (tl >> 1)) & 0x5555555555555555ull;
    const uint32_t nbits = 2u * (p + 1u);                       // 2..128 leading bits to keep
    const uint64_t mh = (nbits >= 64u) ? ~0ull : (~0ull << (64u - nbits));
    const uint64_t ml = (nbits <= 64u) ? 0ull : (~0ull << (128u - nbits));   // nbits == 128 -> shift 0
    return (uint32_t)__popcll( eh & mh ) + (uint32_t)__popcll( el & ml );
}

// counts of all four symbols among symbols [0,p] of a block
__device__ __forceinline__ uint4 count4_in_block(const uint4 b, const uint32_t p)
{
    return make_uint4( count_in_block( b, p, 0 ), count_in_block( b, p, 1 ),
                       count_in_block( b, p, 2 ), count_in_block( b, p, 3 ) );
}

__device__ __forceinline__ uint32_t bwt_symbol(const uint4 b, const uint32_t p)
{
    const uint32_t w = pick4( b.x, b.y, b.z, b.w, p >> 4 );
    return (w >> (30u - 2u * (p & 15u))) & 3u;
}

// Resolve a BWT-matrix row k to either a constant (no memory access) or a text index:
// k == -1 -> 0 ; k == length -> count(c) ; rows >= primary shift down by one because '$' is not
// stored (fmindex_inl.h:38-46).  Returns true when a block must be read; *kt is the text index.
__device__ __forceinline__ bool resolve_row(const DevIndex& f, uint32_t k, const uint32_t c, uint32_t* kt, uint32_t* fixed)
{
    if (k == 0xFFFFFFFFu) { *fixed = 0; return false; }
    if (k == f.length)    { *fixed = count_of( f, c ); return false; }
    if (k >= f.primary) --k;
    if (k == 0xFFFFFFFFu) { *fixed = 0; return false; }         // row 0 == primary
    *kt = k;
    return true;
}

__device__ __forceinline__ uint32_t rank_row(const DevIndex& f, const uint32_t k, const uint32_t c)
{
    uint32_t kt, v;
    if (!resolve_row( f, k, c, &kt, &v )) return v;
    const uint32_t blk = kt >> 6;
    const uint4 b = f.rec[2u * blk], o = f.rec[2u * blk + 1u];
    return pick4( o.x, o.y, o.z, o.w, c ) + count_in_block( b, kt & 63u, c );
}

// one backward-search step: (x,y) -> (L2[c] + rank(x-1,c) + 1, L2[c] + rank(y,c)).
// *nblocks (optional) accumulates the distinct 32-byte records touched.
template <bool COUNT>
__device__ __forceinline__ void search_step(const DevIndex& f, uint32_t& x, uint32_t& y, const uint32_t c, uint32_t& nblocks, uint32_t* nsectors = nullptr)
{
    uint32_t kl = 0, kh = 0, vl = 0, vh = 0;
    const bool ml = resolve_row( f, x - 1u, c, &kl, &vl );
    const bool mh = resolve_row( f, y,      c, &kh, &vh );
    const uint32_t bl = kl >> 6, bh = kh >> 6;

    uint4 b_l = make_uint4(0,0,0,0), o_l = b_l, b_h = b_l, o_h = b_l;
    if (ml) { b_l = f.rec[2u * bl]; o_l = f.rec[2u * bl + 1u]; }
    if (mh)
    {
        if (ml && bh == bl) { b_h = b_l; o_h = o_l; }
        else                { b_h = f.rec[2u * bh]; o_h = f.rec[2u * bh + 1u]; }
    }
    if (ml) vl = pick4( o_l.x, o_l.y, o_l.z, o_l.w, c ) + count_in_block( b_l, kl & 63u, c );
    if (mh) vh = pick4( o_h.x, o_h.y, o_h.z, o_h.w, c ) + count_in_block( b_h, kh & 63u, c );
    if (COUNT) nblocks += (ml ? 1u : 0u) + ((mh && !(ml && bh == bl)) ? 1u : 0u);
    if (nsectors) *nsectors += (ml ? 1u : 0u) + ((mh && !(ml && (bh >> 1) == (bl >> 1))) ? 1u : 0u);   // distinct 64-byte sectors (two records each)

    const uint32_t base = L2_of( f, c );
    x = base + vl + 1u;
    y = base + vh;
}

// one LF step of locate(): j -> L2[c] + rank(fmi,j,c) with c = bwt[j] (or 0 at the primary row)
__device__ __forceinline__ uint32_t lf_step(const DevIndex& f, const uint32_t j)
{
    if (j == f.primary) return 0;
    const uint32_t kt  = (j < f.primary) ? j : j - 1u;
    const uint32_t blk = kt >> 6;
    const uint4 b = f.rec[2u * blk], o = f.rec[2u * blk + 1u];
    const uint32_t c = bwt_symbol( b, kt & 63u );
    return L2_of( f, c ) + pick4( o.x, o.y, o.z, o.w, c ) + count_in_block( b, kt & 63u, c );
}

} // namespace nvbio_amd
