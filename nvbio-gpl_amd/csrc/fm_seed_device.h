// fm_seed_device.h -- device helpers of the direct seed pass (gfx950): a whole seed in one register, and the
// direct k-mer table whose entries carry the text to the left of a k-mer's occurrence(s).
//
// What is computed is the reference's match() / match_range over a seed (nvbio/fmindex/fmindex_inl.h:181-278,
// nvBowtie/bowtie2/cuda/mapping_inl.h:73-86) followed, for a search that ends on ONE suffix-array row, by locate() of that
// row (fmindex_inl.h:360-394).  How: the first k symbols of the seed (in scan order) index a table built once per index
// (fm_index.hip: build_kmer_table); the entry of a k-mer with one occurrence holds that occurrence's text position p and the
// DCTX text symbols to its left, text[p-1], text[p-2], ..., so that the rest of the seed is verified against the entry itself:
// one gather for the whole search.  A k-mer with 2..7 occurrences points to a small group (position + left context of every
// occurrence, one 32- or 64-byte sector); only k-mers with more occurrences fall back to rank steps.
#pragma once
#include "fm_device.h"

namespace nvbio_amd {

// ---------------------------------------------------------------------------------------------
// direct table, format 2 (DevIndex::dmark / dctx / side):
//   entry (lo, hi) of a k-mer in scan order
//     hi <  dmark                : the SA range (lo, hi) of the k-mer, as match() leaves it (empty ranges keep the
//                                  reference's early-exit values)
//     hi >= dmark, lo <  dmark   : ONE occurrence: lo = its text position p (SA[row]; row 0 of the matrix -> length),
//                                  hi - dmark = left context: symbol text[p-1-t] at bits [2(dctx-1-t), +2), t < min(p, dctx)
//     hi >= dmark, lo >= dmark   : m = hi - dmark in 2..7 occurrences: group g = lo - dmark, in units of 4 slots, in `side`:
//                                  slot 0 = the SA range (x, y) of the k-mer, slots 1..m = (position, dmark | context) of rows
//                                  x .. y in that order; groups of 2..3 rows take 4 slots (32 bytes, 32-byte aligned), of
//                                  4..7 rows 8 slots (64 bytes, 64-byte aligned)
//   dmark = 0xC0000000 and dctx = 15 when every row and position is below 0xC0000000 - 2 (texts up to 3.22 G symbols);
//   otherwise dmark = 0xFFFFFFFF, dctx = 0: only one-row entries are rewritten, to (p, 0xFFFFFFFF) (format 1).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t DTAB_MARK      = 0xC0000000u;
constexpr uint32_t DTAB_CTX       = 15u;
constexpr uint32_t DTAB_SIDE_MAX  = 7u;          // largest range resolved through a side group

// text[p-1-t] for t < min(p, DTAB_CTX), symbol t at bits [2(DTAB_CTX-1-t), +2)
__device__ __forceinline__ uint32_t left_context(const uint32_t* __restrict__ text, const uint32_t p)
{
    uint32_t ctx = 0;
    const uint32_t nsym = p < DTAB_CTX ? p : DTAB_CTX;
    if (nsym == 0) return 0;
    // the symbols text[p-nsym, p) lie in at most two consecutive words
    const uint32_t lo = p - nsym, w0 = lo >> 4, w1 = (p - 1u) >> 4;
    const uint32_t a = text[w0], b = (w1 != w0) ? text[w1] : 0u;
    const uint64_t two = ((uint64_t)a << 32) | b;                           // symbols [16 w0, 16 w0 + 32), MSB first
    // text[p-1-t] = symbol index (p-1-t) - 16 w0 of `two`
    #pragma unroll
    for (uint32_t t = 0; t < DTAB_CTX; ++t)
        if (t < nsym)
        {
            const uint32_t k = (p - 1u - t) - (w0 << 4);
            ctx |= (uint32_t)((two >> (62u - 2u * k)) & 3u) << (2u * (DTAB_CTX - 1u - t));
        }
    return ctx;
}

// ---------------------------------------------------------------------------------------------
// A seed of up to 32 symbols as one 64-bit value in SCAN order: symbol s (the s-th symbol match() consumes) at bits
// [2(len-1-s), +2), i.e. the first symbol scanned is the most significant.  Returns false if the seed holds a symbol > 3.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t squeeze_nibbles(uint64_t x)           // 16 nibbles -> their low 2 bits, order kept (32 bits)
{
    x &= 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}

__device__ __forceinline__ uint64_t reverse_symbols(uint64_t v, const uint32_t len)     // symbol order reversed, low-aligned
{
    v = __brevll( v );                                                     // bits reversed: symbols reversed, each with its two bits swapped
    v = ((v & 0x5555555555555555ull) << 1) | ((v >> 1) & 0x5555555555555555ull);
    return v >> (64u - 2u * len);
}

// the (up to 5) packed words that hold a seed: loaded in one go, decoded later (the pipelined seed pass loads the words of
// the tile after next while it works on the current one)
struct SeedWords { uint32_t w0, w1, w2, w3, w4; };

template <int BITS>
__device__ __forceinline__ SeedWords load_seed_words(const void* __restrict__ symbols, const uint32_t begin, const uint32_t len)
{
    constexpr uint32_t LOG = (BITS == 4) ? 3u : 4u;                          // symbols per word: 8 (4-bit) or 16 (2-bit)
    const uint32_t* w = (const uint32_t*)symbols + (begin >> LOG);
    const uint32_t nw = ((begin + len - 1u) >> LOG) - (begin >> LOG) + 1u;   // words that hold the seed: only those are touched
    SeedWords W;
    W.w0 = w[0];
    W.w1 = nw > 1u ? w[1] : 0u;
    W.w2 = nw > 2u ? w[2] : 0u;
    W.w3 = (BITS == 4 && nw > 3u) ? w[3] : 0u;
    W.w4 = (BITS == 4 && nw > 4u) ? w[4] : 0u;
    return W;
}

template <int BITS>
__device__ __forceinline__ bool seed_bits_from_words(const SeedWords& W, const uint32_t begin, const uint32_t len, const bool fwd, const bool comp, uint64_t& V)
{
    // stream order first: symbol j of the seed at bits [2(len-1-j), +2)
    uint64_t P; bool clean = true;
    if (BITS == 4)
    {
        const uint32_t sh = (begin & 7u) * 4u;
        const uint64_t A = ((uint64_t)W.w0 << 32) | W.w1, B = ((uint64_t)W.w2 << 32) | W.w3, C = (uint64_t)W.w4 << 32;
        const uint64_t X = sh ? (A << sh) | (B >> (64u - sh)) : A;             // nibbles 0..15 of the seed
        const uint64_t Y = sh ? (B << sh) | (C >> (64u - sh)) : B;             // nibbles 16..31
        const uint64_t mx = len >= 16u ? ~0ull : ~0ull << (64u - 4u * len);
        const uint64_t my = len <= 16u ? 0ull : (len >= 32u ? ~0ull : ~0ull << (64u - 4u * (len - 16u)));
        clean = (((X & mx) | (Y & my)) & 0xCCCCCCCCCCCCCCCCull) == 0ull;
        P = ((squeeze_nibbles( X ) << 32) | squeeze_nibbles( Y )) >> (64u - 2u * len);
    }
    else
    {
        const uint32_t sh = (begin & 15u) * 2u;
        const uint64_t A = ((uint64_t)W.w0 << 32) | W.w1, C = (uint64_t)W.w2 << 32;
        P = (sh ? (A << sh) | (C >> (64u - sh)) : A) >> (64u - 2u * len);
    }
    V = fwd ? P : reverse_symbols( P, len );
    if (comp) V ^= (len >= 32u) ? ~0ull : ((1ull << (2u * len)) - 1ull);
    return clean;
}

template <int BITS>
__device__ __forceinline__ bool seed_bits(const void* __restrict__ symbols, const uint32_t begin, const uint32_t len, const bool fwd, const bool comp,
                                          uint64_t& V)
{
    if (BITS == 8)
    {
        uint64_t P = 0; bool clean = true;
        const uint8_t* b = (const uint8_t*)symbols + begin;
        for (uint32_t j = 0; j < len; ++j) { const uint32_t c = b[j]; clean = clean && c < 4u; P = (P << 2) | (c & 3u); }
        V = fwd ? P : reverse_symbols( P, len );
        if (comp) V ^= (len >= 32u) ? ~0ull : ((1ull << (2u * len)) - 1ull);
        return clean;
    }
    const SeedWords W = load_seed_words<(BITS == 8 ? 4 : BITS)>( symbols, begin, len );
    return seed_bits_from_words<(BITS == 8 ? 4 : BITS)>( W, begin, len, fwd, comp, V );
}

// the r symbols that follow the first `done` ones in scan order, first one most significant (the layout of a left context
// shifted down to r symbols)
__device__ __forceinline__ uint32_t seed_rest(const uint64_t V, const uint32_t len, const uint32_t done)
{
    const uint32_t r = len - done;
    return (uint32_t)(V & ((1ull << (2u * r)) - 1ull));                    // r <= DTAB_CTX < 16 where this is used
}

// does the occurrence at text position p with left context `ctx` continue with the r symbols `rest`?
__device__ __forceinline__ bool context_matches(const uint32_t p, const uint32_t ctx, const uint32_t rest, const uint32_t r)
{
    return p >= r && (ctx >> (2u * (DTAB_CTX - r))) == rest;               // r >= 1: shift < 32
}

} // namespace nvbio_amd
