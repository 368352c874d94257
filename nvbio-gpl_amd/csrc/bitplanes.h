// bitplanes.h -- bit planes of packed reads and text windows for the ungapped shortcuts (gotoh_banded.hip, gotoh_full.hip,
// gotoh_traceback.hip): bit i of (lo, hi, n) = low bit, high bit, "is N" of pattern row i; bit k of (tlo, thi) = the two
// bits of window symbol k.  Built a packed word at a time: bit-reverse the big-endian word so that symbol 0 sits lowest,
// squeeze every BITS-th bit together (3-4 shift/mask steps), drop the 8 / 16 plane bits at their storage offset, then
// shift the planes so that bit 0 is row 0 (reversed reads: mirror first).  Every word is loaded unconditionally (index
// clamped into the string, so always valid): no branches between the loads, they all issue back to back; plane bits
// outside the pattern / window are garbage and must be masked by the caller.
#pragma once
#include "common.h"

namespace nvbio_amd {

constexpr uint32_t PLANE_MAX_READ = 161u;                        // 3 x 64 bits hold 161 symbols at any storage offset

template <int RBITS> struct ReadWords { static constexpr int N = (161 + 15) * RBITS / 32 + 2; uint32_t w[N]; };
struct TextWords13 { uint32_t w[13]; };

template <int RBITS>
__device__ __forceinline__ void load_read_words(const void* reads, const uint32_t first, const uint32_t M, ReadWords<RBITS>& out)
{
    constexpr uint32_t RPW = 32u / RBITS;
    const uint32_t* __restrict__ rwords = (const uint32_t*)reads;
    const uint32_t rw0 = (first & ~(RPW - 1u)) / RPW, rw_last = (first + M - 1u) / RPW;
    #pragma unroll
    for (int j = 0; j < ReadWords<RBITS>::N; ++j) { const uint32_t widx = rw0 + (uint32_t)j; out.w[j] = rwords[widx < rw_last ? widx : rw_last]; }
}

// T = number of window symbols that will be looked at (<= 192 + 15 - offset)
__device__ __forceinline__ void load_text_words13(const void* text, const uint32_t tb, const uint32_t T, TextWords13& out)
{
    const uint32_t* __restrict__ twords = (const uint32_t*)text;
    const uint32_t tw0 = (tb & ~15u) >> 4, tw_last = (tb + T - 1u) >> 4;
    #pragma unroll
    for (int j = 0; j < 13; ++j) { const uint32_t widx = tw0 + (uint32_t)j; out.w[j] = twords[widx < tw_last ? widx : tw_last]; }
}

__device__ __forceinline__ void shr192(uint64_t (&v)[3], const uint32_t sh)
{
    const uint32_t ws = sh >> 6, bs = sh & 63u;
    uint64_t x[5] = { v[0], v[1], v[2], 0ull, 0ull };
    uint64_t y[4];
    #pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = ws == 0u ? x[k] : (ws == 1u ? x[k + 1 < 5 ? k + 1 : 4] : (ws == 2u ? (k + 2 < 5 ? x[k + 2] : 0ull) : 0ull));
    #pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = bs ? ((y[k] >> bs) | (y[k + 1] << (64u - bs))) : y[k];
}
__device__ __forceinline__ void mirror192(uint64_t (&v)[3])
{
    const uint64_t a = __brevll( v[2] ), c = __brevll( v[0] );
    v[1] = __brevll( v[1] ); v[0] = a; v[2] = c;
}

// every 2nd bit of w (positions 0,2,..,30) squeezed into 16 bits
__device__ __forceinline__ uint32_t squeeze2(uint32_t x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u; x = (x | (x >> 2)) & 0x0F0F0F0Fu; x = (x | (x >> 4)) & 0x00FF00FFu; x = (x | (x >> 8)) & 0xFFFFu;
    return x;
}
// every 4th bit of w (positions 0,4,..,28) squeezed into 8 bits
__device__ __forceinline__ uint32_t squeeze4(uint32_t x)
{
    x &= 0x11111111u;
    x = (x | (x >> 3)) & 0x03030303u; x = (x | (x >> 6)) & 0x000F000Fu; x = (x | (x >> 12)) & 0xFFu;
    return x;
}

template <int RBITS>
__device__ __forceinline__ void read_planes192(const ReadWords<RBITS>& rw, const uint32_t first, const uint32_t M, const bool rev, const bool comp,
                                               uint64_t (&rlo)[3], uint64_t (&rhi)[3], uint64_t (&rn)[3])
{
    constexpr uint32_t RPW = 32u / RBITS;
    const uint32_t roff = first & (RPW - 1u);
    #pragma unroll
    for (int k = 0; k < 3; ++k) { rlo[k] = 0; rhi[k] = 0; rn[k] = 0; }
    #pragma unroll
    for (int j = 0; j < ReadWords<RBITS>::N; ++j)
    {
        // (the array covers 161 symbols at any offset; a wave of 150-symbol reads skips the words past their ends)
        if (!__any( (uint32_t)j * RPW < roff + M )) continue;
        const uint32_t w = __brev( rw.w[j] );
        uint32_t lo, hi, nn;
        if (RBITS == 4)
        {
            // after the reversal symbol k holds value bits (b3,b2,b1,b0) at bit positions (4k, 4k+1, 4k+2, 4k+3)
            lo = squeeze4( w >> 3 ); hi = squeeze4( w >> 2 ); nn = squeeze4( (w >> 1) | w );
        }
        else { lo = squeeze2( w >> 1 ); hi = squeeze2( w ); nn = 0; }
        const int bitpos = j * (int)RPW;                         // compile-time: no dynamic plane index
        if (bitpos < 192)
        {
            rlo[bitpos >> 6] |= (uint64_t)lo << (bitpos & 63);
            rhi[bitpos >> 6] |= (uint64_t)hi << (bitpos & 63);
            rn [bitpos >> 6] |= (uint64_t)nn << (bitpos & 63);
        }
    }
    // bit p of the planes = storage symbol (first - roff) + p.  Forward: row i = p - roff.  Reversed: row i = roff + M-1 - p.
    if (rev)
    {
        mirror192( rlo ); mirror192( rhi ); mirror192( rn );    // bit r now = storage symbol base + 191 - r
        const uint32_t sh = 192u - roff - M;                     // row i = bit i + sh
        shr192( rlo, sh ); shr192( rhi, sh ); shr192( rn, sh );
    }
    else { shr192( rlo, roff ); shr192( rhi, roff ); shr192( rn, roff ); }
    if (comp)                                                    // 3 - q for q < 4: flip both bits of the non-N rows
    {
        #pragma unroll
        for (int k = 0; k < 3; ++k) { rlo[k] ^= ~rn[k]; rhi[k] ^= ~rn[k]; }
    }
}

// 208 plane bits (13 packed words), shifted so that bit 0 = window symbol 0; the top word holds the remainder
__device__ __forceinline__ void text_planes208(const TextWords13& tw, const uint32_t tb, uint64_t (&tlo)[4], uint64_t (&thi)[4])
{
    #pragma unroll
    for (int k = 0; k < 4; ++k) { tlo[k] = 0; thi[k] = 0; }
    #pragma unroll
    for (int j = 0; j < 13; ++j)
    {
        const uint32_t w = __brev( tw.w[j] );
        const uint32_t lo = squeeze2( w >> 1 ), hi = squeeze2( w );
        const int bitpos = j * 16;
        tlo[bitpos >> 6] |= (uint64_t)lo << (bitpos & 63);
        thi[bitpos >> 6] |= (uint64_t)hi << (bitpos & 63);
    }
    const uint32_t toff = tb & 15u;
    if (toff)
    {
        #pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            tlo[k] = (tlo[k] >> toff) | (tlo[k + 1] << (64u - toff));
            thi[k] = (thi[k] >> toff) | (thi[k + 1] << (64u - toff));
        }
        tlo[3] >>= toff; thi[3] >>= toff;
    }
}

} // namespace nvbio_amd
