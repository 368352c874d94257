// fm_index.hip -- FM-index seed pass for gfx950: match / rank / locate / filter, behind the C ABI.
//
// Reference behaviour reproduced (file:line relative to the reference tree):
//   match / match_reverse                 nvbio/fmindex/fmindex_inl.h:181-278
//   nvBowtie match_range + exact seeds    nvBowtie/bowtie2/cuda/mapping_inl.h:73-86,193-282
//   rank / rank4                          nvbio/fmindex/fmindex_inl.h:27-173
//   locate / locate_ssa_iterator / lookup nvbio/fmindex/fmindex_inl.h:360-460
//   FMIndexFilter<device_tag>             nvbio/fmindex/filter_inl.h:261-393
//
// MI355X design: the pass is a dependent chain of random 32-byte gathers, bound by HBM
// request rate, not by arithmetic.  One lane owns one query; kernels keep <= 64 VGPRs so that
// 8 waves per SIMD (2048 lanes per CU) are in flight to cover the ~2 us loaded HBM latency, the
// two ends of a range share one record when they fall in the same 64-symbol block, and the first
// k steps (where the range still spans the whole index and both ends miss every cache) are
// replaced by one 8-byte lookup in a 4^k-entry table of SA ranges built once per index
// (k = 12 -> 128 MiB: served from the 256 MiB Infinity Cache / HBM; the table is exact, it stores
// the reference's own early-exit values, so ranges are identical with and without it).
#include "fm_device.h"
#include "fm_seed_device.h"
#include "seed_hits_device.h"
#include <hipcub/hipcub.hpp>
#include <new>
#include <stdlib.h>

namespace nvbio_amd {

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct FMIndexImpl
{
    int                  device;
    nvbio_fm_index_view  view;        // host copy (device pointers inside)
    uint2*               ktab;        // owned
    uint32_t             kmer;
    uint2*               dtab;        // owned, optional: the table of the direct seed pass (positions for k-mers with few occurrences)
    uint32_t             dkmer;
    uint2*               side;        // owned, optional: its groups for k-mers with 2..7 occurrences
    uint32_t             dmark, dctx;
    uint2*               ctab;        // owned, optional (NVBIO_FM_TABLE_CANONICAL, instead of dtab): the two-strand table (fm_canon_inl.h)
    uint2*               cside;       // owned: its groups
    uint32_t             ckmer;
    uint32_t             cwide;       // its entries are 16 bytes (two rows)
    uint32_t             table_flags; // NVBIO_FM_TABLE_* of the build
    bool                 owns_arrays; // bwt_occ / ssa allocated by nvbio_fm_index_build
    uint64_t             owned_bytes;
    uint32_t*            isa;         // owned, optional (verify mode)
    uint32_t*            text;        // owned, optional (verify mode)

    DevIndex dev() const
    {
        DevIndex d;
        d.length = view.length; d.primary = view.primary;
        d.L2_0 = view.L2[0]; d.L2_1 = view.L2[1]; d.L2_2 = view.L2[2]; d.L2_3 = view.L2[3]; d.L2_4 = view.L2[4];
        d.rec  = (const uint4*)view.bwt_occ_dev;
        d.ssa  = view.ssa_dev;
        d.sa_log = 0; while ((1u << d.sa_log) < (view.sa_int ? view.sa_int : 16u)) ++d.sa_log;
        d.ktab = ktab; d.kmer = kmer; d.dtab = dtab; d.dkmer = dkmer;
        d.side = side; d.dmark = dmark; d.dctx = dctx;
        d.isa = isa; d.text = text;
        d.ctab = ctab; d.cside = cside; d.ckmer = ckmer;
        return d;
    }
};

struct StringSetDev
{
    const void*     symbols;
    const uint32_t* offsets;
    uint32_t        ranges;      // offsets are [n+1] begin/end pairs
    uint32_t        fixed_len;
    uint32_t        stride;
    uint32_t        n;
    uint32_t        spr;         // seeds per string (0: plain string set)
    uint32_t        interval;    // seed interval
    const uint32_t* intervals;   // ragged seed sets: the seed interval of every string (offsets then hold n_strings + 1 entries)
};

// false: query i is a seed id of a ragged seed set beyond its string's last seed -- it matches nothing
__device__ __forceinline__ bool string_bounds(const StringSetDev& q, const uint32_t i, uint32_t& begin, uint32_t& len)
{
    if (q.spr)
    {
        const uint32_t r = i / q.spr, j = i - r * q.spr;
        const uint32_t base = q.offsets ? q.offsets[r] : r * q.stride;
        begin = base + j * (q.intervals ? q.intervals[r] : q.interval);
        len   = q.fixed_len;
        if (q.intervals && begin + len > q.offsets[r + 1]) return false;
    }
    else if (q.offsets)
    {
        begin = q.offsets[i];
        len   = q.ranges ? q.offsets[i + 1] - begin : q.fixed_len;
    }
    else { begin = i * q.stride; len = q.fixed_len; }
    return true;
}

// ---------------------------------------------------------------------------------------------
// match
// ---------------------------------------------------------------------------------------------
// TABLE: start from the k-mer table.  A template parameter (not just a flag) so that launches through the
// reference's plain algorithm (NVBIO_FM_NO_KMER_TABLE) are a different kernel in profiler output.
// DIRECT (nvbio_fm_match_direct): a search whose range has collapsed to ONE row before the pattern is
// exhausted is finished on the text: the only place the match can continue is to the left of SA[row], so the
// rest of the pattern is compared with the text there (2 dependent gathers -- SA, text -- instead of one per
// remaining symbol plus the SA lookup of a later locate).  Such a query reports its single hit as the TEXT
// POSITION (ranges[i] = (pos, pos), direct[i] = 1) -- the position locate() would return for the final row;
// a mismatch reports the empty range (1,0).  Needs the full SA and the text (handles built with sa_int = 1).
// The direct table (fm_seed_device.h) shortens this further: its entries hold the position and the text to the
// left of a k-mer's occurrence(s), so that most searches end with the table gather itself.
#ifndef NVB_NT_TABLE
#define NVB_NT_TABLE 1
#endif
constexpr bool NT_TABLE = NVB_NT_TABLE != 0;

struct SearchState
{
    uint32_t x, y, s;          // SA range and number of symbols consumed
    bool     have_pos;         // the search is down to one occurrence whose text position is known (tpos)
    uint32_t tpos;
    bool     have_ctx;         // ... and so are the DevIndex::dctx symbols to its left (ctx)
    uint32_t ctx;
    bool     is_pos;           // result: (x, y) = (pos, pos) is a text position
    uint32_t sectors;          // accounting (seed pass, COUNT mode): distinct 64-byte sectors this search gathered from the index
};

__device__ __forceinline__ uint2 load_table_entry(const uint2* __restrict__ tab, const uint64_t key, const bool nt)
{
    // a table entry has no reuse: a non-temporal load keeps it from displacing the text and bwt_occ lines the caches can hold
    if (nt && NT_TABLE)
    {
        const unsigned long long v = __builtin_nontemporal_load( (const unsigned long long*)tab + key );
        return make_uint2( (uint32_t)v, (uint32_t)(v >> 32) );
    }
    return tab[key];
}

// decode the direct-table entry `e` of the first st.s symbols (see fm_seed_device.h for the format)
template <class Sym>
__device__ __forceinline__ void direct_entry(const DevIndex& f, const uint2 e, Sym& sym, const uint32_t len, SearchState& st)
{
    if (e.y < f.dmark) { st.x = e.x; st.y = e.y; return; }                 // a plain SA range
    if (e.x < f.dmark)                                                     // one occurrence
    {
        st.have_pos = true; st.tpos = e.x; st.x = st.y = 0u;
        st.have_ctx = f.dctx != 0u; st.ctx = e.y - f.dmark;
        return;
    }
    // 2..7 occurrences: header (x, y) + one (position, context) slot per row, in one or two sectors
    const uint32_t m  = e.y - f.dmark;
    const uint4*   g4 = (const uint4*)(f.side + 4ull * (e.x - f.dmark));
    st.sectors += 1u;                                                      // a 32-byte group lies in one sector, a 64-byte group is one
    const uint4 q0 = g4[0], q1 = g4[1];
    uint4 q2 = make_uint4( 0, 0, 0, 0 ), q3 = q2;
    if (m > 3u) { q2 = g4[2]; q3 = g4[3]; }
    const uint32_t r = len - st.s;
    st.x = q0.x; st.y = q0.y;                                              // the k-mer's SA range
    if (r > f.dctx) return;                                                // longer than the stored context: rank steps
    uint32_t rest = 0; bool clean = true;
    for (uint32_t t = 0; t < r; ++t) { const uint32_t c = sym( st.s + t ); clean = clean && c < 4u; rest = (rest << 2) | (c & 3u); }
    if (!clean) { st.x = 1u; st.y = 0u; return; }                          // an N: no match (fmindex_inl.h:227-228)
    uint32_t hits = 0, pos = 0;
#define NVB_SIDE_ROW(j, P, C) if ((j) <= m && context_matches( (P), (C) - f.dmark, rest, r )) { ++hits; pos = (P); }
    NVB_SIDE_ROW( 1u, q0.z, q0.w ) NVB_SIDE_ROW( 2u, q1.x, q1.y ) NVB_SIDE_ROW( 3u, q1.z, q1.w )
    NVB_SIDE_ROW( 4u, q2.x, q2.y ) NVB_SIDE_ROW( 5u, q2.z, q2.w ) NVB_SIDE_ROW( 6u, q3.x, q3.y ) NVB_SIDE_ROW( 7u, q3.z, q3.w )
#undef NVB_SIDE_ROW
    if (hits == 0u)      { st.x = 1u; st.y = 0u; }                         // no occurrence continues with the rest of the pattern
    else if (hits == 1u) { st.x = st.y = pos - r; st.is_pos = true; st.s = len; }
    // several occurrences continue (a repeat longer than the pattern): their final SA range needs the rank steps
}

// the search from state st on: rank steps, and with DIRECT the finish of a one-row range on the text
template <bool COUNT, bool DIRECT, class Sym>
__device__ __forceinline__ void search_tail(const DevIndex& f, Sym& sym, const uint32_t len, const bool verify, SearchState& st, uint32_t& nblk)
{
    uint32_t x = st.x, y = st.y, s = st.s;
    if (st.is_pos) return;
    for (; s < len && x <= y && !st.have_pos; ++s)
    {
        // DIRECT: a lane whose range has collapsed leaves the loop and waits for its neighbours, so that the
        // whole wave runs the two gathers of the tail below once, together, instead of once per collapse time
        if (DIRECT && x == y) break;
        if (verify && x == y && len - s >= 2u)
        {
            // The range is ONE row: the match can only continue along the text to the left of
            // SA[x].  Compare the rest of the pattern with the text there and jump to the row of
            // the position reached (ISA): 3 gathers instead of one per remaining symbol.  On a
            // mismatch the loop resumes at the offending symbol from the exact row the reference
            // would be at, so the (empty) range it returns is the reference's.
            const uint32_t sv = f.ssa[x];
            const uint32_t p  = (sv == 0xFFFFFFFFu) ? f.length : sv;      // row 0 is the empty suffix
            const uint32_t r  = len - s;
            // the row reached if everything matches is requested together with the text words
            // (speculatively: both depend only on p), so the tail costs two dependent rounds
            const uint32_t full = (p >= r) ? f.isa[p - r] : 0u;
            SymbolReader<2> tr( f.text );
            uint32_t t = 0;
            while (t < r && t < p)
            {
                const uint32_t c = sym( s + t );
                if (c > 3u || c != tr.get( p - 1u - t )) break;
                ++t;
            }
            if (t == r)  { x = y = full; s = len; }
            else if (t)  { x = y = f.isa[p - t]; s += t; }
            if (s >= len) break;
        }
        const uint32_t c = sym( s );
        if (c > 3u) { x = 1u; y = 0u; break; }              // an N: no match (fmindex_inl.h:227-228)
        search_step<COUNT>( f, x, y, c, nblk, &st.sectors );
    }
    if (DIRECT && ((s < len && x == y) || st.have_pos))
    {
        uint32_t p = st.tpos;
        if (!st.have_pos)
        {
            const uint32_t sv = NT_TABLE ? __builtin_nontemporal_load( f.ssa + x ) : f.ssa[x];
            p = (sv == 0xFFFFFFFFu) ? f.length : sv;                  // row 0 is the empty suffix
            st.sectors += 1u;
        }
        const uint32_t r  = len - s;
        bool ok = (p >= r);
        if (ok && r > 0u)
        {
            const uint32_t w0 = (p - r) >> 4;
            if (st.have_pos && st.have_ctx && r <= f.dctx)
            {
                // the table entry carries the text to the left of the occurrence: no further gather
                uint32_t rest = 0;
                for (uint32_t t = 0; t < r; ++t) { const uint32_t c = sym( s + t ); ok = ok && c < 4u; rest = (rest << 2) | (c & 3u); }
                ok = ok && context_matches( p, st.ctx, rest, r );
            }
            else if (r <= 16u && w0 + 2u <= ((f.length + 15u) >> 4))  // both words inside the text copy
            {
                // the r <= 16 symbols text[p-r, p) lie in at most two consecutive words: ONE 8-byte gather (4-byte aligned)
                // instead of a second, dependent 4-byte one whenever they straddle a word boundary
                struct __attribute__((packed, aligned(4))) W2 { uint32_t a, b; };
                const W2 w = *(const W2*)(f.text + w0);
                st.sectors += ((w0 & 15u) == 15u) ? 2u : 1u;
                for (uint32_t t = 0; t < r && ok; ++t)
                {
                    const uint32_t i  = p - 1u - t;
                    const uint32_t tw = ((i >> 4) == w0) ? w.a : w.b;
                    const uint32_t c  = sym( s + t );
                    ok = (c < 4u) && (c == ((tw >> (30u - 2u * (i & 15u))) & 3u));
                }
            }
            else
            {
                SymbolReader<2> tr( f.text );
                st.sectors += ((p - 1u) >> 8) - ((p - r) >> 8) + 1u;        // 256 symbols per sector
                for (uint32_t t = 0; t < r && ok; ++t)
                {
                    const uint32_t c = sym( s + t );
                    ok = (c < 4u) && (c == tr.get( p - 1u - t ));
                }
            }
        }
        if (ok) { x = y = p - r; st.is_pos = true; }
        else    { x = 1u; y = 0u; }
    }
    st.x = x; st.y = y; st.s = s;
}

template <int BITS, bool COUNT, bool TABLE, bool DIRECT>
__device__ __forceinline__ void match_one(const DevIndex& f, const StringSetDev& q, const uint32_t flags, const bool tab, const bool verify,
                                          const uint32_t i, uint32_t& x_out, uint32_t& y_out, uint32_t& nblk_out, bool& is_pos_out,
                                          uint32_t* sectors_out = nullptr)
{
    const bool fwd  = (flags & NVBIO_FM_SCAN_FORWARD) != 0;
    const bool comp = (flags & NVBIO_FM_COMPLEMENT) != 0;
    uint32_t begin, len;
    if (!string_bounds( q, i, begin, len ))
    {
        x_out = 1u; y_out = 0u; nblk_out = 0u; is_pos_out = false;
        if (sectors_out) *sectors_out = 0u;
        return;
    }
    SymbolReader<BITS> rd( q.symbols );

    // symbol s in scan order
    auto sym = [&](const uint32_t s) -> uint32_t {
        const uint32_t c = rd.get( fwd ? begin + s : begin + len - 1u - s );
        return (comp && c < 4u) ? 3u - c : c;
    };

    SearchState st;
    st.x = 0; st.y = f.length; st.s = 0; st.have_pos = false; st.tpos = 0; st.have_ctx = false; st.ctx = 0; st.is_pos = false; st.sectors = 0;
    uint32_t nblk = 0;

    // DIRECT: the handle's second table (one symbol longer) resolves a k-mer with few occurrences to text positions
    const bool     use_d = DIRECT && f.dtab != nullptr && len >= f.dkmer;
    const uint32_t tk    = use_d ? f.dkmer : f.kmer;
    if (tab && len >= tk)
    {
        uint64_t key = 0; bool ok = true;                    // 34 bits at k = 17
        for (uint32_t t = 0; t < tk; ++t)
        {
            const uint32_t c = sym( t );
            ok = ok && (c < 4u);
            key = (key << 2) | (c & 3u);
        }
        if (ok)
        {
            const uint2 e = load_table_entry( use_d ? f.dtab : f.ktab, key, use_d );
            st.s = tk; st.sectors += 1u;
            if (use_d) direct_entry( f, e, sym, len, st );
            else       { st.x = e.x; st.y = e.y; }
        }
    }
    search_tail<COUNT,DIRECT>( f, sym, len, verify, st, nblk );
    x_out = st.x; y_out = st.y; nblk_out = nblk; is_pos_out = st.is_pos;
    if (sectors_out) *sectors_out = st.sectors;
}

template <int BITS, bool COUNT, bool TABLE, bool DIRECT = false>
__global__ void __launch_bounds__(256)
fm_match_kernel(const DevIndex f, const StringSetDev q, const uint32_t flags, uint2* __restrict__ ranges, uint32_t* __restrict__ blocks,
                uint8_t* __restrict__ direct = nullptr)
{
    const bool tab    = TABLE && (f.ktab != nullptr) && !(flags & NVBIO_FM_NO_KMER_TABLE) && !COUNT;
    const bool verify = (f.isa != nullptr) && (f.sa_log == 0) && !(flags & NVBIO_FM_NO_VERIFY) && !COUNT;

    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < q.n; i += gridDim.x * blockDim.x)
    {
        uint32_t x, y, nblk; bool is_pos;
        match_one<BITS,COUNT,TABLE,DIRECT>( f, q, flags, tab, verify, i, x, y, nblk, is_pos );
        ranges[i] = make_uint2( x, y );
        if (COUNT) blocks[i] = nblk;
        if (DIRECT) direct[i] = is_pos ? 1u : 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// Seed pass straight to candidate diagonals (nvbio_fm_match_seed_diagonals): the direct search over the seeds of a read set,
// followed -- in the same kernel -- by what FMIndexFilter::rank's scan, FMIndexFilter::locate, hit_to_diagonal
// (examples/fmmap/fmmap.cu:92-117) and the adjacent-duplicate removal do for a seed with ONE hit:
//   * a seed that ended on one SA row is resolved to its text position (known from the direct table for nearly all of them;
//     one SA gather otherwise -- the handle holds the full SA);
//   * its diagonal key (read << 34 | strand << 33 | position - offset of the seed in the read + 1024) is formed;
//   * keys equal to their predecessor are dropped (consecutive seeds of a read that agree on the diagonal).
// One WAVE owns a tile of whole reads (64 / seeds_per_read of them, one seed per lane): the compaction is a ballot and a
// popcount, the predecessor's key comes through ds_bpermute -- no LDS, no barrier -- and the surviving keys go to the tile's
// own 64 slots of a scratch array with their count beside them.  An exclusive scan of the 1.4 M tile counts and one small
// copy kernel then make the list dense, IN SEED ORDER (a deterministic output; a first version appended each block's keys
// with one atomic add on a global counter: 350 k returning atomics on one address serialise at about 11 ns each, most of
// that kernel's 4.7 ms).
// Seeds that end on SEVERAL rows (repeats) are appended to a residual list (seed id, range) for the ordinary
// scan + locate path; that list's order is arbitrary.  counts[0] = keys written, counts[1] = residual seeds.
// ---------------------------------------------------------------------------------------------
struct SeedTiles
{
    uint32_t reads;            // strings in the set
    uint32_t rpt;              // reads per tile (64 / seeds per read; 1 when a read has more than 64 seeds)
    uint32_t n_tiles;
};

// One round of a tile's output: every lane's search result (single: x = text position; several rows: the range (x, y)) to
//   * the tile's keys, in seed order, a key equal to its predecessor dropped -- ballot + popcount, the predecessor's key
//     through ds_bpermute: no LDS, no barrier;
//   * the residual list, one atomic per wave that has any seed on several rows.
// p_fw: the seed's offset in its read (j x the read's seed interval); read_len: that read's length
__device__ __forceinline__ void emit_seed_results(const uint32_t p_fw, const uint32_t len, const uint32_t read_len, const uint32_t strand, const uint32_t lane,
                                                  const bool valid, const bool single, const uint32_t x, const uint32_t y, const uint32_t rid,
                                                  const uint32_t i, uint64_t* __restrict__ tile_slots, uint32_t& n_out, uint64_t& last_key,
                                                  uint2* __restrict__ res_ranges, uint32_t* __restrict__ res_ids, unsigned int* __restrict__ counts)
{
    const bool multi = valid && !single && x < y;
    uint64_t key = 0;
    if (single)
    {
        uint32_t p = p_fw;
        if (strand) p = read_len - p - len;
        key = ((uint64_t)rid << 34) | ((uint64_t)(strand & 1u) << 33) | ((uint64_t)x + 1024u - p);
    }
    const uint64_t m1 = __ballot( single );
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t prev_mask = m1 & below;
    const int      prev_lane = prev_mask ? 63 - __clzll( (long long)prev_mask ) : 0;
    const uint32_t pk_lo = (uint32_t)__shfl( (int)(uint32_t)key, prev_lane ), pk_hi = (uint32_t)__shfl( (int)(uint32_t)(key >> 32), prev_lane );
    const uint64_t prev_key = prev_mask ? (((uint64_t)pk_hi << 32) | pk_lo) : last_key;
    const bool     keep = single && key != prev_key;
    const uint64_t m3 = __ballot( keep );
    if (keep) tile_slots[n_out + (uint32_t)__popcll( m3 & below )] = key;
    n_out += (uint32_t)__popcll( m3 );
    if (m1)                                                  // the last single key of this round, for the next round's first
    {
        const int hl = 63 - __clzll( (long long)m1 );
        const uint32_t lo = (uint32_t)__shfl( (int)(uint32_t)key, hl ), hi = (uint32_t)__shfl( (int)(uint32_t)(key >> 32), hl );
        last_key = ((uint64_t)hi << 32) | lo;
    }
    const uint64_t m2 = __ballot( multi );
    if (m2)
    {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd( &counts[1], (unsigned int)__popcll( m2 ) );
        base = (uint32_t)__shfl( (int)base, 0 );
        if (multi)
        {
            const uint32_t r = base + (uint32_t)__popcll( m2 & below );
            res_ranges[r] = make_uint2( x, y ); res_ids[r] = i;
        }
    }
}

// COUNT: an accounting launch (untimed; bench.py's roofline): additionally sums, into sectors_out[0], the distinct 64-byte
// sectors every search gathers from the index (table entry, group, bwt_occ records, SA word, text words)
template <int BITS, bool COUNT = false>
__global__ void __launch_bounds__(256)
fm_seed_tiles_kernel(const DevIndex f, const StringSetDev q, const SeedTiles tl, const uint32_t flags, const uint32_t read_len, const uint32_t strand,
                     uint64_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_counts, uint2* __restrict__ res_ranges,
                     uint32_t* __restrict__ res_ids, unsigned int* __restrict__ counts, unsigned long long* __restrict__ sectors_out = nullptr)
{
    const bool fwd  = (flags & NVBIO_FM_SCAN_FORWARD) != 0;
    const bool comp = (flags & NVBIO_FM_COMPLEMENT) != 0;
    const bool tab  = (f.ktab != nullptr) && !(flags & NVBIO_FM_NO_KMER_TABLE);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t len = q.fixed_len;
    // seeds of a read beyond the 64 lanes of a wave take further rounds of the same tile (rpt == 1 then)
    const uint32_t rounds = (q.spr + 63u) / 64u;

    for (uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6); tile < tl.n_tiles; tile += n_waves)
    {
        uint32_t n_out = 0;                                           // keys of this tile written so far (wave-uniform)
        uint64_t last_key = ~0ull;                                    // the tile's last surviving key (wave-uniform)
        for (uint32_t round = 0; round < rounds; ++round)
        {
            const uint32_t lr  = lane / q.spr;                        // read of the lane within the tile (rounds == 1)
            const uint32_t rid = tile * tl.rpt + (rounds == 1u ? lr : 0u);
            const uint32_t j   = rounds == 1u ? lane - lr * q.spr : round * 64u + lane;
            const bool valid   = (rounds == 1u ? lr < tl.rpt : true) && j < q.spr && rid < tl.reads;
            const uint32_t i   = rid * q.spr + j;                     // seed id

            uint32_t x = 1u, y = 0u, sectors = 0u; bool single = false;
            if (valid)
            {
                const uint32_t begin = (q.offsets ? q.offsets[rid] : rid * q.stride) + j * q.interval;
                uint64_t V = 0;
                const bool fast = len <= 32u && f.dtab != nullptr && len >= f.dkmer && tab;
                if (fast)
                {
                    if (seed_bits<BITS>( q.symbols, begin, len, fwd, comp, V ))       // a seed with an N matches nothing
                    {
                        auto sym = [&](const uint32_t s) -> uint32_t { return (uint32_t)(V >> (2u * (len - 1u - s))) & 3u; };
                        SearchState st;
                        st.x = 0; st.y = f.length; st.s = f.dkmer; st.have_pos = false; st.tpos = 0; st.have_ctx = false; st.ctx = 0; st.is_pos = false; st.sectors = 1u;
                        const uint2 e = load_table_entry( f.dtab, V >> (2u * (len - f.dkmer)), true );
                        direct_entry( f, e, sym, len, st );
                        uint32_t nblk = 0;
                        search_tail<false,true>( f, sym, len, false, st, nblk );
                        x = st.x; y = st.y; single = st.is_pos; sectors = st.sectors;
                    }
                }
                else
                {
                    uint32_t nblk;
                    match_one<BITS,false,true,true>( f, q, flags, tab, false, i, x, y, nblk, single, &sectors );
                }
                if (!single && x == y)                               // one row left when the pattern ran out: its position
                {
                    const uint32_t sv = f.ssa[x];
                    x = (sv == 0xFFFFFFFFu) ? f.length : sv; single = true; ++sectors;
                }
            }
            if (COUNT)
            {
                uint32_t tot = sectors;
                #pragma unroll
                for (int d = 32; d > 0; d >>= 1) tot += (uint32_t)__shfl_xor( (int)tot, d );
                if (lane == 0 && tot) atomicAdd( sectors_out, (unsigned long long)tot );
            }
            emit_seed_results( j * q.interval, len, read_len, strand, lane, valid, single, x, y, rid, i, tile_keys + (uint64_t)tile * 64u * rounds, n_out, last_key,
                               res_ranges, res_ids, counts );
        }
        if (lane == 0) tile_counts[tile] = n_out;
    }
}

// The same pass as a three-stage software pipeline over the tiles a wave owns (seeds of up to 32 symbols whose remainder after
// the table's k fits the stored context; at most 64 seeds per read): while tile t is resolved from its table entries -- the
// group gather of the few seeds whose k-mer has 2..7 occurrences is issued first -- the table entries of tile t+1 are
// requested from the seed bits decoded out of words loaded one iteration earlier, and the packed words of tile t+2 are
// loaded.  The three dependent round trips of a seed (read words -> table entry -> group) thus overlap across tiles: one
// memory latency per iteration instead of three.
template <int BITS>
__global__ void __launch_bounds__(256)
fm_seed_pipe_kernel(const DevIndex f, const StringSetDev q, const SeedTiles tl, const uint32_t flags, const uint32_t read_len, const uint32_t strand,
                    uint64_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_counts, uint2* __restrict__ res_ranges,
                    uint32_t* __restrict__ res_ids, unsigned int* __restrict__ counts)
{
    const bool fwd  = (flags & NVBIO_FM_SCAN_FORWARD) != 0;
    const bool comp = (flags & NVBIO_FM_COMPLEMENT) != 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t len = q.fixed_len, r = len - f.dkmer;
    const uint32_t lr = lane / q.spr, j = lane - lr * q.spr;          // the lane's read within a tile and its seed: the same for every tile
    const bool lane_ok = lr < tl.rpt;

    auto seed_begin = [&](const uint32_t rid) -> uint32_t { return (q.offsets ? q.offsets[rid] : rid * q.stride) + j * q.interval; };

    uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= tl.n_tiles) return;
    // prologue: words of this tile and of the next one, table entry of this tile
    SeedWords W = { 0, 0, 0, 0, 0 }; uint32_t w_begin = 0; bool w_valid = false;     // stage 1 -> 2: tile + n_waves after the prologue
    uint64_t V = 0; uint2 e = make_uint2( 1u, 0u ); bool e_valid = false;            // stage 2 -> 3: the current tile
    {
        const uint32_t rid = tile * tl.rpt + lr;
        if (lane_ok && rid < tl.reads)
        {
            const uint32_t b0 = seed_begin( rid );
            const SeedWords W0 = load_seed_words<BITS>( q.symbols, b0, len );
            e_valid = seed_bits_from_words<BITS>( W0, b0, len, fwd, comp, V );
            if (e_valid) e = load_table_entry( f.dtab, V >> (2u * r), true );
        }
        const uint32_t t1 = tile + n_waves, rid1 = t1 * tl.rpt + lr;
        if (t1 < tl.n_tiles && lane_ok && rid1 < tl.reads) { w_begin = seed_begin( rid1 ); W = load_seed_words<BITS>( q.symbols, w_begin, len ); w_valid = true; }
    }
    for (; tile < tl.n_tiles; tile += n_waves)
    {
        const uint32_t rid = tile * tl.rpt + lr;
        const bool valid = lane_ok && rid < tl.reads;
        // ---- stage 3a: what kind of entry; request the group of a 2..7-occurrence k-mer ----
        const uint32_t rest = (uint32_t)(V & ((1ull << (2u * r)) - 1ull));
        const bool is_rng   = e_valid && e.y < f.dmark;
        const bool is_one   = e_valid && e.y >= f.dmark && e.x < f.dmark;
        const bool is_group = e_valid && e.y >= f.dmark && e.x >= f.dmark;
        uint4 q0 = make_uint4( 0, 0, 0, 0 ), q1 = q0, q2 = q0, q3 = q0;
        const uint32_t m = e.y - f.dmark;
        if (is_group)
        {
            const uint4* g4 = (const uint4*)(f.side + 4ull * (e.x - f.dmark));
            q0 = g4[0]; q1 = g4[1];
            if (m > 3u) { q2 = g4[2]; q3 = g4[3]; }
        }
        // ---- stage 2 for the next tile: seed bits from the words loaded one iteration ago, its table entry requested ----
        uint64_t Vn = 0; uint2 en = make_uint2( 1u, 0u ); bool en_valid = false;
        if (w_valid)
        {
            en_valid = seed_bits_from_words<BITS>( W, w_begin, len, fwd, comp, Vn );
            if (en_valid) en = load_table_entry( f.dtab, Vn >> (2u * r), true );
        }
        // ---- stage 1 for the tile after next: its packed words ----
        {
            const uint32_t t2 = tile + 2u * n_waves, rid2 = t2 * tl.rpt + lr;
            w_valid = t2 < tl.n_tiles && lane_ok && rid2 < tl.reads;
            if (w_valid) { w_begin = seed_begin( rid2 ); W = load_seed_words<BITS>( q.symbols, w_begin, len ); }
        }
        // ---- stage 3b: resolve the current tile ----
        uint32_t x = 1u, y = 0u; bool single = false;
        if (is_one)
        {
            if (context_matches( e.x, e.y - f.dmark, rest, r )) { x = y = e.x - r; single = true; }
        }
        else if (is_rng) { x = e.x; y = e.y; }
        else if (is_group)
        {
            uint32_t hits = 0, pos = 0;
#define NVB_SIDE_ROW(jj, P, C) if ((jj) <= m && context_matches( (P), (C) - f.dmark, rest, r )) { ++hits; pos = (P); }
            NVB_SIDE_ROW( 1u, q0.z, q0.w ) NVB_SIDE_ROW( 2u, q1.x, q1.y ) NVB_SIDE_ROW( 3u, q1.z, q1.w )
            NVB_SIDE_ROW( 4u, q2.x, q2.y ) NVB_SIDE_ROW( 5u, q2.z, q2.w ) NVB_SIDE_ROW( 6u, q3.x, q3.y ) NVB_SIDE_ROW( 7u, q3.z, q3.w )
#undef NVB_SIDE_ROW
            if (hits == 1u)     { x = y = pos - r; single = true; }
            else if (hits > 1u) { x = q0.x; y = q0.y; }             // a repeat longer than the seed: its SA range needs the rank steps
        }
        if (!single && x <= y)                                       // k-mers with many occurrences (and true repeats): rank steps, then the text
        {
            auto sym = [&](const uint32_t s) -> uint32_t { return (uint32_t)(V >> (2u * (len - 1u - s))) & 3u; };
            SearchState st;
            st.x = x; st.y = y; st.s = f.dkmer; st.have_pos = false; st.tpos = 0; st.have_ctx = false; st.ctx = 0; st.is_pos = false; st.sectors = 0;
            uint32_t nblk = 0;
            search_tail<false,true>( f, sym, len, false, st, nblk );
            x = st.x; y = st.y; single = st.is_pos;
            if (!single && x == y)                                   // one row left when the pattern ran out: its position
            {
                const uint32_t sv = f.ssa[x];
                x = (sv == 0xFFFFFFFFu) ? f.length : sv; single = true;
            }
        }
        uint32_t n_out = 0; uint64_t last_key = ~0ull;
        emit_seed_results( j * q.interval, len, read_len, strand, lane, valid, single, x, y, rid, rid * q.spr + j, tile_keys + (uint64_t)tile * 64u, n_out, last_key,
                           res_ranges, res_ids, counts );
        if (lane == 0) tile_counts[tile] = n_out;
        V = Vn; e = en; e_valid = en_valid;
    }
}

// keys of tile t -> keys_out[offsets[t] ...]; the last tile also writes the total.  A few LANES per tile: a tile holds a handful of
// keys (3.8 per strand on the benchmark), copied one by one while counts and offsets are read coalesced
// (one wave per tile, the first version, spent 0.2 ms per launch starting 1.4 M waves that copied four keys each).
__global__ void __launch_bounds__(256)
fm_seed_compact_kernel(const uint64_t* __restrict__ tile_keys, const uint32_t* __restrict__ tile_counts, const uint32_t* __restrict__ tile_offsets,
                       const uint32_t n_tiles, const uint32_t slots, uint64_t* __restrict__ keys_out, unsigned int* __restrict__ counts)
{
    // four lanes per tile, each copying every fourth key: a tile of the two-strand pass holds ~8 keys (one lane per tile took 0.25 ms
    // there, one per tile and strand 0.17), of the per-strand pass ~4
    const uint64_t total = 4ull * n_tiles;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint32_t tile = (uint32_t)(t >> 2), part = (uint32_t)(t & 3u);
        const uint32_t n = tile_counts[tile], off = tile_offsets[tile];
        const uint64_t* src = tile_keys + (uint64_t)tile * slots;
        for (uint32_t k = part; k < n; k += 4u) keys_out[off + k] = src[k];
        if (tile == n_tiles - 1u && part == 0u) counts[0] = off + n;
    }
}

// ---------------------------------------------------------------------------------------------
// hamming_backtrack (nvbio/fmindex/backtrack.h:51-157): approximate matching by backtracking -- the last `seed` symbols of
// the pattern are matched exactly, the rest may differ from the text in up to `mismatches` positions.  Same traversal as the
// reference (children pushed for c = 0..3, popped last-in first-out; a branch that has used all its mismatches finishes with
// an exact match of what is left), so the ranges reach the delegate in the reference's order.  The delegate here is the
// reference benchmark's CountDelegate (nvbio-test/fmindex_test.cu:720-737) plus, optionally, the list of ranges.
// The stack lives in private memory (128 entries as the reference's count_core, :757).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rank4_pair(const DevIndex& f, const uint32_t l, const uint32_t r, uint32_t lo[4], uint32_t hi[4])
{
    #pragma unroll
    for (uint32_t c = 0; c < 4; ++c) { lo[c] = rank_row( f, l, c ); hi[c] = rank_row( f, r, c ); }
}

constexpr uint32_t BACKTRACK_STACK = 128u;

template <int BITS>
__global__ void __launch_bounds__(128)
fm_hamming_backtrack_kernel(const DevIndex f, const StringSetDev q, const uint32_t seed_len, const uint32_t mismatches, const bool quirks,
                            uint32_t* __restrict__ counts, uint32_t* __restrict__ n_ranges, uint2* __restrict__ ranges, const uint32_t max_ranges,
                            uint32_t* __restrict__ overflow)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < q.n; i += gridDim.x * blockDim.x)
    {
        uint32_t begin, len;
        string_bounds( q, i, begin, len );
        SymbolReader<BITS> rd( q.symbols );
        auto pat = [&](const uint32_t k) -> uint32_t { return rd.get( (uint32_t)(begin + k) ); };   // 32-bit index arithmetic, as PackedStream's
        // match( fmi, pattern + off, n, range ): backward over pattern[off + n - 1 .. off]; the reference's loop index is an
        // int32 (fmindex_inl.h:223), so a length above 2^31 (quirks mode only) runs no step at all
        auto match_from = [&](uint32_t x, uint32_t y, const uint32_t off, const uint32_t n, uint32_t& ox, uint32_t& oy) {
            uint32_t nb = 0;
            for (uint32_t k = (n > 0x80000000u ? 0u : n); k > 0u && x <= y; --k)
            {
                const uint32_t c = pat( off + k - 1u );
                if (c > 3u) { x = 1u; y = 0u; break; }
                search_step<false>( f, x, y, c, nb );
            }
            ox = x; oy = y;
        };
        uint32_t total = 0, nr = 0;
        auto report = [&](const uint32_t x, const uint32_t y) {
            total += y + 1u - x;
            if (ranges && nr < max_ranges) ranges[(size_t)i * max_ranges + nr] = make_uint2( x, y );
            ++nr;
        };
        const uint32_t seed = seed_len < len ? seed_len : len;
        if (mismatches == 0u || seed == len)
        {
            uint32_t x, y; match_from( 0u, f.length, 0u, len, x, y );
            if (x <= y) report( x, y );
        }
        else
        {
            uint32_t rx, ry; match_from( 0u, f.length, len - seed, seed, rx, ry );
            if (rx <= ry)
            {
                uint4 stack[BACKTRACK_STACK];
                uint32_t sp = 0;
                uint32_t lo[4], hi[4];
                rank4_pair( f, rx - 1u, ry, lo, hi );
                const uint32_t c0 = pat( len - seed - 1u );
                for (uint32_t c = 0; c < 4u; ++c)
                    if (lo[c] < hi[c])
                        stack[sp++] = make_uint4( L2_of( f, c ) + lo[c] + 1u, L2_of( f, c ) + hi[c], (c == c0) ? 0u : 1u, len - seed - 1u );
                while (sp)
                {
                    const uint4 e = stack[--sp];
                    const uint32_t cost = e.z, l = e.w;
                    if (l == 0u)
                    {
                        if (e.x <= e.y) report( e.x, e.y );
                        // The reference does not stop here (backtrack.h:110-116 falls through): a branch that arrives with all
                        // its mismatches used reports the range a second time, one that arrives with some left goes on past
                        // the start of the pattern, through whatever precedes it in the stream (index l - 1 = 0xFFFFFFFF wraps),
                        // and every branch it spawns there ends in a "match" over a length of 2^32 - k symbols that runs no
                        // step and reports.  Default: stop, as the documentation describes.  quirks: do as the code does.
                        if (!quirks) continue;
                    }
                    if (cost < mismatches)
                    {
                        rank4_pair( f, e.x - 1u, e.y, lo, hi );
                        const uint32_t cp = pat( l - 1u );
                        for (uint32_t c = 0; c < 4u; ++c)
                            if (lo[c] < hi[c])
                            {
                                if (sp < BACKTRACK_STACK)
                                    stack[sp++] = make_uint4( L2_of( f, c ) + lo[c] + 1u, L2_of( f, c ) + hi[c], cost + (c == cp ? 0u : 1u), l - 1u );
                                else atomicAdd( overflow, 1u );
                            }
                    }
                    else
                    {
                        uint32_t x, y; match_from( e.x, e.y, 0u, l, x, y );
                        if (x <= y) report( x, y );
                    }
                }
            }
        }
        counts[i] = total;
        if (n_ranges) n_ranges[i] = nr;
    }
}

// ---------------------------------------------------------------------------------------------
// nvBowtie's approximate seed mapper: seed_mapper<APPROX_MAPPING>::enact + map<find_exact> inside map_kernel's loop
// (nvBowtie/bowtie2/cuda/mapping_inl.h:114-184,288-342,485-556).  Every seed is searched four times -- forwards in the forward index
// and backwards in the index of the REVERSED text, as it stands and complemented -- each search matching its first half exactly and
// allowing one substitution in the rest; every non-empty range goes to the read's hit deque under the max_hits rule, in the
// reference's order.  One lane = one read, as in the reference (the deque is a per-read state machine).
// ---------------------------------------------------------------------------------------------
template <class Q>
__device__ __forceinline__ void map_match_range(const DevIndex& f, uint32_t& x, uint32_t& y, Q& query, const uint32_t begin, const uint32_t end)
{
    uint32_t nb = 0;
    for (uint32_t i = begin; i < end && x <= y; ++i)
    {
        const uint32_t c = query( i );
        if (c > 3u) { x = 1u; y = 0u; return; }
        search_step<false>( f, x, y, c, nb );
    }
}

template <class Q>
__device__ void map_one_mismatch(const DevIndex& f, Q& query, const uint32_t len1, const uint32_t len2, const bool find_exact, const uint32_t flags,
                                 HitHeap& heap, const uint32_t max_hits, uint32_t& range_sum, uint32_t& range_count)
{
    uint32_t bx = 0u, by = f.length;
    map_match_range( f, bx, by, query, 0u, len1 );
    for (uint32_t i = len1; i < len2 && bx <= by; ++i)
    {
        const uint32_t c = query( i );
        uint32_t lo[4], hi[4];
        rank4_pair( f, bx - 1u, by, lo, hi );
        for (uint32_t sub = 0; sub < 4u; ++sub)
            if (sub != c && hi[sub] > lo[sub])
            {
                uint32_t x = L2_of( f, sub ) + lo[sub] + 1u, y = L2_of( f, sub ) + hi[sub];
                map_match_range( f, x, y, query, i + 1u, len2 );
                if (x <= y) push_seed_hit( heap, max_hits, x, y, flags, range_sum, range_count );
            }
        if (c < 4u) { bx = L2_of( f, c ) + lo[c] + 1u; by = L2_of( f, c ) + hi[c]; }
        else        { bx = 1u; by = 0u; break; }
    }
    if (find_exact && bx <= by) push_seed_hit( heap, max_hits, bx, by, flags, range_sum, range_count );
}

template <int BITS>
__global__ void __launch_bounds__(128)
fm_map_approx_kernel(const DevIndex f, const DevIndex rf, const void* __restrict__ symbols, const uint32_t* __restrict__ queue, const uint32_t n_reads,
                     const uint32_t spr, const uint32_t first_off, const uint32_t interval, const uint32_t seed_len, const uint32_t read_len,
                     const uint32_t max_hits, const uint32_t rep_seeds, const uint32_t cap, uint2* __restrict__ deques, uint32_t* __restrict__ sizes,
                     uint8_t* __restrict__ reseed)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_reads; t += gridDim.x * blockDim.x)
    {
        const uint32_t r = queue ? queue[t] : t;
        HitHeap heap; heap.a = deques + (uint64_t)r * cap; heap.n = 0;
        uint32_t range_sum = 0, range_count = 0;
        SymbolReader<BITS> rd( symbols );
        const uint32_t base = r * read_len;                             // reads of one length, back to back
        for (uint32_t j = 0; j < spr; ++j)
        {
            const uint32_t pos = first_off + j * interval;
            auto fq  = [&](const uint32_t i) -> uint32_t { return rd.get( base + pos + i ); };
            auto rq  = [&](const uint32_t i) -> uint32_t { return rd.get( base + pos + seed_len - 1u - i ); };
            auto cfq = [&](const uint32_t i) -> uint32_t { const uint32_t c = fq( i ); return c < 4u ? 3u - c : c; };
            auto crq = [&](const uint32_t i) -> uint32_t { const uint32_t c = rq( i ); return c < 4u ? 3u - c : c; };
            // SeedHit::build_flags( readtype, indexdir, pos ): pos << 20 | rc << 30 | indexdir << 31
            map_one_mismatch( f,  fq,  seed_len / 2u,        seed_len, true,  ((read_len - pos - seed_len) & 0x3FFu) << 20,                       heap, max_hits, range_sum, range_count );
            map_one_mismatch( rf, rq,  (seed_len + 1u) / 2u, seed_len, false, (((read_len - pos - 1u) & 0x3FFu) << 20) | (1u << 31),              heap, max_hits, range_sum, range_count );
            map_one_mismatch( rf, cfq, seed_len / 2u,        seed_len, true,  (((pos + seed_len - 1u) & 0x3FFu) << 20) | (1u << 30) | (1u << 31), heap, max_hits, range_sum, range_count );
            map_one_mismatch( f,  crq, (seed_len + 1u) / 2u, seed_len, false, ((pos & 0x3FFu) << 20) | (1u << 30),                                heap, max_hits, range_sum, range_count );
        }
        sizes[r] = heap.n;
        if (reseed) reseed[r] = (range_count == 0u || range_sum >= rep_seeds * range_count) ? 1 : 0;
    }
}

// level j of the k-mer table from level j-1: entry (key<<2 | c) = one search step on entry key
// (or the entry itself when it is already empty: the reference's loop would have stopped there)
__global__ void __launch_bounds__(256)
fm_ktab_level_kernel(const DevIndex f, const uint2* __restrict__ prev, uint2* __restrict__ next, const uint64_t n_next)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_next; e += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint2 r = prev[e >> 2];
        uint32_t x = r.x, y = r.y, nb = 0;
        if (x <= y) search_step<false>( f, x, y, (uint32_t)(e & 3u), nb );
        next[e] = make_uint2( x, y );
    }
}
__global__ void fm_ktab_root_kernel(uint2* root, const uint32_t length) { root[0] = make_uint2( 0u, length ); }

// ---------------------------------------------------------------------------------------------
// rank / rank4
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
fm_rank_kernel(const DevIndex f, const uint32_t* __restrict__ rows, const uint8_t* __restrict__ syms, const uint32_t n, uint32_t* __restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = rank_row( f, rows[i], syms[i] & 3u );
}
__global__ void __launch_bounds__(256)
fm_rank4_kernel(const DevIndex f, const uint32_t* __restrict__ rows, const uint32_t n, uint4* __restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        uint32_t k = rows[i];
        uint4 r;
        if (k == 0xFFFFFFFFu)    r = make_uint4( 0, 0, 0, 0 );
        else if (k == f.length)  r = make_uint4( f.L2_1 - f.L2_0, f.L2_2 - f.L2_1, f.L2_3 - f.L2_2, f.L2_4 - f.L2_3 );
        else
        {
            if (k >= f.primary) --k;
            if (k == 0xFFFFFFFFu) r = make_uint4( 0, 0, 0, 0 );
            else
            {
                const uint4 b = f.rec[2u * (k >> 6)], o = f.rec[2u * (k >> 6) + 1u];
                const uint4 cnt = count4_in_block( b, k & 63u );
                r = make_uint4( o.x + cnt.x, o.y + cnt.y, o.z + cnt.z, o.w + cnt.w );
            }
        }
        out[i] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// locate
// ---------------------------------------------------------------------------------------------
// The LF walk takes 0..sa_int-1 dependent gathers per row, so a wave that waits for its slowest
// lane keeps most lanes idle.  Both locate kernels therefore REFILL lanes: a lane that reaches a
// sampled row writes its result and immediately starts its next row while its neighbours keep
// walking, so every iteration of the wave-uniform loop issues (nearly) 64 gathers.
//
// MODE 0: pos = locate(row); MODE 1: (j,t) = locate_ssa_iterator(row)
template <int MODE>
__global__ void __launch_bounds__(256)
fm_locate_kernel(const DevIndex f, const uint32_t* rows, const uint32_t n, uint32_t* pos, uint2* jt)
{
    const uint32_t mask   = (1u << f.sa_log) - 1u;
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool     have = i < n;
    uint32_t j = have ? rows[i] : 0u, t = 0;
    while (__any( have ))
    {
        if (have)
        {
            if ((j & mask) == 0)                                 // SSA_index_multiple_context<K>::has (ssa_inl.h:491-495)
            {
                if (MODE == 0) pos[i] = f.ssa[j >> f.sa_log] + t;
                else           jt[i]  = make_uint2( j, t );
                i += stride; have = i < n;
                if (have) { j = rows[i]; t = 0; }
            }
            else { j = lf_step( f, j ); ++t; }
        }
    }
}
__global__ void __launch_bounds__(256)
fm_inv_psi_kernel(const DevIndex f, const uint32_t* rows, const uint32_t n, uint32_t* out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = lf_step( f, rows[i] );
}
__global__ void __launch_bounds__(256)
fm_lookup_kernel(const DevIndex f, const uint2* __restrict__ jt, const uint32_t n, uint32_t* __restrict__ pos)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        pos[i] = f.ssa[jt[i].x >> f.sa_log] + jt[i].y;
}

// ---------------------------------------------------------------------------------------------
// filter
// ---------------------------------------------------------------------------------------------
struct RangeSize
{
    __host__ __device__ __forceinline__ uint64_t operator()(const uint2 r) const { return (uint64_t)(uint32_t)(1u + r.y - r.x); }
};

// hits[h-begin] = (locate(range.x + local), query) for the global hit index h (filter_inl.h:66-118,359-392).
// A workgroup owns a tile of consecutive hit indices; their queries form a contiguous slice of
// `slots`, found once per tile (two binary searches by two lanes), so that each hit's own
// upper_bound runs over a few cached entries instead of log2(n_queries) HBM round trips.
constexpr uint32_t FILTER_TILE = 256u * 8u;

__device__ __forceinline__ uint32_t upper_bound_u64(const uint64_t* __restrict__ a, uint32_t lo, uint32_t hi, const uint64_t v)
{
    while (lo < hi)
    {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] <= v) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// seed enumeration of a hit's query id -> its diagonal key (hit_to_diagonal, examples/fmmap/fmmap.cu:92-117; see
// nvbio_hits_to_diagonals): with KEYS the expansion writes the 8-byte key instead of the (position, query) pair
// qid: optional seed id of every query; read_offsets / intervals: ragged reads (every read's length and seed interval)
struct DiagSpec { uint32_t spr, interval, seed_len, read_len, strand; const uint32_t* qid; const uint32_t* read_offsets; const uint32_t* intervals; };

template <bool KEYS>
__global__ void __launch_bounds__(256)
fm_filter_locate_kernel(const DevIndex f, const uint2* __restrict__ ranges, const uint64_t* __restrict__ slots, const uint32_t n_queries,
                        const uint64_t begin, const uint64_t end, uint2* __restrict__ hits, const uint8_t* __restrict__ direct,
                        const DiagSpec ds, uint64_t* __restrict__ keys)
{
    __shared__ uint32_t s_q[2];
    const uint32_t mask    = (1u << f.sa_log) - 1u;
    const uint64_t n_tiles = (end - begin + FILTER_TILE - 1u) / FILTER_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    {
        const uint64_t t_first = begin + tile * FILTER_TILE;
        const uint64_t t_end   = (t_first + FILTER_TILE < end) ? t_first + FILTER_TILE : end;
        __syncthreads();
        if (threadIdx.x < 2)
            s_q[threadIdx.x] = upper_bound_u64( slots, 0u, n_queries, threadIdx.x ? t_end - 1u : t_first );
        __syncthreads();
        const uint32_t q_lo = s_q[0], q_hi = s_q[1] + 1u < n_queries ? s_q[1] + 1u : n_queries;

        uint64_t h    = t_first + threadIdx.x;
        bool     have = h < t_end;
        uint32_t slot = 0, j = 0, t = 0;
        bool     is_pos = false;                                 // the range of this query already holds a text position
        auto start = [&]() {
            slot = upper_bound_u64( slots, q_lo, q_hi, h );
            const uint64_t base = slot ? slots[slot - 1u] : 0ull;
            j = ranges[slot].x + (uint32_t)(h - base); t = 0;
            is_pos = direct && direct[slot];
        };
        if (have) start();
        while (__any( have ))
        {
            if (have)
            {
                if (is_pos || (j & mask) == 0)
                {
                    const uint32_t pos = is_pos ? j : f.ssa[j >> f.sa_log] + t;
                    if (KEYS)
                    {
                        // bit 31 of a query id flips the strand: the two residual lists of the two-strand seed pass go through one call
                        const uint32_t qv  = ds.qid ? ds.qid[slot] : slot;
                        const uint32_t sid = ds.qid ? (qv & 0x7FFFFFFFu) : qv;
                        const uint32_t str = (ds.strand ^ (ds.qid ? qv >> 31 : 0u)) & 1u;
                        const uint32_t rid = sid / ds.spr;
                        uint32_t       p   = (sid - rid * ds.spr) * (ds.intervals ? ds.intervals[rid] : ds.interval);
                        if (str) p = (ds.read_offsets ? ds.read_offsets[rid + 1] - ds.read_offsets[rid] : ds.read_len) - p - ds.seed_len;
                        keys[h - begin] = ((uint64_t)rid << 34) | ((uint64_t)str << 33) | ((uint64_t)pos + 1024u - p);
                    }
                    else hits[h - begin] = make_uint2( pos, slot );
                    h += 256u; have = h < t_end;
                    if (have) start();
                }
                else { j = lf_step( f, j ); ++t; }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// The residual seeds of a seed pass under a seed-hit cap (nvbio_fm_residual_diagonals): entry e = (SA range, seed id | strand << 31);
// the first `cap` rows of every range are located (the handle holds the full suffix array: the rows of a range are consecutive words)
// and turned into diagonal keys.  Entries arrive SORTED by id: consecutive seeds of a read are neighbours, and a key equal to the key
// the previous entry leaves at the same row t is dropped (the seeds of a read that lies in a repeat list the same loci, row for row,
// because the suffixes that decide the order of a repeat's copies in the suffix array begin behind the repeat whatever the seed's offset
// in it).  One lane per entry, two sweeps over its rows (count, then write: the second finds the SA words in L2), one returning atomic
// per workgroup and 256 entries for the output offset.  A duplicate that survives only costs a repeated extension.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
residual_locate_kernel(const DevIndex f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ ids, const uint32_t n, const uint32_t cap,
                       const DiagSpec ds, uint64_t* __restrict__ keys, unsigned int* __restrict__ n_keys)
{
    __shared__ uint32_t s_wave[4], s_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u)              // (block-uniform trip count)
    {
        const uint32_t e = base + threadIdx.x;
        const bool have = e < n;
        uint32_t x = 1u, rows = 0u, xp = 1u, rows_p = 0u; uint64_t hi = 0, hi_p = ~0ull; uint32_t p = 0, pp = 0;
        auto describe = [&](const uint32_t ee, uint32_t& x0, uint32_t& nr, uint64_t& khi, uint32_t& off) {
            const uint2 r = ranges[ee];
            x0 = r.x; nr = r.x <= r.y ? (r.y - r.x + 1u < cap ? r.y - r.x + 1u : cap) : 0u;
            const uint32_t qv  = ids[ee];
            const uint32_t sid = qv & 0x7FFFFFFFu, str = (ds.strand ^ (qv >> 31)) & 1u;
            const uint32_t rid = sid / ds.spr;
            off = (sid - rid * ds.spr) * (ds.intervals ? ds.intervals[rid] : ds.interval);
            if (str) off = (ds.read_offsets ? ds.read_offsets[rid + 1] - ds.read_offsets[rid] : ds.read_len) - off - ds.seed_len;
            khi = ((uint64_t)rid << 34) | ((uint64_t)str << 33);
        };
        if (have) { describe( e, x, rows, hi, p ); if (e > 0u) describe( e - 1u, xp, rows_p, hi_p, pp ); }
        const bool cmp = have && hi == hi_p;                                                 // the previous entry is a seed of the same read and strand
        auto key_at = [&](const uint32_t x0, const uint32_t t, const uint64_t khi, const uint32_t off) -> uint64_t {
            const uint32_t sv = f.ssa[x0 + t];
            return khi | ((uint64_t)((sv == 0xFFFFFFFFu) ? f.length : sv) + 1024u - off);
        };
        uint32_t cnt = 0;
        uint64_t kept = 0;                                                                    // bit t: row t's key is written (cap <= 64)
        for (uint32_t t = 0; t < rows; ++t)
        {
            const uint64_t k = key_at( x, t, hi, p );
            const bool dup = cmp && t < rows_p && key_at( xp, t, hi_p, pp ) == k;
            if (!dup) { kept |= 1ull << t; ++cnt; }
        }
        uint32_t incl = cnt;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t a = (uint32_t)__shfl_up( (int)incl, d ); if (lane >= (uint32_t)d) incl += a; }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const uint32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            s_base = tot ? atomicAdd( n_keys, tot ) : 0u;
        }
        __syncthreads();
        uint32_t o = s_base + incl - cnt;
        for (uint32_t w = 0; w < wave; ++w) o += s_wave[w];
        for (uint32_t t = 0; t < rows; ++t)
            if ((kept >> t) & 1ull) keys[o++] = key_at( x, t, hi, p );
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static nvbio_status make_set(const nvbio_string_set* s, StringSetDev* d)
{
    NVB_REQUIRE( s != nullptr, "queries is NULL" );
    NVB_REQUIRE( s->symbol_bits == 2 || s->symbol_bits == 4 || s->symbol_bits == 8, "symbol_bits must be 2, 4 or 8" );
    NVB_REQUIRE( s->n == 0 || s->symbols_dev != nullptr, "symbols_dev is NULL" );
    NVB_REQUIRE( !(s->offsets_are_ranges && s->offsets_dev == nullptr), "offsets_are_ranges without offsets_dev" );
    d->symbols = s->symbols_dev; d->offsets = s->offsets_dev; d->ranges = s->offsets_are_ranges;
    d->fixed_len = s->fixed_len; d->stride = s->stride; d->n = s->n;
    d->spr = s->seeds_per_string; d->interval = s->seed_interval; d->intervals = s->seed_intervals_dev;
    NVB_REQUIRE( d->intervals == nullptr || (d->spr > 0 && d->offsets != nullptr), "seed_intervals_dev needs seeds_per_string > 0 and offsets_dev (n_strings + 1 entries)" );
    return NVBIO_OK;
}

// in place: every one-row range of a table becomes (text position of that row, 0xFFFFFFFF)  [direct table, format 1]
__global__ void __launch_bounds__(256)
fm_dtab_kernel(uint2* __restrict__ tab, const uint32_t* __restrict__ sa, const uint32_t length, const uint64_t n)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint2 r = tab[e];
        if (r.x == r.y)
        {
            const uint32_t sv = sa[r.x];
            tab[e] = make_uint2( sv == 0xFFFFFFFFu ? length : sv, 0xFFFFFFFFu );
        }
    }
}

// ---- direct table, format 2 (fm_seed_device.h): positions + left context in the entries, groups for 2..7 rows ----
constexpr uint32_t DT_TILE = 1024u;                           // entries per tile: 256 threads x 4 consecutive entries

// 0: plain range (empty, or more than DTAB_SIDE_MAX rows)   1: one row   2: 2..3 rows (4 slots)   3: 4..7 rows (8 slots)
__device__ __forceinline__ uint32_t dtab_class(const uint2 r, const bool groups)
{
    if (r.x > r.y) return 0u;
    const uint32_t d = r.y - r.x;
    if (d == 0u) return 1u;
    if (!groups) return 0u;
    return d <= 2u ? 2u : (d < DTAB_SIDE_MAX ? 3u : 0u);
}

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u;
    #pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1)
    {
        const uint32_t o = (uint32_t)__shfl_up( (int)v, d );
        if (lane >= d) v += o;
    }
    return v;
}

// tile t: cnt_small[t] / cnt_large[t] = its k-mers with 2..3 / 4..7 rows
__global__ void __launch_bounds__(256)
fm_dtab_count_kernel(const uint2* __restrict__ tab, const uint64_t n, const uint32_t n_tiles, uint32_t* __restrict__ cnt_small, uint32_t* __restrict__ cnt_large)
{
    __shared__ uint32_t s_a[4], s_b[4];
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    {
        const uint64_t e0 = (uint64_t)tile * DT_TILE + threadIdx.x * 4u;
        uint32_t a = 0, b = 0;
        #pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (e0 + k < n) { const uint32_t c = dtab_class( tab[e0 + k], true ); a += (c == 2u); b += (c == 3u); }
        a = wave_inclusive_sum( a ); b = wave_inclusive_sum( b );
        if ((threadIdx.x & 63u) == 63u) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
        __syncthreads();
        if (threadIdx.x == 0) { cnt_small[tile] = s_a[0] + s_a[1] + s_a[2] + s_a[3]; cnt_large[tile] = s_b[0] + s_b[1] + s_b[2] + s_b[3]; }
        __syncthreads();
    }
}

// rewrite the entries in place and fill the groups; off_small / off_large = exclusive scans of the tile counts;
// large groups occupy side slots [0, 8 tot_large), small ones follow
__global__ void __launch_bounds__(256)
fm_dtab_fill_kernel(uint2* __restrict__ tab, const uint64_t n, const uint32_t n_tiles, const uint32_t* __restrict__ off_small,
                    const uint32_t* __restrict__ off_large, const uint32_t tot_large, const uint32_t* __restrict__ sa,
                    const uint32_t* __restrict__ text, const uint32_t length, uint2* __restrict__ side)
{
    __shared__ uint32_t s_a[4], s_b[4];
    const bool groups = side != nullptr;
    auto position = [&](const uint32_t row) -> uint32_t { const uint32_t sv = sa[row]; return sv == 0xFFFFFFFFu ? length : sv; };
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    {
        const uint64_t e0 = (uint64_t)tile * DT_TILE + threadIdx.x * 4u;
        uint2 r[4]; uint32_t c[4];
        uint32_t a = 0, b = 0;
        #pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
        {
            c[k] = 0u; r[k] = make_uint2( 1u, 0u );
            if (e0 + k < n) { r[k] = tab[e0 + k]; c[k] = dtab_class( r[k], groups ); a += (c[k] == 2u); b += (c[k] == 3u); }
        }
        const uint32_t ia = wave_inclusive_sum( a ), ib = wave_inclusive_sum( b );
        if ((threadIdx.x & 63u) == 63u) { s_a[threadIdx.x >> 6] = ia; s_b[threadIdx.x >> 6] = ib; }
        __syncthreads();
        uint32_t ga = ia - a, gb = ib - b;                       // groups of this tile before this thread's
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { ga += s_a[w]; gb += s_b[w]; }
        if (groups) { ga += off_small[tile]; gb += off_large[tile]; }
        __syncthreads();
        #pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
        {
            if (c[k] == 1u)
            {
                const uint32_t p = position( r[k].x );
                tab[e0 + k] = make_uint2( p, DTAB_MARK | left_context( text, p ) );
            }
            else if (c[k] >= 2u)
            {
                const uint32_t rows  = r[k].y - r[k].x + 1u;
                const uint32_t slots = c[k] == 2u ? 4u : 8u;
                const uint32_t idx4  = c[k] == 2u ? 2u * tot_large + ga++ : 2u * gb++;       // group address in units of 4 slots
                uint2* g = side + 4ull * idx4;
                g[0] = r[k];
                for (uint32_t j = 1; j < slots; ++j)
                {
                    uint2 v = make_uint2( 0u, DTAB_MARK );
                    if (j <= rows) { const uint32_t p = position( r[k].x + j - 1u ); v = make_uint2( p, DTAB_MARK | left_context( text, p ) ); }
                    g[j] = v;
                }
                tab[e0 + k] = make_uint2( DTAB_MARK | idx4, DTAB_MARK | rows );
            }
        }
    }
}

// rewrite level k of the table (in `tab`) into the direct table; format 2 when the text is short enough for the marker
static nvbio_status build_direct_table(FMIndexImpl* idx, uint2* tab, const uint64_t entries, hipStream_t stream)
{
    const uint32_t length = idx->view.length;
    idx->side = nullptr; idx->dmark = 0xFFFFFFFFu; idx->dctx = 0;
    if ((uint64_t)length + 2u > DTAB_MARK || (idx->table_flags & NVBIO_FM_TABLE_NO_CONTEXT))
    {
        hipLaunchKernelGGL( fm_dtab_kernel, dim3( grid_for( entries ) ), dim3(256), 0, stream, tab, idx->view.ssa_dev, length, entries );
        NVB_HIP( hipGetLastError() );
        return NVBIO_OK;
    }
    const uint32_t n_tiles = (uint32_t)((entries + DT_TILE - 1u) / DT_TILE);
    const dim3 grid( n_tiles < 256u * 64u ? n_tiles : 256u * 64u ), block( 256 );
    uint32_t* cnt = nullptr; void* temp = nullptr; uint2* side = nullptr;
    size_t temp_bytes = 0;
    uint32_t tot_small = 0, tot_large = 0;
    const bool want_groups = !(idx->table_flags & NVBIO_FM_TABLE_NO_GROUPS);
    nvbio_status st = NVBIO_OK;
    if (want_groups)
    {
        if (hipMalloc( (void**)&cnt, 4ull * n_tiles * sizeof(uint32_t) ) != hipSuccess) { (void)hipGetLastError(); set_error( "direct table: out of device memory" ); return NVBIO_ERR_NOMEM; }
        uint32_t *cs = cnt, *cl = cnt + n_tiles, *os = cnt + 2ull * n_tiles, *ol = cnt + 3ull * n_tiles;
        hipLaunchKernelGGL( fm_dtab_count_kernel, grid, block, 0, stream, (const uint2*)tab, entries, n_tiles, cs, cl );
        hipError_t e = hipcub::DeviceScan::ExclusiveSum( nullptr, temp_bytes, cs, os, (int)n_tiles, stream );
        if (e == hipSuccess) e = hipMalloc( &temp, temp_bytes ? temp_bytes : 16 );
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum( temp, temp_bytes, cs, os, (int)n_tiles, stream );
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum( temp, temp_bytes, cl, ol, (int)n_tiles, stream );
        uint32_t last[4] = { 0, 0, 0, 0 };
        if (e == hipSuccess) e = hipMemcpyAsync( &last[0], cs + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
        if (e == hipSuccess) e = hipMemcpyAsync( &last[1], os + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
        if (e == hipSuccess) e = hipMemcpyAsync( &last[2], cl + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
        if (e == hipSuccess) e = hipMemcpyAsync( &last[3], ol + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
        if (e == hipSuccess) e = hipStreamSynchronize( stream );
        if (e != hipSuccess) { (void)hipGetLastError(); set_error( "direct table: counting pass failed: %s", hipGetErrorString( e ) ); st = NVBIO_ERR_HIP; }
        tot_small = last[0] + last[1]; tot_large = last[2] + last[3];
        const uint64_t units = 2ull * tot_large + tot_small;                  // groups in units of 4 slots (32 bytes)
        if (st == NVBIO_OK && units > 0 && units < (1ull << 30))
        {
            if (hipMalloc( (void**)&side, units * 32ull ) != hipSuccess) { (void)hipGetLastError(); side = nullptr; }   // no memory: no groups
        }
    }
    if (st == NVBIO_OK)
    {
        hipLaunchKernelGGL( fm_dtab_fill_kernel, grid, block, 0, stream, tab, entries, n_tiles, cnt ? cnt + 2ull * n_tiles : nullptr,
                            cnt ? cnt + 3ull * n_tiles : nullptr, tot_large, idx->view.ssa_dev, (const uint32_t*)idx->text, length, side );
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize( stream ) != hipSuccess) { set_error( "direct table: fill pass failed" ); st = NVBIO_ERR_HIP; }
    }
    if (cnt)  (void)hipFree( cnt );
    if (temp) (void)hipFree( temp );
    if (st != NVBIO_OK) { if (side) (void)hipFree( side ); return st; }
    idx->side = side; idx->dmark = DTAB_MARK; idx->dctx = DTAB_CTX;
    if (side) idx->owned_bytes += (2ull * tot_large + tot_small) * 32ull;
    return NVBIO_OK;
}

// The k-mer table is built level by level (level j from level j-1, one search step per entry), alternating between two
// buffers.  A handle that holds the full SA and the text (direct-capable) keeps the LAST TWO levels: level k-1 stays the
// plain table of match() (SA ranges), level k becomes the table of the direct seed pass, the entries of k-mers with few
// occurrences rewritten in place to text positions (fm_seed_device.h).  Other handles keep level k as the plain table.
// (k = 17: 32 + 128 GiB; k = 16: 8 + 32 GiB.)
#include "fm_canon_inl.h"

// NVBIO_FM_TABLE_CANONICAL: the plain table of (k-1)-mers (match()) + the canonical two-strand table of k-mers built from it
// (k = 17: 32 + 64 GiB, and nothing larger while building)
static nvbio_status build_canonical_tables(FMIndexImpl* idx, uint32_t k, hipStream_t stream)
{
    const uint32_t kk = k - 1u;
    const uint64_t entries = 1ull << (2 * kk);
    uint2 *a = nullptr, *b = nullptr;
    if (hipMalloc( (void**)&a, entries * sizeof(uint2) ) != hipSuccess) { (void)hipGetLastError(); set_error( "k-mer table: out of device memory" ); return NVBIO_ERR_NOMEM; }
    if (hipMalloc( (void**)&b, (entries / 4) * sizeof(uint2) ) != hipSuccess) { (void)hipGetLastError(); (void)hipFree( a ); set_error( "k-mer table: out of device memory" ); return NVBIO_ERR_NOMEM; }
    uint2* cur = (kk % 2 == 0) ? a : b;
    uint2* oth = (kk % 2 == 0) ? b : a;
    DevIndex f = idx->dev(); f.ktab = nullptr; f.kmer = 0; f.dtab = nullptr; f.dkmer = 0;
    hipLaunchKernelGGL( fm_ktab_root_kernel, dim3(1), dim3(1), 0, stream, cur, idx->view.length );
    for (uint32_t j = 1; j <= kk; ++j)
    {
        const uint64_t n_next = 1ull << (2 * j);
        hipLaunchKernelGGL( fm_ktab_level_kernel, dim3( grid_for( n_next ) ), dim3(256), 0, stream, f, (const uint2*)cur, oth, n_next );
        uint2* t = cur; cur = oth; oth = t;
    }
    const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize( stream ) == hipSuccess;
    (void)hipFree( b );                                          // level kk - 1
    if (!ok) { (void)hipFree( a ); set_error( "k-mer table build failed" ); return NVBIO_ERR_HIP; }
    idx->ktab = a; idx->kmer = kk;
    idx->owned_bytes += entries * sizeof(uint2);
    const nvbio_status st = build_canonical_table( idx, k, (idx->table_flags & NVBIO_FM_TABLE_CANONICAL_WIDE) != 0, stream );
    if (st != NVBIO_OK) { (void)hipFree( a ); idx->ktab = nullptr; idx->kmer = 0; }
    return st;
}

static nvbio_status build_kmer_table(FMIndexImpl* idx, uint32_t k, hipStream_t stream)
{
    idx->ktab = nullptr; idx->kmer = 0; idx->dtab = nullptr; idx->dkmer = 0; idx->side = nullptr; idx->dmark = 0xFFFFFFFFu; idx->dctx = 0;
    idx->ctab = nullptr; idx->cside = nullptr; idx->ckmer = 0; idx->cwide = 0;
    if (k == 0) return NVBIO_OK;
    const bool direct = idx->text && idx->view.ssa_dev && idx->view.sa_int == 1 && k >= 2 && !(idx->table_flags & NVBIO_FM_TABLE_NO_DIRECT);
    if (idx->table_flags & (NVBIO_FM_TABLE_CANONICAL | NVBIO_FM_TABLE_CANONICAL_WIDE))
    {
        if (!direct || (k & 1u) == 0u || k < 3u)
        {
            set_error( "NVBIO_FM_TABLE_CANONICAL needs the full suffix array, the text (sa_int = 1) and an odd kmer_len >= 3" );
            return NVBIO_ERR_UNSUPPORTED;
        }
        return build_canonical_tables( idx, k, stream );
    }
    const uint64_t entries = 1ull << (2 * k);
    uint2 *a = nullptr, *b = nullptr;
    if (hipMalloc( (void**)&a, entries * sizeof(uint2) ) != hipSuccess) { (void)hipGetLastError(); set_error( "k-mer table: out of device memory" ); return NVBIO_ERR_NOMEM; }
    if (hipMalloc( (void**)&b, (entries / 4) * sizeof(uint2) ) != hipSuccess) { (void)hipGetLastError(); (void)hipFree( a ); set_error( "k-mer table: out of device memory" ); return NVBIO_ERR_NOMEM; }
    // levels alternate between the two buffers so that level k lands in `a` (the large one) and level k-1 in `b`
    uint2* cur = (k % 2 == 0) ? a : b;
    uint2* oth = (k % 2 == 0) ? b : a;
    DevIndex f = idx->dev(); f.ktab = nullptr; f.kmer = 0; f.dtab = nullptr; f.dkmer = 0;
    hipLaunchKernelGGL( fm_ktab_root_kernel, dim3(1), dim3(1), 0, stream, cur, idx->view.length );
    for (uint32_t j = 1; j <= k; ++j)
    {
        const uint64_t n_next = 1ull << (2 * j);
        hipLaunchKernelGGL( fm_ktab_level_kernel, dim3( grid_for( n_next ) ), dim3(256), 0, stream, f, (const uint2*)cur, oth, n_next );
        uint2* t = cur; cur = oth; oth = t;
    }
    // cur == a (level k), oth == b (level k-1) by construction
    nvbio_status st = NVBIO_OK;
    if (hipGetLastError() != hipSuccess) { set_error( "k-mer table build failed" ); st = NVBIO_ERR_HIP; }
    if (st == NVBIO_OK && direct) st = build_direct_table( idx, a, entries, stream );
    if (st == NVBIO_OK && hipStreamSynchronize( stream ) != hipSuccess) { set_error( "k-mer table build failed" ); st = NVBIO_ERR_HIP; }
    if (st != NVBIO_OK)
    {
        (void)hipFree( a ); (void)hipFree( b );
        return st;
    }
    if (direct)
    {
        idx->ktab = b; idx->kmer = k - 1u; idx->dtab = a; idx->dkmer = k;
        idx->owned_bytes += (entries + entries / 4) * sizeof(uint2);
    }
    else
    {
        (void)hipFree( b );
        idx->ktab = a; idx->kmer = k;
        idx->owned_bytes += entries * sizeof(uint2);
    }
    return NVBIO_OK;
}

nvbio_status fm_index_adopt(const nvbio_fm_index_view* view, int device, uint32_t kmer_len, bool owns, hipStream_t stream, nvbio_fm_index_t* out,
                            uint32_t* isa, uint32_t* text, uint32_t table_flags)
{
    FMIndexImpl* idx = new (std::nothrow) FMIndexImpl;
    if (!idx) { set_error( "out of host memory" ); return NVBIO_ERR_NOMEM; }
    idx->device = device; idx->view = *view; idx->ktab = nullptr; idx->kmer = 0; idx->dtab = nullptr; idx->dkmer = 0; idx->isa = nullptr; idx->text = nullptr;
    idx->side = nullptr; idx->dmark = 0xFFFFFFFFu; idx->dctx = 0; idx->table_flags = table_flags;
    idx->ctab = nullptr; idx->cside = nullptr; idx->ckmer = 0; idx->cwide = 0;
    if (idx->view.sa_int == 0) idx->view.sa_int = 16;
    idx->owns_arrays = owns; idx->owned_bytes = owns ? (view->bwt_occ_words + view->ssa_words) * 4ull : 0ull;
    idx->isa = isa; idx->text = text;
    if (isa)  idx->owned_bytes += ((uint64_t)view->length + 1u) * 4ull;
    if (text) idx->owned_bytes += (((uint64_t)view->length + 15u) / 16u) * 4ull;
    const nvbio_status st = build_kmer_table( idx, kmer_len, stream );
    if (st != NVBIO_OK)
    {
        if (owns) { (void)hipFree( (void*)view->bwt_occ_dev ); (void)hipFree( (void*)view->ssa_dev ); }
        if (isa)  (void)hipFree( isa );
        if (text) (void)hipFree( text );
        delete idx;
        return st;
    }
    *out = (nvbio_fm_index_t)idx;
    return NVBIO_OK;
}

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" {

nvbio_status nvbio_fm_index_create(const nvbio_fm_index_view* view, int device, uint32_t kmer_len, void* stream, nvbio_fm_index_t* out)
{
    NVB_REQUIRE( view && out, "view/out is NULL" );
    NVB_REQUIRE( view->bwt_occ_dev != nullptr, "bwt_occ_dev is NULL" );
    NVB_REQUIRE( ((uintptr_t)view->bwt_occ_dev & 31u) == 0, "bwt_occ_dev must be 32-byte aligned" );
    NVB_REQUIRE( kmer_len <= 17, "kmer_len must be <= 17" );
    NVB_REQUIRE( view->L2[4] == view->length, "L2[4] must equal length" );
    const uint32_t K = view->sa_int ? view->sa_int : 16u;
    NVB_REQUIRE( K <= 64 && (K & (K - 1u)) == 0, "sa_int must be a power of two in [1,64]" );
    NVB_REQUIRE( view->primary <= view->length, "primary out of range" );
    const uint64_t need = 2ull * ((((uint64_t)view->length + 15u) / 16u + 3u) & ~3ull);
    NVB_REQUIRE( view->bwt_occ_words >= need, "bwt_occ_words too small for length" );
    NVB_REQUIRE( view->ssa_dev == nullptr || view->ssa_words >= (uint64_t)view->length / K + 1u, "ssa_words too small for length" );
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    return fm_index_adopt( view, device, kmer_len, false, (hipStream_t)stream, out, nullptr, nullptr, 0u );
}

nvbio_status nvbio_fm_index_destroy(nvbio_fm_index_t index)
{
    if (!index) return NVBIO_OK;
    FMIndexImpl* idx = (FMIndexImpl*)index;
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    if (idx->ktab) (void)hipFree( idx->ktab );
    if (idx->dtab) (void)hipFree( idx->dtab );
    if (idx->side) (void)hipFree( idx->side );
    if (idx->ctab)  (void)hipFree( idx->ctab );
    if (idx->cside) (void)hipFree( idx->cside );
    if (idx->isa)  (void)hipFree( idx->isa );
    if (idx->text) (void)hipFree( idx->text );
    if (idx->owns_arrays) { (void)hipFree( (void*)idx->view.bwt_occ_dev ); (void)hipFree( (void*)idx->view.ssa_dev ); }
    delete idx;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_index_get_view(nvbio_fm_index_t index, nvbio_fm_index_view* view)
{
    NVB_REQUIRE( index && view, "index/view is NULL" );
    *view = ((FMIndexImpl*)index)->view;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_index_export(nvbio_fm_index_t index, uint32_t* bwt_occ_out_dev, uint32_t* ssa_out_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    if (bwt_occ_out_dev)
        NVB_HIP( hipMemcpyAsync( bwt_occ_out_dev, idx->view.bwt_occ_dev, idx->view.bwt_occ_words * 4ull, hipMemcpyDeviceToDevice, (hipStream_t)stream ) );
    if (ssa_out_dev && idx->view.ssa_dev)
        NVB_HIP( hipMemcpyAsync( ssa_out_dev, idx->view.ssa_dev, idx->view.ssa_words * 4ull, hipMemcpyDeviceToDevice, (hipStream_t)stream ) );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_index_device_bytes(nvbio_fm_index_t index, uint64_t* bytes)
{
    NVB_REQUIRE( index && bytes, "index/bytes is NULL" );
    *bytes = ((FMIndexImpl*)index)->owned_bytes;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_match(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t flags,
                            nvbio_uint2* ranges_dev, uint32_t* blocks_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    StringSetDev q; NVB_CHECK( make_set( queries, &q ) );
    if (q.n == 0) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev != nullptr, "ranges_dev is NULL" );
    NVB_REQUIRE( blocks_dev == nullptr || (flags & NVBIO_FM_NO_KMER_TABLE), "blocks_dev requires NVBIO_FM_NO_KMER_TABLE" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    const DevIndex f = idx->dev();
    // grid-stride launch, 4x oversubscribed (32 blocks of 256 per CU): measured 8.35 ms per 90 M seeds,
    // against 9.8 ms for an exactly resident grid (8 per CU: tail imbalance) and 8.4 ms for 6 per CU
    const dim3 grid( grid_for( q.n ) ), block( 256 );
    hipStream_t s = (hipStream_t)stream;
    const bool use_table = (f.ktab != nullptr) && !(flags & NVBIO_FM_NO_KMER_TABLE);
#define NVB_LAUNCH_MATCH(BITS)                                                                                              \
    if (blocks_dev)    hipLaunchKernelGGL( (fm_match_kernel<BITS,true,false>),  grid, block, 0, s, f, q, flags, (uint2*)ranges_dev, blocks_dev ); \
    else if (use_table) hipLaunchKernelGGL( (fm_match_kernel<BITS,false,true>),  grid, block, 0, s, f, q, flags, (uint2*)ranges_dev, blocks_dev ); \
    else                hipLaunchKernelGGL( (fm_match_kernel<BITS,false,false>), grid, block, 0, s, f, q, flags, (uint2*)ranges_dev, blocks_dev )
    switch (queries->symbol_bits)
    {
    case 2: NVB_LAUNCH_MATCH(2); break;
    case 4: NVB_LAUNCH_MATCH(4); break;
    default: NVB_LAUNCH_MATCH(8); break;
    }
#undef NVB_LAUNCH_MATCH
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_rank(nvbio_fm_index_t index, const uint32_t* rows_dev, const uint8_t* syms_dev, uint32_t n, uint32_t* out_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( rows_dev && syms_dev && out_dev, "NULL device pointer" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( fm_rank_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, idx->dev(), rows_dev, syms_dev, n, out_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_rank4(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n, uint32_t* out_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( rows_dev && out_dev, "NULL device pointer" );
    NVB_REQUIRE( ((uintptr_t)out_dev & 15u) == 0, "out_dev must be 16-byte aligned" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( fm_rank4_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, idx->dev(), rows_dev, n, (uint4*)out_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

static nvbio_status locate_common(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n, uint32_t* pos_dev, nvbio_uint2* jt_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( rows_dev && (pos_dev || jt_dev), "NULL device pointer" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    NVB_REQUIRE( jt_dev || idx->view.ssa_dev, "index has no sampled suffix array" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    if (pos_dev) hipLaunchKernelGGL( fm_locate_kernel<0>, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, idx->dev(), rows_dev, n, pos_dev, (uint2*)nullptr );
    else         hipLaunchKernelGGL( fm_locate_kernel<1>, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, idx->dev(), rows_dev, n, (uint32_t*)nullptr, (uint2*)jt_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_basic_inv_psi(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n, uint32_t* out_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( rows_dev && out_dev, "NULL device pointer" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( fm_inv_psi_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, idx->dev(), rows_dev, n, out_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_locate(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n, uint32_t* pos_dev, void* stream)
{
    NVB_REQUIRE( pos_dev != nullptr || n == 0, "pos_dev is NULL" );
    return locate_common( index, rows_dev, n, pos_dev, nullptr, stream );
}
nvbio_status nvbio_fm_locate_init(nvbio_fm_index_t index, const uint32_t* rows_dev, uint32_t n, nvbio_uint2* jt_dev, void* stream)
{
    NVB_REQUIRE( jt_dev != nullptr || n == 0, "jt_dev is NULL" );
    return locate_common( index, rows_dev, n, nullptr, jt_dev, stream );
}
nvbio_status nvbio_fm_locate_lookup(nvbio_fm_index_t index, const nvbio_uint2* jt_dev, uint32_t n, uint32_t* pos_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( jt_dev && pos_dev, "NULL device pointer" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    NVB_REQUIRE( idx->view.ssa_dev, "index has no sampled suffix array" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( fm_lookup_kernel, dim3( grid_for( n ) ), dim3(256), 0, (hipStream_t)stream, idx->dev(), (const uint2*)jt_dev, n, pos_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_filter_scan(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, uint32_t n_queries,
                                  uint64_t* slots_dev, uint64_t* n_hits, void* stream)
{
    NVB_REQUIRE( index && n_hits, "NULL argument" );
    *n_hits = 0;
    if (n_queries == 0) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev && slots_dev, "NULL device pointer" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;

    hipcub::TransformInputIterator<uint64_t, RangeSize, const uint2*> sizes( (const uint2*)ranges_dev, RangeSize() );
    size_t temp_bytes = 0;
    NVB_HIP( hipcub::DeviceScan::InclusiveSum( nullptr, temp_bytes, sizes, slots_dev, (int)n_queries, s ) );
    void* temp = nullptr;
    if (scratch_alloc( &temp, temp_bytes ? temp_bytes : 16, s ) != hipSuccess) { set_error( "filter_scan: out of device memory" ); return NVBIO_ERR_NOMEM; }
    const hipError_t e = hipcub::DeviceScan::InclusiveSum( temp, temp_bytes, sizes, slots_dev, (int)n_queries, s );
    scratch_free( temp, s );
    if (e != hipSuccess) { set_error( "filter_scan: scan failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    NVB_HIP( hipMemcpyAsync( n_hits, slots_dev + (n_queries - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, s ) );
    NVB_HIP( hipStreamSynchronize( s ) );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_filter_rank(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t flags,
                                  nvbio_uint2* ranges_dev, uint64_t* slots_dev, uint64_t* n_hits, void* stream)
{
    NVB_REQUIRE( index && queries && n_hits, "NULL argument" );
    *n_hits = 0;
    if (queries->n == 0) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev && slots_dev, "NULL device pointer" );
    NVB_CHECK( nvbio_fm_match( index, queries, flags, ranges_dev, nullptr, stream ) );
    return nvbio_fm_filter_scan( index, ranges_dev, queries->n, slots_dev, n_hits, stream );
}

nvbio_status nvbio_fm_filter_locate(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                    uint32_t n_queries, uint64_t begin, uint64_t end, nvbio_uint2* hits_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (end <= begin) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev && slots_dev && hits_dev, "NULL device pointer" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    NVB_REQUIRE( idx->view.ssa_dev, "index has no sampled suffix array" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( fm_filter_locate_kernel<false>, dim3( grid_for( (end - begin + FILTER_TILE - 1u) / FILTER_TILE * 256u ) ), dim3(256), 0, (hipStream_t)stream,
                        idx->dev(), (const uint2*)ranges_dev, slots_dev, n_queries, begin, end, (uint2*)hits_dev, (const uint8_t*)nullptr, DiagSpec{}, (uint64_t*)nullptr );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_index_supports_direct(nvbio_fm_index_t index, int* yes)
{
    NVB_REQUIRE( index && yes, "index/yes is NULL" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    *yes = (idx->text && idx->view.ssa_dev && idx->view.sa_int == 1) ? 1 : 0;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_match_direct(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t flags,
                                   nvbio_uint2* ranges_dev, uint8_t* direct_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    StringSetDev q; NVB_CHECK( make_set( queries, &q ) );
    if (q.n == 0) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev && direct_dev, "NULL device pointer" );
    if (!(idx->text && idx->view.ssa_dev && idx->view.sa_int == 1))
    {
        set_error( "nvbio_fm_match_direct needs the full suffix array and the text: build the index with sa_int = 1" );
        return NVBIO_ERR_UNSUPPORTED;
    }
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    const DevIndex f = idx->dev();
    const dim3 grid( grid_for( q.n ) ), block( 256 );
    hipStream_t s = (hipStream_t)stream;
    const bool use_table = (f.ktab != nullptr) && !(flags & NVBIO_FM_NO_KMER_TABLE);
#define NVB_LAUNCH_DIRECT(BITS)                                                                                              \
    if (use_table) hipLaunchKernelGGL( (fm_match_kernel<BITS,false,true,true>),  grid, block, 0, s, f, q, flags, (uint2*)ranges_dev, (uint32_t*)nullptr, direct_dev ); \
    else           hipLaunchKernelGGL( (fm_match_kernel<BITS,false,false,true>), grid, block, 0, s, f, q, flags, (uint2*)ranges_dev, (uint32_t*)nullptr, direct_dev )
    switch (queries->symbol_bits)
    {
    case 2: NVB_LAUNCH_DIRECT(2); break;
    case 4: NVB_LAUNCH_DIRECT(4); break;
    default: NVB_LAUNCH_DIRECT(8); break;
    }
#undef NVB_LAUNCH_DIRECT
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_hamming_backtrack(nvbio_fm_index_t index, const nvbio_string_set* queries, uint32_t seed_len, uint32_t mismatches, uint32_t flags,
                                        uint32_t* counts_dev, uint32_t* n_ranges_dev, nvbio_uint2* ranges_dev, uint32_t max_ranges, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    StringSetDev q; NVB_CHECK( make_set( queries, &q ) );
    if (q.n == 0) return NVBIO_OK;
    NVB_REQUIRE( counts_dev != nullptr, "counts_dev is NULL" );
    NVB_REQUIRE( ranges_dev == nullptr || max_ranges > 0, "ranges_dev without max_ranges" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    uint32_t* overflow = nullptr;
    NVB_HIP( scratch_alloc( (void**)&overflow, sizeof(uint32_t), s ) );
    NVB_HIP( hipMemsetAsync( overflow, 0, sizeof(uint32_t), s ) );
    const DevIndex f = idx->dev();
    const dim3 grid( grid_for( q.n, 128 ) ), block( 128 );
#define NVB_LAUNCH_BT(BITS) hipLaunchKernelGGL( (fm_hamming_backtrack_kernel<BITS>), grid, block, 0, s, f, q, seed_len, mismatches, (flags & NVBIO_BACKTRACK_REFERENCE_QUIRKS) != 0, counts_dev, n_ranges_dev, \
                                                (uint2*)ranges_dev, max_ranges, overflow )
    switch (queries->symbol_bits)
    {
    case 2: NVB_LAUNCH_BT(2); break;
    case 4: NVB_LAUNCH_BT(4); break;
    default: NVB_LAUNCH_BT(8); break;
    }
#undef NVB_LAUNCH_BT
    uint32_t h_over = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync( &h_over, overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, s );
    if (e == hipSuccess) e = hipStreamSynchronize( s );
    scratch_free( overflow, s );
    if (e != hipSuccess) { set_error( "hamming_backtrack failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    if (h_over) { set_error( "hamming_backtrack: the 128-entry stack of the reference's benchmark overflowed for %u branches", h_over ); return NVBIO_ERR_UNSUPPORTED; }
    return NVBIO_OK;
}

// scratch layout of nvbio_fm_match_seed_diagonals: tile keys | tile counts | tile offsets | scan temp
struct SeedScratch { SeedTiles tl; uint32_t slots; uint64_t keys_bytes, counts_bytes, scan_bytes, total; };

static nvbio_status seed_scratch_layout(const nvbio_string_set* seeds, SeedScratch* L)
{
    NVB_REQUIRE( seeds != nullptr, "seeds is NULL" );
    const uint32_t spr = seeds->seeds_per_string;
    NVB_REQUIRE( spr > 0, "the string set must enumerate seeds (seeds_per_string > 0)" );
    NVB_REQUIRE( seeds->n % spr == 0, "n must be a multiple of seeds_per_string" );
    L->tl.reads   = seeds->n / spr;
    L->tl.rpt     = spr <= 64u ? 64u / spr : 1u;
    L->tl.n_tiles = (L->tl.reads + L->tl.rpt - 1u) / L->tl.rpt;
    L->slots      = 64u * ((spr + 63u) / 64u);
    L->keys_bytes   = ((uint64_t)L->tl.n_tiles * L->slots * sizeof(uint64_t) + 255u) & ~255ull;
    L->counts_bytes = ((uint64_t)(L->tl.n_tiles + 1u) * sizeof(uint32_t) + 255u) & ~255ull;
    size_t scan = 0;
    if (L->tl.n_tiles)
        NVB_HIP( hipcub::DeviceScan::ExclusiveSum( nullptr, scan, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)L->tl.n_tiles, (hipStream_t)0 ) );
    L->scan_bytes = ((uint64_t)scan + 255u) & ~255ull;
    L->total = L->keys_bytes + 2u * L->counts_bytes + L->scan_bytes + 256u;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_match_seed_diagonals_temp_bytes(const nvbio_string_set* seeds, uint64_t* bytes)
{
    NVB_REQUIRE( bytes != nullptr, "bytes is NULL" );
    SeedScratch L; NVB_CHECK( seed_scratch_layout( seeds, &L ) );
    *bytes = L.total;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_match_seed_diagonals(nvbio_fm_index_t index, const nvbio_string_set* seeds, uint32_t flags, uint32_t read_len,
                                           uint32_t strand, uint64_t* keys_dev, nvbio_uint2* residual_ranges_dev, uint32_t* residual_ids_dev,
                                           uint32_t* counts_dev, void* temp_dev, uint64_t temp_bytes, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    StringSetDev q; NVB_CHECK( make_set( seeds, &q ) );
    NVB_REQUIRE( counts_dev != nullptr, "counts_dev is NULL" );
    NVB_REQUIRE( q.spr > 0, "the string set must enumerate seeds (seeds_per_string > 0)" );
    NVB_REQUIRE( q.intervals == nullptr, "ragged seed sets (seed_intervals_dev) go through nvbio_fm_match_seed_diagonals_both" );
    NVB_REQUIRE( (uint64_t)(q.spr - 1u) * q.interval + q.fixed_len <= read_len, "seeds do not fit the read" );
    NVB_REQUIRE( q.fixed_len > 0, "empty seeds" );
    if (!(idx->text && idx->view.ssa_dev && idx->view.sa_int == 1))
    {
        set_error( "nvbio_fm_match_seed_diagonals needs the full suffix array and the text: build the index with sa_int = 1" );
        return NVBIO_ERR_UNSUPPORTED;
    }
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const bool count = (flags & NVBIO_FM_COUNT_SECTORS) != 0;
    NVB_REQUIRE( !count || ((uintptr_t)counts_dev & 7u) == 0, "counts_dev must be 8-byte aligned with NVBIO_FM_COUNT_SECTORS" );
    NVB_HIP( hipMemsetAsync( counts_dev, 0, (count ? 4 : 2) * sizeof(uint32_t), s ) );
    if (q.n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && residual_ranges_dev && residual_ids_dev, "NULL device pointer" );
    SeedScratch L; NVB_CHECK( seed_scratch_layout( seeds, &L ) );
    uint8_t* temp = (uint8_t*)temp_dev;
    bool own_temp = false;
    if (temp == nullptr)
    {
        if (scratch_alloc( (void**)&temp, L.total, s ) != hipSuccess) { (void)hipGetLastError(); set_error( "seed pass: out of device memory for %llu bytes of scratch", (unsigned long long)L.total ); return NVBIO_ERR_NOMEM; }
        own_temp = true;
    }
    else NVB_REQUIRE( temp_bytes >= L.total, "temp_bytes too small (nvbio_fm_match_seed_diagonals_temp_bytes)" );
    uint8_t* base = (uint8_t*)(((uintptr_t)temp + 255u) & ~(uintptr_t)255u);
    uint64_t* tile_keys    = (uint64_t*)base;
    uint32_t* tile_counts  = (uint32_t*)(base + L.keys_bytes);
    uint32_t* tile_offsets = (uint32_t*)(base + L.keys_bytes + L.counts_bytes);
    void*     scan_temp    = base + L.keys_bytes + 2u * L.counts_bytes;
    const DevIndex f = idx->dev();
    // one wave per tile of whole reads; the grid keeps every wave slot of the chip busy a few times over (flags bits 16..31:
    // workgroups in units of 64, a tuning/testing knob -- results do not depend on it)
    unsigned blocks = (L.tl.n_tiles + 3u) / 4u;
    const unsigned cap = (flags >> 16) ? (flags >> 16) * 64u : 256u * 64u;
    if (blocks > cap) blocks = cap;
    const dim3 grid( blocks ), block( 256 );
    // the pipelined kernel serves the production shape: packed seeds of up to 32 symbols resolved by the direct table's contexts
    const bool pipe = !count && !(flags & NVBIO_FM_NO_PIPELINE) && seeds->symbol_bits != 8 && q.spr <= 64u && q.fixed_len <= 32u && f.dtab != nullptr &&
                      f.ktab != nullptr && !(flags & NVBIO_FM_NO_KMER_TABLE) && f.dctx != 0u && q.fixed_len >= f.dkmer && q.fixed_len - f.dkmer <= f.dctx;
#define NVB_LAUNCH_SD(BITS) \
    if (pipe)  hipLaunchKernelGGL( (fm_seed_pipe_kernel<(BITS == 8 ? 4 : BITS)>), grid, block, 0, s, f, q, L.tl, flags & 0xFFFFu, read_len, strand, tile_keys, tile_counts, \
                                   (uint2*)residual_ranges_dev, residual_ids_dev, (unsigned int*)counts_dev );                                              \
    else if (count) hipLaunchKernelGGL( (fm_seed_tiles_kernel<BITS,true>), grid, block, 0, s, f, q, L.tl, flags & 0xFFFFu, read_len, strand, tile_keys, tile_counts, \
                                   (uint2*)residual_ranges_dev, residual_ids_dev, (unsigned int*)counts_dev, (unsigned long long*)(counts_dev + 2) );       \
    else       hipLaunchKernelGGL( (fm_seed_tiles_kernel<BITS,false>), grid, block, 0, s, f, q, L.tl, flags & 0xFFFFu, read_len, strand, tile_keys, tile_counts, \
                                   (uint2*)residual_ranges_dev, residual_ids_dev, (unsigned int*)counts_dev, (unsigned long long*)nullptr )
    switch (seeds->symbol_bits)
    {
    case 2: NVB_LAUNCH_SD(2); break;
    case 4: NVB_LAUNCH_SD(4); break;
    default: NVB_LAUNCH_SD(8); break;
    }
#undef NVB_LAUNCH_SD
    hipError_t e = hipGetLastError();
    size_t scan_bytes = L.scan_bytes;
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum( scan_temp, scan_bytes, (const uint32_t*)tile_counts, tile_offsets, (int)L.tl.n_tiles, s );
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL( fm_seed_compact_kernel, dim3( grid_for( 4ull * L.tl.n_tiles ) ), block, 0, s, (const uint64_t*)tile_keys, (const uint32_t*)tile_counts,
                            (const uint32_t*)tile_offsets, L.tl.n_tiles, L.slots, keys_dev, (unsigned int*)counts_dev );
        e = hipGetLastError();
    }
    if (own_temp) scratch_free( temp, s );
    if (e != hipSuccess) { set_error( "seed pass failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    return NVBIO_OK;
}

int nvbio_fm_index_is_canonical(nvbio_fm_index_t index)
{
    return index != nullptr && ((FMIndexImpl*)index)->ctab != nullptr ? (int)((FMIndexImpl*)index)->ckmer : 0;
}

// scratch of the two-strand pass: 128 key slots and 2 counts per tile; with NVBIO_FM_DEFER_HEAVY also 128 deferred slots (uint32) and 2 counts per
// tile, the dense list of deferred searches and its length
struct SeedBothScratch { SeedScratch L; uint64_t defer_bytes, list_bytes; };
static nvbio_status seed_both_layout(const nvbio_string_set* seeds, SeedScratch* L, SeedBothScratch* B = nullptr)
{
    NVB_CHECK( seed_scratch_layout( seeds, L ) );
    NVB_REQUIRE( seeds->seeds_per_string <= 64u, "the two-strand seed pass takes at most 64 seeds per read" );
    L->slots        = 128u;
    L->keys_bytes   = ((uint64_t)L->tl.n_tiles * 128u * sizeof(uint64_t) + 255u) & ~255ull;
    L->counts_bytes = ((uint64_t)(L->tl.n_tiles + 1u) * sizeof(uint32_t) + 255u) & ~255ull;
    size_t scan = 0;
    if (L->tl.n_tiles)
        NVB_HIP( hipcub::DeviceScan::ExclusiveSum( nullptr, scan, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)L->tl.n_tiles, (hipStream_t)0 ) );
    L->scan_bytes = ((uint64_t)scan + 255u) & ~255ull;
    const uint64_t defer_bytes = ((uint64_t)L->tl.n_tiles * 128u * sizeof(uint32_t) + 255u) & ~255ull;
    const uint64_t list_bytes  = ((uint64_t)L->tl.n_tiles * 128u * sizeof(uint32_t) + 255u) & ~255ull;
    // keys | counts | offsets | scan temp | deferred slots | deferred counts | deferred offsets | dense deferred list | its length
    L->total = L->keys_bytes + 2u * L->counts_bytes + L->scan_bytes + defer_bytes + 2u * L->counts_bytes + list_bytes + 256u + 256u;
    if (B) { B->L = *L; B->defer_bytes = defer_bytes; B->list_bytes = list_bytes; }
    return NVBIO_OK;
}

nvbio_status nvbio_fm_match_seed_diagonals_both_temp_bytes(const nvbio_string_set* seeds, uint64_t* bytes)
{
    NVB_REQUIRE( bytes != nullptr, "bytes is NULL" );
    SeedScratch L; NVB_CHECK( seed_both_layout( seeds, &L ) );
    *bytes = L.total;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_match_seed_diagonals_both_keys_capacity(const nvbio_string_set* seeds, uint64_t* n_keys)
{
    NVB_REQUIRE( n_keys != nullptr, "n_keys is NULL" );
    SeedScratch L; NVB_CHECK( seed_both_layout( seeds, &L ) );
    // a tile of 64 / seeds_per_string reads can leave up to 64 keys per strand, and (NVBIO_FM_DEFER_HEAVY) one more per deferred search
    *n_keys = 256ull * L.tl.n_tiles;
    return NVBIO_OK;
}

nvbio_status nvbio_fm_match_seed_diagonals_both(nvbio_fm_index_t index, const nvbio_string_set* seeds, uint32_t flags, uint32_t read_len,
                                                uint64_t* keys_dev, nvbio_uint2* residual_ranges_dev, uint32_t* residual_ids_dev,
                                                uint32_t residual_capacity, uint32_t* counts_dev, void* temp_dev, uint64_t temp_bytes,
                                                void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    StringSetDev q; NVB_CHECK( make_set( seeds, &q ) );
    NVB_REQUIRE( counts_dev != nullptr, "counts_dev is NULL" );
    if (idx->ctab == nullptr)
    {
        set_error( "nvbio_fm_match_seed_diagonals_both needs the canonical table: build the index with NVBIO_FM_TABLE_CANONICAL" );
        return NVBIO_ERR_UNSUPPORTED;
    }
    NVB_REQUIRE( q.spr > 0 && q.spr <= 64u, "the string set must enumerate 1..64 seeds per read (seeds_per_string)" );
    NVB_REQUIRE( seeds->symbol_bits == 2 || seeds->symbol_bits == 4, "packed seeds (2 or 4 bits per symbol)" );
    NVB_REQUIRE( q.fixed_len >= idx->ckmer && q.fixed_len - idx->ckmer <= CTAB_FLANK, "seed length outside [kmer_len, kmer_len + 7]" );
    NVB_REQUIRE( q.intervals != nullptr || (uint64_t)(q.spr - 1u) * q.interval + q.fixed_len <= read_len, "seeds do not fit the read" );
    NVB_REQUIRE( residual_capacity >= q.n, "residual_capacity must be at least the number of seeds" );
    NVB_REQUIRE( q.n < (1u << 31), "at most 2^31 - 1 seeds per call (bit 31 of a seed id carries the strand)" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const bool count = (flags & NVBIO_FM_COUNT_SECTORS) != 0;
    const bool defer = (flags & NVBIO_FM_DEFER_HEAVY) != 0 && !count;
    NVB_REQUIRE( !count || ((uintptr_t)counts_dev & 7u) == 0, "counts_dev must be 8-byte aligned with NVBIO_FM_COUNT_SECTORS" );
    NVB_HIP( hipMemsetAsync( counts_dev, 0, (count ? 6 : 4) * sizeof(uint32_t), s ) );
    if (q.n == 0) return NVBIO_OK;
    NVB_REQUIRE( keys_dev && residual_ranges_dev && residual_ids_dev, "NULL device pointer" );
    SeedScratch L; SeedBothScratch B; NVB_CHECK( seed_both_layout( seeds, &L, &B ) );
    uint8_t* temp = (uint8_t*)temp_dev;
    bool own_temp = false;
    if (temp == nullptr)
    {
        if (scratch_alloc( (void**)&temp, L.total, s ) != hipSuccess) { (void)hipGetLastError(); set_error( "seed pass: out of device memory for %llu bytes of scratch", (unsigned long long)L.total ); return NVBIO_ERR_NOMEM; }
        own_temp = true;
    }
    else NVB_REQUIRE( temp_bytes >= L.total, "temp_bytes too small (nvbio_fm_match_seed_diagonals_both_temp_bytes)" );
    uint8_t* base = (uint8_t*)(((uintptr_t)temp + 255u) & ~(uintptr_t)255u);
    uint64_t* tile_keys    = (uint64_t*)base;
    uint32_t* tile_counts  = (uint32_t*)(base + L.keys_bytes);
    uint32_t* tile_offsets = (uint32_t*)(base + L.keys_bytes + L.counts_bytes);
    void*     scan_temp    = base + L.keys_bytes + 2u * L.counts_bytes;
    uint8_t*  dbase        = base + L.keys_bytes + 2u * L.counts_bytes + L.scan_bytes;
    uint32_t* tile_defer    = (uint32_t*)dbase;
    uint32_t* defer_counts  = (uint32_t*)(dbase + B.defer_bytes);
    uint32_t* defer_offsets = (uint32_t*)(dbase + B.defer_bytes + L.counts_bytes);
    uint32_t* defer_list    = (uint32_t*)(dbase + B.defer_bytes + 2u * L.counts_bytes);
    uint32_t* defer_n       = (uint32_t*)(dbase + B.defer_bytes + 2u * L.counts_bytes + B.list_bytes);
    const DevIndex f = idx->dev();
    unsigned blocks = (L.tl.n_tiles + 3u) / 4u;
    const unsigned cap = (flags >> 16) ? (flags >> 16) * 64u : 256u * 64u;
    if (blocks > cap) blocks = cap;
    const dim3 grid( blocks ), block( 256 );
    // flags bits 8..11: a seed with up to that many hits on a strand leaves them all as keys (0/1: only one-hit seeds do)
    uint32_t inline_max = (flags >> 8) & 15u;
    inline_max = inline_max < 1u ? 1u : (inline_max > CTAB_INLINE ? CTAB_INLINE : inline_max);
#define NVB_LAUNCH_SBD(BITS, CNT, W, D) hipLaunchKernelGGL( (fm_seed_both_kernel<BITS,CNT,W,D>), grid, block, 0, s, f, q, L.tl, read_len, inline_max, tile_keys, tile_counts, \
                                    (uint2*)residual_ranges_dev, residual_ids_dev, residual_capacity, (unsigned int*)counts_dev,                         \
                                    CNT ? (unsigned long long*)(counts_dev + 4) : (unsigned long long*)nullptr, tile_defer, defer_counts )
#define NVB_LAUNCH_SBW(BITS, W) do { if (count) NVB_LAUNCH_SBD( BITS, true, W, false ); else if (defer) NVB_LAUNCH_SBD( BITS, false, W, true ); \
                                     else NVB_LAUNCH_SBD( BITS, false, W, false ); } while (0)
#define NVB_LAUNCH_SB(BITS) do { if (idx->cwide) NVB_LAUNCH_SBW( BITS, true ); else NVB_LAUNCH_SBW( BITS, false ); } while (0)
    if (seeds->symbol_bits == 2) NVB_LAUNCH_SB( 2 ); else NVB_LAUNCH_SB( 4 );
#undef NVB_LAUNCH_SB
#undef NVB_LAUNCH_SBW
#undef NVB_LAUNCH_SBD
    hipError_t e = hipGetLastError();
    size_t scan_bytes = L.scan_bytes;
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum( scan_temp, scan_bytes, (const uint32_t*)tile_counts, tile_offsets, (int)L.tl.n_tiles, s );
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL( fm_seed_compact_kernel, dim3( grid_for( 4ull * L.tl.n_tiles ) ), block, 0, s, (const uint64_t*)tile_keys, (const uint32_t*)tile_counts,
                            (const uint32_t*)tile_offsets, L.tl.n_tiles, 128u, keys_dev, (unsigned int*)counts_dev );
        e = hipGetLastError();
    }
    if (e == hipSuccess && defer)
    {
        // the deferred searches: their slots made dense, then a launch of their own that appends to the keys and the residual lists
        scan_bytes = L.scan_bytes;
        e = hipcub::DeviceScan::ExclusiveSum( scan_temp, scan_bytes, (const uint32_t*)defer_counts, defer_offsets, (int)L.tl.n_tiles, s );
        if (e == hipSuccess)
        {
            hipLaunchKernelGGL( fm_seed_defer_compact_kernel, dim3( grid_for( L.tl.n_tiles ) ), block, 0, s, (const uint32_t*)tile_defer, (const uint32_t*)defer_counts,
                                (const uint32_t*)defer_offsets, L.tl.n_tiles, defer_list, defer_n );
            const uint64_t max_chunks = (128ull * L.tl.n_tiles + 256u * HEAVY_PER_LANE - 1u) / (256u * HEAVY_PER_LANE);
            const dim3 hgrid( (unsigned)(max_chunks < 4096u ? (max_chunks ? max_chunks : 1u) : 4096u) );
            if (seeds->symbol_bits == 2)
                hipLaunchKernelGGL( (fm_seed_heavy_kernel<2>), hgrid, block, 0, s, f, q, read_len, (const uint32_t*)defer_list, (const uint32_t*)defer_n, keys_dev,
                                    (uint2*)residual_ranges_dev, residual_ids_dev, residual_capacity, (unsigned int*)counts_dev );
            else
                hipLaunchKernelGGL( (fm_seed_heavy_kernel<4>), hgrid, block, 0, s, f, q, read_len, (const uint32_t*)defer_list, (const uint32_t*)defer_n, keys_dev,
                                    (uint2*)residual_ranges_dev, residual_ids_dev, residual_capacity, (unsigned int*)counts_dev );
            e = hipGetLastError();
        }
    }
    if (own_temp) scratch_free( temp, s );
    if (e != hipSuccess) { set_error( "seed pass failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    return NVBIO_OK;
}

nvbio_status nvbio_fm_filter_locate_direct(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                           const uint8_t* direct_dev, uint32_t n_queries, uint64_t begin, uint64_t end,
                                           nvbio_uint2* hits_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (end <= begin) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev && slots_dev && hits_dev && direct_dev, "NULL device pointer" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    NVB_REQUIRE( idx->view.ssa_dev, "index has no sampled suffix array" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipLaunchKernelGGL( fm_filter_locate_kernel<false>, dim3( grid_for( (end - begin + FILTER_TILE - 1u) / FILTER_TILE * 256u ) ), dim3(256), 0, (hipStream_t)stream,
                        idx->dev(), (const uint2*)ranges_dev, slots_dev, n_queries, begin, end, (uint2*)hits_dev, direct_dev, DiagSpec{}, (uint64_t*)nullptr );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

static nvbio_status filter_locate_diagonals(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                            const uint8_t* direct_dev, uint32_t n_queries, uint64_t begin, uint64_t end,
                                            uint32_t seeds_per_read, uint32_t seed_interval, uint32_t seed_len, uint32_t read_len,
                                            uint32_t strand, const uint32_t* query_ids_dev, const uint32_t* read_offsets_dev,
                                            const uint32_t* seed_intervals_dev, uint64_t* keys_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr, "index is NULL" );
    if (end <= begin) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev && slots_dev && keys_dev, "NULL device pointer" );
    NVB_REQUIRE( seeds_per_read > 0, "seeds_per_read must be positive" );
    NVB_REQUIRE( (read_offsets_dev == nullptr) == (seed_intervals_dev == nullptr), "ragged reads need both read_offsets_dev and seed_intervals_dev" );
    NVB_REQUIRE( read_offsets_dev != nullptr || (uint64_t)(seeds_per_read - 1u) * seed_interval + seed_len <= read_len, "seeds do not fit the read" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    NVB_REQUIRE( idx->view.ssa_dev, "index has no sampled suffix array" );
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    const DiagSpec ds = { seeds_per_read, seed_interval, seed_len, read_len, strand, query_ids_dev, read_offsets_dev, seed_intervals_dev };
    hipLaunchKernelGGL( fm_filter_locate_kernel<true>, dim3( grid_for( (end - begin + FILTER_TILE - 1u) / FILTER_TILE * 256u ) ), dim3(256), 0, (hipStream_t)stream,
                        idx->dev(), (const uint2*)ranges_dev, slots_dev, n_queries, begin, end, (uint2*)nullptr, direct_dev, ds, keys_dev );
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_fm_filter_locate_diagonals(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                              const uint8_t* direct_dev, uint32_t n_queries, uint64_t begin, uint64_t end,
                                              uint32_t seeds_per_read, uint32_t seed_interval, uint32_t seed_len, uint32_t read_len,
                                              uint32_t strand, const uint32_t* query_ids_dev, uint64_t* keys_dev, void* stream)
{
    return filter_locate_diagonals( index, ranges_dev, slots_dev, direct_dev, n_queries, begin, end, seeds_per_read, seed_interval, seed_len, read_len,
                                    strand, query_ids_dev, nullptr, nullptr, keys_dev, stream );
}

nvbio_status nvbio_fm_filter_locate_diagonals_ragged(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint64_t* slots_dev,
                                                     const uint8_t* direct_dev, uint32_t n_queries, uint64_t begin, uint64_t end,
                                                     uint32_t seeds_per_read, uint32_t seed_len, const uint32_t* read_offsets_dev,
                                                     const uint32_t* seed_intervals_dev, uint32_t strand, const uint32_t* query_ids_dev,
                                                     uint64_t* keys_dev, void* stream)
{
    NVB_REQUIRE( read_offsets_dev && seed_intervals_dev, "read_offsets_dev and seed_intervals_dev are required" );
    return filter_locate_diagonals( index, ranges_dev, slots_dev, direct_dev, n_queries, begin, end, seeds_per_read, 0u, seed_len, 0u,
                                    strand, query_ids_dev, read_offsets_dev, seed_intervals_dev, keys_dev, stream );
}



nvbio_status nvbio_fm_residual_diagonals(nvbio_fm_index_t index, const nvbio_uint2* ranges_dev, const uint32_t* ids_dev, uint32_t n, uint32_t cap,
                                         uint32_t seeds_per_read, uint32_t seed_interval, uint32_t seed_len, uint32_t read_len,
                                         const uint32_t* read_offsets_dev, const uint32_t* seed_intervals_dev,
                                         uint64_t* keys_dev, uint32_t* n_keys_dev, void* stream)
{
    NVB_REQUIRE( index != nullptr && n_keys_dev != nullptr, "NULL argument" );
    FMIndexImpl* idx = (FMIndexImpl*)index;
    DeviceGuard g( idx->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    NVB_HIP( hipMemsetAsync( n_keys_dev, 0, sizeof(uint32_t), s ) );
    if (n == 0) return NVBIO_OK;
    NVB_REQUIRE( ranges_dev && ids_dev && keys_dev, "NULL device pointer" );
    NVB_REQUIRE( cap >= 1u && cap <= 64u, "cap must be in 1..64" );
    NVB_REQUIRE( (uint64_t)n * cap < (1ull << 31), "n * cap must stay below 2^31" );
    NVB_REQUIRE( seeds_per_read > 0, "seeds_per_read must be positive" );
    NVB_REQUIRE( (read_offsets_dev == nullptr) == (seed_intervals_dev == nullptr), "ragged reads need both read_offsets_dev and seed_intervals_dev" );
    if (!(idx->view.ssa_dev && idx->view.sa_int == 1))
    {
        set_error( "nvbio_fm_residual_diagonals needs the full suffix array: build the index with sa_int = 1" );
        return NVBIO_ERR_UNSUPPORTED;
    }
    // scratch: sorted ids | sorted ranges | sort temp
    const uint64_t ids_bytes = ((uint64_t)n * 4u + 255u) & ~255ull, rng_bytes = ((uint64_t)n * 8u + 255u) & ~255ull;
    size_t sort_bytes = 0;
    NVB_HIP( hipcub::DeviceRadixSort::SortPairs( nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n, 0, 32, s ) );
    uint8_t* aux = nullptr;
    if (scratch_alloc( (void**)&aux, ids_bytes + rng_bytes + sort_bytes + 256u, s ) != hipSuccess)
    {
        (void)hipGetLastError();
        set_error( "residual diagonals: out of device memory" );
        return NVBIO_ERR_NOMEM;
    }
    uint32_t* ids_s = (uint32_t*)aux;
    uint64_t* rng_s = (uint64_t*)(aux + ids_bytes);
    void*     tmp   = aux + ids_bytes + rng_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs( tmp, sort_bytes, ids_dev, ids_s, (const uint64_t*)ranges_dev, rng_s, (int)n, 0, 32, s );
    if (e == hipSuccess)
    {
        const DiagSpec ds = { seeds_per_read, seed_interval, seed_len, read_len, 0u, nullptr, read_offsets_dev, seed_intervals_dev };
        hipLaunchKernelGGL( residual_locate_kernel, dim3( grid_for( n ) ), dim3(256), 0, s, idx->dev(), (const uint2*)rng_s, (const uint32_t*)ids_s, n, cap, ds,
                            keys_dev, (unsigned int*)n_keys_dev );
        e = hipGetLastError();
    }
    scratch_free( aux, s );
    if (e != hipSuccess) { set_error( "residual diagonals failed: %s", hipGetErrorString( e ) ); return NVBIO_ERR_HIP; }
    return NVBIO_OK;
}

nvbio_status nvbio_seed_hits_map_approx(nvbio_fm_index_t index, nvbio_fm_index_t reverse_index, const void* reads_dev, uint32_t read_bits,
                                        const uint32_t* read_queue_dev, uint32_t n_reads, const nvbio_seed_hits_params* p, nvbio_uint2* deques_dev,
                                        uint32_t* sizes_dev, uint8_t* reseed_dev, void* stream)
{
    NVB_REQUIRE( index && reverse_index && p, "NULL argument" );
    if (n_reads == 0) return NVBIO_OK;
    NVB_REQUIRE( reads_dev && deques_dev && sizes_dev, "NULL device pointer" );
    NVB_REQUIRE( read_bits == 2 || read_bits == 4 || read_bits == 8, "read_bits must be 2, 4 or 8" );
    NVB_REQUIRE( p->max_hits > 0 && p->seeds_per_read > 0 && p->seed_len > 0, "max_hits, seeds_per_read and seed_len must be positive" );
    NVB_REQUIRE( (uint64_t)p->first_offset + (uint64_t)(p->seeds_per_read - 1u) * p->seed_interval + p->seed_len <= p->read_len, "seeds do not fit the read" );
    NVB_REQUIRE( p->read_len < 1024u, "SeedHit keeps the seed position in 10 bits (seed_hit.h:217)" );
    FMIndexImpl *fi = (FMIndexImpl*)index, *ri = (FMIndexImpl*)reverse_index;
    NVB_REQUIRE( fi->device == ri->device, "both indices must live on one device" );
    // up to 3 (len2 - len1) + 1 ranges per search, four searches per seed
    const uint64_t worst = 4ull * p->seeds_per_read * (3ull * ((p->seed_len + 1u) / 2u) + 1ull);
    const uint32_t cap = (uint32_t)((worst < p->max_hits ? worst : p->max_hits) + 1u);
    DeviceGuard g( fi->device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    DevIndex f = fi->dev(), rf = ri->dev();
    const dim3 grid( grid_for( n_reads, 128 ) ), block( 128 );
#define NVB_LAUNCH_MA(BITS) hipLaunchKernelGGL( (fm_map_approx_kernel<BITS>), grid, block, 0, (hipStream_t)stream, f, rf, reads_dev, read_queue_dev, n_reads, \
                                                p->seeds_per_read, p->first_offset, p->seed_interval, p->seed_len, p->read_len, p->max_hits, p->rep_seeds, cap,   \
                                                (uint2*)deques_dev, sizes_dev, reseed_dev )
    switch (read_bits) { case 2: NVB_LAUNCH_MA(2); break; case 4: NVB_LAUNCH_MA(4); break; default: NVB_LAUNCH_MA(8); break; }
#undef NVB_LAUNCH_MA
    NVB_HIP( hipGetLastError() );
    return NVBIO_OK;
}

nvbio_status nvbio_seed_hits_approx_capacity(uint32_t seeds_per_read, uint32_t seed_len, uint32_t max_hits, uint32_t* capacity)
{
    NVB_REQUIRE( capacity != nullptr, "capacity is NULL" );
    const uint64_t worst = 4ull * seeds_per_read * (3ull * ((seed_len + 1u) / 2u) + 1ull);
    *capacity = (uint32_t)((worst < max_hits ? worst : max_hits) + 1u);
    return NVBIO_OK;
}

} // extern "C"
