// ---------------------------------------------------------------------------------------------
// fm_canon_inl.h -- the two-strand seed pass over a CANONICAL k-mer table (included by fm_index.hip, inside namespace nvbio_amd).
//
// What is computed: for every seed window S of a read, match() of S (the forward strand) AND match() of its reverse complement
// (nvBowtie searches both, mapping_inl.h:288-414; the fmmap-shaped pipeline runs the seed pass once per strand with
// NVBIO_FM_SCAN_FORWARD | NVBIO_FM_COMPLEMENT), each followed by locate() of a search that ends on one row -- the outputs of two
// launches of the per-strand pass (fm_seed_pipe_kernel), from ONE table gather per window instead of one per window and strand.
//
// How.  Let k = ckmer (odd), r = len - k, W = the last k symbols of S and A = its first r symbols (S = A W).  An occurrence of W in the
// text at p is a forward hit at p - r iff text[p-r, p) = A; an occurrence of rc(W) at p is a reverse-strand hit at p iff
// text[p+k, p+k+r) = rc(A) (rc(S) = rc(W) rc(A)).  W and rc(W) share ONE table entry -- the entry of whichever of the two has its
// middle symbol in {A, C} (k is odd: the middle symbols of W and rc(W) are complements, so exactly one qualifies; dropping that
// symbol's high bit gives a dense index of 2^(2k-1) entries: 64 GiB at k = 17 where the per-strand table takes 128) -- and that
// entry lists the occurrences of BOTH orientations, each with the 7 text symbols before and the 7 after it, taken in the
// orientation of the canonical k-mer C:
//   entry (lo, hi):  hi <  MARK, lo >  hi : no occurrence of either orientation
//                    hi <  MARK, lo <= hi : more than 8 occurrences: the search falls back to rank steps (per strand)
//                    hi >= MARK, lo <  MARK : ONE occurrence: lo = text position p of the k-mer,
//                                             hi - MARK = o << 28 | before << 14 | after, where o = 0 if text[p, p+k) = C and 1 if it is
//                                             rc(C); before[t] (bits 2(6-t) of the field) = the symbol t+1 places before C in C's
//                                             orientation: text[p-1-t] (o = 0) or 3 - text[p+k+t] (o = 1); after[t] likewise the symbol
//                                             t+1 places after C: text[p+k+t] (o = 0) or 3 - text[p-1-t] (o = 1); symbols outside the
//                                             text read as 0 and are never trusted (the position checks below exclude them)
//                    hi >= MARK, lo >= MARK : m = hi - MARK in 2..8 occurrences, rows (position, MARK | payload) at cside + 4 (lo - MARK):
//                                             2..4 rows in 32 bytes, 5..8 in 64 (aligned): the occurrences of C in SA order, then those of rc(C)
// A window whose last k symbols are W looks up C = W (qo = 0) or C = rc(W) (qo = 1).  Row (p, o): it is an occurrence of W if o = qo and of
// rc(W) otherwise; qo = 0 compares `before` with A reversed (the seed's remaining symbols in scan order), qo = 1 compares `after` with their
// complement (rc(A) follows C = rc(W)).  A forward hit needs p >= r, a reverse one p + len <= length.
//
// Entries of 16 bytes (NVBIO_FM_TABLE_CANONICAL_WIDE, 128 GiB at k = 17): two slots.  Slot 0 is as above; a k-mer with exactly TWO
// occurrences has the second row in slot 1 (hi >= MARK there; (0, 0) otherwise), and groups begin at three occurrences (3..4 rows in 32
// bytes, 5..8 in 64).  On a 3 Gbp text one seed window in four has a k-mer that occurs once more somewhere, in either orientation:
// 1.25 gathered sectors per window with 8-byte entries, 1.04 with 16-byte ones.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t CTAB_FLANK    = 7u;            // symbols stored on each side of an occurrence: seeds of up to k + 7 symbols
constexpr uint32_t CTAB_ROWS_MAX = 8u;
constexpr uint32_t CTAB_INLINE   = 4u;            // a seed with up to this many hits on a strand can leave them all as keys (inline_max)

__device__ __forceinline__ uint64_t canon_revcomp(const uint64_t key, const uint32_t k)
{
    return reverse_symbols( key, k ) ^ ((1ull << (2u * k)) - 1ull);
}
__device__ __forceinline__ uint64_t canon_index(const uint64_t ckey, const uint32_t k)      // ckey: bit k (middle symbol's high bit) is 0
{
    return ((ckey >> (k + 1u)) << k) | (ckey & ((1ull << k) - 1ull));
}
__device__ __forceinline__ uint64_t canon_key(const uint64_t idx, const uint32_t k)
{
    return ((idx >> k) << (k + 1u)) | (idx & ((1ull << k) - 1ull));
}
__device__ __forceinline__ uint32_t text_symbol(const uint32_t* __restrict__ text, const uint32_t i)
{
    return (text[i >> 4] >> (30u - 2u * (i & 15u))) & 3u;
}

// the SA range of the k-mer `key` (scan order) from the table of (k-1)-mers: one search step, as fm_ktab_level_kernel
__device__ __forceinline__ uint2 canon_range(const DevIndex& f, const uint64_t key)
{
    const uint2 r = f.ktab[key >> 2];
    uint32_t x = r.x, y = r.y, nb = 0;
    if (x <= y) search_step<false>( f, x, y, (uint32_t)(key & 3u), nb );
    return make_uint2( x, y );
}
__device__ __forceinline__ uint32_t range_rows(const uint2 r) { return r.x <= r.y ? r.y - r.x + 1u : 0u; }

// 0: empty / heavy   1: in the entry itself (one occurrence; two with 16-byte entries)   2: up to 4 rows (4 slots)   3: 5..8 (8 slots)
__device__ __forceinline__ uint32_t canon_class(const uint32_t m, const bool wide)
{
    if (m == 1u || (wide && m == 2u)) return 1u;
    return m >= 2u && m <= 4u ? 2u : (m >= 5u && m <= CTAB_ROWS_MAX ? 3u : 0u);
}

// the row of an occurrence: o = 0: text[p, p+k) is the canonical k-mer; o = 1: its reverse complement
__device__ __forceinline__ uint2 canon_row(const uint32_t* __restrict__ text, const uint32_t length, const uint32_t k, const uint32_t p, const uint32_t o)
{
    uint32_t before = 0, after = 0;
    #pragma unroll
    for (uint32_t t = 0; t < CTAB_FLANK; ++t)
    {
        const bool has_l = p >= t + 1u, has_r = (uint64_t)p + k + t < length;
        const uint32_t l = has_l ? text_symbol( text, p - 1u - t ) : 0u;
        const uint32_t g = has_r ? text_symbol( text, p + k + t ) : 0u;
        const uint32_t b = o ? (has_r ? 3u - g : 0u) : l;
        const uint32_t a = o ? (has_l ? 3u - l : 0u) : g;
        before |= b << (2u * (CTAB_FLANK - 1u - t));
        after  |= a << (2u * (CTAB_FLANK - 1u - t));
    }
    return make_uint2( p, DTAB_MARK | (o << 28) | (before << 14) | after );
}

// pass 1: tab[idx] = SA range of the reverse complement of canonical k-mer idx (kept for pass 2); per-tile group counts
__global__ void __launch_bounds__(256)
fm_ctab_count_kernel(const DevIndex f, const uint32_t k, const uint32_t wide, uint2* __restrict__ tab, const uint64_t n, const uint32_t n_tiles,
                     uint32_t* __restrict__ cnt_small, uint32_t* __restrict__ cnt_large)
{
    const uint32_t stride = wide ? 2u : 1u;                      // slots per entry
    __shared__ uint32_t s_a[4], s_b[4];
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    {
        const uint64_t e0 = (uint64_t)tile * DT_TILE + threadIdx.x * 4u;
        uint32_t a = 0, b = 0;
        #pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
            if (e0 + j < n)
            {
                const uint64_t key = canon_key( e0 + j, k );
                const uint2 r1 = canon_range( f, key ), r2 = canon_range( f, canon_revcomp( key, k ) );
                tab[(e0 + j) * stride] = r2;
                const uint32_t c = canon_class( range_rows( r1 ) + range_rows( r2 ), wide != 0u );
                a += (c == 2u); b += (c == 3u);
            }
        a = wave_inclusive_sum( a ); b = wave_inclusive_sum( b );
        if ((threadIdx.x & 63u) == 63u) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
        __syncthreads();
        if (threadIdx.x == 0) { cnt_small[tile] = s_a[0] + s_a[1] + s_a[2] + s_a[3]; cnt_large[tile] = s_b[0] + s_b[1] + s_b[2] + s_b[3]; }
        __syncthreads();
    }
}

// pass 2: the entries and the groups; large groups occupy side slots [0, 8 tot_large), small ones follow
__global__ void __launch_bounds__(256)
fm_ctab_fill_kernel(const DevIndex f, const uint32_t k, const uint32_t wide, uint2* __restrict__ tab, const uint64_t n, const uint32_t n_tiles,
                    const uint32_t* __restrict__ off_small, const uint32_t* __restrict__ off_large, const uint32_t tot_large, uint2* __restrict__ side)
{
    __shared__ uint32_t s_a[4], s_b[4];
    const uint32_t stride = wide ? 2u : 1u;
    auto position = [&](const uint32_t row) -> uint32_t { const uint32_t sv = f.ssa[row]; return sv == 0xFFFFFFFFu ? f.length : sv; };
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    {
        const uint64_t e0 = (uint64_t)tile * DT_TILE + threadIdx.x * 4u;
        uint2 r1[4], r2[4]; uint32_t c[4];
        uint32_t a = 0, b = 0;
        #pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
        {
            c[j] = 0u; r1[j] = r2[j] = make_uint2( 1u, 0u );
            if (e0 + j < n)
            {
                r1[j] = canon_range( f, canon_key( e0 + j, k ) ); r2[j] = tab[(e0 + j) * stride];
                c[j] = canon_class( range_rows( r1[j] ) + range_rows( r2[j] ), wide != 0u );
                a += (c[j] == 2u); b += (c[j] == 3u);
            }
        }
        const uint32_t ia = wave_inclusive_sum( a ), ib = wave_inclusive_sum( b );
        if ((threadIdx.x & 63u) == 63u) { s_a[threadIdx.x >> 6] = ia; s_b[threadIdx.x >> 6] = ib; }
        __syncthreads();
        uint32_t ga = ia - a, gb = ib - b;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { ga += s_a[w]; gb += s_b[w]; }
        ga += off_small[tile]; gb += off_large[tile];
        __syncthreads();
        #pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
        {
            if (e0 + j >= n) continue;
            const uint32_t n1 = range_rows( r1[j] ), n2 = range_rows( r2[j] ), m = n1 + n2;
            uint2* ent = tab + (e0 + j) * stride;
            if (wide) ent[1] = make_uint2( 0u, 0u );                         // no second row
            auto occurrence = [&](const uint32_t s) -> uint2 {               // row s of the entry: the occurrences of C in SA order, then those of rc(C)
                return s < n1 ? canon_row( f.text, f.length, k, position( r1[j].x + s ), 0u )
                              : canon_row( f.text, f.length, k, position( r2[j].x + (s - n1) ), 1u );
            };
            if (c[j] == 1u)
            {
                ent[0] = occurrence( 0u );
                if (m == 2u) ent[1] = occurrence( 1u );
            }
            else if (c[j] >= 2u)
            {
                const uint32_t slots = c[j] == 2u ? 4u : 8u;
                const uint32_t idx4  = c[j] == 2u ? 2u * tot_large + ga++ : 2u * gb++;
                uint2* g = side + 4ull * idx4;
                for (uint32_t s = 0; s < slots; ++s) g[s] = s < m ? occurrence( s ) : make_uint2( 0u, DTAB_MARK );
                ent[0] = make_uint2( DTAB_MARK | idx4, DTAB_MARK | m );
            }
            else ent[0] = m ? make_uint2( 0u, 1u ) : make_uint2( 1u, 0u );             // heavy : empty
        }
    }
}

// the canonical table of k-mers from the plain table of (k-1)-mers the handle keeps (idx->ktab)
static nvbio_status build_canonical_table(FMIndexImpl* idx, const uint32_t k, const bool wide, hipStream_t stream)
{
    const uint32_t length = idx->view.length;
    if ((uint64_t)length + 2u > DTAB_MARK) { set_error( "canonical table: the text is too long for the position marker" ); return NVBIO_ERR_UNSUPPORTED; }
    const uint64_t entries = 1ull << (2u * k - 1u);
    const uint32_t n_tiles = (uint32_t)((entries + DT_TILE - 1u) / DT_TILE);
    const dim3 grid( n_tiles < 256u * 64u ? n_tiles : 256u * 64u ), block( 256 );
    uint2* tab = nullptr; uint32_t* cnt = nullptr; void* temp = nullptr; uint2* side = nullptr;
    size_t temp_bytes = 0;
    const uint64_t tab_bytes = entries * sizeof(uint2) * (wide ? 2u : 1u);
    if (hipMalloc( (void**)&tab, tab_bytes ) != hipSuccess) { (void)hipGetLastError(); set_error( "canonical table: out of device memory" ); return NVBIO_ERR_NOMEM; }
    if (hipMalloc( (void**)&cnt, 4ull * n_tiles * sizeof(uint32_t) ) != hipSuccess) { (void)hipGetLastError(); (void)hipFree( tab ); set_error( "canonical table: out of device memory" ); return NVBIO_ERR_NOMEM; }
    uint32_t *cs = cnt, *cl = cnt + n_tiles, *os = cnt + 2ull * n_tiles, *ol = cnt + 3ull * n_tiles;
    const DevIndex f = idx->dev();
    hipLaunchKernelGGL( fm_ctab_count_kernel, grid, block, 0, stream, f, k, wide ? 1u : 0u, tab, entries, n_tiles, cs, cl );
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum( nullptr, temp_bytes, cs, os, (int)n_tiles, stream );
    if (e == hipSuccess) e = hipMalloc( &temp, temp_bytes ? temp_bytes : 16 );
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum( temp, temp_bytes, cs, os, (int)n_tiles, stream );
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum( temp, temp_bytes, cl, ol, (int)n_tiles, stream );
    uint32_t last[4] = { 0, 0, 0, 0 };
    if (e == hipSuccess) e = hipMemcpyAsync( &last[0], cs + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
    if (e == hipSuccess) e = hipMemcpyAsync( &last[1], os + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
    if (e == hipSuccess) e = hipMemcpyAsync( &last[2], cl + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
    if (e == hipSuccess) e = hipMemcpyAsync( &last[3], ol + n_tiles - 1u, 4, hipMemcpyDeviceToHost, stream );
    if (e == hipSuccess) e = hipStreamSynchronize( stream );
    nvbio_status st = NVBIO_OK;
    if (e != hipSuccess) { (void)hipGetLastError(); set_error( "canonical table: counting pass failed: %s", hipGetErrorString( e ) ); st = NVBIO_ERR_HIP; }
    const uint32_t tot_small = last[0] + last[1], tot_large = last[2] + last[3];
    const uint64_t units = 2ull * tot_large + tot_small;                      // groups in units of 4 slots (32 bytes)
    if (st == NVBIO_OK && units >= (1ull << 30)) { set_error( "canonical table: too many groups" ); st = NVBIO_ERR_UNSUPPORTED; }
    if (st == NVBIO_OK && hipMalloc( (void**)&side, (units ? units : 1u) * 32ull ) != hipSuccess) { (void)hipGetLastError(); set_error( "canonical table: out of device memory" ); st = NVBIO_ERR_NOMEM; }
    if (st == NVBIO_OK)
    {
        hipLaunchKernelGGL( fm_ctab_fill_kernel, grid, block, 0, stream, f, k, wide ? 1u : 0u, tab, entries, n_tiles, (const uint32_t*)os, (const uint32_t*)ol, tot_large, side );
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize( stream ) != hipSuccess) { set_error( "canonical table: fill pass failed" ); st = NVBIO_ERR_HIP; }
    }
    (void)hipFree( cnt );
    if (temp) (void)hipFree( temp );
    if (st != NVBIO_OK) { (void)hipFree( tab ); if (side) (void)hipFree( side ); return st; }
    idx->ctab = tab; idx->cside = side; idx->ckmer = k; idx->cwide = wide ? 1u : 0u;
    idx->owned_bytes += tab_bytes + (units ? units : 1u) * 32ull;
    return NVBIO_OK;
}

// ---------------------------------------------------------------------------------------------
// the seed pass of both strands: one wave per tile of whole reads, software-pipelined over the tiles a wave owns as
// fm_seed_pipe_kernel (words of tile t+2 | table entry of tile t+1 | resolution of tile t).  A tile has 128 key slots: the keys of the
// forward strand (at most 64), those of the reverse strand right behind them; tile_counts[tile] = both.  Seeds whose k-mer has more than
// 8 occurrences, or with several hits on a strand (a repeat longer than the seed), need the ordinary search of that strand
// (plain table + rank steps + finish on the text): what ends on several rows goes to that strand's residual
// list: ranges / ids [0, cap) forward, [cap, 2 cap) reverse; counts[1], counts[2].
// DEFER = false: that search runs inside this kernel, in the lane that needs it (every wave that holds one such lane pays its ten
//   dependent gathers: fine on a unique-ish genome where one window in 10^5 needs it, 2.6x the kernel's time when 5 % do).
// DEFER = true (NVBIO_FM_DEFER_HEAVY): the lane only records (seed id | strand << 31) in its tile's 128 deferred slots
//   (tile_defer / defer_counts, compacted like the keys); fm_seed_heavy_kernel then runs those searches as a dense launch of their own.
// Ragged reads (q.intervals): read r has its own length (offsets[r+1] - offsets[r]) and seed interval; lane (read, j) holds a seed iff
//   j x interval + len fits the read; q.spr is the largest seed count of a read (the stride of seed ids).
// ---------------------------------------------------------------------------------------------
// WIDE: 16-byte entries (NVBIO_FM_TABLE_CANONICAL_WIDE): a k-mer with TWO occurrences has both rows in its entry, groups start at three
template <int BITS, bool COUNT, bool WIDE, bool DEFER>
__global__ void __launch_bounds__(256)
fm_seed_both_kernel(const DevIndex f, const StringSetDev q, const SeedTiles tl, const uint32_t read_len, const uint32_t inline_max,
                    uint64_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_counts, uint2* __restrict__ res_ranges,
                    uint32_t* __restrict__ res_ids, const uint32_t res_cap, unsigned int* __restrict__ counts, unsigned long long* __restrict__ sectors_out,
                    uint32_t* __restrict__ tile_defer, uint32_t* __restrict__ defer_counts)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t k = f.ckmer, len = q.fixed_len, r = len - k;
    const uint32_t lr = lane / q.spr, j = lane - lr * q.spr;
    const bool lane_ok = lr < tl.rpt;
    const bool ragged = q.intervals != nullptr;
    const uint32_t rmask = (1u << (2u * r)) - 1u;                            // r <= 7
    const uint32_t fshift = 2u * (CTAB_FLANK - r);
    DevIndex fplain = f; fplain.dtab = nullptr; fplain.dkmer = 0;            // the fallback searches use the plain table only

    // the seed of this lane in read rid: where it begins in the symbol stream, its offset in the read, the read's length; false: no such seed
    auto seed_geom = [&](const uint32_t rid, uint32_t& begin, uint32_t& p_fw, uint32_t& rlen) -> bool {
        const uint32_t base = q.offsets ? q.offsets[rid] : rid * q.stride;
        p_fw  = j * (ragged ? q.intervals[rid] : q.interval);
        rlen  = ragged ? q.offsets[rid + 1] - base : read_len;
        begin = base + p_fw;
        return !ragged || p_fw + len <= rlen;
    };
    auto entry_of = [&](const uint64_t V, uint32_t& qo) -> uint4 {
        const uint64_t key = V >> (2u * r);
        qo = (uint32_t)(key >> k) & 1u;
        const uint64_t idx = canon_index( qo ? canon_revcomp( key, k ) : key, k );
        if (WIDE)
        {
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const u4v v = __builtin_nontemporal_load( (const u4v*)f.ctab + idx );
            return make_uint4( v.x, v.y, v.z, v.w );
        }
        const uint2 e = load_table_entry( f.ctab, idx, true );
        return make_uint4( e.x, e.y, 0u, 0u );
    };

    uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= tl.n_tiles) return;
    // stage 1 -> 2 (the tile after the current one): packed words, where they begin, the seed's offset in its read, the read's length
    SeedWords W = { 0, 0, 0, 0, 0 }; uint32_t w_begin = 0, w_pfw = 0, w_rlen = 0; bool w_valid = false;
    // stage 2 -> 3 (the current tile): seed bits, table entry, orientation
    uint64_t V = 0; uint4 e = make_uint4( 1u, 0u, 0u, 0u ); bool e_valid = false; uint32_t qo = 0, c_pfw = 0, c_rlen = 0; bool c_seed = false;
    {
        const uint32_t rid = tile * tl.rpt + lr;
        if (lane_ok && rid < tl.reads)
        {
            uint32_t b0;
            c_seed = seed_geom( rid, b0, c_pfw, c_rlen );
            if (c_seed)
            {
                const SeedWords W0 = load_seed_words<BITS>( q.symbols, b0, len );
                e_valid = seed_bits_from_words<BITS>( W0, b0, len, false, false, V );
                if (e_valid) e = entry_of( V, qo );
            }
        }
        const uint32_t t1 = tile + n_waves, rid1 = t1 * tl.rpt + lr;
        if (t1 < tl.n_tiles && lane_ok && rid1 < tl.reads)
        {
            w_valid = seed_geom( rid1, w_begin, w_pfw, w_rlen );
            if (w_valid) W = load_seed_words<BITS>( q.symbols, w_begin, len );
        }
    }
    for (; tile < tl.n_tiles; tile += n_waves)
    {
        const uint32_t rid = tile * tl.rpt + lr;
        const bool valid = lane_ok && rid < tl.reads && c_seed;
        // ---- stage 3a: the kind of entry; the group of a k-mer with 2..8 occurrences is requested ----
        const bool is_one   = e_valid && e.y >= DTAB_MARK && e.x < DTAB_MARK;
        const bool is_group = e_valid && e.y >= DTAB_MARK && e.x >= DTAB_MARK;
        const bool is_heavy = e_valid && e.y < DTAB_MARK && e.x <= e.y;
        uint4 q0 = make_uint4( 0, 0, 0, 0 ), q1 = q0, q2 = q0, q3 = q0;
        const uint32_t m = e.y - DTAB_MARK;
        if (is_group)
        {
            const uint4* g4 = (const uint4*)(f.cside + 4ull * (e.x - DTAB_MARK));
            q0 = g4[0]; q1 = g4[1];
            if (m > 4u) { q2 = g4[2]; q3 = g4[3]; }
        }
        // ---- stage 2 for the next tile ----
        uint64_t Vn = 0; uint4 en = make_uint4( 1u, 0u, 0u, 0u ); bool en_valid = false; uint32_t qon = 0;
        const uint32_t n_pfw = w_pfw, n_rlen = w_rlen; const bool n_seed = w_valid;
        if (w_valid)
        {
            en_valid = seed_bits_from_words<BITS>( W, w_begin, len, false, false, Vn );
            if (en_valid) en = entry_of( Vn, qon );
        }
        // ---- stage 1 for the tile after next ----
        {
            const uint32_t t2 = tile + 2u * n_waves, rid2 = t2 * tl.rpt + lr;
            w_valid = t2 < tl.n_tiles && lane_ok && rid2 < tl.reads;
            if (w_valid) w_valid = seed_geom( rid2, w_begin, w_pfw, w_rlen );
            if (w_valid) W = load_seed_words<BITS>( q.symbols, w_begin, len );
        }
        // ---- stage 3b: resolve the current tile ----
        const uint32_t rest = (uint32_t)V & rmask;                           // the seed's first r symbols, reversed (scan order)
        const uint32_t want = qo ? (~rest & rmask) : rest;
        // per strand: the hits found among the entry's rows (the first CTAB_INLINE positions are kept)
        uint32_t hits[2] = { 0, 0 }, pos[2][CTAB_INLINE] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
        auto row = [&](const uint32_t P, const uint32_t C) {
            const uint32_t pay = C - DTAB_MARK;
            const uint32_t o = (pay >> 28) & 1u;
            const uint32_t flank = qo ? (pay & 0x3FFFu) : ((pay >> 14) & 0x3FFFu);
            if ((flank >> fshift) != want) return;
            const bool rev = (o ^ qo) != 0u;
            if (rev ? ((uint64_t)P + len > f.length) : (P < r)) return;
            const uint32_t v = rev ? P : P - r;
            #pragma unroll
            for (int s = 0; s < 2; ++s)
                if ((s == 1) == rev)
                {
                    #pragma unroll
                    for (uint32_t t = 0; t < CTAB_INLINE; ++t) pos[s][t] = (hits[s] == t) ? v : pos[s][t];
                    ++hits[s];
                }
        };
        if (is_one) { row( e.x, e.y ); if (WIDE && e.w >= DTAB_MARK) row( e.z, e.w ); }
        else if (is_group)
        {
#define NVB_CROW(jj, P, C) if ((jj) < m) row( (P), (C) );
            NVB_CROW( 0u, q0.x, q0.y ) NVB_CROW( 1u, q0.z, q0.w ) NVB_CROW( 2u, q1.x, q1.y ) NVB_CROW( 3u, q1.z, q1.w )
            NVB_CROW( 4u, q2.x, q2.y ) NVB_CROW( 5u, q2.z, q2.w ) NVB_CROW( 6u, q3.x, q3.y ) NVB_CROW( 7u, q3.z, q3.w )
#undef NVB_CROW
        }
        uint32_t sectors = e_valid ? (is_group ? 2u : 1u) : 0u;
        const uint32_t i = rid * q.spr + j;
        uint32_t tile_fill = 0;                                              // keys of this tile written so far: the reverse strand's follow the forward strand's
        uint32_t defer_fill = 0;                                             // deferred searches of this tile recorded so far (DEFER)
        #pragma unroll
        for (int s = 0; s < 2; ++s)
        {
            // what this strand leaves: cnt keys (positions pos[s][0 .. cnt)), or the range (rx, ry) of several rows for the residual list
            uint32_t cnt = (valid && e_valid && !is_heavy && hits[s] <= inline_max) ? hits[s] : 0u;
            uint32_t rx = 1u, ry = 0u;
            bool searched = false, deferred = false;
            const uint32_t sflags = s ? (NVBIO_FM_SCAN_FORWARD | NVBIO_FM_COMPLEMENT) : 0u;
            auto plain_search = [&]() -> bool {                              // the ordinary search of this strand: range (rx, ry) or a position
                uint32_t nblk, sec = 0; bool single = false;
                match_one<BITS,false,true,true>( fplain, q, sflags, f.ktab != nullptr, false, i, rx, ry, nblk, single, &sec );
                sectors += sec; searched = true;
                return single;
            };
            if (valid && e_valid && (is_heavy || hits[s] > inline_max))
            {
                cnt = 0u;
                if (DEFER) deferred = true;
                else if (plain_search()) { pos[s][0] = rx; cnt = 1u; }
                else if (rx <= ry && ry - rx < inline_max)                   // 1 .. inline_max rows: their positions from the suffix array
                {
                    cnt = ry - rx + 1u;
                    #pragma unroll
                    for (uint32_t t = 0; t < CTAB_INLINE; ++t)
                        if (t < cnt) { const uint32_t sv = f.ssa[rx + t]; pos[s][t] = (sv == 0xFFFFFFFFu) ? f.length : sv; }
                    sectors += cnt;
                }
            }
            // a tile's keys of one strand must fit its 64 slots: if the seeds with several keys would overflow them, those seeds go to the
            // residual list instead (a wave-uniform decision; the range of a seed resolved from its entry's rows is searched for then)
            uint32_t total = (uint32_t)__popcll( __ballot( cnt == 1u ) );
            #pragma unroll
            for (uint32_t t = 0; t < CTAB_INLINE; ++t) total += (uint32_t)__popcll( __ballot( cnt >= 2u && t < cnt ) );
            if (total > 64u && cnt >= 2u)
            {
                if (DEFER) { deferred = true; rx = 1u; ry = 0u; }
                else if (!searched) (void)plain_search();
                cnt = 0u;
            }
            uint64_t* slots = tile_keys + (uint64_t)tile * 128u + tile_fill;
            uint32_t n_out = 0; uint64_t last = ~0ull;
            const uint32_t x1 = cnt == 1u ? pos[s][0] : (cnt == 0u ? rx : 1u), y1 = cnt == 1u ? pos[s][0] : (cnt == 0u ? ry : 0u);
            emit_seed_results( c_pfw, len, c_rlen, (uint32_t)s, lane, valid, cnt == 1u, x1, y1, rid, i, slots, n_out, last,
                               res_ranges + (s ? res_cap : 0u), res_ids + (s ? res_cap : 0u), counts + s );
            // seeds with 2 .. inline_max hits on this strand: their keys behind the tile's one-hit keys, no duplicate removal
            if (__ballot( cnt >= 2u ))
            {
                uint32_t pq = c_pfw;
                if (s) pq = c_rlen - pq - len;
                #pragma unroll
                for (uint32_t t = 0; t < CTAB_INLINE; ++t)
                {
                    const bool on = cnt >= 2u && t < cnt;
                    const uint64_t mk = __ballot( on );
                    if (on) slots[n_out + (uint32_t)__popcll( mk & ((1ull << lane) - 1ull) )] =
                                ((uint64_t)rid << 34) | ((uint64_t)s << 33) | ((uint64_t)pos[s][t] + 1024u - pq);
                    n_out += (uint32_t)__popcll( mk );
                }
            }
            tile_fill += n_out;
            if (DEFER)
            {
                const uint64_t md = __ballot( deferred );
                if (md)
                {
                    if (deferred) tile_defer[(uint64_t)tile * 128u + defer_fill + (uint32_t)__popcll( md & ((1ull << lane) - 1ull) )] = i | ((uint32_t)s << 31);
                    defer_fill += (uint32_t)__popcll( md );
                }
            }
        }
        if (lane == 0) { tile_counts[tile] = tile_fill; if (DEFER) defer_counts[tile] = defer_fill; }
        if (COUNT)
        {
            uint32_t tot = valid ? sectors : 0u;
            #pragma unroll
            for (int d = 32; d > 0; d >>= 1) tot += (uint32_t)__shfl_xor( (int)tot, d );
            if (lane == 0 && tot) atomicAdd( sectors_out, (unsigned long long)tot );
        }
        V = Vn; e = en; e_valid = en_valid; qo = qon; c_pfw = n_pfw; c_rlen = n_rlen; c_seed = n_seed;
    }
}

// The deferred searches of fm_seed_both_kernel<.., DEFER = true> as a dense launch: entry t of `list` (n = *n_list) is
// (seed id | strand << 31); the ordinary search of that strand (plain table + rank steps + finish on the text) ends on
//   one text occurrence       -> its diagonal key, appended behind the compacted keys (keys_out + counts[0], counts[0] advanced)
//   several rows (a repeat)   -> that strand's residual list (counts[1 + strand]), as the in-line search would
//   nothing                   -> nothing.
// Every lane runs four searches, then the workgroup appends what they left with ONE returning atomic per output list (a wave-level
// append per list would put 2 x 10^5 atomics on two neighbouring words for 9 M searches: at ~11 ns each, longer than the searches).
constexpr uint32_t HEAVY_PER_LANE = 4u;
template <int BITS>
__global__ void __launch_bounds__(256)
fm_seed_heavy_kernel(const DevIndex f, const StringSetDev q, const uint32_t read_len, const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                     uint64_t* __restrict__ keys_out, uint2* __restrict__ res_ranges, uint32_t* __restrict__ res_ids, const uint32_t res_cap,
                     unsigned int* __restrict__ counts)
{
    __shared__ uint32_t s_cnt[4][3], s_base[3];
    const uint32_t n = *n_list;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t len = q.fixed_len;
    const bool ragged = q.intervals != nullptr;
    DevIndex fplain = f; fplain.dtab = nullptr; fplain.dkmer = 0;
    const uint32_t chunk = 256u * HEAVY_PER_LANE;
    for (uint32_t c0 = blockIdx.x * chunk; c0 < n; c0 += gridDim.x * chunk)           // (block-uniform trip count: the barriers below are safe)
    {
        uint32_t kind[HEAVY_PER_LANE], vx[HEAVY_PER_LANE], vy[HEAVY_PER_LANE], ent[HEAVY_PER_LANE];
        uint32_t nk = 0, nr0 = 0, nr1 = 0;                                            // this lane's keys / forward residuals / reverse residuals
        #pragma unroll
        for (uint32_t u = 0; u < HEAVY_PER_LANE; ++u)
        {
            const uint32_t t = c0 + u * 256u + threadIdx.x;
            kind[u] = 0u; vx[u] = 1u; vy[u] = 0u; ent[u] = 0u;
            if (t < n)
            {
                ent[u] = list[t];
                const uint32_t i = ent[u] & 0x7FFFFFFFu, s = ent[u] >> 31;
                uint32_t rx, ry, nblk; bool single;
                match_one<BITS,false,true,true>( fplain, q, s ? (NVBIO_FM_SCAN_FORWARD | NVBIO_FM_COMPLEMENT) : 0u, f.ktab != nullptr, false, i, rx, ry, nblk, single );
                if (!single && rx == ry) { const uint32_t sv = f.ssa[rx]; rx = ry = (sv == 0xFFFFFFFFu) ? f.length : sv; single = true; }
                if (single)        { kind[u] = 1u; vx[u] = rx; ++nk; }
                else if (rx < ry)  { kind[u] = 2u + s; vx[u] = rx; vy[u] = ry; if (s) ++nr1; else ++nr0; }
            }
        }
        // exclusive prefix of the three counts within the wave, wave totals to LDS, one atomic per list and workgroup
        uint32_t pk = nk, p0 = nr0, p1 = nr1;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            const uint32_t a = (uint32_t)__shfl_up( (int)pk, d ), b = (uint32_t)__shfl_up( (int)p0, d ), c = (uint32_t)__shfl_up( (int)p1, d );
            if (lane >= (uint32_t)d) { pk += a; p0 += b; p1 += c; }
        }
        if (lane == 63u) { s_cnt[wave][0] = pk; s_cnt[wave][1] = p0; s_cnt[wave][2] = p1; }
        __syncthreads();
        if (threadIdx.x < 3u)
        {
            const uint32_t tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
            s_base[threadIdx.x] = tot ? atomicAdd( &counts[threadIdx.x], tot ) : 0u;
        }
        __syncthreads();
        uint32_t ok = s_base[0] + (pk - nk), o0 = s_base[1] + (p0 - nr0), o1 = s_base[2] + (p1 - nr1);
        for (uint32_t w = 0; w < wave; ++w) { ok += s_cnt[w][0]; o0 += s_cnt[w][1]; o1 += s_cnt[w][2]; }
        #pragma unroll
        for (uint32_t u = 0; u < HEAVY_PER_LANE; ++u)
        {
            const uint32_t i = ent[u] & 0x7FFFFFFFu, s = ent[u] >> 31;
            if (kind[u] == 1u)
            {
                const uint32_t rid = i / q.spr, j = i - rid * q.spr;
                uint32_t pq = j * (ragged ? q.intervals[rid] : q.interval);
                const uint32_t rlen = ragged ? q.offsets[rid + 1] - q.offsets[rid] : read_len;
                if (s) pq = rlen - pq - len;
                keys_out[ok++] = ((uint64_t)rid << 34) | ((uint64_t)s << 33) | ((uint64_t)vx[u] + 1024u - pq);
            }
            else if (kind[u] == 2u) { res_ranges[o0] = make_uint2( vx[u], vy[u] ); res_ids[o0] = i; ++o0; }
            else if (kind[u] == 3u) { res_ranges[res_cap + o1] = make_uint2( vx[u], vy[u] ); res_ids[res_cap + o1] = i; ++o1; }
        }
        __syncthreads();                                                              // s_cnt / s_base are reused by the next chunk
    }
}

// the deferred slots of every tile made dense (as fm_seed_compact_kernel does with the keys); counts_out[0] = their number
__global__ void __launch_bounds__(256)
fm_seed_defer_compact_kernel(const uint32_t* __restrict__ tile_defer, const uint32_t* __restrict__ defer_counts, const uint32_t* __restrict__ defer_offsets,
                             const uint32_t n_tiles, uint32_t* __restrict__ list, uint32_t* __restrict__ n_out)
{
    for (uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x; tile < n_tiles; tile += gridDim.x * blockDim.x)
    {
        const uint32_t n = defer_counts[tile], off = defer_offsets[tile];
        const uint32_t* src = tile_defer + (uint64_t)tile * 128u;
        for (uint32_t k = 0; k < n; ++k) list[off + k] = src[k];
        if (tile == n_tiles - 1u) n_out[0] = off + n;
    }
}
