"""nvbio-gpl_amd -- MI355X-native seed-and-extend core behind NVBIO's operator API.

The product is the shared library ``lib/libnvbio_amd.so`` (hand-written HIP kernels for gfx950
behind the C ABI of ``include/nvbio_amd.h``).  This package is the thin host-side mirror of the
reference's interface for the path, for Python callers (tests, bench): the same names and
argument meaning as the reference's C++ templates --

    FMIndex                      nvbio::fm_index / io::FMIndexDataDevice   (nvbio/fmindex/fmindex.h:320-557)
    FMIndexFilter                nvbio::FMIndexFilter<device_tag,...>      (nvbio/fmindex/filter.h:52-231)
    SimpleGotohScheme, GotohAligner, BestSink semantics                    (nvbio/alignment/utils.h:103-123, alignment.h:437-449)
    BatchedBandedAlignmentScore, batch_banded_alignment_score              (nvbio/alignment/batched.h:104-298)

PyTorch is used only as plumbing for device memory and streams.  There is NO CPU fallback:
importing works anywhere (so that symbol checks can run), but every compute entry point raises
``NvbioError`` when the library or a gfx950 device is missing.  Nothing here imports ``oracle``.

The directory name contains a hyphen, so load it with ``__graft_entry__.load_package()``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnvbio_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nvbio_amd.h")

GLOBAL, LOCAL, SEMI_GLOBAL = 0, 1, 2
SCORE_MIN = -(1 << 30)
FM_SCAN_FORWARD, FM_COMPLEMENT, FM_NO_KMER_TABLE, FM_NO_VERIFY, FM_COUNT_SECTORS, FM_NO_PIPELINE, FM_DEFER_HEAVY = 1, 2, 4, 8, 16, 32, 64
FM_TABLE_NO_DIRECT, FM_TABLE_NO_CONTEXT, FM_TABLE_NO_GROUPS, FM_TABLE_CANONICAL, FM_TABLE_CANONICAL_WIDE = 1, 2, 4, 8, 16      # nvbio_fm_build_options::table_flags
READ_REVERSE, READ_COMPLEMENT = 1, 2
TRACEBACK_SINKS_GIVEN = 1
# nvbio_alignment_batch::algo_flags (which exact shortcuts / kernel variants a call may use; results do not depend on them)
ALN_NO_UNGAPPED_SCORE, ALN_NO_THIRD_CHANCE, ALN_NO_PACKED_DP, ALN_FORCE_PACKED_DP, ALN_NO_UNGAPPED_TRACEBACK, ALN_PK_THREE_WAVES = 1, 2, 4, 8, 16, 32
ALN_NO_NARROW_TRACEBACK, ALN_NO_SECOND_CHANCE, ALN_PK_STRIPE8, ALN_NO_NARROW_SCORE, ALN_NO_BAND_ROUTE = 64, 128, 256, 512, 1024
ALN_NO_QUALITY_SHORTCUT, ALN_RAGGED_READS, ALN_NO_LENGTH_SORT, ALN_NO_F16_DP, ALN_NO_COOPERATIVE_DP, ALN_NO_GAP_CHANCE = 2048, 4096, 8192, 16384, 32768, 65536
DEFAULT_ALGO_FLAGS = 0          # what an AlignmentBatch is created with unless told otherwise (tests set it for a whole run)
BACKTRACK_REFERENCE_QUIRKS = 1

_STATUS = {0: "OK", 1: "INVALID", 2: "HIP", 3: "NOMEM", 4: "UNSUPPORTED", 5: "NO_DEVICE"}


class NvbioError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("nvbio_amd: %s: %s" % (_STATUS.get(status, status), msg))
        self.status = status


# ---- C structs ---------------------------------------------------------------------------------
class _View(ctypes.Structure):
    _fields_ = [("length", ctypes.c_uint32), ("primary", ctypes.c_uint32), ("L2", ctypes.c_uint32 * 5),
                ("bwt_occ_dev", ctypes.c_void_p), ("bwt_occ_words", ctypes.c_uint64),
                ("ssa_dev", ctypes.c_void_p), ("ssa_words", ctypes.c_uint64), ("sa_int", ctypes.c_uint32)]


class _BuildOptions(ctypes.Structure):
    _fields_ = [("kmer_len", ctypes.c_uint32), ("sa_int", ctypes.c_uint32), ("max_lcp", ctypes.c_uint32),
                ("verify", ctypes.c_uint32), ("table_flags", ctypes.c_uint32), ("bucket_symbols", ctypes.c_uint32)]


class _StringSet(ctypes.Structure):
    _fields_ = [("symbols_dev", ctypes.c_void_p), ("symbol_bits", ctypes.c_uint32),
                ("offsets_dev", ctypes.c_void_p), ("offsets_are_ranges", ctypes.c_uint32),
                ("fixed_len", ctypes.c_uint32), ("stride", ctypes.c_uint32), ("n", ctypes.c_uint32),
                ("seeds_per_string", ctypes.c_uint32), ("seed_interval", ctypes.c_uint32),
                ("seed_intervals_dev", ctypes.c_void_p)]


class _Scheme(ctypes.Structure):
    _fields_ = [("match", ctypes.c_int32), ("mm_min", ctypes.c_int32), ("mm_max", ctypes.c_int32),
                ("pat_gap_open", ctypes.c_int32), ("pat_gap_ext", ctypes.c_int32),
                ("txt_gap_open", ctypes.c_int32), ("txt_gap_ext", ctypes.c_int32)]


class _SWScheme(ctypes.Structure):
    _fields_ = [("match", ctypes.c_int32), ("mismatch", ctypes.c_int32), ("deletion", ctypes.c_int32),
                ("insertion", ctypes.c_int32)]


class _Batch(ctypes.Structure):
    _fields_ = [("reads_dev", ctypes.c_void_p), ("read_bits", ctypes.c_uint32),
                ("read_offsets_dev", ctypes.c_void_p), ("quals_dev", ctypes.c_void_p),
                ("read_id_dev", ctypes.c_void_p), ("flags_dev", ctypes.c_void_p),
                ("text_dev", ctypes.c_void_p), ("text_bits", ctypes.c_uint32),
                ("win_begin_dev", ctypes.c_void_p), ("win_end_dev", ctypes.c_void_p), ("n", ctypes.c_uint32),
                ("max_read_len", ctypes.c_uint32), ("algo_flags", ctypes.c_uint32)]


class _HitQueues(ctypes.Structure):
    _fields_ = [("idx_queue_dev", ctypes.c_void_p), ("hit_read_id_dev", ctypes.c_void_p), ("hit_seed_dev", ctypes.c_void_p),
                ("hit_loc_dev", ctypes.c_void_p), ("hit_score_dev", ctypes.c_void_p), ("hit_sink_dev", ctypes.c_void_p),
                ("n", ctypes.c_uint32)]


class _SeedHitsParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("seeds_per_read", "first_offset", "seed_interval", "seed_len", "read_len", "max_hits",
                                                "rep_seeds", "max_effort", "min_ext", "max_ext")]


class _RankDict(ctypes.Structure):
    _fields_ = [("text_dev", ctypes.c_void_p), ("word_bits", ctypes.c_uint32), ("occ_dev", ctypes.c_void_p), ("index_bits", ctypes.c_uint32),
                ("K", ctypes.c_uint32), ("length", ctypes.c_uint64)]


_lib = None


def lib():
    """the C-ABI library; raises (loudly) if it has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NvbioError(5, "%s is missing: build it with __graft_entry__.build() "
                                "(there is no CPU fallback)" % LIB_PATH)
        # torch first: it ships its own HIP runtime, and the library must bind to the one that owns the tensors it is handed
        # (loaded the other way round the two runtimes do not see each other's device context: NVBIO_ERR_NO_DEVICE)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.nvbio_amd_last_error.restype = ctypes.c_char_p
        L.nvbio_amd_version.restype = ctypes.c_int
        _lib = L
    return _lib


def release_scratch():
    """nvbio_amd_release_scratch: give the library's idle scratch blocks back to the driver (it keeps them per stream between calls)"""
    _check(lib().nvbio_amd_release_scratch())


def _check(status):
    if status != 0:
        raise NvbioError(status, lib().nvbio_amd_last_error().decode())


def _torch():
    import torch
    return torch


_SCRATCH = {}


def _scratch(device, nbytes, cap=8 << 30):
    """one persistent scratch tensor per device for the DP kernels' boundary columns / direction vectors (at most `cap`
    bytes; the library processes a batch in as many launches as the scratch allows).  Handing the library caller scratch
    keeps multi-GiB allocations out of the call (the library would otherwise take a block of its own scratch cache for them and
    keep it).  All users are ordered on the current stream."""
    torch = _torch()
    nbytes = int(min(max(nbytes, 1 << 20), cap))
    key = str(device)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        _SCRATCH[key] = None
        t = _SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return t


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream_ptr(device):
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def device_count():
    n = ctypes.c_int(0)
    lib().nvbio_amd_device_count(ctypes.byref(n))
    return n.value


def _dev_tensor(a, dtype, device):
    """numpy array or tensor -> contiguous device tensor of the given torch dtype (bit pattern kept)"""
    torch = _torch()
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        t = a
        if t.dtype != dtype:
            t = t.view(dtype) if t.element_size() == torch.empty(0, dtype=dtype).element_size() else t.to(dtype)
        return t.to(device).contiguous()
    a = np.ascontiguousarray(a)
    np_of = {torch.int32: np.int32, torch.uint8: np.uint8, torch.int64: np.int64}[dtype]
    if a.dtype.itemsize == np.dtype(np_of).itemsize:
        a = a.view(np_of)
    else:
        a = a.astype(np_of)
    return torch.from_numpy(a).to(device)


# ---- string sets -------------------------------------------------------------------------------
class PackedStringSet:
    """A set of strings in HBM (the string-set concept of FMIndexFilter::rank, filter.h:52-231).

    symbols : packed big-endian uint32 words (bits 2 or 4) or bytes (bits 8), numpy or tensor
    offsets : None (string i = [i*stride, i*stride+fixed_len)), n start offsets (+ fixed_len),
              or n+1 offsets with ranges=True (io::SequenceData sequence_index)
    """

    def __init__(self, symbols, bits, n, offsets=None, ranges=False, fixed_len=0, stride=None, device="cuda:0",
                 seeds_per_string=0, seed_interval=0, seed_intervals=None):
        torch = _torch()
        self.bits, self.n = int(bits), int(n)
        self.symbols = _dev_tensor(symbols, torch.uint8 if bits == 8 else torch.int32, device)
        self.offsets = _dev_tensor(offsets, torch.int32, device)
        self.ranges, self.fixed_len = bool(ranges), int(fixed_len)
        self.stride = int(fixed_len if stride is None else stride)
        self.device = device
        # seed enumeration: n = n_strings * seeds_per_string queries [base(r) + j*seed_interval, + fixed_len)
        self.seeds_per_string, self.seed_interval = int(seeds_per_string), int(seed_interval)
        # ragged seed sets: one seed interval per string (int32 [n_strings]); offsets then hold n_strings + 1 entries and
        # seeds_per_string is the largest seed count of a string (the stride of seed ids)
        self.seed_intervals = _dev_tensor(seed_intervals, torch.int32, device)

    def c_struct(self):
        return _StringSet(_ptr(self.symbols), self.bits, _ptr(self.offsets), 1 if self.ranges else 0,
                          self.fixed_len, self.stride, self.n, self.seeds_per_string, self.seed_interval, _ptr(self.seed_intervals))


# ---- FM-index ----------------------------------------------------------------------------------
class FMIndex:
    """nvbio::fm_index over the production bwt_occ / ssa layout, resident in HBM."""

    def __init__(self, handle, device, keep=()):
        self._h, self.device, self._keep = handle, device, keep

    @staticmethod
    def _dev_index(device):
        torch = _torch()
        d = torch.device(device)
        return d.index if d.index is not None else 0

    @classmethod
    def from_arrays(cls, length, primary, L2, bwt_occ, ssa, kmer_len=0, sa_int=16, device="cuda:0"):
        """wrap index arrays (numpy or tensors); replaces FMIndexDataDevice (fmindex_impl.cu:740-816)"""
        torch = _torch()
        b = _dev_tensor(bwt_occ, torch.int32, device)
        s = _dev_tensor(ssa, torch.int32, device) if ssa is not None else None
        v = _View()
        v.length, v.primary = int(length), int(primary)
        for i in range(5):
            v.L2[i] = int(L2[i])
        v.bwt_occ_dev, v.bwt_occ_words = b.data_ptr(), b.numel()
        v.ssa_dev, v.ssa_words = (s.data_ptr(), s.numel()) if s is not None else (None, 0)
        v.sa_int = sa_int
        h = ctypes.c_void_p()
        _check(lib().nvbio_fm_index_create(ctypes.byref(v), cls._dev_index(device), ctypes.c_uint32(kmer_len),
                                           _stream_ptr(device), ctypes.byref(h)))
        return cls(h, device, keep=(b, s))

    @classmethod
    def build(cls, text2, length, kmer_len=0, max_lcp=0, sa_int=16, verify=False, device="cuda:0", table_flags=0,
              bucket_symbols=None):
        """build the index on the GPU from a 2-bit packed text (nvbio_fm_index_build).
        table_flags: FM_TABLE_* (which form of the k-mer tables a direct-capable handle keeps; results do not depend on it);
        bucket_symbols: None = automatic, 0..4 forces that many prefix symbols in the suffix sort's bucketing (tests)"""
        torch = _torch()
        t = _dev_tensor(text2, torch.int32, device)
        h = ctypes.c_void_p()
        opts = _BuildOptions(kmer_len, sa_int, max_lcp, 1 if verify else 0, int(table_flags),
                             0 if bucket_symbols is None else 1 + int(bucket_symbols))
        _check(lib().nvbio_fm_index_build(_ptr(t), ctypes.c_uint32(length), cls._dev_index(device),
                                          ctypes.byref(opts), _stream_ptr(device), ctypes.byref(h)))
        return cls(h, device, keep=(t,))

    @classmethod
    def load(cls, bwt_path, sa_path=None, kmer_len=0, device="cuda:0"):
        """load the reference's .bwt / .sa files (io::FMIndexDataHost::load, fmindex_impl.cu:333-...)"""
        h = ctypes.c_void_p()
        _check(lib().nvbio_fm_index_load(bwt_path.encode(), sa_path.encode() if sa_path else None,
                                         cls._dev_index(device), ctypes.c_uint32(kmer_len), _stream_ptr(device),
                                         ctypes.byref(h)))
        return cls(h, device)

    def save(self, bwt_path, sa_path=None):
        """write the index in the reference's .bwt / .sa formats (the role of nvBWT)"""
        _check(lib().nvbio_fm_index_save(self._h, bwt_path.encode(), sa_path.encode() if sa_path else None,
                                         _stream_ptr(self.device)))

    def close(self):
        if self._h is not None:
            _torch().cuda.synchronize(self.device)
            _check(lib().nvbio_fm_index_destroy(self._h))
            self._h = None
            self._seed_bufs = None                   # the pipeline's per-strand seed-pass buffers (pipeline.seed_and_extend)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view(self):
        v = _View()
        _check(lib().nvbio_fm_index_get_view(self._h, ctypes.byref(v)))
        return v

    @property
    def length(self):
        return self.view().length

    @property
    def primary(self):
        return self.view().primary

    def arrays(self):
        """(bwt_occ, ssa) copied out into fresh device int32 tensors (nvbio_fm_index_export)"""
        torch = _torch()
        v = self.view()
        b = torch.empty(int(v.bwt_occ_words), dtype=torch.int32, device=self.device)
        s = torch.empty(int(v.ssa_words), dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_index_export(self._h, _ptr(b), _ptr(s), _stream_ptr(self.device)))
        return b, s

    def device_bytes(self):
        b = ctypes.c_uint64()
        _check(lib().nvbio_fm_index_device_bytes(self._h, ctypes.byref(b)))
        return b.value

    # -- free functions of nvbio/fmindex/fmindex.h:370-557 ---------------------------------------
    def match(self, queries, flags=0, want_blocks=False):
        """ranges[n,2] (int32 tensor holding uint32 bits) = match()/match_reverse() of every query"""
        torch = _torch()
        ranges = torch.empty((queries.n, 2), dtype=torch.int32, device=self.device)
        blocks = torch.empty(queries.n, dtype=torch.int32, device=self.device) if want_blocks else None
        qs = queries.c_struct()
        _check(lib().nvbio_fm_match(self._h, ctypes.byref(qs), ctypes.c_uint32(flags), _ptr(ranges), _ptr(blocks),
                                    _stream_ptr(self.device)))
        return (ranges, blocks) if want_blocks else ranges

    def supports_direct(self):
        yes = ctypes.c_int(0)
        _check(lib().nvbio_fm_index_supports_direct(self._h, ctypes.byref(yes)))
        return bool(yes.value)

    def match_direct(self, queries, flags=0):
        """(ranges, direct): as match(), but searches that collapse to one SA row finish on the text and report
        the occurrence's text position, flagged in direct (nvbio_fm_match_direct; needs an index built with sa_int=1)"""
        torch = _torch()
        ranges = torch.empty((queries.n, 2), dtype=torch.int32, device=self.device)
        direct = torch.empty(queries.n, dtype=torch.uint8, device=self.device)
        qs = queries.c_struct()
        _check(lib().nvbio_fm_match_direct(self._h, ctypes.byref(qs), ctypes.c_uint32(flags), _ptr(ranges), _ptr(direct),
                                           _stream_ptr(self.device)))
        return ranges, direct

    def hamming_backtrack(self, queries, seed_len, mismatches, quirks=False, max_ranges=0):
        """nvbio::hamming_backtrack with a counting delegate (nvbio_fm_hamming_backtrack) -> (counts, n_ranges, ranges or None)"""
        torch = _torch()
        counts = torch.empty(queries.n, dtype=torch.int32, device=self.device)
        nr = torch.empty(queries.n, dtype=torch.int32, device=self.device)
        rg = torch.zeros((queries.n, max_ranges, 2), dtype=torch.int32, device=self.device) if max_ranges else None
        qs = queries.c_struct()
        _check(lib().nvbio_fm_hamming_backtrack(self._h, ctypes.byref(qs), ctypes.c_uint32(seed_len), ctypes.c_uint32(mismatches),
                                                ctypes.c_uint32(BACKTRACK_REFERENCE_QUIRKS if quirks else 0), _ptr(counts), _ptr(nr), _ptr(rg),
                                                ctypes.c_uint32(max_ranges), _stream_ptr(self.device)))
        return counts, nr, rg

    def match_seed_diagonals(self, seeds, flags, read_len, strand, buffers=None, grid_blocks=0):
        """the seed pass of one strand straight to candidate diagonals (nvbio_fm_match_seed_diagonals) -> the buffers dict:
        "keys" int64 (the first counts[0] are valid, in seed order), "ranges" int32 [., 2] and "ids" int32 (the first counts[1]:
        the residual seeds on several SA rows), "counts" int32 [2] (on the device: the caller reads them).
        buffers: optional dict reused across calls (the output arrays and the call's scratch);
        grid_blocks: cap of the launch in workgroups, a multiple of 64 (0 = the library's choice; results do not depend on it)"""
        torch = _torch()
        n = seeds.n
        if buffers is None:
            buffers = {}
        qs = seeds.c_struct()
        if buffers.get("n") != n or buffers.get("spr") != seeds.seeds_per_string:
            buffers["n"], buffers["spr"] = n, seeds.seeds_per_string
            # keys: only as many as there can be candidates are ever written; sized for the worst case like the residual arrays
            buffers["keys"] = torch.empty(n, dtype=torch.int64, device=self.device)
            buffers["ranges"] = torch.empty((n, 2), dtype=torch.int32, device=self.device)
            buffers["ids"] = torch.empty(n, dtype=torch.int32, device=self.device)
            buffers["counts"] = torch.empty(4, dtype=torch.int32, device=self.device)     # [2:4]: FM_COUNT_SECTORS' uint64
            nb = ctypes.c_uint64(0)
            _check(lib().nvbio_fm_match_seed_diagonals_temp_bytes(ctypes.byref(qs), ctypes.byref(nb)))
            buffers["temp"] = torch.empty(nb.value, dtype=torch.uint8, device=self.device)
        assert grid_blocks % 64 == 0 and grid_blocks < (1 << 22)
        _check(lib().nvbio_fm_match_seed_diagonals(self._h, ctypes.byref(qs), ctypes.c_uint32(flags | ((grid_blocks // 64) << 16)),
                                                   ctypes.c_uint32(read_len), ctypes.c_uint32(strand), _ptr(buffers["keys"]),
                                                   _ptr(buffers["ranges"]), _ptr(buffers["ids"]), _ptr(buffers["counts"]),
                                                   _ptr(buffers["temp"]), ctypes.c_uint64(buffers["temp"].numel()),
                                                   _stream_ptr(self.device)))
        return buffers

    @property
    def canonical_kmer(self):
        """k of the canonical two-strand table the handle holds (built with FM_TABLE_CANONICAL; seeds of k .. k + 7 symbols), 0 if none"""
        fn = lib().nvbio_fm_index_is_canonical
        fn.restype = ctypes.c_int
        return int(fn(self._h))

    @property
    def canonical(self):
        return self.canonical_kmer != 0

    def match_seed_diagonals_both(self, seeds, read_len, buffers=None, flags=0, grid_blocks=0, inline_hits=0, defer_heavy=False):
        """the seed pass of BOTH strands in one launch over the canonical table (nvbio_fm_match_seed_diagonals_both) -> the buffers dict:
        "keys" int64 (the first counts[0]: both strands, tile by tile), "ranges" int32 [2 n, 2] / "ids" int32 [2 n]: residual seeds on
        several rows, forward strand in [0, counts[1]), reverse strand in [n, n + counts[2]); "counts" int32 [6] on the device.
        inline_hits (2..4): a search that ends on up to that many rows leaves all their keys in "keys" instead of a residual entry."""
        torch = _torch()
        n = seeds.n
        if buffers is None:
            buffers = {}
        qs = seeds.c_struct()
        if buffers.get("n") != n or buffers.get("spr") != seeds.seeds_per_string:
            buffers["n"], buffers["spr"] = n, seeds.seeds_per_string
            cap = ctypes.c_uint64(0)
            _check(lib().nvbio_fm_match_seed_diagonals_both_keys_capacity(ctypes.byref(qs), ctypes.byref(cap)))
            buffers["keys"] = torch.empty(max(int(cap.value), 1), dtype=torch.int64, device=self.device)
            buffers["ranges"] = torch.empty((2 * n, 2), dtype=torch.int32, device=self.device)
            buffers["ids"] = torch.empty(2 * n, dtype=torch.int32, device=self.device)
            buffers["counts"] = torch.empty(6, dtype=torch.int32, device=self.device)     # [4:6]: FM_COUNT_SECTORS' uint64
            nb = ctypes.c_uint64(0)
            _check(lib().nvbio_fm_match_seed_diagonals_both_temp_bytes(ctypes.byref(qs), ctypes.byref(nb)))
            buffers["temp"] = torch.empty(nb.value, dtype=torch.uint8, device=self.device)
        assert grid_blocks % 64 == 0 and grid_blocks < (1 << 22)
        _check(lib().nvbio_fm_match_seed_diagonals_both(
            self._h, ctypes.byref(qs), ctypes.c_uint32(flags | (FM_DEFER_HEAVY if defer_heavy else 0) | ((int(inline_hits) & 15) << 8) | ((grid_blocks // 64) << 16)),
            ctypes.c_uint32(read_len),
            _ptr(buffers["keys"]), _ptr(buffers["ranges"]), _ptr(buffers["ids"]), ctypes.c_uint32(n), _ptr(buffers["counts"]),
            _ptr(buffers["temp"]), ctypes.c_uint64(buffers["temp"].numel()), _stream_ptr(self.device)))
        return buffers

    def residual_diagonals(self, ranges, ids, cap, seeds_per_read, seed_interval, seed_len, read_len, read_offsets=None, seed_intervals=None):
        """nvbio_fm_residual_diagonals: the first `cap` rows of every residual range located and turned into diagonal keys (duplicates of
        neighbouring seeds dropped) -> (keys int64 [n * cap], n_keys int32 [1] on the device)"""
        torch = _torch()
        n = int(ids.numel())
        keys = torch.empty(max(n * int(cap), 1), dtype=torch.int64, device=self.device)
        n_keys = torch.zeros(1, dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_residual_diagonals(
            self._h, _ptr(ranges), _ptr(ids), ctypes.c_uint32(n), ctypes.c_uint32(int(cap)), ctypes.c_uint32(seeds_per_read),
            ctypes.c_uint32(seed_interval), ctypes.c_uint32(seed_len), ctypes.c_uint32(read_len), _ptr(read_offsets), _ptr(seed_intervals),
            _ptr(keys), _ptr(n_keys), _stream_ptr(self.device)))
        return keys, n_keys

    def rank(self, rows, syms):
        torch = _torch()
        rows = _dev_tensor(rows, torch.int32, self.device)
        syms = _dev_tensor(syms, torch.uint8, self.device)
        out = torch.empty(rows.numel(), dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_rank(self._h, _ptr(rows), _ptr(syms), ctypes.c_uint32(rows.numel()), _ptr(out),
                                   _stream_ptr(self.device)))
        return out

    def rank4(self, rows):
        torch = _torch()
        rows = _dev_tensor(rows, torch.int32, self.device)
        out = torch.empty((rows.numel(), 4), dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_rank4(self._h, _ptr(rows), ctypes.c_uint32(rows.numel()), _ptr(out),
                                    _stream_ptr(self.device)))
        return out

    def basic_inv_psi(self, rows):
        torch = _torch()
        rows = _dev_tensor(rows, torch.int32, self.device)
        out = torch.empty(rows.numel(), dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_basic_inv_psi(self._h, _ptr(rows), ctypes.c_uint32(rows.numel()), _ptr(out),
                                            _stream_ptr(self.device)))
        return out

    def locate(self, rows):
        torch = _torch()
        rows = _dev_tensor(rows, torch.int32, self.device)
        pos = torch.empty(rows.numel(), dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_locate(self._h, _ptr(rows), ctypes.c_uint32(rows.numel()), _ptr(pos),
                                     _stream_ptr(self.device)))
        return pos

    def locate_ssa_iterator(self, rows):
        torch = _torch()
        rows = _dev_tensor(rows, torch.int32, self.device)
        jt = torch.empty((rows.numel(), 2), dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_locate_init(self._h, _ptr(rows), ctypes.c_uint32(rows.numel()), _ptr(jt),
                                          _stream_ptr(self.device)))
        return jt

    def lookup_ssa_iterator(self, jt):
        torch = _torch()
        pos = torch.empty(jt.shape[0], dtype=torch.int32, device=self.device)
        _check(lib().nvbio_fm_locate_lookup(self._h, _ptr(jt), ctypes.c_uint32(jt.shape[0]), _ptr(pos),
                                            _stream_ptr(self.device)))
        return pos


class FMIndexFilter:
    """nvbio::FMIndexFilter<device_tag, fm_index_type> (nvbio/fmindex/filter.h:136-231)."""

    def __init__(self):
        self._index = None
        self._ranges = self._slots = self._direct = None
        self._n_hits = 0
        self._n_queries = 0

    def rank(self, index, string_set, flags=0):
        """enact the filter; returns the total number of hits (filter_inl.h:261-293)"""
        torch = _torch()
        self._index, self._n_queries, self._direct = index, string_set.n, None
        self._ranges = torch.empty((string_set.n, 2), dtype=torch.int32, device=index.device)
        self._slots = torch.empty(string_set.n, dtype=torch.int64, device=index.device)
        total = ctypes.c_uint64(0)
        qs = string_set.c_struct()
        _check(lib().nvbio_fm_filter_rank(index._h, ctypes.byref(qs), ctypes.c_uint32(flags), _ptr(self._ranges),
                                          _ptr(self._slots), ctypes.byref(total), _stream_ptr(index.device)))
        self._n_hits = total.value
        return self._n_hits

    def rank_ranges(self, index, ranges, direct=None):
        """the scan half of rank() over ranges already computed by FMIndex.match (nvbio_fm_filter_scan), or by
        FMIndex.match_direct (then pass its `direct` flags: locate() copies the positions those ranges hold)"""
        torch = _torch()
        self._index, self._n_queries, self._ranges, self._direct = index, ranges.shape[0], ranges, direct
        self._slots = torch.empty(ranges.shape[0], dtype=torch.int64, device=index.device)
        total = ctypes.c_uint64(0)
        _check(lib().nvbio_fm_filter_scan(index._h, _ptr(ranges), ctypes.c_uint32(ranges.shape[0]), _ptr(self._slots),
                                          ctypes.byref(total), _stream_ptr(index.device)))
        self._n_hits = total.value
        return self._n_hits

    def locate(self, begin, end, hits=None):
        """hits[h-begin] = (text_pos, query_id) for hit indices [begin,end) (filter_inl.h:299-393)"""
        torch = _torch()
        if hits is None:
            hits = torch.empty((end - begin, 2), dtype=torch.int32, device=self._index.device)
        if self._direct is not None:
            _check(lib().nvbio_fm_filter_locate_direct(self._index._h, _ptr(self._ranges), _ptr(self._slots), _ptr(self._direct),
                                                       ctypes.c_uint32(self._n_queries), ctypes.c_uint64(begin),
                                                       ctypes.c_uint64(end), _ptr(hits), _stream_ptr(self._index.device)))
            return hits
        _check(lib().nvbio_fm_filter_locate(self._index._h, _ptr(self._ranges), _ptr(self._slots),
                                            ctypes.c_uint32(self._n_queries), ctypes.c_uint64(begin),
                                            ctypes.c_uint64(end), _ptr(hits), _stream_ptr(self._index.device)))
        return hits

    def locate_diagonals(self, begin, end, seeds_per_read, seed_interval, seed_len, read_len, strand, query_ids=None, read_offsets=None,
                         seed_intervals=None):
        """locate() and hits_to_diagonals() in one pass (nvbio_fm_filter_locate_diagonals): int64 keys
        read << 34 | strand << 33 | diagonal + 1024 of hit indices [begin, end); query_ids: the seed id of every range when
        the ranges are a compacted subset (FMIndex.match_seed_diagonals' residual list)"""
        torch = _torch()
        keys = torch.empty(end - begin, dtype=torch.int64, device=self._index.device)
        if read_offsets is not None:                 # ragged reads: every read's length and seed interval
            _check(lib().nvbio_fm_filter_locate_diagonals_ragged(
                self._index._h, _ptr(self._ranges), _ptr(self._slots), _ptr(self._direct), ctypes.c_uint32(self._n_queries),
                ctypes.c_uint64(begin), ctypes.c_uint64(end), ctypes.c_uint32(seeds_per_read), ctypes.c_uint32(seed_len), _ptr(read_offsets),
                _ptr(seed_intervals), ctypes.c_uint32(strand), _ptr(query_ids), _ptr(keys), _stream_ptr(self._index.device)))
            return keys
        _check(lib().nvbio_fm_filter_locate_diagonals(
            self._index._h, _ptr(self._ranges), _ptr(self._slots), _ptr(self._direct), ctypes.c_uint32(self._n_queries),
            ctypes.c_uint64(begin), ctypes.c_uint64(end), ctypes.c_uint32(seeds_per_read), ctypes.c_uint32(seed_interval),
            ctypes.c_uint32(seed_len), ctypes.c_uint32(read_len), ctypes.c_uint32(strand), _ptr(query_ids), _ptr(keys),
            _stream_ptr(self._index.device)))
        return keys

    def n_hits(self):
        return self._n_hits

    def ranges(self):
        return self._ranges

    def slots(self):
        return self._slots


# ---- alignment ---------------------------------------------------------------------------------
class GotohScheme:
    """the Gotoh scoring-scheme concept as data (see nvbio_gotoh_scheme in include/nvbio_amd.h)"""

    def __init__(self, match, mm_min, mm_max, pat_gap_open, pat_gap_ext, txt_gap_open, txt_gap_ext):
        self.c = _Scheme(match, mm_min, mm_max, pat_gap_open, pat_gap_ext, txt_gap_open, txt_gap_ext)


def SimpleGotohScheme(match, mismatch, gap_open, gap_ext):
    """aln::SimpleGotohScheme (nvbio/alignment/utils.h:103-123)"""
    return GotohScheme(match, -mismatch, -mismatch, gap_open, gap_ext, gap_open, gap_ext)


def EditDistanceScheme():
    """the scheme under which the Gotoh kernels compute the reference's edit-distance aligner: EditDistanceSWScheme
    (match 0, mismatch -1, insertion = deletion = -1; nvbio/alignment/ed/ed_banded_inl.h:37-69 runs the banded
    linear-gap SW with it), and with open = extension the affine recurrences give the same H in every cell -- pinned
    against the reference in tests/golden/ed_golden.npz"""
    return GotohScheme(0, 1, 1, -1, -1, -1, -1)


class SimpleSmithWatermanScheme:
    """aln::SimpleSmithWatermanScheme(match, mismatch, deletion, insertion) (nvbio/alignment/utils.h:81-98), signed scores"""

    def __init__(self, match, mismatch, deletion, insertion):
        self.c = _SWScheme(match, mismatch, deletion, insertion)


class SmithWatermanAligner:
    """aln::SmithWatermanAligner<TYPE, scheme> (nvbio/alignment/alignment.h:508-545): linear gaps.  `.sw` is what the
    scoring entry points take (nvbio_banded_sw_score / nvbio_full_sw_score); `.scheme` is the Gotoh form with open =
    extension, which exists when deletion == insertion and is what the banded traceback takes."""

    def __init__(self, type, scheme):
        self.type, self.sw = type, scheme
        c = scheme.c
        self.scheme = (GotohScheme(c.match, -c.mismatch, -c.mismatch, c.deletion, c.deletion, c.deletion, c.deletion)
                       if c.deletion == c.insertion else None)


def make_smith_waterman_aligner(type, scheme):
    """aln::make_smith_waterman_aligner<TYPE>(scheme) (nvbio/alignment/alignment.h:529)"""
    return SmithWatermanAligner(type, scheme)


def make_edit_distance_aligner(type):
    """aln::make_edit_distance_aligner<TYPE>() (nvbio/alignment/alignment.h:382): the Smith-Waterman aligner with
    EditDistanceSWScheme (0, -1, -1, -1) (ed/ed_utils.h:36-43) -- the aligner of examples/fmmap/fmmap.cu:346-359 and of
    nvBowtie --scoring ed (banded), and the full-matrix one of ed/ed_inl.h"""
    return SmithWatermanAligner(type, SimpleSmithWatermanScheme(0, -1, -1, -1))


class GotohAligner:
    """aln::GotohAligner<TYPE, scheme> (nvbio/alignment/alignment.h:437-449)"""

    def __init__(self, type, scheme):
        self.type, self.scheme = type, scheme


def make_gotoh_aligner(type, scheme):
    return GotohAligner(type, scheme)


class AlignmentBatch:
    """The flattened stream of alignment jobs (see nvbio_alignment_batch)."""

    def __init__(self, reads, read_bits, read_offsets, text, text_bits, win_begin, win_end, quals=None, read_id=None,
                 flags=None, device="cuda:0", max_read_len=0, algo_flags=None):
        torch = _torch()
        self.device = device
        self.algo_flags = DEFAULT_ALGO_FLAGS if algo_flags is None else int(algo_flags)
        self.read_bits, self.text_bits = int(read_bits), int(text_bits)
        self.reads = _dev_tensor(reads, torch.uint8 if read_bits == 8 else torch.int32, device)
        self.read_offsets = _dev_tensor(read_offsets, torch.int32, device)
        self.text = _dev_tensor(text, torch.uint8 if text_bits == 8 else torch.int32, device)
        self.win_begin = _dev_tensor(win_begin, torch.int32, device)
        self.win_end = _dev_tensor(win_end, torch.int32, device)
        self.quals = _dev_tensor(quals, torch.uint8, device)
        self.read_id = _dev_tensor(read_id, torch.int32, device)
        self.flags = _dev_tensor(flags, torch.uint8, device)
        self.n = int(self.win_begin.numel())
        self.max_read_len = int(max_read_len)       # max_pattern_length() of the stream concept; 0 = unknown

    def size(self):
        return self.n

    def c_struct(self):
        return _Batch(_ptr(self.reads), self.read_bits, _ptr(self.read_offsets), _ptr(self.quals), _ptr(self.read_id),
                      _ptr(self.flags), _ptr(self.text), self.text_bits, _ptr(self.win_begin), _ptr(self.win_end),
                      self.n, self.max_read_len, self.algo_flags)


DEVICE_THREAD_SCHEDULER = "DeviceThreadScheduler"
DEVICE_STAGED_THREAD_SCHEDULER = "DeviceStagedThreadScheduler"     # nvbio/alignment/batched.h: the 32-row-window work queue


class BatchedBandedAlignmentScore:
    """aln::BatchedBandedAlignmentScore<BAND_LEN, stream, scheduler> (nvbio/alignment/batched.h:298,
    batched_banded_inl.h:90-157): enact() scores every job of the stream.  scheduler =
    DEVICE_STAGED_THREAD_SCHEDULER gives the results of the staged specialization (batched_banded_inl.h:165-236): 32-row windows
    with the min_score exit; min_scores (int32 per job, or one int) is then the stream's context->min_score."""

    def __init__(self, band_len, aligner, scheduler=DEVICE_THREAD_SCHEDULER):
        self.band_len, self.aligner, self.scheduler = int(band_len), aligner, scheduler
        if scheduler not in (DEVICE_THREAD_SCHEDULER, DEVICE_STAGED_THREAD_SCHEDULER):
            raise NvbioError(1, "unknown scheduler %r" % (scheduler,))

    def min_temp_storage(self, max_pattern_len, max_text_len, stream_size):
        # as the reference's Host/Device thread schedulers (batched_banded_inl.h:100-104); its staged scheduler asks for one band of
        # short2 checkpoints per queue slot (:176-191) -- this library keeps the band in registers and needs none
        return 0

    max_temp_storage = min_temp_storage

    def enact(self, batch, scores=None, sinks=None, min_scores=None):
        torch = _torch()
        if scores is None:
            scores = torch.empty(batch.n, dtype=torch.int32, device=batch.device)
        if sinks is None:
            sinks = torch.empty((batch.n, 2), dtype=torch.int32, device=batch.device)
        bs = batch.c_struct()
        sw = getattr(self.aligner, "sw", None)
        if self.scheduler == DEVICE_STAGED_THREAD_SCHEDULER:
            if sw is not None:
                raise NvbioError(4, "the staged scheduler is instantiated for Gotoh aligners")
            per_job = min_scores is not None and not isinstance(min_scores, int)
            if per_job and (min_scores.dtype != torch.int32 or min_scores.numel() != batch.n):
                raise NvbioError(1, "min_scores: int32 per job")
            _check(lib().nvbio_banded_gotoh_score_staged(
                FMIndex._dev_index(batch.device), ctypes.c_uint32(self.band_len), ctypes.c_int(self.aligner.type),
                ctypes.byref(self.aligner.scheme.c), ctypes.byref(bs), _ptr(min_scores) if per_job else None,
                ctypes.c_int32(SCORE_MIN if min_scores is None else (0 if per_job else int(min_scores))),
                _ptr(scores), _ptr(sinks), _stream_ptr(batch.device)))
            return scores, sinks
        fn = lib().nvbio_banded_sw_score if sw is not None else lib().nvbio_banded_gotoh_score
        _check(fn(FMIndex._dev_index(batch.device), ctypes.c_uint32(self.band_len), ctypes.c_int(self.aligner.type),
                  ctypes.byref(sw.c if sw is not None else self.aligner.scheme.c),
                  ctypes.byref(bs), _ptr(scores), _ptr(sinks), _stream_ptr(batch.device)))
        return scores, sinks


def batch_banded_alignment_score(band_len, aligner, batch):
    """aln::batch_banded_alignment_score<BAND_LEN> (nvbio/alignment/batched.h:185, batched_inl.h:1046-1086)"""
    return BatchedBandedAlignmentScore(band_len, aligner).enact(batch)


class BatchedBandedAlignmentTraceback:
    """aln::BatchedBandedAlignmentTraceback<BAND_LEN,CHECKPOINTS,stream> (nvbio/alignment/batched.h,
    batched_banded_inl.h) with nvBowtie's run-length Backtracker (alignment_utils.h:115-157) as the stream's
    backtracer: per job the Alignment {score, source, sink} and its io::Cigar elements in backtracking order"""

    def __init__(self, band_len, aligner):
        self.band_len, self.aligner = int(band_len), aligner

    def min_temp_storage(self, batch):
        out = ctypes.c_uint64(0)
        bs = batch.c_struct()
        _check(lib().nvbio_banded_gotoh_traceback_temp_bytes(ctypes.byref(bs), ctypes.c_uint32(self.band_len), ctypes.byref(out)))
        return int(out.value)

    def enact(self, batch, cigar_stride=64, temp=None, scores=None, sinks=None):
        """scores / sinks: optional results of BatchedBandedAlignmentScore for the same batch (the scoring pass is
        then skipped, NVBIO_TRACEBACK_SINKS_GIVEN)"""
        torch = _torch()
        n, dev = batch.n, batch.device
        given = scores is not None and sinks is not None
        if not given:
            scores = torch.empty(n, dtype=torch.int32, device=dev)
            sinks = torch.empty((n, 2), dtype=torch.int32, device=dev)
        sources = torch.empty((n, 2), dtype=torch.int32, device=dev)
        cigars = torch.zeros((n, cigar_stride), dtype=torch.int16, device=dev)
        lens = torch.empty(n, dtype=torch.int32, device=dev)
        bs = batch.c_struct()
        sw = getattr(self.aligner, "sw", None)                      # the linear-gap Smith-Waterman / edit-distance aligners
        fn = lib().nvbio_banded_sw_traceback if sw is not None else lib().nvbio_banded_gotoh_traceback
        _check(fn(
            FMIndex._dev_index(dev), ctypes.c_uint32(self.band_len), ctypes.c_int(self.aligner.type),
            ctypes.byref(sw.c if sw is not None else self.aligner.scheme.c), ctypes.byref(bs), _ptr(scores), _ptr(sources), _ptr(sinks), _ptr(cigars),
            ctypes.c_uint32(cigar_stride), _ptr(lens), ctypes.c_uint32(TRACEBACK_SINKS_GIVEN if given else 0), _ptr(temp),
            ctypes.c_uint64(0 if temp is None else temp.numel() * temp.element_size()), _stream_ptr(dev)))
        return scores, sources, sinks, cigars, lens


class BatchedAlignmentTraceback:
    """aln::BatchedAlignmentTraceback<CHECKPOINTS,stream> (full-matrix DP; nvbio/alignment/batched.h:395-411) with
    nvBowtie's run-length Backtracker: Alignment {score, source, sink} (x = text, y = pattern) and io::Cigar runs"""

    def __init__(self, aligner):
        self.aligner = aligner

    def min_temp_storage(self, batch, max_pattern_len, max_text_len):
        out = ctypes.c_uint64(0)
        bs = batch.c_struct()
        _check(lib().nvbio_full_gotoh_traceback_temp_bytes(ctypes.byref(bs), ctypes.c_uint32(max_pattern_len),
                                                           ctypes.c_uint32(max_text_len), ctypes.byref(out)))
        return int(out.value)

    def enact(self, batch, max_pattern_len, max_text_len, min_scores=None, cigar_stride=64, temp=None, scores=None, sinks=None):
        torch = _torch()
        n, dev = batch.n, batch.device
        given = scores is not None and sinks is not None
        if not given:
            scores = torch.empty(n, dtype=torch.int32, device=dev)
            sinks = torch.empty((n, 2), dtype=torch.int32, device=dev)
        sources = torch.empty((n, 2), dtype=torch.int32, device=dev)
        cigars = torch.zeros((n, cigar_stride), dtype=torch.int16, device=dev)
        lens = torch.empty(n, dtype=torch.int32, device=dev)
        ms = _dev_tensor(min_scores, torch.int32, dev)
        bs = batch.c_struct()
        if temp is None and n:
            temp = _scratch(dev, self.min_temp_storage(batch, max_pattern_len, max_text_len))
        sw = getattr(self.aligner, "sw", None)                      # the linear-gap Smith-Waterman / edit-distance aligners
        fn = lib().nvbio_full_sw_traceback if sw is not None else lib().nvbio_full_gotoh_traceback
        _check(fn(
            FMIndex._dev_index(dev), ctypes.c_int(self.aligner.type), ctypes.byref(sw.c if sw is not None else self.aligner.scheme.c), ctypes.byref(bs),
            ctypes.c_uint32(max_pattern_len), ctypes.c_uint32(max_text_len), _ptr(ms), _ptr(scores), _ptr(sources), _ptr(sinks),
            _ptr(cigars), ctypes.c_uint32(cigar_stride), _ptr(lens), ctypes.c_uint32(TRACEBACK_SINKS_GIVEN if given else 0),
            _ptr(temp), ctypes.c_uint64(0 if temp is None else temp.numel() * temp.element_size()), _stream_ptr(dev)))
        return scores, sources, sinks, cigars, lens


def finish_alignment(batch, sources, cigars, cigar_lens, mds_stride=0):
    """nvBowtie finish_alignment (traceback_inl.h:536-705) on the outputs of a traceback: (ed, mds, mds_lens) --
    edit distance per job and, with mds_stride > 0, the MDS byte streams [n, mds_stride]"""
    torch = _torch()
    n, dev = batch.n, batch.device
    ed = torch.empty(n, dtype=torch.int32, device=dev)
    mds = torch.zeros((n, mds_stride), dtype=torch.uint8, device=dev) if mds_stride else None
    ml = torch.empty(n, dtype=torch.int32, device=dev) if mds_stride else None
    bs = batch.c_struct()
    _check(lib().nvbio_finish_alignment(FMIndex._dev_index(dev), ctypes.byref(bs), _ptr(sources), _ptr(cigars), ctypes.c_uint32(cigars.shape[1]),
                                        _ptr(cigar_lens), _ptr(ed), _ptr(mds), ctypes.c_uint32(mds_stride), _ptr(ml), _stream_ptr(dev)))
    return ed, mds, ml


def cigar_string(cigar_row, length, forward=True):
    """render one alignment's io::Cigar elements ('3M2D147M'); forward=True reverses the backtracking order"""
    els = [int(c) & 0xFFFF for c in cigar_row[:length]]
    if forward:
        els = els[::-1]
    return "".join("%d%s" % (c >> 2, "MIDS"[c & 3]) for c in els)


class BatchedAlignmentScore:
    """aln::BatchedAlignmentScore<stream, scheduler> (full-matrix DP; batched.h:274, batched_inl.h:221-592)"""

    def __init__(self, aligner, text_blocking=True):
        self.aligner, self.text_blocking = aligner, bool(text_blocking)

    def enact(self, batch, max_pattern_len, max_text_len, min_scores=None, scores=None, sinks=None):
        torch = _torch()
        if scores is None:
            scores = torch.empty(batch.n, dtype=torch.int32, device=batch.device)
        if sinks is None:
            sinks = torch.empty((batch.n, 2), dtype=torch.int32, device=batch.device)
        ms = _dev_tensor(min_scores, torch.int32, batch.device)
        bs = batch.c_struct()
        sw = getattr(self.aligner, "sw", None)
        fn = lib().nvbio_full_sw_score if sw is not None else lib().nvbio_full_gotoh_score
        rows = max_pattern_len if self.text_blocking else max_text_len
        temp = _scratch(batch.device, batch.n * max(rows, 1) * 4)
        _check(fn(FMIndex._dev_index(batch.device), ctypes.c_int(self.aligner.type),
                  ctypes.c_int(1 if self.text_blocking else 0),
                  ctypes.byref(sw.c if sw is not None else self.aligner.scheme.c), ctypes.byref(bs),
                  ctypes.c_uint32(max_pattern_len), ctypes.c_uint32(max_text_len), _ptr(ms),
                  _ptr(scores), _ptr(sinks), _ptr(temp), ctypes.c_uint64(temp.numel()), _stream_ptr(batch.device)))
        return scores, sinks


def batch_banded_alignment_score_best2(band_len, aligner, batch, distinct_dist=0):
    """banded scoring into aln::Best2Sink<int32>(distinct_dist) (nvbio/alignment/sink.h:96-116)
    -> (scores, sinks, scores2, sinks2)"""
    torch = _torch()
    sc = [torch.empty(batch.n, dtype=torch.int32, device=batch.device) for _ in range(2)]
    sk = [torch.empty((batch.n, 2), dtype=torch.int32, device=batch.device) for _ in range(2)]
    bs = batch.c_struct()
    _check(lib().nvbio_banded_gotoh_score_best2(FMIndex._dev_index(batch.device), ctypes.c_uint32(band_len), ctypes.c_int(aligner.type),
                                                ctypes.byref(aligner.scheme.c), ctypes.byref(bs), ctypes.c_uint32(distinct_dist),
                                                _ptr(sc[0]), _ptr(sk[0]), _ptr(sc[1]), _ptr(sk[1]), _stream_ptr(batch.device)))
    return sc[0], sk[0], sc[1], sk[1]


def batch_alignment_score_best2(aligner, batch, max_pattern_len, max_text_len, distinct_dist=0, text_blocking=False, min_scores=None):
    """full-matrix scoring into aln::Best2Sink<int32>(distinct_dist) -> (scores, sinks, scores2, sinks2)"""
    torch = _torch()
    sc = [torch.empty(batch.n, dtype=torch.int32, device=batch.device) for _ in range(2)]
    sk = [torch.empty((batch.n, 2), dtype=torch.int32, device=batch.device) for _ in range(2)]
    ms = _dev_tensor(min_scores, torch.int32, batch.device)
    bs = batch.c_struct()
    _check(lib().nvbio_full_gotoh_score_best2(FMIndex._dev_index(batch.device), ctypes.c_int(aligner.type), ctypes.c_int(1 if text_blocking else 0),
                                              ctypes.byref(aligner.scheme.c), ctypes.byref(bs), ctypes.c_uint32(max_pattern_len),
                                              ctypes.c_uint32(max_text_len), _ptr(ms), ctypes.c_uint32(distinct_dist),
                                              _ptr(sc[0]), _ptr(sk[0]), _ptr(sc[1]), _ptr(sk[1]), _stream_ptr(batch.device)))
    return sc[0], sk[0], sc[1], sk[1]


def hits_to_diagonals(hits, seeds_per_read, seed_interval, seed_len, read_len, strand):
    """hit_to_diagonal (examples/fmmap/fmmap.cu:92-117): int64 keys read<<34 | strand<<33 | diagonal+1024"""
    torch = _torch()
    keys = torch.empty(hits.shape[0], dtype=torch.int64, device=hits.device)
    _check(lib().nvbio_hits_to_diagonals(FMIndex._dev_index(hits.device), _ptr(hits), ctypes.c_uint64(hits.shape[0]),
                                         ctypes.c_uint32(seeds_per_read), ctypes.c_uint32(seed_interval),
                                         ctypes.c_uint32(seed_len), ctypes.c_uint32(read_len), ctypes.c_uint32(strand),
                                         _ptr(keys), _stream_ptr(hits.device)))
    return keys


def best_candidate_reduce(keys, scores, sinks, win_begin, best):
    """best[read] = max(best[read], selection key of each candidate) by 64-bit atomic max (nvbio_best_candidate_reduce);
    best: int64 tensor, zero-initialised by the caller, one entry per read"""
    _check(lib().nvbio_best_candidate_reduce(FMIndex._dev_index(keys.device), _ptr(keys), _ptr(scores), _ptr(sinks), _ptr(win_begin),
                                             ctypes.c_uint64(keys.shape[0]), _ptr(best), _stream_ptr(keys.device)))
    return best


def best_candidate_unpack(best):
    """(score int32, end position int64 or -1, strand uint8) per read from the keys of best_candidate_reduce"""
    torch = _torch()
    n, dev = best.shape[0], best.device
    score = torch.empty(n, dtype=torch.int32, device=dev); pos = torch.empty(n, dtype=torch.int64, device=dev)
    rc = torch.empty(n, dtype=torch.uint8, device=dev)
    _check(lib().nvbio_best_candidate_unpack(FMIndex._dev_index(dev), _ptr(best), ctypes.c_uint32(n), _ptr(score), _ptr(pos), _ptr(rc),
                                             _stream_ptr(dev)))
    return score, pos, rc


def best_candidate_windows(keys, scores, sinks, win_begin, best, best_wb, best_locus=None):
    """nvbio_best_candidate_windows: atomic max of the window begin (and locus) of the candidates that ARE their read's best (`best`:
    the final keys of best_candidate_reduce) into best_wb / best_locus (int64 per read, initialised to -1 by the caller)"""
    if keys.numel() == 0:
        return
    _check(lib().nvbio_best_candidate_windows(FMIndex._dev_index(keys.device), _ptr(keys), _ptr(scores), _ptr(sinks), _ptr(win_begin),
                                              ctypes.c_uint64(keys.numel()), _ptr(best), _ptr(best_wb), _ptr(best_locus),
                                              _stream_ptr(keys.device)))


def traceback_best_batch(best, best_wb, read_len, band, genome_len, min_score, read_offsets=None, min_scores=None):
    """nvbio_traceback_best_batch: the per-read arrays of the traceback batch of every read's best alignment ->
    (flags uint8, win_begin int32, win_end int32, scores int32, sinks int32 [R, 2]); unaligned reads get the empty window"""
    torch = _torch()
    n, dev = best.shape[0], best.device
    flags = torch.empty(n, dtype=torch.uint8, device=dev)
    wb = torch.empty(n, dtype=torch.int32, device=dev); we = torch.empty(n, dtype=torch.int32, device=dev)
    scores = torch.empty(n, dtype=torch.int32, device=dev); sinks = torch.empty((n, 2), dtype=torch.int32, device=dev)
    if read_offsets is not None:                     # ragged reads (read_len / min_score are ignored)
        _check(lib().nvbio_traceback_best_batch_ragged(FMIndex._dev_index(dev), _ptr(best), _ptr(best_wb), ctypes.c_uint32(n), _ptr(read_offsets),
                                                       ctypes.c_uint32(band), ctypes.c_uint32(genome_len), _ptr(min_scores), _ptr(flags),
                                                       _ptr(wb), _ptr(we), _ptr(scores), _ptr(sinks), _stream_ptr(dev)))
        return flags, wb, we, scores, sinks
    _check(lib().nvbio_traceback_best_batch(FMIndex._dev_index(dev), _ptr(best), _ptr(best_wb), ctypes.c_uint32(n), ctypes.c_uint32(read_len),
                                            ctypes.c_uint32(band), ctypes.c_uint32(genome_len), ctypes.c_int32(min_score), _ptr(flags),
                                            _ptr(wb), _ptr(we), _ptr(scores), _ptr(sinks), _stream_ptr(dev)))
    return flags, wb, we, scores, sinks


PE_POLICY_FF, PE_POLICY_FR, PE_POLICY_RF, PE_POLICY_RR = 0, 1, 2, 3


def max_text_gaps(scheme, min_score, pattern_len):
    """aln::max_text_gaps for a Gotoh aligner (nvbio/alignment/utils_inl.h:141-162): the largest number of reference
    gaps an alignment of pattern_len symbols can hold without dropping below min_score"""
    c = scheme.c
    score = pattern_len * c.match
    if score < min_score:
        return 0
    score += c.txt_gap_open
    n = 0
    while score >= min_score and n < pattern_len:
        score += c.txt_gap_ext
        n += 1
    return (n - 1) & 0xFFFFFFFF


class _MapqParams(ctypes.Structure):
    _fields_ = [("version", ctypes.c_int32), ("monotone", ctypes.c_int32), ("perfect_score", ctypes.c_int32),
                ("min_score", ctypes.c_int32)]


def second_candidate_reduce(keys, scores, sinks, wb, best, distinct_dist, worst_score, second, read_offsets=None, min_scores=None):
    """nvBowtie's second-best alignment per read (score_reduce, reduce_inl.h:65-140; see nvbio_second_candidate_reduce):
    best must already hold the final per-read maxima over ALL candidates; second is zero-initialised by the caller"""
    if read_offsets is not None:                     # ragged reads: distinct_dist = read_len / 2 and worst_score = min_score - 1 per read
        _check(lib().nvbio_second_candidate_reduce_ragged(FMIndex._dev_index(keys.device), _ptr(keys), _ptr(scores), _ptr(sinks), _ptr(wb),
                                                          ctypes.c_uint64(keys.shape[0]), _ptr(best), _ptr(read_offsets), _ptr(min_scores),
                                                          _ptr(second), _stream_ptr(keys.device)))
        return second
    _check(lib().nvbio_second_candidate_reduce(FMIndex._dev_index(keys.device), _ptr(keys), _ptr(scores), _ptr(sinks), _ptr(wb),
                                               ctypes.c_uint64(keys.shape[0]), _ptr(best), ctypes.c_uint32(distinct_dist),
                                               ctypes.c_int32(worst_score), _ptr(second), _stream_ptr(keys.device)))
    return second


def mapq(best, second, perfect_score, min_score, monotone, version=2, read_offsets=None, min_scores=None, match=0):
    """Bowtie2's mapping quality per read from the best / second-best selection keys (BowtieMapq2 / BowtieMapq3,
    nvBowtie/bowtie2/cuda/mapq.h) -> (mapq uint8 [R], second_score int32 [R])"""
    torch = _torch()
    R = best.shape[0]
    q = torch.empty(R, dtype=torch.uint8, device=best.device)
    ss = torch.empty(R, dtype=torch.int32, device=best.device)
    if read_offsets is not None:                     # ragged reads: perfect score = match x length, min score per read
        _check(lib().nvbio_mapq_ragged(FMIndex._dev_index(best.device), _ptr(best), _ptr(second) if second is not None else None,
                                       ctypes.c_uint32(R), ctypes.c_int32(int(version)), ctypes.c_int32(int(match)), _ptr(read_offsets),
                                       _ptr(min_scores), _ptr(ss), _ptr(q), _stream_ptr(best.device)))
        return q, ss
    prm = _MapqParams(int(version), 1 if monotone else 0, int(perfect_score), int(min_score))
    _check(lib().nvbio_mapq(FMIndex._dev_index(best.device), _ptr(best), _ptr(second) if second is not None else None,
                            ctypes.c_uint32(R), ctypes.byref(prm), _ptr(ss), _ptr(q), _stream_ptr(best.device)))
    return q, ss


def opposite_mate_windows(g_pos, anchor_rc, anchor_len, opposite_gapped_len, anchor, genome_len, policy=PE_POLICY_FR,
                          min_frag_len=0, max_frag_len=500, overlap=True):
    """BestOppositeScoreStream::init_context's window (nvBowtie score_inl.h:389-425): (win_begin, win_end, flags, valid)"""
    torch = _torch()
    n, dev = g_pos.shape[0], g_pos.device
    wb = torch.empty(n, dtype=torch.int32, device=dev); we = torch.empty(n, dtype=torch.int32, device=dev)
    flags = torch.empty(n, dtype=torch.uint8, device=dev); valid = torch.empty(n, dtype=torch.uint8, device=dev)
    _check(lib().nvbio_opposite_mate_windows(
        FMIndex._dev_index(dev), _ptr(g_pos), _ptr(anchor_rc), ctypes.c_uint32(n), ctypes.c_uint32(anchor_len),
        ctypes.c_uint32(opposite_gapped_len), ctypes.c_uint32(anchor), ctypes.c_uint32(policy), ctypes.c_uint32(min_frag_len),
        ctypes.c_uint32(max_frag_len), ctypes.c_uint32(1 if overlap else 0), ctypes.c_uint32(genome_len), _ptr(wb), _ptr(we),
        _ptr(flags), _ptr(valid), _stream_ptr(dev)))
    return wb, we, flags, valid


def diagonals_to_windows(keys, band, read_len, genome_len, read_offsets=None):
    """genome_infixes + nvBowtie's window rule: (read_id, flags, win_begin, win_end) of every candidate key"""
    torch = _torch()
    n, dev = keys.numel(), keys.device
    rid = torch.empty(n, dtype=torch.int32, device=dev)
    fl = torch.empty(n, dtype=torch.uint8, device=dev)
    wb = torch.empty(n, dtype=torch.int32, device=dev)
    we = torch.empty(n, dtype=torch.int32, device=dev)
    if read_offsets is not None:                     # ragged reads: window end = begin + band + the read's own length
        _check(lib().nvbio_diagonals_to_windows_ragged(FMIndex._dev_index(dev), _ptr(keys), ctypes.c_uint64(n), ctypes.c_uint32(band),
                                                       _ptr(read_offsets), ctypes.c_uint32(genome_len), _ptr(rid), _ptr(fl),
                                                       _ptr(wb), _ptr(we), _stream_ptr(dev)))
        return rid, fl, wb, we
    _check(lib().nvbio_diagonals_to_windows(FMIndex._dev_index(dev), _ptr(keys), ctypes.c_uint64(n), ctypes.c_uint32(band),
                                            ctypes.c_uint32(read_len), ctypes.c_uint32(genome_len), _ptr(rid), _ptr(fl),
                                            _ptr(wb), _ptr(we), _stream_ptr(dev)))
    return rid, fl, wb, we


def u32(t):
    """int32 device/host tensor -> numpy uint32 (results are uint32 bit patterns)"""
    return t.detach().cpu().numpy().view(np.uint32)


# ---- nvBowtie's scoring stream as data (nvbio_hit_queues) ------------------------------------------------------------------------
class HitQueues:
    """the members of nvBowtie's scoring pipeline a BestScoreStream reads and writes (pipeline_states.h:49-115,
    scoring_queues.h:211-289), as int32 device tensors: idx_queue (or None), read_id, seed (packed_seed words), loc, score, sink"""

    def __init__(self, read_id, seed, loc, idx_queue=None, device="cuda:0"):
        torch = _torch()
        self.device = device
        self.read_id = _dev_tensor(read_id, torch.int32, device)
        self.seed = _dev_tensor(seed, torch.int32, device)
        self.loc = _dev_tensor(loc, torch.int32, device)
        self.idx_queue = _dev_tensor(idx_queue, torch.int32, device)
        self.n = int(self.idx_queue.numel() if self.idx_queue is not None else self.read_id.numel())     # may be lowered: the first n slots count
        self.score = torch.zeros(self.read_id.numel(), dtype=torch.int32, device=device)
        self.sink = torch.zeros(self.read_id.numel(), dtype=torch.int32, device=device)

    def c_struct(self):
        return _HitQueues(_ptr(self.idx_queue), _ptr(self.read_id), _ptr(self.seed), _ptr(self.loc), _ptr(self.score), _ptr(self.sink), self.n)


def score_stream_flatten(hits, read_index, band_len, genome_len, reads_reversed=True):
    """BestScoreStream::init_context + load_strings' orientation for the whole stream (nvbio_score_stream_flatten) ->
    (read_id, flags, win_begin, win_end): the per-job arrays of an AlignmentBatch"""
    torch = _torch()
    dev = hits.device
    ri = _dev_tensor(read_index, torch.int32, dev)
    rid = torch.empty(hits.n, dtype=torch.int32, device=dev)
    flags = torch.empty(hits.n, dtype=torch.uint8, device=dev)
    wb = torch.empty(hits.n, dtype=torch.int32, device=dev)
    we = torch.empty(hits.n, dtype=torch.int32, device=dev)
    hq = hits.c_struct()
    _check(lib().nvbio_score_stream_flatten(FMIndex._dev_index(dev), ctypes.byref(hq), _ptr(ri), ctypes.c_uint32(band_len),
                                            ctypes.c_uint32(genome_len), ctypes.c_uint32(1 if reads_reversed else 0), _ptr(rid), _ptr(flags),
                                            _ptr(wb), _ptr(we), _stream_ptr(dev)))
    return rid, flags, wb, we


def score_stream_output(hits, scores, sinks, win_begin, worst_score=-65536):
    """BestScoreStream::output for the whole stream (nvbio_score_stream_output): fills hits.score / hits.sink"""
    hq = hits.c_struct()
    _check(lib().nvbio_score_stream_output(FMIndex._dev_index(hits.device), ctypes.byref(hq), _ptr(scores), _ptr(sinks), _ptr(win_begin),
                                           ctypes.c_int32(worst_score), _stream_ptr(hits.device)))


# ---- nvBowtie's seed-hit deques, selection and effort-limited reduction (include/nvbio_amd.h: nvbio_seed_hits_*) -----------------
class SeedHitsParams:
    """nvbio_seed_hits_params with nvBowtie's defaults (bowtie2_cuda_driver.cu:86-141)"""

    def __init__(self, seeds_per_read, seed_interval, seed_len, read_len, first_offset=0, max_hits=100, rep_seeds=1000, max_effort=15,
                 min_ext=30, max_ext=400):
        self.c = _SeedHitsParams(seeds_per_read, first_offset, seed_interval, seed_len, read_len, max_hits, rep_seeds, max_effort, min_ext, max_ext)

    def capacity(self):
        cap = ctypes.c_uint32(0)
        _check(lib().nvbio_seed_hits_capacity(self.c.seeds_per_read, self.c.max_hits, ctypes.byref(cap)))
        return cap.value


def seed_hits_map(fw_ranges, rc_ranges, params, n_reads, deques, sizes, reseed=None, read_queue=None):
    """the exact seed mapper's deque bookkeeping (nvbio_seed_hits_map): fills deques [R, capacity, 2], sizes [R], reseed [R] for the
    n_reads reads of read_queue (None: reads 0..n_reads-1)"""
    dev = deques.device
    _check(lib().nvbio_seed_hits_map(FMIndex._dev_index(dev), _ptr(fw_ranges), _ptr(rc_ranges), _ptr(read_queue), ctypes.c_uint32(n_reads),
                                     ctypes.byref(params.c), _ptr(deques), _ptr(sizes), _ptr(reseed), _stream_ptr(dev)))


def seed_hits_approx_capacity(params):
    cap = ctypes.c_uint32(0)
    _check(lib().nvbio_seed_hits_approx_capacity(params.c.seeds_per_read, params.c.seed_len, params.c.max_hits, ctypes.byref(cap)))
    return cap.value


def seed_hits_map_approx(fmi, rfmi, reads, read_bits, params, n_reads, deques, sizes, reseed=None, read_queue=None):
    """nvBowtie's approximate seed mapper (nvbio_seed_hits_map_approx): fmi = the forward index, rfmi = the index of the reversed text;
    reads = the stored (reversed) reads of params.read_len symbols back to back; deques [R, seed_hits_approx_capacity, 2]"""
    dev = deques.device
    _check(lib().nvbio_seed_hits_map_approx(fmi._h, rfmi._h, _ptr(reads), ctypes.c_uint32(read_bits), _ptr(read_queue), ctypes.c_uint32(n_reads),
                                            ctypes.byref(params.c), _ptr(deques), _ptr(sizes), _ptr(reseed), _stream_ptr(dev)))


def seed_hits_select(active, trys, params, deques, sizes, hits, active_out, count):
    """select_kernel (nvbio_seed_hits_select): active int32 [n] (read | top_flag << 31); fills hits.read_id / loc / seed and active_out;
    count int32 [1] on the device receives the number of slots written"""
    dev = deques.device
    hq = hits.c_struct()
    _check(lib().nvbio_seed_hits_select(FMIndex._dev_index(dev), _ptr(active), ctypes.c_uint32(active.numel()), _ptr(trys),
                                        ctypes.c_uint32(params.capacity()), _ptr(deques), _ptr(sizes), _ptr(active_out), ctypes.byref(hq),
                                        _ptr(count), _stream_ptr(dev)))


def seed_hits_loc(positions, hits):
    hq = hits.c_struct()
    _check(lib().nvbio_seed_hits_loc(FMIndex._dev_index(hits.device), _ptr(positions), ctypes.byref(hq), _stream_ptr(hits.device)))


def score_reduce_effort(active, hits, read_len, n_ext, params, best, best_rc, trys, sizes):
    """score_reduce_kernel with the best-approx effort rules (nvbio_score_reduce_effort): best int32 [R, 4] = (a1 score, a1 locus,
    a2 score, a2 locus), best_rc uint8 [R], trys int32 [R], sizes int32 [R] (erased deques get 0)"""
    hq = hits.c_struct()
    _check(lib().nvbio_score_reduce_effort(FMIndex._dev_index(hits.device), _ptr(active), ctypes.byref(hq), ctypes.c_uint32(read_len),
                                           ctypes.c_uint32(n_ext), ctypes.byref(params.c), _ptr(best), _ptr(best_rc), _ptr(trys), _ptr(sizes),
                                           _stream_ptr(hits.device)))


# ---- the generic rank dictionary (nvbio_rank_dictionary_*) -------------------------------------------------------------------------
class RankDictionary:
    """nvbio::rank_dictionary<2, K, PackedStream<const uint32*|const uint64*, uint8, 2, true>, occ, count_table> over plain word storage
    with a separate occurrence table and 32- or 64-bit indices (nvbio/fmindex/rank_dictionary_inl.h:206-336).  text: int32 tensor of
    32-bit words or int64 tensor of 64-bit words (bit patterns), on the device."""

    def __init__(self, text, length, K, index_bits):
        torch = _torch()
        self.text, self.length, self.K, self.index_bits = text, int(length), int(K), int(index_bits)
        self.word_bits = 32 if text.dtype == torch.int32 else 64
        self.device = text.device
        d = self._c(None)
        n = ctypes.c_uint64(0)
        _check(lib().nvbio_rank_dictionary_occ_entries(ctypes.byref(d), ctypes.byref(n)))
        self.occ = torch.zeros(max(n.value, 4), dtype=torch.int32 if index_bits == 32 else torch.int64, device=self.device)
        counts = (ctypes.c_uint64 * 4)()
        _check(lib().nvbio_rank_dictionary_build(FMIndex._dev_index(self.device), ctypes.byref(d), _ptr(self.occ), counts, _stream_ptr(self.device)))
        self.counts = [int(c) for c in counts]

    def _c(self, occ):
        return _RankDict(_ptr(self.text), self.word_bits, _ptr(occ), self.index_bits, self.K, self.length)

    def rank(self, idx, syms):
        """idx: tensor of index_bits-wide indices; syms uint8 -> ranks (same dtype as idx)"""
        torch = _torch()
        out = torch.empty_like(idx)
        d = self._c(self.occ)
        _check(lib().nvbio_rank_dictionary_rank(FMIndex._dev_index(self.device), ctypes.byref(d), _ptr(idx), _ptr(syms), ctypes.c_uint32(idx.numel()),
                                                _ptr(out), _stream_ptr(self.device)))
        return out

    def rank4(self, idx):
        torch = _torch()
        out = torch.empty((idx.numel(), 4), dtype=idx.dtype, device=idx.device)
        d = self._c(self.occ)
        _check(lib().nvbio_rank_dictionary_rank4(FMIndex._dev_index(self.device), ctypes.byref(d), _ptr(idx), ctypes.c_uint32(idx.numel()), _ptr(out),
                                                 _stream_ptr(self.device)))
        return out


def batch_banded_myers_score(band, aln_type, batch, min_score=SCORE_MIN):
    """aln::batch_banded_alignment_score<BAND>( make_edit_distance_aligner<TYPE, MyersTag<5>>(), ... ) (nvbio_banded_myers_score): the aligner of
    examples/fmmap -> (scores = -(edit distance), sinks)"""
    torch = _torch()
    scores = torch.empty(batch.n, dtype=torch.int32, device=batch.device)
    sinks = torch.empty((batch.n, 2), dtype=torch.int32, device=batch.device)
    bs = batch.c_struct()
    _check(lib().nvbio_banded_myers_score(FMIndex._dev_index(batch.device), ctypes.c_uint32(band), ctypes.c_int(aln_type), ctypes.byref(bs),
                                          ctypes.c_int32(min_score), _ptr(scores), _ptr(sinks), _stream_ptr(batch.device)))
    return scores, sinks
