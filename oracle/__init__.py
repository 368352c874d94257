"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for the plain-C restatement of the reference's CPU algorithm
(oracle/nvbio_oracle.c) and, when it has been built in the development container,
for the reference's own host code (oracle/_ref/libnvbio_ref.so, built by oracle/Makefile
from the sources under /root/reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (nvbio-gpl_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

GLOBAL, LOCAL, SEMI_GLOBAL = 0, 1, 2
SCORE_MIN = -(1 << 30)

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _c8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def _c32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint32)


# the banded edit-distance aligner is the banded linear-gap Smith-Waterman with (match 0, mismatch -1, gap -1)
# (ed/ed_banded_inl.h:37-69, EditDistanceSWScheme); with equal open and extension costs the Gotoh recurrences
# produce the same H in every cell (F <= H and E <= H make "extend" never better than "open"), hence the same
# scores and sinks -- checked against the reference in tests/test_oracle_vs_reference.py and pinned in ed_golden.npz
ED_SCHEME = (0, 1, 1, -1, -1, -1, -1)
ED_SW = (0, -1, -1, -1)      # EditDistanceSWScheme (ed/ed_utils.h:36-43) as (match, mismatch, deletion, insertion)


def cigar_from_ops(ops, clip_before, clip_after):
    """nvBowtie's Backtracker (alignment_utils.h:115-157) over a recorded op string: run-length io::Cigar
    elements (type | len << 2) in backtracking order, soft clips only when non-zero"""
    out = []
    if clip_before:
        out.append(3 | (clip_before << 2))
    prev = 255
    for op in ops:
        op = int(op)
        if op == prev:
            out[-1] += 4
        else:
            out.append(op | (1 << 2)); prev = op
    if clip_after:
        out.append(3 | (clip_after << 2))
    return np.array(out, dtype=np.uint16)


def cigar_string(cigar, reverse=False):
    """'147M2D3M'-style rendering of io::Cigar elements (backtracking order unless reverse)"""
    els = list(cigar)[::-1] if reverse else list(cigar)
    return "".join("%d%s" % (int(c) >> 2, "MIDS"[int(c) & 3]) for c in els)


def build(force=False):
    """compile liboracle.so (and _ref/libnvbio_ref.so when /root/reference exists)"""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("nvbio_oracle.c", "nvbio_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libnvbio_ref.so")
    ref_src = os.path.join(_HERE, "ref", "nvbio_ref.cpp")
    if os.path.isdir("/root/reference/nvbio") and (
            force or not os.path.exists(ref_so) or os.path.getmtime(ref_so) < os.path.getmtime(ref_src)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class Scheme(ctypes.Structure):
    """orc_gotoh_scheme"""
    _fields_ = [("match", ctypes.c_int32), ("mm_min", ctypes.c_int32), ("mm_max", ctypes.c_int32),
                ("pat_gap_open", ctypes.c_int32), ("pat_gap_ext", ctypes.c_int32),
                ("txt_gap_open", ctypes.c_int32), ("txt_gap_ext", ctypes.c_int32)]

    @staticmethod
    def simple(match, mismatch, gap_open, gap_ext):
        """aln::SimpleGotohScheme(match, mismatch, gap_open, gap_ext) (nvbio/alignment/utils.h:103-123)"""
        return Scheme(match, -mismatch, -mismatch, gap_open, gap_ext, gap_open, gap_ext)

    def as_array(self):
        return np.array([self.match, self.mm_min, self.mm_max, self.pat_gap_open, self.pat_gap_ext,
                         self.txt_gap_open, self.txt_gap_ext], dtype=np.int32)


class FMView(ctypes.Structure):
    """orc_fm_index"""
    _fields_ = [("length", ctypes.c_uint32), ("primary", ctypes.c_uint32), ("L2", ctypes.c_uint32 * 5),
                ("bwt_occ", _u32p), ("ssa", _u32p)]


class HostIndex:
    """an FM-index held in numpy arrays (production layout)"""

    def __init__(self, n, primary, L2, bwt_occ, ssa, sa=None, text=None):
        self.n, self.primary = int(n), int(primary)
        self.L2 = np.asarray(L2, dtype=np.uint32)
        self.bwt_occ = np.ascontiguousarray(bwt_occ, dtype=np.uint32)
        self.ssa = np.ascontiguousarray(ssa, dtype=np.uint32)
        self.sa, self.text = sa, text
        self.handle = None

    def view(self):
        v = FMView()
        v.length, v.primary = self.n, self.primary
        for i in range(5):
            v.L2[i] = int(self.L2[i])
        v.bwt_occ = self.bwt_occ.ctypes.data_as(_u32p)
        v.ssa = self.ssa.ctypes.data_as(_u32p)
        return v


class Oracle:
    """the C restatement"""

    def __init__(self):
        build()
        self.lib = L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.orc_bwt_words.restype = ctypes.c_uint32
        L.orc_fm_build.restype = ctypes.c_uint32
        L.orc_rank.restype = ctypes.c_uint32
        L.orc_dict_rank.restype = ctypes.c_uint32
        L.orc_locate.restype = ctypes.c_uint32
        L.orc_basic_inv_psi.restype = ctypes.c_uint32
        L.orc_filter_rank.restype = ctypes.c_uint64
        L.orc_mismatch.restype = ctypes.c_int32
        L.orc_popc_2bit.restype = ctypes.c_uint32
        L.orc_popc_2bit_hi.restype = ctypes.c_uint32
        L.orc_get2.restype = ctypes.c_uint8
        L.orc_get4.restype = ctypes.c_uint8

    def num_threads(self):
        return int(self.lib.orc_num_threads())

    def set_num_threads(self, n):
        self.lib.orc_set_num_threads(int(n))

    # ---- packing -------------------------------------------------------------------------
    def pack2(self, syms):
        syms = _c8(syms)
        words = np.zeros((len(syms) + 15) // 16 + 4, dtype=np.uint32)
        self.lib.orc_pack2(_p(syms, _u8p), ctypes.c_uint64(len(syms)), _p(words, _u32p))
        return words

    def pack4(self, syms):
        syms = _c8(syms)
        words = np.zeros((len(syms) + 7) // 8 + 4, dtype=np.uint32)
        self.lib.orc_pack4(_p(syms, _u8p), ctypes.c_uint64(len(syms)), _p(words, _u32p))
        return words

    def get2(self, words, i):
        return int(self.lib.orc_get2(_p(words, _u32p), ctypes.c_uint64(i)))

    def get4(self, words, i):
        return int(self.lib.orc_get4(_p(words, _u32p), ctypes.c_uint64(i)))

    # ---- index ---------------------------------------------------------------------------
    def suffix_sort(self, text):
        text = _c8(text)
        sa = np.zeros(len(text) + 1, dtype=np.uint32)
        self.lib.orc_suffix_sort(_p(text, _u8p), ctypes.c_uint32(len(text)), _p(sa, _u32p))
        return sa

    def build_index(self, text, sa=None):
        text = _c8(text)
        n = len(text)
        if sa is None:
            sa = self.suffix_sort(text)
        sa = _c32(sa)
        words = int(self.lib.orc_bwt_words(ctypes.c_uint32(n)))
        bwt_occ = np.zeros(2 * words, dtype=np.uint32)
        ssa = np.zeros((n + 16) // 16, dtype=np.uint32)
        L2 = np.zeros(5, dtype=np.uint32)
        primary = self.lib.orc_fm_build(_p(text, _u8p), ctypes.c_uint32(n), _p(sa, _u32p), _p(bwt_occ, _u32p),
                                        _p(ssa, _u32p), _p(L2, _u32p))
        return HostIndex(n, primary, L2, bwt_occ, ssa, sa=sa, text=text)

    # ---- queries -------------------------------------------------------------------------
    def rank(self, idx, k, c):
        v = idx.view()
        return int(self.lib.orc_rank(ctypes.byref(v), ctypes.c_uint32(k & 0xFFFFFFFF), ctypes.c_uint32(c)))

    def rank2(self, idx, l, r, c):
        v = idx.view()
        out = np.zeros(2, dtype=np.uint32)
        self.lib.orc_rank2(ctypes.byref(v), ctypes.c_uint32(l & 0xFFFFFFFF), ctypes.c_uint32(r & 0xFFFFFFFF),
                           ctypes.c_uint32(c), _p(out, _u32p))
        return out

    def rank4(self, idx, k):
        v = idx.view()
        out = np.zeros(4, dtype=np.uint32)
        self.lib.orc_rank4(ctypes.byref(v), ctypes.c_uint32(k & 0xFFFFFFFF), _p(out, _u32p))
        return out

    def match_batch(self, idx, syms, offsets, reverse=False, want_blocks=False):
        v = idx.view()
        syms, offsets = _c8(syms), _c32(offsets)
        n = len(offsets) - 1
        ranges = np.zeros((n, 2), dtype=np.uint32)
        blocks = np.zeros(n, dtype=np.uint32) if want_blocks else None
        self.lib.orc_match_batch(ctypes.byref(v), _p(syms, _u8p), _p(offsets, _u32p), ctypes.c_uint32(n),
                                 ctypes.c_int(1 if reverse else 0), _p(ranges, _u32p), _p(blocks, _u32p))
        return (ranges, blocks) if want_blocks else ranges

    def basic_inv_psi(self, idx, rows):
        v = idx.view()
        return np.array([self.lib.orc_basic_inv_psi(ctypes.byref(v), ctypes.c_uint32(int(r))) for r in rows], dtype=np.uint32)

    def locate_batch(self, idx, rows):
        v = idx.view()
        rows = _c32(rows)
        pos = np.zeros(len(rows), dtype=np.uint32)
        self.lib.orc_locate_batch(ctypes.byref(v), _p(rows, _u32p), ctypes.c_uint32(len(rows)), _p(pos, _u32p))
        return pos

    def locate_ssa_batch(self, idx, rows):
        v = idx.view()
        out = np.zeros((len(rows), 2), dtype=np.uint32)
        tmp = np.zeros(2, dtype=np.uint32)
        for i, r in enumerate(rows):
            self.lib.orc_locate_ssa(ctypes.byref(v), ctypes.c_uint32(int(r)), _p(tmp, _u32p))
            out[i] = tmp
        return out

    def filter_rank(self, idx, syms, offsets):
        v = idx.view()
        syms, offsets = _c8(syms), _c32(offsets)
        n = len(offsets) - 1
        ranges = np.zeros((n, 2), dtype=np.uint32)
        slots = np.zeros(n, dtype=np.uint64)
        total = self.lib.orc_filter_rank(ctypes.byref(v), _p(syms, _u8p), _p(offsets, _u32p), ctypes.c_uint32(n),
                                         _p(ranges, _u32p), _p(slots, _u64p))
        return int(total), ranges, slots

    def filter_locate(self, idx, ranges, slots, begin, end):
        v = idx.view()
        hits = np.zeros((end - begin, 2), dtype=np.uint32)
        self.lib.orc_filter_locate(ctypes.byref(v), _p(ranges, _u32p), _p(slots, _u64p),
                                   ctypes.c_uint32(len(slots)), ctypes.c_uint64(begin), ctypes.c_uint64(end),
                                   _p(hits, _u32p))
        return hits

    # ---- DP ------------------------------------------------------------------------------
    def mismatch(self, scheme, q):
        return int(self.lib.orc_mismatch(ctypes.byref(scheme), ctypes.c_uint32(q)))

    def banded_gotoh(self, band, typ, scheme, pat, txt, quals=None):
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.orc_banded_gotoh(ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.byref(scheme), _p(pat, _u8p),
                                       _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                       ctypes.c_uint32(len(txt)), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def banded_gotoh_staged(self, band, typ, scheme, pat, txt, min_score, quals=None):
        """the staged scheduler's windowed scoring -> (ran_to_end, score, sink)"""
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.orc_banded_gotoh_staged(ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.byref(scheme), _p(pat, _u8p),
                                              _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                              ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc),
                                              _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def banded_gotoh_traceback(self, band, typ, scheme, pat, txt, quals=None, cap=4096):
        """-> (traced, score, source, sink, cigar uint16[], ops uint8[]) in backtracking order"""
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        cig = np.zeros(cap, dtype=np.uint16); ops = np.zeros(cap, dtype=np.uint8)
        cl = ctypes.c_uint32(); no = ctypes.c_uint32()
        ok = self.lib.orc_banded_gotoh_traceback(
            ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.byref(scheme), _p(pat, _u8p), _p(quals, _u8p),
            ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.byref(sc), _p(src, _u32p),
            _p(sk, _u32p), cig.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(cap), ctypes.byref(cl),
            _p(ops, _u8p), ctypes.c_uint32(cap), ctypes.byref(no))
        assert cl.value <= cap and no.value <= cap
        return ok, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), cig[:cl.value].copy(), ops[:no.value].copy()

    def banded_gotoh_traceback_packed_batch(self, band, typ, scheme, reads4, read_offsets, genome2, win_begin, win_end,
                                            cigar_stride, read_id=None, flags=None, quals=None):
        reads4, read_offsets, genome2 = _c32(reads4), _c32(read_offsets), _c32(genome2)
        win_begin, win_end, read_id = _c32(win_begin), _c32(win_end), _c32(read_id)
        flags, quals = _c8(flags), _c8(quals)
        n = len(win_begin)
        scores = np.zeros(n, dtype=np.int32)
        sources = np.zeros((n, 2), dtype=np.uint32); sinks = np.zeros((n, 2), dtype=np.uint32)
        cigars = np.zeros((n, cigar_stride), dtype=np.uint16); lens = np.zeros(n, dtype=np.uint32)
        self.lib.orc_banded_gotoh_traceback_packed_batch(
            ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.byref(scheme), _p(reads4, _u32p),
            _p(read_offsets, _u32p), _p(quals, _u8p), _p(read_id, _u32p), _p(flags, _u8p), _p(genome2, _u32p),
            _p(win_begin, _u32p), _p(win_end, _u32p), ctypes.c_uint32(n), _p(scores, _i32p), _p(sources, _u32p),
            _p(sinks, _u32p), cigars.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(cigar_stride), _p(lens, _u32p))
        return scores, sources, sinks, cigars, lens

    def banded_sw_traceback(self, band, typ, sw, pat, txt, cap=4096):
        """linear-gap Smith-Waterman aligner -> (traced, score, source, sink, cigar uint16[] in backtracking order)"""
        pat, txt = _c8(pat), _c8(txt)
        sw = np.ascontiguousarray(np.asarray(sw, dtype=np.int32))
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        cig = np.zeros(cap, dtype=np.uint16); cl = ctypes.c_uint32()
        ok = self.lib.orc_banded_sw_traceback(ctypes.c_uint32(band), ctypes.c_int(typ), _p(sw, _i32p), _p(pat, _u8p), ctypes.c_uint32(len(pat)),
                                              _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.byref(sc), _p(src, _u32p), _p(sk, _u32p),
                                              cig.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(cap), ctypes.byref(cl))
        assert cl.value <= cap
        return ok, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), cig[:cl.value].copy()

    def full_sw_traceback(self, typ, sw, pat, txt, min_score=SCORE_MIN, cap=4096):
        """full-matrix traceback of the linear-gap Smith-Waterman aligner -> (traced, score, source, sink, cigar uint16[] in backtracking order)"""
        pat, txt = _c8(pat), _c8(txt)
        sw = np.ascontiguousarray(np.asarray(sw, dtype=np.int32))
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        cig = np.zeros(cap, dtype=np.uint16); cl = ctypes.c_uint32()
        ok = self.lib.orc_full_sw_traceback(ctypes.c_int(typ), _p(sw, _i32p), _p(pat, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                            ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc), _p(src, _u32p), _p(sk, _u32p),
                                            cig.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(cap), ctypes.byref(cl))
        assert cl.value <= cap
        return ok, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), cig[:cl.value].copy()

    def finish_alignment(self, pat, txt, cigar, cigar_offset, cap=1024):
        """nvBowtie finish_alignment: (edit distance, MDS bytes) of a traced alignment"""
        pat, txt = _c8(pat), _c8(txt)
        cigar = np.ascontiguousarray(cigar, dtype=np.uint16)
        ed = ctypes.c_uint32(); ml = ctypes.c_uint32()
        mds = np.zeros(cap, dtype=np.uint8)
        self.lib.orc_finish_alignment(_p(pat, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)),
                                      cigar.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(cigar)), ctypes.c_uint32(cigar_offset),
                                      ctypes.byref(ed), _p(mds, _u8p), ctypes.c_uint32(cap), ctypes.byref(ml))
        assert ml.value <= cap
        return ed.value, mds[:ml.value].copy()

    def full_gotoh_traceback(self, typ, scheme, pat, txt, quals=None, min_score=SCORE_MIN, cap=4096):
        """-> (traced, score, source, sink, cigar uint16[]) -- x = text, y = pattern; cigar in backtracking order"""
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        cig = np.zeros(cap, dtype=np.uint16); cl = ctypes.c_uint32()
        ok = self.lib.orc_full_gotoh_traceback(
            ctypes.c_int(typ), ctypes.byref(scheme), _p(pat, _u8p), _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
            ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc), _p(src, _u32p), _p(sk, _u32p),
            cig.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(cap), ctypes.byref(cl))
        assert cl.value <= cap
        return ok, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), cig[:cl.value].copy()

    def full_gotoh(self, typ, blocking, scheme, pat, txt, quals=None, min_score=SCORE_MIN):
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.orc_full_gotoh(ctypes.c_int(typ), ctypes.c_int(blocking), ctypes.byref(scheme), _p(pat, _u8p),
                                     _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                     ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc),
                                     _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def banded_gotoh_best2(self, band, typ, scheme, pat, txt, quals=None, dist=0):
        """banded Gotoh reporting into Best2Sink<int32>(dist) -> (ok, (score1, x1, y1, score2, x2, y2))"""
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        out = np.zeros(6, dtype=np.int64)
        ok = self.lib.orc_banded_gotoh_best2(ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.byref(scheme), _p(pat, _u8p), _p(quals, _u8p),
                                             ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.c_uint32(dist),
                                             out.ctypes.data_as(ctypes.c_void_p))
        return ok, tuple(int(v) for v in out)

    def full_gotoh_best2(self, typ, blocking, scheme, pat, txt, quals=None, min_score=SCORE_MIN, dist=0):
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        out = np.zeros(6, dtype=np.int64)
        ok = self.lib.orc_full_gotoh_best2(ctypes.c_int(typ), ctypes.c_int(blocking), ctypes.byref(scheme), _p(pat, _u8p), _p(quals, _u8p),
                                           ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score),
                                           ctypes.c_uint32(dist), out.ctypes.data_as(ctypes.c_void_p))
        return ok, tuple(int(v) for v in out)

    def banded_sw(self, band, typ, sw, pat, txt):
        """linear-gap Smith-Waterman in a band; sw = (match, mismatch, deletion, insertion), signed"""
        pat, txt = _c8(pat), _c8(txt)
        arr = np.array(sw, dtype=np.int32)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.orc_banded_sw(ctypes.c_uint32(band), ctypes.c_int(typ), _p(arr, _i32p), _p(pat, _u8p), ctypes.c_uint32(len(pat)),
                                    _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def full_sw(self, typ, blocking, sw, pat, txt, min_score=SCORE_MIN):
        """full-matrix linear-gap Smith-Waterman (16-wide stripes); the edit-distance aligner is sw = ED_SW"""
        pat, txt = _c8(pat), _c8(txt)
        arr = np.array(sw, dtype=np.int32)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.orc_full_sw(ctypes.c_int(typ), ctypes.c_int(blocking), _p(arr, _i32p), _p(pat, _u8p), ctypes.c_uint32(len(pat)),
                                  _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def hamming_backtrack(self, idx, stream, begin, length, seed, mismatches, quirks=False, cap=64):
        """hamming_backtrack with a counting delegate -> (count, n_ranges, ranges uint32 [min(n, cap), 2]); stream: uint8 symbols"""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        cnt = ctypes.c_uint32(0)
        rg = np.zeros((cap, 2), dtype=np.uint32)
        self.lib.orc_hamming_backtrack.restype = ctypes.c_uint32
        v = idx.view()
        n = self.lib.orc_hamming_backtrack(ctypes.byref(v), _p(stream, _u8p), ctypes.c_uint32(begin), ctypes.c_uint32(length),
                                           ctypes.c_uint32(seed), ctypes.c_uint32(mismatches), ctypes.c_int(1 if quirks else 0),
                                           ctypes.byref(cnt), _p(rg, _u32p), ctypes.c_uint32(cap))
        return cnt.value, int(n), rg[:min(int(n), cap)].copy()

    def score_reduce(self, scores, pos, rc, read_len, worst_score):
        """nvBowtie's score_reduce_kernel for one read over candidates in the order given
        -> (a1 aligned, a1 score, a1 pos, a1 rc, a2 aligned, a2 score, a2 pos, a2 rc)"""
        scores = np.ascontiguousarray(scores, dtype=np.int32); pos = np.ascontiguousarray(pos, dtype=np.uint32)
        rc = np.ascontiguousarray(rc, dtype=np.uint8)
        out = np.zeros(8, dtype=np.int64)
        self.lib.orc_score_reduce(_p(scores, _i32p), _p(pos, _u32p), _p(rc, _u8p), ctypes.c_uint32(len(scores)),
                                  ctypes.c_uint32(read_len), ctypes.c_int32(worst_score), out.ctypes.data_as(ctypes.c_void_p))
        return tuple(int(v) for v in out)

    def hit_deque_run(self, ops, begins, bits, max_hits):
        """a sequence of operations on nvBowtie's per-read seed-hit deque (0 push under the max_hits rule, 1 pop_top, 2 pop_bottom,
        3 select's in-place row pop) -> (heap array [size, 2] as it lies in memory, rows returned by the select ops)"""
        ops, begins, bits = _c32(ops), _c32(begins), _c32(bits)
        n = len(ops)
        heap = np.zeros((n + 1, 2), dtype=np.uint32); size = ctypes.c_uint32(0); rows = np.zeros(max(n, 1), dtype=np.uint32)
        self.lib.orc_hit_deque_run(_p(ops, _u32p), _p(begins, _u32p), _p(bits, _u32p), ctypes.c_uint32(n), ctypes.c_uint32(max_hits),
                    _p(heap, _u32p), ctypes.byref(size), _p(rows, _u32p))
        return heap[:size.value].copy(), rows[:n].copy()
    def map_exact_read(self, fw, rc, seed_off, read_len, seed_len, max_hits, rep_seeds):
        """seed_mapper<EXACT_MAPPING> + map_kernel bookkeeping for one read (mapping_inl.h:193-282,485-556) -> (deque [n, 2], reseed)"""
        fw = np.ascontiguousarray(fw, dtype=np.uint32); rc = np.ascontiguousarray(rc, dtype=np.uint32); seed_off = _c32(seed_off)
        deque = np.zeros((max(2 * len(seed_off), 1), 2), dtype=np.uint32); size = ctypes.c_uint32(0)
        reseed = self.lib.orc_map_exact_read(_p(fw, _u32p), _p(rc, _u32p), _p(seed_off, _u32p), ctypes.c_uint32(len(seed_off)),
                                             ctypes.c_uint32(read_len), ctypes.c_uint32(seed_len), ctypes.c_uint32(max_hits),
                                             ctypes.c_uint32(rep_seeds), _p(deque, _u32p), ctypes.byref(size))
        return deque[:size.value].copy(), bool(reseed)

    def map_approx_read(self, idx, ridx, stored, seed_off, seed_len, max_hits, rep_seeds):
        """seed_mapper<APPROX_MAPPING> for one read (mapping_inl.h:114-184,288-342): idx / ridx = the forward index and the index of the
        reversed text, stored = the read as nvBowtie stores it (reversed), one symbol per byte -> (deque [n, 2], reseed)"""
        v, rv = idx.view(), ridx.view()
        stored = _c8(stored); seed_off = _c32(seed_off)
        cap = 4 * len(seed_off) * (3 * ((seed_len + 1) // 2) + 1) + 1
        deque = np.zeros((cap, 2), dtype=np.uint32); size = ctypes.c_uint32(0)
        reseed = self.lib.orc_map_approx_read(ctypes.byref(v), ctypes.byref(rv), _p(stored, _u8p), ctypes.c_uint32(len(stored)), _p(seed_off, _u32p),
                                              ctypes.c_uint32(len(seed_off)), ctypes.c_uint32(seed_len), ctypes.c_uint32(max_hits),
                                              ctypes.c_uint32(rep_seeds), _p(deque, _u32p), ctypes.byref(size))
        return deque[:size.value].copy(), bool(reseed)

    def select_read(self, deque, top_flag):
        """select_kernel for one read (select_inl.h:62-130); deque [n, 2] is updated in place (returned with its new size)
        -> (selected, sa_pos, packed_seed, top_flag, deque)"""
        buf = np.zeros((len(deque) + 1, 2), dtype=np.uint32); buf[:len(deque)] = deque
        size = ctypes.c_uint32(len(deque)); tf = ctypes.c_uint32(top_flag); row = ctypes.c_uint32(0); seed = ctypes.c_uint32(0)
        ok = self.lib.orc_select_read(_p(buf, _u32p), ctypes.byref(size), ctypes.byref(tf), ctypes.byref(row), ctypes.byref(seed))
        return bool(ok), row.value, seed.value, tf.value, buf[:size.value].copy()

    def score_reduce_effort(self, best, trys, score, g_pos, read_rc, top_flag, read_len, ext, max_effort, min_ext, max_ext):
        """score_reduce_kernel for one hit with ReduceBestApproxContext (reduce_inl.h:65-140, reduce.h:55-99); best = [a1 score, a1 pos,
        a1 rc, a2 score, a2 pos, a2 rc] -> (best, trys, erase)"""
        b = np.array(best, dtype=np.int64); t = ctypes.c_uint32(trys)
        erase = self.lib.orc_score_reduce_effort(b.ctypes.data_as(ctypes.c_void_p), ctypes.byref(t), ctypes.c_int32(score), ctypes.c_uint32(g_pos),
                                                 ctypes.c_uint32(read_rc), ctypes.c_uint32(top_flag), ctypes.c_uint32(read_len), ctypes.c_uint32(ext),
                                                 ctypes.c_uint32(max_effort), ctypes.c_uint32(min_ext), ctypes.c_uint32(max_ext))
        return [int(v) for v in b], t.value, bool(erase)

    def rank_generic(self, text_words, word_bits, length, K, index_bits, idx, sym):
        """the generic rank dictionary (rank_dictionary_inl.h:206-336): build_occurrence_table<K> + rank(dict, i, c) for every (idx, sym) pair
        -> (occ [blocks, 4], counts [4], ranks)"""
        tw = np.ascontiguousarray(text_words, dtype=np.uint32 if word_bits == 32 else np.uint64)
        nb = (length + K - 1) // K
        occ = np.zeros((max(nb, 1), 4), dtype=np.uint32 if index_bits == 32 else np.uint64)
        cnt = np.zeros(4, dtype=np.uint64)
        self.lib.orc_rank_generic.restype = ctypes.c_uint64
        self.lib.orc_rank_generic_build(tw.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(word_bits), ctypes.c_uint64(length), ctypes.c_uint32(K),
                                        ctypes.c_uint32(index_bits), occ.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
        out = np.array([self.lib.orc_rank_generic(tw.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(word_bits), occ.ctypes.data_as(ctypes.c_void_p),
                                                  ctypes.c_uint32(index_bits), ctypes.c_uint32(K), ctypes.c_uint64(int(i)), ctypes.c_uint32(int(c)))
                        for i, c in zip(idx, sym)], dtype=np.uint64)
        return occ, cnt, out

    def banded_myers(self, band, typ, pat, txt, min_score=SCORE_MIN):
        """aln::banded_alignment_score<BAND>( EditDistanceAligner<TYPE, MyersTag<5>>, ... ) (myers/myers_banded_inl.h:247-342)
        -> (ok, score = -(edit distance), sink)"""
        pat, txt = _c8(pat), _c8(txt)
        sc = ctypes.c_int32(); sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.orc_banded_myers(ctypes.c_uint32(band), ctypes.c_int(typ), _p(pat, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                         ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def mapq(self, version, monotone, perfect_score, min_score, best_score, has_second, second_score):
        """BowtieMapq2 / BowtieMapq3, single-end (nvBowtie/bowtie2/cuda/mapq.h)"""
        return int(self.lib.orc_mapq(ctypes.c_int(version), ctypes.c_int(1 if monotone else 0), ctypes.c_int32(perfect_score),
                                     ctypes.c_int32(min_score), ctypes.c_int32(best_score), ctypes.c_int(1 if has_second else 0),
                                     ctypes.c_int32(second_score)))

    def banded_gotoh_batch(self, band, typ, scheme, pats, pat_off, txts, txt_off, quals=None):
        pats, txts, quals = _c8(pats), _c8(txts), _c8(quals)
        pat_off, txt_off = _c32(pat_off), _c32(txt_off)
        n = len(pat_off) - 1
        scores = np.zeros(n, dtype=np.int32)
        sinks = np.zeros((n, 2), dtype=np.uint32)
        self.lib.orc_banded_gotoh_batch(ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.byref(scheme),
                                        _p(pats, _u8p), _p(quals, _u8p), _p(pat_off, _u32p), _p(txts, _u8p),
                                        _p(txt_off, _u32p), ctypes.c_uint32(n), _p(scores, _i32p), _p(sinks, _u32p))
        return scores, sinks

    def full_gotoh_batch(self, typ, blocking, scheme, pats, pat_off, txts, txt_off, quals=None, min_score=SCORE_MIN):
        pats, txts, quals = _c8(pats), _c8(txts), _c8(quals)
        pat_off, txt_off = _c32(pat_off), _c32(txt_off)
        n = len(pat_off) - 1
        scores = np.zeros(n, dtype=np.int32)
        sinks = np.zeros((n, 2), dtype=np.uint32)
        self.lib.orc_full_gotoh_batch(ctypes.c_int(typ), ctypes.c_int(blocking), ctypes.byref(scheme),
                                      _p(pats, _u8p), _p(quals, _u8p), _p(pat_off, _u32p), _p(txts, _u8p),
                                      _p(txt_off, _u32p), ctypes.c_uint32(n), ctypes.c_int32(min_score),
                                      _p(scores, _i32p), _p(sinks, _u32p))
        return scores, sinks

    def banded_gotoh_packed_batch(self, band, typ, scheme, reads4, read_offsets, genome2, win_begin, win_end,
                                  read_id=None, flags=None, quals=None):
        reads4, read_offsets, genome2 = _c32(reads4), _c32(read_offsets), _c32(genome2)
        win_begin, win_end, read_id = _c32(win_begin), _c32(win_end), _c32(read_id)
        flags, quals = _c8(flags), _c8(quals)
        n = len(win_begin)
        scores = np.zeros(n, dtype=np.int32)
        sinks = np.zeros((n, 2), dtype=np.uint32)
        self.lib.orc_banded_gotoh_packed_batch(
            ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.byref(scheme), _p(reads4, _u32p),
            _p(read_offsets, _u32p), _p(quals, _u8p), _p(read_id, _u32p), _p(flags, _u8p), _p(genome2, _u32p),
            _p(win_begin, _u32p), _p(win_end, _u32p), ctypes.c_uint32(n), _p(scores, _i32p), _p(sinks, _u32p))
        return scores, sinks


class Reference:
    """the reference's own host code (development container only)"""

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libnvbio_ref.so"))

    def __init__(self):
        self.lib = L = ctypes.CDLL(os.path.join(_HERE, "_ref", "libnvbio_ref.so"))
        L.ref_fm_create.restype = ctypes.c_void_p
        L.ref_fm_adopt.restype = ctypes.c_void_p
        L.ref_fm_primary.restype = ctypes.c_uint32
        L.ref_fm_words.restype = ctypes.c_uint32
        L.ref_fm_ssa_words.restype = ctypes.c_uint32
        L.ref_fm_rank.restype = ctypes.c_uint32

    def hit_deque_run(self, ops, begins, bits, max_hits):
        """a sequence of operations on nvBowtie's per-read seed-hit deque (0 push under the max_hits rule, 1 pop_top, 2 pop_bottom,
        3 select's in-place row pop) -> (heap array [size, 2] as it lies in memory, rows returned by the select ops)"""
        ops, begins, bits = _c32(ops), _c32(begins), _c32(bits)
        n = len(ops)
        heap = np.zeros((n + 1, 2), dtype=np.uint32); size = ctypes.c_uint32(0); rows = np.zeros(max(n, 1), dtype=np.uint32)
        self.lib.ref_hit_deque_run(_p(ops, _u32p), _p(begins, _u32p), _p(bits, _u32p), ctypes.c_uint32(n), ctypes.c_uint32(max_hits),
                    _p(heap, _u32p), ctypes.byref(size), _p(rows, _u32p))
        return heap[:size.value].copy(), rows[:n].copy()

    def rank_generic(self, text_words, word_bits, length, idx, sym):
        """the reference's generic rank dictionary in its two test configurations (32-bit words / K 64 / uint32, 64-bit words / K 128 / uint64)
        -> (occ [blocks, 4], counts [4], rank [n], rank4 [n, 4])"""
        K = 64 if word_bits == 32 else 128
        tw = np.ascontiguousarray(text_words, dtype=np.uint32 if word_bits == 32 else np.uint64)
        idx = np.ascontiguousarray(idx, dtype=np.uint64); sym = _c8(sym)
        nb = (length + K - 1) // K
        occ = np.zeros((max(nb, 1), 4), dtype=np.uint32 if word_bits == 32 else np.uint64)
        cnt = np.zeros(4, dtype=np.uint64); r = np.zeros(len(idx), dtype=np.uint64); r4 = np.zeros((len(idx), 4), dtype=np.uint64)
        self.lib.ref_rank_generic(ctypes.c_uint32(word_bits), tw.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(length), idx.ctypes.data_as(ctypes.c_void_p),
                                  _p(sym, _u8p), ctypes.c_uint32(len(idx)), occ.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p),
                                  r.ctypes.data_as(ctypes.c_void_p), r4.ctypes.data_as(ctypes.c_void_p))
        return occ, cnt, r, r4

    def banded_myers(self, band, typ, pat, txt, min_score=SCORE_MIN):
        """aln::banded_alignment_score<BAND>( EditDistanceAligner<TYPE, MyersTag<5>>, ... ) (myers/myers_banded_inl.h:247-342)
        -> (ok, score = -(edit distance), sink)"""
        pat, txt = _c8(pat), _c8(txt)
        sc = ctypes.c_int32(); sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.ref_banded_myers(ctypes.c_uint32(band), ctypes.c_int(typ), _p(pat, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                         ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def build_index(self, text):
        text = _c8(text)
        n = len(text)
        h = ctypes.c_void_p(self.lib.ref_fm_create(_p(text, _u8p), ctypes.c_uint32(n)))
        words = int(self.lib.ref_fm_words(h))
        bwt_occ = np.zeros(2 * words, dtype=np.uint32)
        sa = np.zeros(n + 1, dtype=np.int32)
        ssa = np.zeros(int(self.lib.ref_fm_ssa_words(h)), dtype=np.uint32)
        L2 = np.zeros(5, dtype=np.uint32)
        self.lib.ref_fm_export(h, None, None, _p(bwt_occ, _u32p), _p(sa, _i32p), _p(ssa, _u32p))
        self.lib.ref_fm_L2(h, _p(L2, _u32p))
        idx = HostIndex(n, self.lib.ref_fm_primary(h), L2, bwt_occ, ssa, sa=sa.view(np.uint32), text=text)
        idx.handle = h
        return idx

    def adopt_index(self, hidx):
        """hand an index held in numpy arrays (e.g. built on the GPU) to the reference's host code"""
        h = ctypes.c_void_p(self.lib.ref_fm_adopt(
            ctypes.c_uint32(hidx.n), ctypes.c_uint32(hidx.primary), _p(_c32(hidx.L2), _u32p), _p(hidx.bwt_occ, _u32p),
            ctypes.c_uint64(len(hidx.bwt_occ)), _p(hidx.ssa, _u32p), ctypes.c_uint64(len(hidx.ssa))))
        hidx.handle = h
        return hidx

    def num_threads(self):
        return int(self.lib.ref_num_threads())

    def set_num_threads(self, n):
        self.lib.ref_set_num_threads(int(n))

    def banded_gotoh_batch(self, band, typ, scheme, pats, pat_off, txts, txt_off, quals=None):
        pats, txts, quals = _c8(pats), _c8(txts), _c8(quals)
        pat_off, txt_off = _c32(pat_off), _c32(txt_off)
        n = len(pat_off) - 1
        arr = scheme.as_array()
        scores = np.zeros(n, dtype=np.int32)
        sinks = np.zeros((n, 2), dtype=np.uint32)
        self.lib.ref_banded_gotoh_ex_batch(ctypes.c_uint32(band), ctypes.c_int(typ), _p(arr, _i32p), _p(pats, _u8p),
                                           _p(quals, _u8p), _p(pat_off, _u32p), _p(txts, _u8p), _p(txt_off, _u32p),
                                           ctypes.c_uint32(n), ctypes.c_int32(SCORE_MIN), _p(scores, _i32p),
                                           _p(sinks, _u32p))
        return scores, sinks

    def destroy(self, idx):
        self.lib.ref_fm_destroy(idx.handle)
        idx.handle = None

    def rank(self, idx, k, c):
        return int(self.lib.ref_fm_rank(idx.handle, ctypes.c_uint32(k & 0xFFFFFFFF), ctypes.c_uint32(c)))

    def rank2(self, idx, l, r, c):
        out = np.zeros(2, dtype=np.uint32)
        self.lib.ref_fm_rank_range(idx.handle, ctypes.c_uint32(l & 0xFFFFFFFF), ctypes.c_uint32(r & 0xFFFFFFFF),
                                   ctypes.c_uint32(c), _p(out, _u32p))
        return out

    def rank4(self, idx, k):
        out = np.zeros(4, dtype=np.uint32)
        self.lib.ref_fm_rank4(idx.handle, ctypes.c_uint32(k & 0xFFFFFFFF), _p(out, _u32p))
        return out

    def match_batch(self, idx, syms, offsets, reverse=False):
        syms, offsets = _c8(syms), _c32(offsets)
        n = len(offsets) - 1
        ranges = np.zeros((n, 2), dtype=np.uint32)
        self.lib.ref_fm_match(idx.handle, _p(syms, _u8p), _p(offsets, _u32p), ctypes.c_uint32(n), _p(ranges, _u32p),
                              ctypes.c_int(1 if reverse else 0))
        return ranges

    def locate_batch(self, idx, rows):
        rows = _c32(rows)
        pos = np.zeros(len(rows), dtype=np.uint32)
        self.lib.ref_fm_locate(idx.handle, _p(rows, _u32p), ctypes.c_uint32(len(rows)), _p(pos, _u32p))
        return pos

    def locate_ssa_batch(self, idx, rows):
        rows = _c32(rows)
        out = np.zeros((len(rows), 2), dtype=np.uint32)
        self.lib.ref_fm_locate_ssa(idx.handle, _p(rows, _u32p), ctypes.c_uint32(len(rows)), _p(out, _u32p))
        return out

    def banded_gotoh(self, band, typ, scheme, pat, txt, quals=None, simple=False):
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        if simple:
            ok = self.lib.ref_banded_gotoh(ctypes.c_uint32(band), ctypes.c_int(typ), ctypes.c_int(scheme.match),
                                           ctypes.c_int(-scheme.mm_min), ctypes.c_int(scheme.pat_gap_open),
                                           ctypes.c_int(scheme.pat_gap_ext), _p(pat, _u8p), ctypes.c_uint32(len(pat)),
                                           _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.c_int32(SCORE_MIN),
                                           ctypes.byref(sc), _p(sk, _u32p))
        else:
            arr = scheme.as_array()
            ok = self.lib.ref_banded_gotoh_ex(ctypes.c_uint32(band), ctypes.c_int(typ), _p(arr, _i32p), _p(pat, _u8p),
                                              _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                              ctypes.c_uint32(len(txt)), ctypes.c_int32(SCORE_MIN), ctypes.byref(sc),
                                              _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def banded_gotoh_staged(self, band, typ, scheme, pat, txt, min_score, quals=None):
        """StagedAlignmentUnitBase::run over BandedScoreUnit (batched_stream.h:117-285) -> (ran_to_end, score, sink, windows).
        The windowed overload reloads its text cache from text[window_begin + j], j < BAND-1, unconditionally
        (gotoh_banded_inl.h:432-433): the text is handed over with BAND sentinel bytes (255) behind it so that the read is
        defined -- the value the continuous loop puts there (:569-570)."""
        pat, quals = _c8(pat), _c8(quals)
        n = len(txt)
        txt = np.concatenate([np.asarray(txt, dtype=np.uint8), np.full(band + 1, 255, np.uint8)])
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        arr = scheme.as_array()
        r = self.lib.ref_banded_gotoh_staged_ex(ctypes.c_uint32(band), ctypes.c_int(typ), _p(arr, _i32p), _p(pat, _u8p),
                                                _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                                ctypes.c_uint32(n), ctypes.c_int32(min_score), ctypes.byref(sc),
                                                _p(sk, _u32p))
        return r & 1, sc.value, (int(sk[0]), int(sk[1])), r >> 1

    def banded_ed(self, band, typ, pat, txt):
        """aln::banded_alignment_score<BAND>( EditDistanceAligner<TYPE> ) (ed/ed_banded_inl.h:37-69)"""
        pat, txt = _c8(pat), _c8(txt)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.ref_banded_ed(ctypes.c_uint32(band), ctypes.c_int(typ), _p(pat, _u8p), ctypes.c_uint32(len(pat)),
                                    _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def hamming_backtrack(self, idx, stream_words, begin, length, seed, mismatches, cap=64):
        """nvbio::hamming_backtrack over a 2-bit PackedStream pattern, as the reference's benchmark (fmindex_test.cu:744-767)"""
        w = np.ascontiguousarray(stream_words, dtype=np.uint32)
        cnt = ctypes.c_uint32(0)
        rg = np.zeros((cap, 2), dtype=np.uint32)
        self.lib.ref_hamming_backtrack.restype = ctypes.c_uint32
        n = self.lib.ref_hamming_backtrack(idx.handle, _p(w, _u32p), ctypes.c_uint32(begin), ctypes.c_uint32(length),
                                           ctypes.c_uint32(seed), ctypes.c_uint32(mismatches), ctypes.byref(cnt), _p(rg, _u32p),
                                           ctypes.c_uint32(cap))
        return cnt.value, int(n), rg[:min(int(n), cap)].copy()

    def banded_gotoh_best2(self, band, typ, scheme, pat, txt, quals=None, dist=0):
        """aln::banded_alignment_score<BAND> with aln::Best2Sink<int32>(dist) (sink.h:96-116) -> (ok, (s1, x1, y1, s2, x2, y2))"""
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        arr = scheme.as_array()
        out = np.zeros(6, dtype=np.int64)
        ok = self.lib.ref_banded_gotoh_best2(ctypes.c_uint32(band), ctypes.c_int(typ), _p(arr, _i32p), _p(pat, _u8p), _p(quals, _u8p),
                                             ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.c_uint32(dist),
                                             out.ctypes.data_as(ctypes.c_void_p))
        return ok, tuple(int(v) for v in out)

    def full_gotoh_best2(self, typ, blocking, scheme, pat, txt, quals=None, min_score=SCORE_MIN, dist=0):
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        arr = scheme.as_array()
        out = np.zeros(6, dtype=np.int64)
        ok = self.lib.ref_full_gotoh_best2(ctypes.c_int(typ), ctypes.c_int(blocking), _p(arr, _i32p), _p(pat, _u8p), _p(quals, _u8p),
                                           ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score),
                                           ctypes.c_uint32(dist), out.ctypes.data_as(ctypes.c_void_p))
        return ok, tuple(int(v) for v in out)

    def banded_sw(self, band, typ, sw, pat, txt, min_score=SCORE_MIN):
        """aln::banded_alignment_score<BAND>( SmithWatermanAligner<TYPE,SimpleSmithWatermanScheme> ), sw = (match,
        mismatch, deletion, insertion) as signed scores (sw/sw_banded_inl.h:281-520)"""
        pat, txt = _c8(pat), _c8(txt)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.ref_banded_sw(ctypes.c_uint32(band), ctypes.c_int(typ), *[ctypes.c_int(int(v)) for v in sw],
                                    _p(pat, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)),
                                    ctypes.c_int32(min_score), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def full_sw(self, typ, blocking, sw, pat, txt, min_score=SCORE_MIN):
        """aln::alignment_score( SmithWatermanAligner<TYPE,SimpleSmithWatermanScheme,blocking tag> ) (sw/sw_inl.h)"""
        pat, txt = _c8(pat), _c8(txt)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.ref_full_sw(ctypes.c_int(typ), ctypes.c_int(blocking), *[ctypes.c_int(int(v)) for v in sw],
                                  _p(pat, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(len(txt)),
                                  ctypes.c_int32(min_score), ctypes.byref(sc), _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def full_ed(self, typ, blocking, pat, txt, min_score=SCORE_MIN):
        """aln::alignment_score( EditDistanceAligner<TYPE,blocking tag> ) (ed/ed_inl.h)"""
        pat, txt = _c8(pat), _c8(txt)
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.ref_full_ed(ctypes.c_int(typ), ctypes.c_int(blocking), _p(pat, _u8p), ctypes.c_uint32(len(pat)),
                                  _p(txt, _u8p), ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc),
                                  _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))

    def banded_gotoh_traceback(self, band, typ, scheme, pat, txt, quals=None, cap=4096):
        """the reference's banded_alignment_traceback<BAND,1024,16> with a recording backtracer
        -> (n_clip_calls, score, source, sink, ops uint8[] in backtracking order, (clip_before, clip_after)).
        The reference re-reads text[window_begin .. window_begin+BAND-2] unconditionally when it recomputes a
        checkpointed block (gotoh_banded_inl.h:432-433), i.e. past the end of a clipped text; the text is
        handed over followed by the same 255 sentinel its score pass substitutes there (:569), which keeps the
        two passes consistent (in nvBowtie those reads land in the genome after the window)."""
        n_txt = len(txt)
        pat, quals = _c8(pat), _c8(quals)
        txt = np.ascontiguousarray(np.concatenate([np.asarray(txt, dtype=np.uint8), np.full(64, 255, dtype=np.uint8)]))
        arr = scheme.as_array()
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        ops = np.zeros(cap, dtype=np.uint8); clips = np.zeros(2, dtype=np.uint32)
        no = ctypes.c_uint32()
        r = self.lib.ref_banded_gotoh_traceback_ex(
            ctypes.c_uint32(band), ctypes.c_int(typ), _p(arr, _i32p), _p(pat, _u8p), _p(quals, _u8p),
            ctypes.c_uint32(len(pat)), _p(txt, _u8p), ctypes.c_uint32(n_txt), ctypes.c_int32(SCORE_MIN),
            ctypes.byref(sc), _p(src, _u32p), _p(sk, _u32p), _p(ops, _u8p), ctypes.c_uint32(cap), ctypes.byref(no),
            _p(clips, _u32p))
        assert no.value <= cap
        return r, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), ops[:no.value].copy(), (int(clips[0]), int(clips[1]))

    def banded_sw_traceback(self, band, typ, sw, pat, txt, cap=4096):
        """the reference's banded_alignment_traceback<BAND,1024,16> for SmithWatermanAligner<TYPE> with a recording backtracer
        -> (n_clip_calls, score, source, sink, ops uint8[] in backtracking order, (clip_before, clip_after)); the text is handed over
        followed by sentinel bytes (the checkpointed recomputation re-reads text[window_begin + j] unconditionally)"""
        pat = _c8(pat)
        n_txt = len(txt)
        txt = np.ascontiguousarray(np.concatenate([np.asarray(txt, dtype=np.uint8), np.full(64, 255, dtype=np.uint8)]))
        sw = np.ascontiguousarray(np.asarray(sw, dtype=np.int32))
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        ops = np.zeros(cap, dtype=np.uint8); clips = np.zeros(2, dtype=np.uint32)
        no = ctypes.c_uint32()
        r = self.lib.ref_banded_sw_traceback(ctypes.c_uint32(band), ctypes.c_int(typ), _p(sw, _i32p), _p(pat, _u8p), ctypes.c_uint32(len(pat)),
                                             _p(txt, _u8p), ctypes.c_uint32(n_txt), ctypes.c_int32(SCORE_MIN), ctypes.byref(sc), _p(src, _u32p),
                                             _p(sk, _u32p), _p(ops, _u8p), ctypes.c_uint32(cap), ctypes.byref(no), _p(clips, _u32p))
        assert no.value <= cap
        return r, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), ops[:no.value].copy(), (int(clips[0]), int(clips[1]))

    def full_sw_traceback(self, typ, sw, pat, txt, min_score=SCORE_MIN, cap=8192):
        """the reference's alignment_traceback<256,1024,64> for SmithWatermanAligner<TYPE> with a recording backtracer
        -> (n_clip_calls, score, source, sink, ops uint8[] in backtracking order, (clip_before, clip_after))"""
        pat, txt = _c8(pat), _c8(txt)
        sw = np.ascontiguousarray(np.asarray(sw, dtype=np.int32))
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        ops = np.zeros(cap, dtype=np.uint8); clips = np.zeros(2, dtype=np.uint32)
        no = ctypes.c_uint32()
        r = self.lib.ref_full_sw_traceback(ctypes.c_int(typ), _p(sw, _i32p), _p(pat, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                           ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc), _p(src, _u32p), _p(sk, _u32p),
                                           _p(ops, _u8p), ctypes.c_uint32(cap), ctypes.byref(no), _p(clips, _u32p))
        assert no.value <= cap
        return r, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), ops[:no.value].copy(), (int(clips[0]), int(clips[1]))

    def full_gotoh_traceback(self, typ, scheme, pat, txt, quals=None, min_score=SCORE_MIN, cap=8192):
        """the reference's alignment_traceback<256,1024,64> with a recording backtracer
        -> (n_clip_calls, score, source, sink, ops uint8[] in backtracking order, (clip_before, clip_after))"""
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        arr = scheme.as_array()
        sc = ctypes.c_int32()
        src = np.zeros(2, dtype=np.uint32); sk = np.zeros(2, dtype=np.uint32)
        ops = np.zeros(cap, dtype=np.uint8); clips = np.zeros(2, dtype=np.uint32)
        no = ctypes.c_uint32()
        r = self.lib.ref_full_gotoh_traceback_ex(
            ctypes.c_int(typ), _p(arr, _i32p), _p(pat, _u8p), _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
            ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc), _p(src, _u32p), _p(sk, _u32p), _p(ops, _u8p),
            ctypes.c_uint32(cap), ctypes.byref(no), _p(clips, _u32p))
        assert no.value <= cap
        return r, sc.value, (int(src[0]), int(src[1])), (int(sk[0]), int(sk[1])), ops[:no.value].copy(), (int(clips[0]), int(clips[1]))

    def full_gotoh(self, typ, blocking, scheme, pat, txt, quals=None, min_score=SCORE_MIN):
        pat, txt, quals = _c8(pat), _c8(txt), _c8(quals)
        arr = scheme.as_array()
        sc = ctypes.c_int32()
        sk = np.zeros(2, dtype=np.uint32)
        ok = self.lib.ref_full_gotoh_ex(ctypes.c_int(typ), ctypes.c_int(blocking), _p(arr, _i32p), _p(pat, _u8p),
                                        _p(quals, _u8p), ctypes.c_uint32(len(pat)), _p(txt, _u8p),
                                        ctypes.c_uint32(len(txt)), ctypes.c_int32(min_score), ctypes.byref(sc),
                                        _p(sk, _u32p))
        return ok, sc.value, (int(sk[0]), int(sk[1]))
