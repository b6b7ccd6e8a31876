"""SegmentBatch: reads of many independent segments in, contigs + breakage scores out, everything resident in HBM
between the calls (gasm_batch_* in include/gasm.h).  This is the reads-level surface the reference only has in R
(lib/DeNovoAssembler.R:58-68: get_reads -> get_kmers_from_reads -> get_contigs -> calc_breakscore)."""
import ctypes as C

import numpy as np

from . import qtable
from ._lib import check, default_context, lib
from .api import unpack_kmers


class SegmentBatch:
    def __init__(self, reads, seg_read_off, fixed_len=0, read_off=None, ctx=None):
        """reads: uint8 array (ASCII ACGT) of all reads of all segments, concatenated segment after segment.
        seg_read_off: n_segments+1 read indices.  Either fixed_len > 0 or read_off (n_reads+1 base offsets)."""
        self.ctx = ctx or default_context()
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        seg = np.ascontiguousarray(seg_read_off, dtype=np.uint64)
        self.n_segments = len(seg) - 1
        self.n_reads = int(seg[-1])
        ro = None
        if read_off is not None:
            ro = np.ascontiguousarray(read_off, dtype=np.uint64)
            fixed_len = 0
        h = C.c_void_p()
        check(lib().gasm_batch_create(self.ctx.h, reads.ctypes.data_as(C.c_void_p),
                                      ro.ctypes.data_as(C.c_void_p) if ro is not None else None, self.n_reads, int(fixed_len),
                                      seg.ctypes.data_as(C.c_void_p), self.n_segments, C.byref(h)))
        self.h = h
        self.k = None
        self._table = None

    @classmethod
    def from_strings(cls, segments, ctx=None):
        """segments: list (one per segment) of lists of read strings"""
        seg = np.zeros(len(segments) + 1, dtype=np.uint64)
        off, data = [0], []
        for i, rs in enumerate(segments):
            seg[i + 1] = seg[i] + len(rs)
            for r in rs:
                b = r.encode() if isinstance(r, str) else bytes(r)
                data.append(b)
                off.append(off[-1] + len(b))
        buf = np.frombuffer(b"".join(data), dtype=np.uint8) if data else np.zeros(0, dtype=np.uint8)
        return cls(buf, seg, read_off=np.array(off, dtype=np.uint64), ctx=ctx)

    @classmethod
    def from_packed(cls, words, seg_read_off, fixed_len=0, read_off=None, ctx=None):
        """reads that are 2-bit packed already (gasm_batch_create_packed): a quarter of the bytes over PCIe"""
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        w = np.ascontiguousarray(words, dtype=np.uint64)
        seg = np.ascontiguousarray(seg_read_off, dtype=np.uint64)
        ro = np.ascontiguousarray(read_off, dtype=np.uint64) if read_off is not None else None
        self.n_segments, self.n_reads = len(seg) - 1, int(seg[-1])
        h = C.c_void_p()
        check(lib().gasm_batch_create_packed(self.ctx.h, w.ctypes.data_as(C.c_void_p), ro.ctypes.data_as(C.c_void_p) if ro is not None else None,
                                             self.n_reads, 0 if ro is not None else int(fixed_len), seg.ctypes.data_as(C.c_void_p),
                                             self.n_segments, C.byref(h)))
        self.h, self.k, self._table = h, None, None
        return self

    @classmethod
    def simulate(cls, genomes, read_len, coverage, seed, kmer=8, table=None, ctx=None):
        """reads simulated on the device from the segments' genomes (lib/GenerateReads.R:235-313, gasm_batch_simulate):
        genomes = list of str/bytes/uint8 arrays; table = normalised 69 904-row table for probability-weighted starts
        (the reference's ultrasonication model) or None for uniform starts"""
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        bs = [g.encode() if isinstance(g, str) else (g.tobytes() if hasattr(g, "tobytes") else bytes(g)) for g in genomes]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        t = np.ascontiguousarray(table, dtype=np.float64) if table is not None else None
        h = C.c_void_p()
        check(lib().gasm_batch_simulate(self.ctx.h, b"".join(bs), off.ctypes.data_as(C.c_void_p), len(bs), int(read_len), float(coverage), int(seed),
                                        int(kmer), t.ctypes.data_as(C.c_void_p) if t is not None else None, C.byref(h)))
        self.h, self.k, self._table = h, None, None
        self.n_segments = len(bs)
        self.n_reads = int(lib().gasm_batch_total_reads(h))
        return self

    def read_starts(self):
        """(seg_read_off[n_segments+1], 0-based start of every simulated read in its genome)"""
        so, st = C.c_void_p(), C.c_void_p()
        check(lib().gasm_batch_fetch_read_starts(self.h, C.byref(so), C.byref(st)))
        seg = np.ctypeslib.as_array(C.cast(so, C.POINTER(C.c_uint64)), shape=(self.n_segments + 1,)).copy()
        n = int(seg[-1])
        starts = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        return seg, starts

    @classmethod
    def from_fastq(cls, paths, non_acgt="drop", ctx=None):
        """one FASTQ/FASTA file (plain or .gz) per segment, read and packed by libgasm (gasm_batch_from_files); reads with a
        base outside ACGT are dropped (count in `.dropped_reads`) or, with non_acgt='error', refused"""
        from .seqio import _paths
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        h, dropped = C.c_void_p(), C.c_uint64()
        check(lib().gasm_batch_from_files(self.ctx.h, _paths(paths), len(paths), 1 if non_acgt == "error" else 0, C.byref(h), C.byref(dropped)))
        self.h, self.k, self._table = h, None, None
        self.n_segments = len(paths)
        self.n_reads = int(lib().gasm_batch_total_reads(h))
        self.dropped_reads = int(dropped.value)
        return self

    def build(self, k, genome_len_hint=0):
        check(lib().gasm_batch_build(self.h, int(k), int(genome_len_hint)))
        self.k = int(k)
        return self

    def score(self, kmer=8, table=None):
        t = np.ascontiguousarray(qtable.load_normalised() if table is None else table, dtype=np.float64)
        if t.size != qtable.ROWS:
            raise ValueError(f"table must hold {qtable.ROWS} probabilities")
        self._table = t
        check(lib().gasm_batch_score(self.h, int(kmer), t.ctypes.data_as(C.c_void_p)))
        return self

    def total_kmers(self):
        return int(lib().gasm_batch_total_kmers(self.h))

    def distinct(self):
        """(seg_off[n_segments+1], keys uint64, multiplicities uint32, words)"""
        so, ks, ms, w = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        check(lib().gasm_batch_fetch_distinct(self.h, C.byref(so), C.byref(ks), C.byref(ms), C.byref(w)))
        seg = np.ctypeslib.as_array(C.cast(so, C.POINTER(C.c_uint64)), shape=(self.n_segments + 1,)).copy()
        n = int(seg[-1])
        keys = np.ctypeslib.as_array(C.cast(ks, C.POINTER(C.c_uint64)), shape=(n * w.value,)).copy() if n else np.zeros(0, np.uint64)
        mult = np.ctypeslib.as_array(C.cast(ms, C.POINTER(C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        return seg, keys, mult, w.value

    def distinct_kmers(self, segment):
        seg, keys, mult, w = self.distinct()
        a, b = int(seg[segment]), int(seg[segment + 1])
        return unpack_kmers(keys[a * w:b * w], self.k, w), mult[a:b]

    def graph(self):
        """(edge_flags uint8, edge_next uint32), one entry per distinct k-mer in the order of distinct(): bit 0 of a flag = the
        edge's source node is branching, bit 1 (first out-edge of a node) = the node has two or more in-edges; edge_next =
        the edge the walk continues with, 0xFFFFFFFF at the end of a contig (lib/DeNovoAssembler.cpp:125-189)"""
        seg = self.distinct()[0]
        n = int(seg[-1])
        fl, nx = C.c_void_p(), C.c_void_p()
        check(lib().gasm_batch_fetch_graph(self.h, C.byref(fl), C.byref(nx)))
        if not n:
            return np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        return (np.ctypeslib.as_array(C.cast(fl, C.POINTER(C.c_uint8)), shape=(n,)).copy(),
                np.ctypeslib.as_array(C.cast(nx, C.POINTER(C.c_uint32)), shape=(n,)).copy())

    def contigs_raw(self):
        """(seg_contig_off[n_segments+1], contig base offsets[n_contigs+1], ASCII bytes)"""
        so, off, data = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().gasm_batch_fetch_contigs(self.h, C.byref(so), C.byref(off), C.byref(data)))
        seg = np.ctypeslib.as_array(C.cast(so, C.POINTER(C.c_uint64)), shape=(self.n_segments + 1,)).copy()
        nc = int(seg[-1])
        o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(nc + 1,)).copy()
        raw = C.string_at(data, int(o[-1])) if nc and o[-1] else b""
        return seg, o, raw

    def contigs(self, segment=None):
        seg, o, raw = self.contigs_raw()
        rng = range(self.n_segments) if segment is None else [segment]
        out = [[raw[int(o[c]):int(o[c + 1])].decode() for c in range(int(seg[s]), int(seg[s + 1]))] for s in rng]
        return out if segment is None else out[0]

    def scores(self):
        """dict of per-contig arrays, in contigs_raw() order"""
        ps = [C.c_void_p() for _ in range(5)]
        check(lib().gasm_batch_fetch_scores(self.h, *[C.byref(p) for p in ps]))
        seg, _, _ = self.contigs_raw()
        n = int(seg[-1])

        def arr(p, ct):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)).copy() if n else np.zeros(0, ct)
        return dict(bp_score=arr(ps[0], C.c_double), bp_score_norm_by_break_freqs=arr(ps[1], C.c_double),
                    bp_score_norm_by_len=arr(ps[2], C.c_double), kmer_breaks=arr(ps[3], C.c_int32),
                    sequence_len=arr(ps[4], C.c_int32), seg_contig_off=seg)

    def score_fixed(self):
        """(int64 fixed-point breakage sums per contig, shift): bp_score == fx * 2**-shift exactly"""
        p, sh = C.c_void_p(), C.c_int()
        check(lib().gasm_batch_fetch_score_fixed(self.h, C.byref(p), C.byref(sh)))
        seg, _, _ = self.contigs_raw()
        n = int(seg[-1])
        fx = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64)), shape=(n,)).copy() if n else np.zeros(0, np.int64)
        return fx, sh.value

    def guided(self):
        """breakage-score-guided scaffolds (not in the reference) of a built + scored batch:
        list per segment of dicts(sequence, bp_score, bp_score_norm_by_len, kmer_breaks), longest first"""
        check(lib().gasm_batch_guided(self.h))
        ps = [C.c_void_p() for _ in range(6)]
        check(lib().gasm_batch_fetch_guided(self.h, *[C.byref(p) for p in ps]))
        seg = np.ctypeslib.as_array(C.cast(ps[0], C.POINTER(C.c_uint64)), shape=(self.n_segments + 1,)).copy()
        n = int(seg[-1])
        off = np.ctypeslib.as_array(C.cast(ps[1], C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        raw = C.string_at(ps[2], int(off[-1])) if n and off[-1] else b""
        arr = lambda p, ct: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)).copy() if n else np.zeros(0, ct)
        bp, nl, br = arr(ps[3], C.c_double), arr(ps[4], C.c_double), arr(ps[5], C.c_int32)
        return [[dict(sequence=raw[int(off[i]):int(off[i + 1])].decode(), bp_score=bp[i], bp_score_norm_by_len=nl[i], kmer_breaks=int(br[i]))
                 for i in range(int(seg[s]), int(seg[s + 1]))] for s in range(self.n_segments)]

    def close(self):
        if self.h:
            lib().gasm_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
