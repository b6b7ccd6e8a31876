"""Host-side mirror of the reference's R-callable entry points (same names, argument meaning and error behaviour),
over the C ABI of libgasm (include/gasm.h):

    get_contigs(read_kmers, dbg_kmer, seed)                  lib/DeNovoAssembler.cpp:86-206
    assemble_contigs(contig_matrix, dbg_kmer)                lib/DeNovoAssembler.cpp:215-305
    assemble_contigs_velvet(velvet_contigs, dbg_kmer, seed)  lib/BreakageScorer.cpp:80-174
    calc_breakscore(path, sequencing_reads, true_solution, kmer, bp_kmer, bp_prob)
                                                              lib/DeNovoAssembler.cpp:317-477 (variant="own")
                                                              lib/BreakageScorer.cpp:186-353  (variant="velvet")
    get_kmers_from_reads(reads, dbg_kmer)                    lib/DeNovoAssembler.R:109-130

Where the reference raises an R error (a C++ exception through Rcpp), these raise GasmError / IndexError.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GasmError, check, default_context, lib


def _pack(strs):
    """sequence of str/bytes -> (bytes, uint64 offsets[n+1])"""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strs]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), off


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _strings(data_ptr, off_ptr, n):
    if n == 0:
        return []
    off = np.ctypeslib.as_array(C.cast(off_ptr, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    raw = C.string_at(data_ptr, int(off[-1])) if off[-1] else b""
    return [raw[int(off[i]):int(off[i + 1])].decode() for i in range(n)]


def _arr(ptr, ctype, n):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=ctype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).copy()


def unpack_kmers(keys, k, words=1):
    """2-bit packed k-mers (gasm.h layout: base 0 most significant) -> list of str"""
    keys = np.asarray(keys, dtype=np.uint64).reshape(-1, words)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((keys.shape[0], k), dtype=np.uint8)
    for j in range(k):
        bit = 2 * (k - 1 - j)
        w = words - 1 - bit // 64
        out[:, j] = lut[((keys[:, w] >> np.uint64(bit % 64)) & np.uint64(3)).astype(np.int64)]
    return [r.tobytes().decode() for r in out]


def get_kmers_from_reads(reads, dbg_kmer):
    """Every k-mer of every read, read-major, duplicates kept (lib/DeNovoAssembler.R:109-130).  Host convenience for
    callers of get_contigs; the batch path (SegmentBatch) extracts k-mers on the GPU instead."""
    out = []
    for r in reads:
        out.extend(r[i:i + dbg_kmer] for i in range(len(r) - dbg_kmer + 1))
    return out


class ContigMatrix:
    """get_contigs' return value without materialising rows x contigs strings: the sorted unique contigs plus the
    shuffle matrix as indices.  `rows()` / `as_lists()` give the reference's list-of-character-vectors shape."""

    def __init__(self, contigs, perm, dbg_kmer, distinct_keys, distinct_mult, words):
        self.contigs, self.perm, self.dbg_kmer = contigs, perm, dbg_kmer
        self.distinct_keys, self.distinct_mult, self.words = distinct_keys, distinct_mult, words

    def distinct_kmers(self):
        return unpack_kmers(self.distinct_keys, self.dbg_kmer, self.words)

    def row(self, i):
        return [self.contigs[j] for j in self.perm[i]]

    def as_lists(self):
        return [self.row(i) for i in range(self.perm.shape[0])]


def get_contigs(read_kmers, dbg_kmer, seed, matrix_rows=10000, ctx=None, as_lists=False):
    """read_kmers: list of str (each exactly dbg_kmer long), or a uint8 array of shape (n, dbg_kmer)."""
    ctx = ctx or default_context()
    if isinstance(read_kmers, np.ndarray):
        a = np.ascontiguousarray(read_kmers, dtype=np.uint8)
        if a.ndim != 2 or a.shape[1] != dbg_kmer:
            raise ValueError("read_kmers array must have shape (n, dbg_kmer)")
        n, buf = a.shape[0], a
        p = _ptr(a)
    else:
        n = len(read_kmers)
        for s in read_kmers:
            if len(s) != dbg_kmer:
                raise ValueError(f"every k-mer must be exactly dbg_kmer={dbg_kmer} long (got {len(s)})")
        buf = "".join(read_kmers).encode() if n and isinstance(read_kmers[0], str) else b"".join(read_kmers)
        p = C.cast(C.c_char_p(buf), C.c_void_p)
    h = C.c_void_p()
    check(lib().gasm_get_contigs(ctx.h, p, n, int(dbg_kmer), int(seed), int(matrix_rows), C.byref(h)))
    return _contig_matrix(h, dbg_kmer, as_lists)


def get_contigs_from_reads(reads, dbg_kmer, seed, matrix_rows=10000, ctx=None, as_lists=False):
    """get_kmers_from_reads + get_contigs in one call (gasm_get_contigs_from_reads): the k-mers are taken on the GPU from the
    packed reads instead of being exploded into len(reads) * (read_len - k + 1) strings first (lib/DeNovoAssembler.R:109-130).
    reads: list of str / bytes.  Same ContigMatrix as get_contigs(get_kmers_from_reads(reads, k), k, seed)."""
    ctx = ctx or default_context()
    buf, off = _pack(reads)
    h = C.c_void_p()
    check(lib().gasm_get_contigs_from_reads(ctx.h, buf, _ptr(off), len(reads), int(dbg_kmer), int(seed), int(matrix_rows), C.byref(h)))
    return _contig_matrix(h, dbg_kmer, as_lists)


def _contig_matrix(h, dbg_kmer, as_lists):
    try:
        L = lib()
        nc = L.gasm_contigs_count(h)
        contigs = _strings(L.gasm_contigs_data(h), L.gasm_contigs_offsets(h), nc)
        rows = L.gasm_contigs_rows(h)
        perm = _arr(L.gasm_contigs_perm(h), C.c_uint32, rows * nc).reshape(rows, nc)
        nd, words = L.gasm_contigs_distinct_count(h), L.gasm_contigs_key_words(h)
        dk = _arr(L.gasm_contigs_distinct_keys(h), C.c_uint64, nd * words)
        dm = _arr(L.gasm_contigs_distinct_mult(h), C.c_uint32, nd)
    finally:
        lib().gasm_contigs_free(h)
    m = ContigMatrix(contigs, perm, dbg_kmer, dk, dm, words)
    return m.as_lists() if as_lists else m


def _strlist(h):
    L = lib()
    try:
        return _strings(L.gasm_strlist_data(h), L.gasm_strlist_offsets(h), L.gasm_strlist_count(h))
    finally:
        L.gasm_strlist_free(h)


def _raise(e):
    if e.status == -6:
        raise IndexError(str(e)) from None
    raise e


class Scaffolds:
    """Distinct scaffolds of assemble_contigs left on the GPU (2-bit, the reference's order: longest first).  Pass it to
    calc_breakscore as `path`; `.strings()` makes the text."""

    def __init__(self, h, ctx):
        self.h, self.ctx = h, ctx
        n = lib().gasm_scaffolds_count(h)
        off = _arr(lib().gasm_scaffolds_offsets(h), C.c_uint64, n + 1)
        self.lengths = np.diff(off).astype(np.int64) if n else np.zeros(0, np.int64)
        r = C.c_uint64()
        self.merge_device = {0: None, 1: "gpu", 2: "host", 3: "gpu+host"}[lib().gasm_scaffolds_merge_device(h, C.byref(r))]
        self.rows_on_host = int(r.value)

    def __len__(self):
        return len(self.lengths)

    def strings(self):
        h = C.c_void_p()
        check(lib().gasm_scaffolds_fetch(self.h, C.byref(h)))
        return _strlist(h)

    def close(self):
        if self.h:
            lib().gasm_scaffolds_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def assemble_contigs(contig_matrix, dbg_kmer, ctx=None, on_device=False):
    """contig_matrix: a ContigMatrix, or the reference's list of equally long lists of strings.
    on_device=True: returns a `Scaffolds` handle (nothing is copied back) instead of the list of strings."""
    if isinstance(contig_matrix, ContigMatrix):
        contigs, perm = contig_matrix.contigs, contig_matrix.perm
    else:
        rows = [list(r) for r in contig_matrix]
        width = len(rows[0]) if rows else 0
        if any(len(r) != width for r in rows):
            raise ValueError("all rows of contig_matrix must have the same length")
        contigs = sorted(set(s for r in rows for s in r))
        ix = {s: i for i, s in enumerate(contigs)}
        perm = np.array([[ix[s] for s in r] for r in rows], dtype=np.uint32).reshape(len(rows), width)
    buf, off = _pack(contigs)
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    h = C.c_void_p()
    try:
        if on_device:
            ctx = ctx or default_context()
            check(lib().gasm_assemble_contigs_dev(ctx.h, buf, _ptr(off), len(contigs), _ptr(perm), perm.shape[0], perm.shape[1], int(dbg_kmer),
                                                  C.byref(h)))
            return Scaffolds(h, ctx)
        check(lib().gasm_assemble_contigs(ctx.h if ctx else None, buf, _ptr(off), len(contigs), _ptr(perm), perm.shape[0],
                                          perm.shape[1], int(dbg_kmer), C.byref(h)))
    except GasmError as e:
        _raise(e)
    return _strlist(h)


def assemble_contigs_velvet(velvet_contigs, dbg_kmer, seed, rows=20000, ctx=None, on_device=False):
    buf, off = _pack(velvet_contigs)
    h = C.c_void_p()
    try:
        if on_device:
            ctx = ctx or default_context()
            check(lib().gasm_assemble_contigs_velvet_dev(ctx.h, buf, _ptr(off), len(velvet_contigs), int(dbg_kmer), int(seed), int(rows), C.byref(h)))
            return Scaffolds(h, ctx)
        check(lib().gasm_assemble_contigs_velvet(ctx.h if ctx else None, buf, _ptr(off), len(velvet_contigs), int(dbg_kmer),
                                                 int(seed), int(rows), C.byref(h)))
    except GasmError as e:
        _raise(e)
    return _strlist(h)


def levenshtein(query, target, infix=False):
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    out = C.c_int32()
    check(lib().gasm_levenshtein(q, len(q), t, len(t), int(infix), C.byref(out)))
    return out.value


def calc_breakscore(path, sequencing_reads, true_solution, kmer, bp_kmer, bp_prob, variant="own", with_lev=True,
                    with_freq=True, with_ks=False, ctx=None):
    """Returns a dict with the names of the reference's Rcpp::List (lib/DeNovoAssembler.cpp:467-476 /
    lib/BreakageScorer.cpp:343-353).  path_freq rows follow bp_kmer order (the reference: hash order).
    with_ks adds stat_test_KS: what lib/DeNovoAssembler.R:419-424 computes from path_freq afterwards."""
    ctx = ctx or default_context()
    velvet = variant == "velvet"
    if variant not in ("own", "velvet"):
        raise ValueError("variant must be 'own' or 'velvet'")
    rb, ro = _pack(sequencing_reads)
    kb, ko = _pack(bp_kmer)
    prob = np.ascontiguousarray(bp_prob, dtype=np.float64)
    t = true_solution.encode() if isinstance(true_solution, str) else bytes(true_solution)
    flags = (_lib.WANT_LEV if with_lev else 0) | (_lib.WANT_FREQ if with_freq and not velvet else 0) | (_lib.WANT_KS if with_ks else 0)
    h = C.c_void_p()
    if isinstance(path, Scaffolds):
        # the scaffolds are on the device already (assemble_contigs(..., on_device=True)): no text, no upload
        check(lib().gasm_calc_breakscore_dev(ctx.h, path.h, rb, _ptr(ro), len(sequencing_reads), t, len(t), int(kmer), kb, _ptr(ko), len(bp_kmer),
                                             _ptr(prob), _lib.SCORE_VELVET if velvet else _lib.SCORE_OWN, flags, C.byref(h)))
    else:
        pb, po = _pack(path)
        check(lib().gasm_calc_breakscore(ctx.h, pb, _ptr(po), len(path), rb, _ptr(ro), len(sequencing_reads), t, len(t), int(kmer),
                                         kb, _ptr(ko), len(bp_kmer), _ptr(prob), _lib.SCORE_VELVET if velvet else _lib.SCORE_OWN,
                                         flags, C.byref(h)))
    L = lib()
    try:
        n = L.gasm_scores_count(h)
        out = dict(sequence=(path if isinstance(path, Scaffolds) else list(path)),
                   sequence_len=_arr(L.gasm_scores_sequence_len(h), C.c_int32, n),
                   bp_score=_arr(L.gasm_scores_bp_score(h), C.c_double, n),
                   bp_score_norm_by_break_freqs=_arr(L.gasm_scores_norm_by_break_freqs(h), C.c_double, n),
                   bp_score_norm_by_len=_arr(L.gasm_scores_norm_by_len(h), C.c_double, n),
                   kmer_breaks=_arr(L.gasm_scores_kmer_breaks(h), C.c_int32, n),
                   lev_dist_vs_true=_arr(L.gasm_scores_lev_dist(h), C.c_int32, n))
        if velvet:
            out["path_prob_dist_startpos"] = _arr(L.gasm_scores_startpos(h), C.c_int32, n)
            off = _arr(L.gasm_scores_prob_dist_offsets(h), C.c_uint64, n + 1)
            pd = _arr(L.gasm_scores_prob_dist(h), C.c_double, int(off[-1]) if n else 0)
            out["path_prob_dist"] = [pd[int(off[i]):int(off[i + 1])] for i in range(n)]
        elif with_freq:
            f = _arr(L.gasm_scores_path_freq(h), C.c_double, n * len(bp_kmer))
            out["path_freq"] = f.reshape(n, len(bp_kmer))
        if with_ks:
            out["stat_test_KS"] = _arr(L.gasm_scores_ks(h), C.c_double, n)
        out["lev_device"] = {0: None, 1: "gpu", 2: "host"}[L.gasm_scores_lev_device(h)]      # (not a reference column: who did the work)
    finally:
        L.gasm_scores_free(h)
    return out


def coverage_percent(starts, lens, seq_len, ctx=None):
    """contig_frac_len (lib/DeNovoAssembler.R:432-445): percentage of [1, seq_len] covered by the union of the inclusive
    ranges [start, start + len]"""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(starts, dtype=np.int64)
    b = np.ascontiguousarray(lens, dtype=np.int64)
    out = C.c_double()
    check(lib().gasm_coverage_percent(ctx.h, _ptr(a), _ptr(b), len(a), int(seq_len), C.byref(out)))
    return out.value
