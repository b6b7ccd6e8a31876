"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (genomeassembler_dev_amd/) never does.
See oracle/gasm_oracle.cpp for what is restated and how it is pinned.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborc.so")


def build(force=False):
    src = os.path.join(_HERE, "gasm_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_free.argtypes = [C.c_void_p]
        for name in ("orc_kmers_from_reads", "orc_get_contigs", "orc_assemble_matrix", "orc_assemble_velvet",
                     "orc_calc_breakscore"):
            getattr(L, name).restype = C.c_void_p
        L.orc_time_get_contigs.restype = C.c_uint64
        L.orc_levenshtein.restype = C.c_int
        _lib = L
    return _lib


def _pack(strs):
    """list of str/bytes -> (bytes buffer, uint64 offsets[n+1])"""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strs]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), off


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _blob(ptr, nbytes):
    raw = C.string_at(ptr, nbytes)
    lib().orc_free(ptr)
    out, pos = {}, 0
    while True:
        tag, n = struct.unpack_from("<QQ", raw, pos)
        pos += 16
        if tag == 0:
            break
        out[tag] = raw[pos:pos + n]
        pos += (n + 7) // 8 * 8
    return out


def _strs(b):
    head, _, rest = b.partition(b"\n")
    n = int(head)
    if n == 0:
        return []
    v = rest.split(b"\n")
    assert len(v) == n
    return [x.decode() for x in v]


def _vecs(b):
    (cnt,) = struct.unpack_from("<Q", b, 0)
    lens = np.frombuffer(b, dtype=np.uint64, count=cnt, offset=8).astype(np.int64)
    vals = np.frombuffer(b, dtype=np.float64, offset=8 + 8 * cnt)
    out, p = [], 0
    for l in lens:
        out.append(vals[p:p + l].copy())
        p += l
    return out


def kmers_from_reads(reads, k):
    buf, off = _pack(reads)
    n = C.c_uint64()
    p = lib().orc_kmers_from_reads(buf, _p(off), C.c_uint64(len(reads)), C.c_int(k), C.byref(n))
    return _strs(_blob(p, n.value)[1])


def get_contigs(read_kmers, dbg_kmer, seed, rows=10000):
    """Restates get_contigs (lib/DeNovoAssembler.cpp:86-206).  Returns a dict with the sorted unique contigs, the
    shuffle matrix as indices into them, and the graph detail used for edge-list/degree parity."""
    buf, off = _pack(read_kmers)
    n = C.c_uint64()
    p = lib().orc_get_contigs(buf, _p(off), C.c_uint64(len(read_kmers)), C.c_int(dbg_kmer), C.c_int(seed),
                              C.c_int(rows), C.byref(n))
    b = _blob(p, n.value)
    contigs = _strs(b[1])
    perm = np.frombuffer(b[8], dtype=np.uint32).reshape(rows, len(contigs)) if len(contigs) else \
        np.zeros((rows, 0), dtype=np.uint32)
    return dict(contigs=contigs, edge_prefix=_strs(b[2]), edge_suffix=_strs(b[3]), node=_strs(b[4]),
                node_in=np.frombuffer(b[5], dtype=np.int32), node_out=np.frombuffer(b[6], dtype=np.int32),
                branch=_strs(b[7]), perm=perm, distinct=_strs(b[9]), counts=np.frombuffer(b[10], dtype=np.int64))


def time_get_contigs(read_kmers_buf, off, nk, dbg_kmer, seed, rows=10000):
    return lib().orc_time_get_contigs(read_kmers_buf, _p(off), C.c_uint64(nk), C.c_int(dbg_kmer), C.c_int(seed),
                                      C.c_int(rows))


def time_build_score(reads, k, kmer, bp_kmer, bp_prob):
    """cpu_baseline leg of bench.py: k-mers -> contigs -> scores of one segment, single thread.  Returns (#k-mers, s)."""
    import time
    rb, ro = _pack(reads)
    kb, ko = _pack(bp_kmer)
    prob = np.ascontiguousarray(bp_prob, dtype=np.float64)
    cs = C.c_uint64()
    f = lib().orc_time_build_score
    f.restype = C.c_uint64
    t0 = time.perf_counter()
    n = f(rb, _p(ro), C.c_uint64(len(reads)), C.c_int(k), C.c_int(kmer), kb, _p(ko), C.c_uint64(len(bp_kmer)), _p(prob),
          C.byref(cs))
    return int(n), time.perf_counter() - t0


def build_score(reads, k, kmer, bp_kmer, bp_prob):
    """bench.py: k-mers -> contigs -> scores of one segment in one call, single thread (ctypes releases the GIL, so several
    of these run side by side from Python threads).  Returns dict(n_kmers, seconds (the three stages alone), contigs,
    kmer_breaks, bp_score)."""
    rb, ro = _pack(reads)
    kb, ko = _pack(bp_kmer)
    prob = np.ascontiguousarray(bp_prob, dtype=np.float64)
    sec, n = C.c_double(), C.c_uint64()
    f = lib().orc_build_score
    f.restype = C.c_void_p
    p = f(rb, _p(ro), C.c_uint64(len(reads)), C.c_int(k), C.c_int(kmer), kb, _p(ko), C.c_uint64(len(bp_kmer)), _p(prob), C.byref(sec),
          C.byref(n))
    b = _blob(p, n.value)
    return dict(n_kmers=int(np.frombuffer(b[4], dtype=np.uint64)[0]), seconds=sec.value, contigs=_strs(b[1]),
                kmer_breaks=np.frombuffer(b[2], dtype=np.int32), bp_score=np.frombuffer(b[3], dtype=np.float64))


def assemble_contigs(contigs, perm, dbg_kmer):
    """Restates assemble_contigs(contig_matrix, dbg_kmer) (lib/DeNovoAssembler.cpp:215-305); the matrix is given as
    `perm` (rows × len(contigs) indices)."""
    buf, off = _pack(contigs)
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    n, err = C.c_uint64(), C.c_int()
    p = lib().orc_assemble_matrix(buf, _p(off), C.c_uint64(len(contigs)), _p(perm), C.c_uint64(perm.shape[0]),
                                  C.c_int(dbg_kmer), C.byref(n), C.byref(err))
    b = _blob(p, n.value)
    if err.value:
        raise IndexError("basic_string::substr out of range (contig shorter than the overlap)")
    return _strs(b[1])


def assemble_contigs_velvet(velvet_contigs, dbg_kmer, seed, rows=20000):
    """Restates assemble_contigs(velvet_contigs, dbg_kmer, seed) (lib/BreakageScorer.cpp:80-174)."""
    buf, off = _pack(velvet_contigs)
    n, err = C.c_uint64(), C.c_int()
    p = lib().orc_assemble_velvet(buf, _p(off), C.c_uint64(len(velvet_contigs)), C.c_int(dbg_kmer), C.c_int(seed),
                                  C.c_int(rows), C.byref(n), C.byref(err))
    b = _blob(p, n.value)
    if err.value:
        raise IndexError("basic_string::substr out of range (contig shorter than the overlap)")
    return _strs(b[1])


def calc_breakscore(path, sequencing_reads, true_solution, kmer, bp_kmer, bp_prob, velvet=False, with_lev=True,
                    with_freq=True):
    """Restates calc_breakscore (lib/DeNovoAssembler.cpp:317-477; velvet=True: lib/BreakageScorer.cpp:186-353)."""
    pb, po = _pack(path)
    rb, ro = _pack(sequencing_reads)
    kb, ko = _pack(bp_kmer)
    prob = np.ascontiguousarray(bp_prob, dtype=np.float64)
    t = true_solution.encode() if isinstance(true_solution, str) else bytes(true_solution)
    n = C.c_uint64()
    p = lib().orc_calc_breakscore(pb, _p(po), C.c_uint64(len(path)), rb, _p(ro), C.c_uint64(len(sequencing_reads)), t,
                                  C.c_uint64(len(t)), C.c_int(kmer), kb, _p(ko), C.c_uint64(len(bp_kmer)), _p(prob),
                                  C.c_int(int(velvet)), C.c_int(int(with_lev)), C.c_int(int(with_freq)), C.byref(n))
    b = _blob(p, n.value)
    out = dict(sequence=list(path), sequence_len=np.frombuffer(b[1], dtype=np.int32),
               bp_score=np.frombuffer(b[2], dtype=np.float64),
               bp_score_norm_by_break_freqs=np.frombuffer(b[3], dtype=np.float64),
               bp_score_norm_by_len=np.frombuffer(b[4], dtype=np.float64),
               kmer_breaks=np.frombuffer(b[5], dtype=np.int32), lev_dist_vs_true=np.frombuffer(b[6], dtype=np.int32))
    if 7 in b:
        out["path_freq"] = _vecs(b[7])
        out["path_freq_by_input"] = _vecs(b[8])
    if 9 in b:
        out["path_prob_dist_startpos"] = np.frombuffer(b[9], dtype=np.int32)
        out["path_prob_dist"] = _vecs(b[10])
    return out


def levenshtein(q, t, infix=False):
    q = q.encode() if isinstance(q, str) else q
    t = t.encode() if isinstance(t, str) else t
    return lib().orc_levenshtein(q, C.c_uint64(len(q)), t, C.c_uint64(len(t)), C.c_int(int(infix)))


def count_windows(reads, k, keys):
    rb, ro = _pack(reads)
    kb, ko = _pack(keys)
    out = np.zeros(len(keys), dtype=np.int64)
    lib().orc_count_windows(rb, _p(ro), C.c_uint64(len(reads)), C.c_int(k), kb, _p(ko), C.c_uint64(len(keys)), _p(out))
    return out


def normalise_tables(prob, sizes):
    p = np.array(prob, dtype=np.float64)
    s = np.array(sizes, dtype=np.uint64)
    lib().orc_normalise_tables(_p(p), _p(s), C.c_uint64(len(s)))
    return p


def kmer_from_seq(genome, kmer, bp_kmer, bp_prob):
    """lib/GenerateReads.R:243-259: per-position probability of the genome's kmer-long windows"""
    g = genome.encode() if isinstance(genome, str) else bytes(genome)
    kb, ko = _pack(bp_kmer)
    prob = np.ascontiguousarray(bp_prob, dtype=np.float64)
    out = np.zeros(max(len(g) - kmer + 1, 0), dtype=np.float64)
    lib().orc_kmer_from_seq(g, C.c_uint64(len(g)), C.c_int(kmer), kb, _p(ko), C.c_uint64(len(bp_kmer)), _p(prob), _p(out))
    return out


def ks_statistic(x, y):
    """lib/DeNovoAssembler.R:414-424: D of R's two-sample ks.test after dropping NAs (NaN for an empty sample)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    f = lib().orc_ks_statistic
    f.restype = C.c_double
    return f(_p(x), C.c_uint64(len(x)), _p(y), C.c_uint64(len(y)))


def coverage_percent(starts, lens, seq_len):
    """lib/DeNovoAssembler.R:432-445: contig_frac_len"""
    a = np.ascontiguousarray(starts, dtype=np.int64)
    b = np.ascontiguousarray(lens, dtype=np.int64)
    f = lib().orc_coverage_percent
    f.restype = C.c_double
    return f(_p(a), _p(b), C.c_uint64(len(a)), C.c_int64(seq_len))


def sim_weight_shift(genome_lengths, kmer, bp_kmer, bp_prob):
    """the fixed-point shift of the weighted simulator for a batch of genomes: 52, lowered until the largest weight times the
    most start positions of any genome stays below 2^62 (a segment's running sum must not wrap)"""
    max_p = max((p for key, p in zip(bp_kmer, bp_prob) if len(key) == kmer), default=0.0)
    max_np = max([max(L - kmer + 1, 0) for L in genome_lengths] + [1])
    shift = 52
    while shift > 0 and float(np.ldexp(max_p, shift)) * float(max_np) >= 4611686018427387904.0:
        shift -= 1
    return shift


def simulate_starts(genome, seg_index, read_len, coverage, seed, kmer=8, bp_kmer=None, bp_prob=None, weight_shift=52):
    """lib/GenerateReads.R:235-313 with the build's pinned random stream: kept 0-based read starts, in draw order"""
    g = genome.encode() if isinstance(genome, str) else bytes(genome)
    nd = int(np.ceil(coverage * len(g) / read_len)) + 1
    out = np.zeros(nd, dtype=np.uint32)
    f = lib().orc_simulate_starts
    f.restype = C.c_uint64
    if bp_prob is not None:
        kb, ko = _pack(bp_kmer)
        prob = np.ascontiguousarray(bp_prob, dtype=np.float64)
        n = f(g, C.c_uint64(len(g)), C.c_uint32(seg_index), C.c_uint32(read_len), C.c_double(coverage), C.c_uint64(seed), C.c_int(kmer), kb, _p(ko),
              C.c_uint64(len(bp_kmer)), _p(prob), C.c_int(weight_shift), _p(out))
    else:
        n = f(g, C.c_uint64(len(g)), C.c_uint32(seg_index), C.c_uint32(read_len), C.c_double(coverage), C.c_uint64(seed), C.c_int(kmer), None, None,
              C.c_uint64(0), None, C.c_int(weight_shift), _p(out))
    return out[:int(n)].copy()
