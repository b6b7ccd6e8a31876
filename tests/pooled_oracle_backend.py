"""CPU stand-in for one rank's device work in a pooled build (TEST INFRASTRUCTURE: numpy + the oracle).  It implements the
backend interface of genomeassembler_dev_amd/pooled.py — same wire formats as libgasm (2-bit keys, first base most
significant; 32-bit counts; reads as word-aligned 2-bit pieces) — so that the protocol code itself (ownership, run
directories, offsets in the received buffers, the three exchanges over torch.distributed) runs on a machine without a
GPU, with gloo, and can be compared with a single-process oracle run.  k <= 31 only."""
import numpy as np
import torch

from oracle import orc

_CODE = np.zeros(256, np.uint64)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def keys_of_reads(reads, k):
    """(n_reads, rl) uint8 ASCII -> uint64 keys of every k-mer, read-major"""
    c = _CODE[reads]
    n, rl = reads.shape
    nk = rl - k + 1
    if n == 0 or nk <= 0:
        return np.zeros(0, np.uint64)
    key = np.zeros((n, nk), np.uint64)
    for j in range(k):
        key = (key << np.uint64(2)) | c[:, j:j + nk]
    return key.reshape(-1)


def unpack(keys, k):
    lut = np.frombuffer(b"ACGT", np.uint8)
    out = np.empty((len(keys), k), np.uint8)
    for j in range(k):
        out[:, j] = lut[((keys >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.int64)]
    return [r.tobytes().decode() for r in out]


class OracleBackend:
    words = 1

    def __init__(self, reads, seg_read_off, fixed_len, qkeys):
        self.reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1, fixed_len)
        self.seg_read_off = np.asarray(seg_read_off, dtype=np.int64)
        self.n_segments = len(self.seg_read_off) - 1
        self.fixed_len = int(fixed_len)
        self.qkeys = qkeys
        self.runs = []      # current runs: list of (keys uint64 sorted, counts int64)

    def local_runs(self, k, bbits):
        assert k <= 31
        self.k, self.bbits = k, bbits
        nb = 1 << bbits
        self.runs = []
        for s in range(self.n_segments):
            key = keys_of_reads(self.reads[self.seg_read_off[s]:self.seg_read_off[s + 1]], k)
            u, c = np.unique(key, return_counts=True)
            b = (u >> np.uint64(2 * k - bbits)).astype(np.int64) if bbits else np.zeros(len(u), np.int64)
            for q in range(nb):
                m = b == q
                self.runs.append((u[m], c[m].astype(np.int64)))
        return np.array([len(r[0]) for r in self.runs], dtype=np.uint32)

    def pack_runs(self, bucket_ix, n_records):
        ks = [self.runs[int(i)][0] for i in bucket_ix]
        cs = [self.runs[int(i)][1] for i in bucket_ix]
        keys = np.concatenate(ks) if ks else np.zeros(0, np.uint64)
        cnt = np.concatenate(cs) if cs else np.zeros(0, np.int64)
        assert len(keys) == n_records
        return torch.from_numpy(keys.view(np.int64).copy()), torch.from_numpy(cnt.astype(np.int32))

    def merge_runs(self, n_out, n_src, run_off, run_len, keys, counts):
        keys = keys.numpy().view(np.uint64)
        counts = counts.numpy().astype(np.int64)
        run_off = np.asarray(run_off, np.int64).reshape(n_out, n_src)
        run_len = np.asarray(run_len, np.int64).reshape(n_out, n_src)
        out = []
        for j in range(n_out):
            kk = np.concatenate([keys[run_off[j, s]:run_off[j, s] + run_len[j, s]] for s in range(n_src)]) if n_src else np.zeros(0, np.uint64)
            cc = np.concatenate([counts[run_off[j, s]:run_off[j, s] + run_len[j, s]] for s in range(n_src)]) if n_src else np.zeros(0, np.int64)
            u, inv = np.unique(kk, return_inverse=True)
            c = np.zeros(len(u), np.int64)
            np.add.at(c, inv, cc)
            out.append((u, c))
        self.runs = out
        return np.array([len(r[0]) for r in out], dtype=np.uint32)

    def graph(self, n_local):
        nb = 1 << self.bbits
        assert len(self.runs) == n_local * nb
        self.n_local = n_local
        self.res = []
        for s in range(n_local):
            u = np.concatenate([self.runs[s * nb + q][0] for q in range(nb)]) if nb else np.zeros(0, np.uint64)
            c = np.concatenate([self.runs[s * nb + q][1] for q in range(nb)])
            assert (np.diff(u.astype(object)) > 0).all() if len(u) > 1 else True
            dk = unpack(u, self.k)
            contigs = orc.get_contigs(dk, self.k, 1, rows=1)["contigs"]     # contigs depend on the distinct set only
            self.res.append(dict(distinct=dk, counts=c, contigs=contigs))

    def reads_per_segment(self):
        return np.diff(self.seg_read_off).astype(np.int64)

    def _pack_piece(self, rd):
        """reads (n, rl) -> uint64 words, 32 bases per word, first base most significant, zero-padded"""
        c = _CODE[rd.reshape(-1)]
        n = len(c)
        nw = (n + 31) // 32
        pad = np.zeros(nw * 32, np.uint64)
        pad[:n] = c
        w = np.zeros(nw, np.uint64)
        pad = pad.reshape(nw, 32)
        for j in range(32):
            w = (w << np.uint64(2)) | pad[:, j]
        return w

    def pack_reads(self, seg_lo, seg_hi):
        ws = [self._pack_piece(self.reads[self.seg_read_off[s]:self.seg_read_off[s + 1]]) for s in range(seg_lo, seg_hi)]
        w = np.concatenate(ws) if ws else np.zeros(0, np.uint64)
        return torch.from_numpy(w.view(np.int64).copy())

    def set_reads(self, words, piece_seg, piece_reads, piece_word_off):
        w = words.numpy().view(np.uint64)
        lut = np.frombuffer(b"ACGT", np.uint8)
        self.own_reads = [[] for _ in range(self.n_local)]
        for s, n, o in zip(piece_seg, piece_reads, piece_word_off):
            n, o = int(n), int(o)
            nb_ = n * self.fixed_len
            ww = w[o:o + (nb_ + 31) // 32]
            bases = np.zeros((len(ww), 32), np.uint8)
            for j in range(32):
                bases[:, j] = lut[((ww >> np.uint64(62 - 2 * j)) & np.uint64(3)).astype(np.int64)]
            flat = bases.reshape(-1)[:nb_].reshape(n, self.fixed_len)
            self.own_reads[int(s)] += [r.tobytes().decode() for r in flat]

    def score(self, kmer, table):
        for s in range(self.n_local):
            o = orc.calc_breakscore(self.res[s]["contigs"], self.own_reads[s], "", kmer, self.qkeys, table, with_lev=False, with_freq=False)
            for kk in ("bp_score", "bp_score_norm_by_break_freqs", "bp_score_norm_by_len", "kmer_breaks", "sequence_len"):
                self.res[s][kk] = o[kk]

    def results(self):
        return self.res

    def close(self):
        pass
