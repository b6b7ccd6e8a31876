"""Seeded synthetic inputs: a 50 kb ACGT segment with planted repeat structure so the graph branches the
way a T2T segment does, and fixed-length error-free forward-strand reads.  The reference's own simulator
(lib/GenerateReads.R:235-313) needs R, Biostrings and the BSgenome T2T package, none of which exist here; what is kept
of it: n = ceil(coverage * L / read_len) start positions (:302), starts whose read would run past the end are dropped
(:310-313), optional start weights = normalised 8-mer probability of the window beginning at the start (:243-259)."""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_segment(seed, length=50000, n_short=20, short_len=300, n_long=5, long_len=2000, tandem_unit=6, tandem_len=1000,
                 planted=True):
    """uint8 ASCII array.  planted=True copies one 300-bp block to 20 places, one 2-kb block to 5 places and lays one
    1-kb tandem repeat of a 6-bp unit, at seeded random offsets (recipe printed by bench.py)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    g = _ACGT[rng.integers(0, 4, size=length)]
    if planted and length >= 4 * long_len:
        blk = _ACGT[rng.integers(0, 4, size=short_len)]
        for p in rng.integers(0, length - short_len, size=n_short):
            g[p:p + short_len] = blk
        blk = _ACGT[rng.integers(0, 4, size=long_len)]
        for p in rng.integers(0, length - long_len, size=n_long):
            g[p:p + long_len] = blk
        unit = _ACGT[rng.integers(0, 4, size=tandem_unit)]
        p = int(rng.integers(0, length - tandem_len))
        g[p:p + tandem_len] = np.resize(unit, tandem_len)
    return g


def simulate_reads(genome, read_len, coverage, seed, weights=None):
    """(n_reads, read_len) uint8 array of reads; starts uniform, or drawn with `weights` (one per start position)."""
    L = genome.size
    rng = np.random.Generator(np.random.MT19937(seed))
    n = int(np.ceil(coverage * L / read_len))
    if weights is None:
        starts = rng.integers(0, L, size=n)
    else:
        w = np.asarray(weights, dtype=np.float64)
        starts = rng.choice(L, size=n, p=w / w.sum())
    starts = starts[starts + read_len <= L]
    idx = starts[:, None] + np.arange(read_len)[None, :]
    return genome[idx]


def make_batch(n_segments, seg_len, read_len, coverage, seed0=1234, planted=True):
    """reads of n_segments segments as one (total_reads, read_len) uint8 array + seg_read_off + the genomes"""
    reads, off, genomes = [], [0], []
    for s in range(n_segments):
        g = make_segment(seed0 + s, seg_len, planted=planted)
        r = simulate_reads(g, read_len, coverage, 10_000_019 + seed0 + s)
        genomes.append(g)
        reads.append(r)
        off.append(off[-1] + r.shape[0])
    return np.concatenate(reads, axis=0), np.array(off, dtype=np.uint64), genomes


def pack_2bit(ascii_bases):
    """ASCII ACGT (uint8, any shape, read after read) -> uint64 words, 32 bases per word, first base most significant
    (A=0 C=1 G=2 T=3): the layout gasm_batch_create_packed takes — a quarter of the bytes over PCIe"""
    a = np.ascontiguousarray(ascii_bases, dtype=np.uint8).reshape(-1)
    code = ((a >> 1) & 3) ^ ((a >> 2) & 1)
    n = code.size
    pad = (-n) % 32
    if pad:
        code = np.concatenate([code, np.zeros(pad, dtype=np.uint8)])
    c = code.reshape(-1, 32).astype(np.uint64)
    shifts = (62 - 2 * np.arange(32, dtype=np.uint64)).astype(np.uint64)
    return np.bitwise_or.reduce(c << shifts[None, :], axis=1)
