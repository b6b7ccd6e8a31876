"""Randomised parity soak at bench-like shapes (GPU box): many segments of 8-50 kb, random read length / coverage / k / hint;
every batch built twice (estimates, then the sizes the first build reported) and scored; a few random segments per batch
against the oracle, all segments through size-independent properties.  usage: python tools/soak_large.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable, synth  # noqa: E402
from oracle import orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
keys, prob = qtable.keys(), qtable.load_normalised()
t0, rounds, checked = time.time(), 0, 0
while time.time() - t0 < budget:
    S = int(rng.integers(20, 130))
    L = int(rng.integers(8000, 50001))
    if os.environ.get("SOAK_MANY_SEGMENTS") and rounds % 2 == 0:      # more segments than CUs: the other side of several launch decisions
        S = int(rng.integers(260, 1300))
        L = int(rng.integers(600, 6000))
    k = int(rng.choice([15, 21, 31, 33, 51, 63]))
    rl = int(rng.integers(max(k + 10, 60), 260))
    cov = float(rng.uniform(8, 45))
    while S * L * cov / rl * (rl - k + 1) > 3.5e8:        # keep a batch near the bench's size
        S = max(8, S // 2)
    rl = min(rl, max(k + 2, L // 3))
    hint = int(rng.choice([0, L]))
    reads, seg_off, genomes = synth.make_batch(S, L, rl, cov, seed0=int(rng.integers(1 << 30)), planted=bool(rng.integers(0, 2)))
    tag = f"S={S} L={L} k={k} rl={rl} cov={cov:.1f} hint={hint} reads={reads.shape[0]}"
    print(f"[{time.time() - t0:6.1f} s] batch {rounds + 1}: {tag} ...", flush=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=hint).score(8, prob)
    c1 = b.contigs()
    b.build(k, genome_len_hint=hint).score(8, prob)          # second build: the partition and sizes of the first
    contigs, sc = b.contigs(), b.scores()
    assert contigs == c1, (tag, "second build")
    assert b.total_kmers() == reads.shape[0] * (rl - k + 1)
    seg, dkeys, mult, _w = b.distinct()
    for s in range(S):
        n_reads = int(seg_off[s + 1] - seg_off[s])
        a, e = int(seg[s]), int(seg[s + 1])
        assert int(mult[a:e].sum()) == n_reads * (rl - k + 1), (tag, s, "multiplicities")
        cs = contigs[s]
        assert cs == sorted(set(cs)), (tag, s, "contigs sorted + unique")
        assert sum(len(c) - (k - 1) for c in cs) == e - a, (tag, s, "every edge in exactly one contig")   # (no isolated cycles in these inputs)
        ca, ce = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        assert ce - ca == len(cs) and sc["sequence_len"][ca:ce].tolist() == [len(c) for c in cs], (tag, s, "score rows")
    for s in rng.choice(S, size=min(S, 3), replace=False):
        s = int(s)
        rs = [x.tobytes().decode() for x in reads[int(seg_off[s]):int(seg_off[s + 1])]]
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        assert contigs[s] == ref["contigs"], (tag, s, "contigs")
        dk, dm = b.distinct_kmers(s)
        assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist(), (tag, s, "counts")
        if len(contigs[s]) * len(rs) <= 3_000_000:
            o = orc.calc_breakscore(contigs[s], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
            ca, ce = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
            assert sc["kmer_breaks"][ca:ce].tolist() == o["kmer_breaks"].tolist(), (tag, s, "breaks")
            assert np.abs(sc["bp_score"][ca:ce] - o["bp_score"]).max(initial=0.0) < 1e-9, (tag, s, "score")
        checked += 1
    b.close()
    rounds += 1
    print(f"[{time.time() - t0:6.1f} s] batch {rounds}: ok", flush=True)
print(f"soak_large ok: {rounds} batches, {checked} segments against the oracle in {time.time() - t0:.0f} s")
