"""Randomised check of the breakage-score-guided traversal (row A16) against its CPU restatement: many small one- to three-
segment batches (few reads, so the fixed-point shift is large and the sums have 50+ bits; two- to four-letter alphabets, so
most contigs tie at score 0 and branching is dense).  usage: python tools/soak_guided.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable, synth  # noqa: E402
from oracle import guided_oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
keys, prob = qtable.keys(), qtable.load_normalised()
table = dict(zip(keys, prob.tolist()))
t0, n_b, n_s, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    S = int(rng.integers(1, 4))
    L = int(rng.integers(400, 5000))
    k = int(rng.choice([9, 15, 21, 27, 31, 33, 41]))
    rl = int(rng.integers(k + 8, k + 110))
    cov = float(rng.uniform(4, 30))
    alphabet = str(rng.choice(["ACGT", "AC", "ACG", "ACGT", "CT"]))
    parts, off, segs = [], [0], []
    for s in range(S):
        g = synth.make_segment(int(rng.integers(1 << 30)), L, planted=bool(rng.integers(0, 2)))
        if alphabet != "ACGT":
            lut = np.frombuffer(alphabet.encode(), dtype=np.uint8)
            g = lut[np.frombuffer(g.tobytes(), dtype=np.uint8) % len(lut)]
        r = synth.simulate_reads(g, rl, cov, int(rng.integers(1 << 30)))
        parts.append(r)
        off.append(off[-1] + r.shape[0])
    reads = np.concatenate(parts, axis=0)
    b = ga.SegmentBatch(reads.reshape(-1), np.array(off, dtype=np.uint64), fixed_len=rl)
    b.build(k, genome_len_hint=L).score(8, prob)
    contigs, sc = b.contigs(), b.scores()
    if reads.shape[0] * sum(len(c) for cs in contigs for c in cs) > 6e7:
        b.close()
        continue
    fx, shift = b.score_fixed()
    gd = b.guided()
    for s in range(S):
        rs = [x.tobytes().decode() for x in reads[off[s]:off[s + 1]]]
        a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        ofx = guided_oracle.fixed_sums(contigs[s], rs, table, 8, shift)
        assert ofx == fx[a:e].tolist(), "sums"
        ok = [d["sequence"] for d in gd[s]] == guided_oracle.guided_paths(contigs[s], ofx, k)
        n_s += 1
        if not ok:
            bad += 1
            print(f"MISMATCH: S={S} L={L} k={k} rl={rl} cov={cov:.1f} alphabet={alphabet} shift={shift} contigs={len(contigs[s])} max fx bits={max(ofx).bit_length() if ofx else 0}", flush=True)
    b.close()
    n_b += 1
print(f"soak_guided: {n_b} batches, {n_s} segments, {bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
