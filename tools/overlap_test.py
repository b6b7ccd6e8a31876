"""Experiment: does splitting the batch into independent sub-batches on their own streams (one host thread each) let
the store-bound scatter of one overlap the load-bound de-duplication of another?"""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import qtable, synth

nseg, L, rl, cov, k = 100, 50000, 150, 50, 31
table = qtable.load_normalised()
reads, seg_off, _ = synth.make_batch(nseg, L, rl, cov, seed0=1234, planted=True)
n_kmers = int(seg_off[-1]) * (rl - k + 1)
for nsub in (1, 2, 4):
    ctxs = [ga.Context(0) for _ in range(nsub)]
    bounds = np.linspace(0, nseg, nsub + 1).astype(int)
    batches = []
    for i in range(nsub):
        a, b = int(seg_off[bounds[i]]), int(seg_off[bounds[i + 1]])
        so = (seg_off[bounds[i]:bounds[i + 1] + 1] - seg_off[bounds[i]]).astype(np.uint64)
        batches.append(ga.SegmentBatch(reads[a:b].reshape(-1), so, fixed_len=rl, ctx=ctxs[i]))
    def work(i, steps):
        for _ in range(steps):
            batches[i].build(k, genome_len_hint=L)
            batches[i].score(8, table)
        ctxs[i].sync()
    for steps in (2, 10):
        th = [threading.Thread(target=work, args=(i, steps)) for i in range(nsub)]
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
    print(f"sub-batches {nsub}: {dt / 10 * 1e3:.3f} ms per step of {nseg} segments, {n_kmers * 10 / dt:.3e} k-mers/s", flush=True)
    for b in batches: b.close()
    for c in ctxs: c.close()
