"""Experiment: split the cfg2 batch into independent sub-batches, each on its own stream with its own persistent host
thread, and time K steps wall-clock (barrier before/after), like bench.py does."""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import qtable, synth

nseg, L, rl, cov, k = 100, 50000, 150, 50, 31
table = qtable.load_normalised()
reads, seg_off, _ = synth.make_batch(nseg, L, rl, cov, seed0=1234, planted=True)
n_kmers = int(seg_off[-1]) * (rl - k + 1)
K = 20
for nsub in (1, 2, 3, 4, 6):
    ctxs = [ga.Context(0) for _ in range(nsub)]
    bounds = np.linspace(0, nseg, nsub + 1).astype(int)
    batches = []
    for i in range(nsub):
        a, b = int(seg_off[bounds[i]]), int(seg_off[bounds[i + 1]])
        so = (seg_off[bounds[i]:bounds[i + 1] + 1] - seg_off[bounds[i]]).astype(np.uint64)
        batches.append(ga.SegmentBatch(reads[a:b].reshape(-1), so, fixed_len=rl, ctx=ctxs[i]))
    start = threading.Barrier(nsub + 1); done = threading.Barrier(nsub + 1)
    def work(i):
        for phase in range(2):          # warm-up pass, timed pass
            start.wait()
            for _ in range(3 if phase == 0 else K):
                batches[i].build(k, genome_len_hint=L)
                batches[i].score(8, table)
            ctxs[i].sync()
            done.wait()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nsub)]
    [t.start() for t in th]
    start.wait(); done.wait()          # warm-up
    start.wait(); t0 = time.perf_counter(); done.wait(); dt = time.perf_counter() - t0
    [t.join() for t in th]
    print(f"sub-batches {nsub}: {dt / K * 1e3:.3f} ms per step of {nseg} segments, {n_kmers * K / dt:.3e} k-mers/s", flush=True)
    for b in batches: b.close()
    for c in ctxs: c.close()
