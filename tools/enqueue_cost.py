"""Host time to QUEUE a step (build + score of cfg2) against the time the GPU needs for it: the host runs far ahead.
usage: python tools/enqueue_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import qtable, synth
reads, seg_off, _g = synth.make_batch(100, 50000, 150, 50, seed0=1234, planted=True)
ctx = ga.default_context()
b = ga.SegmentBatch.from_packed(synth.pack_2bit(reads), seg_off, fixed_len=150, ctx=ctx)
table = qtable.load_normalised()
for _ in range(6):
    b.build(31, genome_len_hint=50000).score(8, table); b.distinct()
ctx.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        b.build(31, genome_len_hint=50000).score(8, table)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/200:.4f} ms/step, total {1e3*(t2-t0)/200:.4f} ms/step")
