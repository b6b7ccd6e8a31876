"""cProfile of the pooled mode's host side on one GPU (world 1).  usage: python tools/prof_pooled.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import pooled, qtable, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S, L, rl, cov, k = 100, 50000, 150, 50, 31
reads, seg_off, _ = synth.make_batch(S, L, rl, cov, seed0=1, planted=True)
prob = qtable.load_normalised()
ctx = ga.default_context()
be = pooled.GasmBackend(reads.reshape(-1), seg_off, rl, ctx=ctx)
comm = pooled.VirtualComm(1)
for _ in range(2):
    pooled.pooled_build(comm, {0: be}, S, k, 6, kmer=8, table=prob)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    pooled.pooled_build(comm, {0: be}, S, k, 6, kmer=8, table=prob)
torch.cuda.synchronize()
pr.disable()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step (with the profiler on)")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
