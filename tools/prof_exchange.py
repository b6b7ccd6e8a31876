"""per-kernel device time of one pooled step through gasm_pool_exchange_build with W virtual ranks on one GPU (all ranks' kernels
add up: the sum is what W GPUs do in all, not what one of them takes): python tools/prof_exchange.py [world] [segments]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import pooled, qtable, synth

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 100
L, rl, cov, k, bbits = 50000, 150, 50, 31, 6
reads, seg_off, _ = synth.make_batch(nseg, L, rl, cov, seed0=1234, planted=True)
prob = qtable.load_normalised()
ctx = ga.default_context()
be = []
for r in range(world):
    parts, off = [], [0]
    for s in range(nseg):
        x = reads[int(seg_off[s]):int(seg_off[s + 1])][r::world]
        parts.append(x)
        off.append(off[-1] + x.shape[0])
    be.append(pooled.GasmBackend(np.concatenate(parts, axis=0), np.array(off, dtype=np.uint64), rl))
comm = pooled.Comm.virtual(ctx, world)
for _ in range(2):
    pooled.exchange_build(comm, be, k, bbits, kmer=8, table=prob)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    stats, own = pooled.exchange_build(comm, be, k, bbits, kmer=8, table=prob)
ctx.sync()
dt = (time.perf_counter() - t0) / 5
print(f"world {world}, {nseg} segments: {dt * 1e3:.3f} ms per step for all virtual ranks together = {dt / world * 1e3:.3f} ms per rank's share; "
      f"bytes sent by rank 0: {stats['bytes_sent']}, remote {stats['bytes_sent_remote']}")
ctx.profile(True)
ctx.profile_reset()
for _ in range(3):
    pooled.exchange_build(comm, be, k, bbits, kmer=8, table=prob)
ctx.sync()
p = ctx.profile_read()
ctx.profile(False)
tot = sum(v[0] for v in p.values()) / 3
print(f"kernels {tot:.3f} ms per step (all ranks)")
for n, v in sorted(p.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {n:24s} {v[0] / 3:8.3f} ms  x{v[1] // 3}")
