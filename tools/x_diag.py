"""diagnostic: one small pooled step through gasm_pool_exchange_build with per-stage syncs (GASM_X_SYNC=1), stderr visible"""
import os
import sys

os.environ.setdefault("GASM_X_SYNC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import pooled, qtable, synth

world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
reads, seg_off, _ = synth.make_batch(4, 3000, 60, 20, seed0=1, planted=True)
prob = qtable.load_normalised()


def shard(rank):
    parts, off = [], [0]
    for s in range(len(seg_off) - 1):
        r = reads[int(seg_off[s]):int(seg_off[s + 1])][rank::world]
        parts.append(r)
        off.append(off[-1] + r.shape[0])
    return np.concatenate(parts, axis=0), np.array(off, dtype=np.uint64)


ctx = ga.default_context()
comm = pooled.Comm.virtual(ctx, world)
be = [pooled.GasmBackend(*shard(r), 60) for r in range(world)]
for step in range(2):
    print("step", step, file=sys.stderr, flush=True)
    stats, own = pooled.exchange_build(comm, be, k, 3, kmer=8, table=prob)
    print(stats, own, file=sys.stderr, flush=True)
    for r, b in enumerate(be):
        res = b.results()
        print("rank", r, [len(d["contigs"]) for d in res], file=sys.stderr, flush=True)
print("ok")
