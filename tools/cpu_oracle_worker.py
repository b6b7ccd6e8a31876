"""One oracle run in a process of its own (bench.py's all-cores CPU baseline): a process per host core, each building and scoring
one synthetic segment with oracle/gasm_oracle.cpp — never loads libgasm, never touches a GPU.
usage: cpu_oracle_worker.py <global segment id> <L> <read_len> <coverage> <k> <out.npz>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from genomeassembler_dev_amd import qtable, synth
from oracle import orc  # the checker / CPU baseline

g, L, rl, cov, k = (int(v) for v in sys.argv[1:6])
gen = synth.make_segment(1234 + g, L, planted=True)
reads = synth.simulate_reads(gen, rl, cov, 10_000_019 + 1234 + g)
rs = [r.tobytes().decode() for r in reads]
keys, table = qtable.keys(), qtable.load_normalised()
t0 = time.time()
o = orc.build_score(rs, k, 8, keys, table)
t1 = time.time()
np.savez(sys.argv[6], n_kmers=o["n_kmers"], seconds=o["seconds"], t0=t0, t1=t1, contigs=np.array("\n".join(o["contigs"])),
         kmer_breaks=o["kmer_breaks"], bp_score=o["bp_score"])
