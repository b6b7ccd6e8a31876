"""The reference's experiment for ONE segment, stage by stage, through the drop-in API (lib/DeNovoAssembler.R:109-346):
reads -> k-mers -> get_contigs (+ shuffle matrix) -> assemble_contigs -> calc_breakscore of every scaffold (with
Levenshtein).  usage: python tools/bench_workflow.py [matrix_rows] [segment_len] [coverage]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
cov = int(sys.argv[3]) if len(sys.argv) > 3 else 50

import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable, synth  # noqa: E402

g = synth.make_segment(1234, L, planted=True)
reads = [r.tobytes().decode() for r in synth.simulate_reads(g, 150, cov, 1234 + 10000019)]
truth = g.tobytes().decode()
keys, prob = qtable.keys(), qtable.load_normalised()
k = 31
ga.get_contigs(ga.get_kmers_from_reads(reads[:50], k), k, 1, matrix_rows=2)      # warm up (context, kernels)


def timed(name, f):
    t0 = time.perf_counter()
    r = f()
    print(f"{name:34s} {(time.perf_counter() - t0) * 1e3:10.1f} ms")
    return r


km = timed("get_kmers_from_reads (host)", lambda: ga.get_kmers_from_reads(reads, k))
m = timed(f"get_contigs ({rows} shuffles)", lambda: ga.get_contigs(km, k, 1234, matrix_rows=rows))
m2 = timed("get_contigs_from_reads (one call)", lambda: ga.get_contigs_from_reads(reads, k, 1234, matrix_rows=rows))
assert m2.contigs == m.contigs and (m2.perm == m.perm).all()
sc = timed("assemble_contigs (strings back)", lambda: ga.assemble_contigs(m, k, ctx=ga.default_context()))
print(f"  contigs {len(m.contigs)}, scaffolds {len(sc)}, scaffold bases {sum(map(len, sc))}")
a = timed("calc_breakscore(strings), no lev", lambda: ga.calc_breakscore(sc, reads, truth, 8, keys, prob, with_lev=False, with_freq=False))
timed("calc_breakscore(strings), with lev", lambda: ga.calc_breakscore(sc, reads, truth, 8, keys, prob, with_lev=True, with_freq=False))
# the same experiment with the scaffolds left on the GPU between the two calls (gasm_assemble_contigs_dev / _calc_breakscore_dev)
dv = timed("assemble_contigs (on device)", lambda: ga.assemble_contigs(m, k, on_device=True))
b = timed("calc_breakscore(handle), no lev", lambda: ga.calc_breakscore(dv, reads, truth, 8, keys, prob, with_lev=False, with_freq=False))
c = timed("calc_breakscore(handle), lev + KS", lambda: ga.calc_breakscore(dv, reads, truth, 8, keys, prob, with_lev=True, with_freq=False, with_ks=True))
print(f"  who did the work: greedy merge {dv.merge_device} ({dv.rows_on_host} of {rows} permutations on the host), Levenshtein {c['lev_device']}")
assert len(dv) == len(sc) and (a["kmer_breaks"] == b["kmer_breaks"]).all() and abs(a["bp_score"] - b["bp_score"]).max() == 0.0
os.environ["GASM_ASM_HOST"] = "1"
timed("assemble_contigs (host strings)", lambda: ga.assemble_contigs(m, k))
