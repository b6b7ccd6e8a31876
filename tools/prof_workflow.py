"""per-kernel device time of the one-segment workflow's two heavy calls (assemble_contigs on the device; calc_breakscore with
Levenshtein + KS from the handle): python tools/prof_workflow.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable, synth  # noqa: E402

g = synth.make_segment(1234, 50000, planted=True)
reads = [r.tobytes().decode() for r in synth.simulate_reads(g, 150, 50, 1234 + 10000019)]
truth = g.tobytes().decode()
keys, prob = qtable.keys(), qtable.load_normalised()
k = 31
ctx = ga.default_context()
km = ga.get_kmers_from_reads(reads, k)
m = ga.get_contigs(km, k, 1234, matrix_rows=rows)
for rep in range(2):
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    dv = ga.assemble_contigs(m, k, on_device=True)
    t1 = time.perf_counter()
    p1 = ctx.profile_read()
    ctx.profile_reset()
    b = ga.calc_breakscore(dv, reads, truth, 8, keys, prob, with_lev=True, with_freq=False, with_ks=True)
    t2 = time.perf_counter()
    p2 = ctx.profile_read()
    ctx.profile(False)
    print(f"rep {rep}: assemble_contigs(on device) {1e3 * (t1 - t0):.1f} ms (merge by {dv.merge_device}, {dv.rows_on_host} rows on the host); "
          f"calc_breakscore(lev + KS) {1e3 * (t2 - t1):.1f} ms (lev by {b['lev_device']})")
    for name, p in (("assemble", p1), ("breakscore", p2)):
        tot = sum(v[0] for v in p.values())
        print(f"  {name}: kernels {tot:.2f} ms")
        for n, v in sorted(p.items(), key=lambda kv: -kv[1][0])[:12]:
            print(f"    {n:28s} {v[0]:9.3f} ms  x{v[1]}")
