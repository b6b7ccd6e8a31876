"""The real timeline of overlapping steps (step slots) from the library's own HIP events (GASM_PROF_TIMELINE): which kernels
of which slot run beside which, where a stream waits.  usage: python tools/slot_timeline.py [steps] [first_step_shown] [steps_shown]"""
import os
import sys
import tempfile
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30
shown = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tl = os.path.join(tempfile.gettempdir(), "gasm_timeline.txt")
os.environ["GASM_PROF_TIMELINE"] = tl

import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable, synth  # noqa: E402

reads, seg_off, _g = synth.make_batch(100, 50000, 150, 50, seed0=1234, planted=True)
ctx = ga.default_context()
b = ga.SegmentBatch.from_packed(synth.pack_2bit(reads), seg_off, fixed_len=150, ctx=ctx)
table = qtable.load_normalised()
for _ in range(6):
    b.build(31, genome_len_hint=50000).score(8, table)
    b.distinct()
ctx.sync()
ctx.profile(True)
ctx.profile_reset()
for _ in range(steps):
    b.build(31, genome_len_hint=50000).score(8, table)
ctx.sync()
ctx.profile_read()
ctx.profile(False)
rows = [l.split() for l in open(tl)]
rows = [(r[0], r[1], float(r[2]), float(r[3])) for r in rows if len(r) == 4]
streams = {s: i for i, s in enumerate(sorted({r[0] for r in rows}))}
rows.sort(key=lambda r: r[2])
# a step = from a k_bucket_partition to the k_score_finish of the same stream
starts = [r for r in rows if r[1] == "k_bucket_partition"]
print(f"{len(rows)} launches on {len(streams)} streams; {len(starts)} steps; wall {rows[-1][3] - rows[0][2]:.3f} ms = {(rows[-1][3] - rows[0][2]) / max(1, len(starts)):.4f} ms/step")
t0 = starts[first][2] if first < len(starts) else starts[0][2]
t1 = starts[min(first + shown, len(starts) - 1)][2]
print(f"--- launches that start in [{t0:.3f}, {t1:.3f}) ms: stream, kernel, start, end, duration (us)")
for s, name, a, e in rows:
    if t0 <= a < t1:
        print(f"  s{streams[s]}  {name:22s} {a - t0:8.3f} {e - t0:8.3f} {1e3 * (e - a):8.1f}")
# busy analysis over the steady part: time covered by >= 1 streaming kernel, by 2, by none
big = ("k_bucket_partition", "k_bucket_dedup")
ev = []
lo, hi = starts[min(5, len(starts) - 1)][2], starts[-3][2]
for s, name, a, e in rows:
    if name in big and e > lo and a < hi:
        ev.append((max(a, lo), 1))
        ev.append((min(e, hi), -1))
ev.sort()
cover = defaultdict(float)
n, last = 0, lo
for t, d in ev:
    cover[n] += t - last
    last = t
    n += d
cover[n] += hi - last
tot = hi - lo
print("--- share of the steady wall with 0 / 1 / 2 / 3 streaming kernels (partition, de-duplication) resident: " +
      " / ".join(f"{100 * cover[i] / tot:.1f} %" for i in range(4)))
per = defaultdict(list)
for s, name, a, e in rows:
    if lo <= a < hi:
        per[name].append(e - a)
print("--- mean residence (us): " + ", ".join(f"{k} {1e3 * sum(v) / len(v):.1f}" for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))))
