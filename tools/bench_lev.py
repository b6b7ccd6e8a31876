"""Levenshtein row: k_levenshtein against the threaded host routine, through gasm_calc_breakscore.
usage: python tools/bench_lev.py [n_paths] [path_len] [truth_len] [--host]
Paths are mutated windows of the truth (the shape of scaffolds scored against a true solution)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
P = int(args[0]) if len(args) > 0 else 256
PL = int(args[1]) if len(args) > 1 else 40000
TL = int(args[2]) if len(args) > 2 else 50000
if "--host" in sys.argv:
    os.environ["GASM_LEV_HOST"] = "1"
else:
    os.environ["GASM_LEV_GPU"] = "1"       # measure the kernel whatever the library's own GPU/host choice would be

import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable, synth  # noqa: E402

rng = np.random.default_rng(5)
truth = synth.make_segment(7, TL, planted=False)
paths = []
for p in range(P):
    a = int(rng.integers(0, TL - PL + 1))
    s = truth[a:a + PL].copy()
    idx = rng.integers(0, PL, PL // 200)
    s[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, idx.size)]
    paths.append(s.tobytes().decode())
truth_s = truth.tobytes().decode()
reads = [truth_s[i:i + 100] for i in range(0, TL - 100, 997)]
keys, prob = qtable.keys(), qtable.load_normalised()
for variant in ("own", "velvet"):
    ga.calc_breakscore(paths[:2], reads, truth_s, 8, keys, prob, variant=variant, with_lev=True, with_freq=False)   # warm up
    t0 = time.perf_counter()
    m = ga.calc_breakscore(paths, reads, truth_s, 8, keys, prob, variant=variant, with_lev=True, with_freq=False)
    t1 = time.perf_counter()
    m0 = ga.calc_breakscore(paths, reads, truth_s, 8, keys, prob, variant=variant, with_lev=False, with_freq=False)
    t2 = time.perf_counter()
    dt = (t1 - t0) - (t2 - t1)
    cells = float(P) * PL * TL
    print(f"{'host' if '--host' in sys.argv else 'gpu '} {variant:6s} P={P} |path|={PL} |truth|={TL}: lev part {dt * 1e3:9.1f} ms "
          f"= {cells / dt / 1e12:.3f} T cells/s   (sum of distances {int(m['lev_dist_vs_true'].sum())})")
