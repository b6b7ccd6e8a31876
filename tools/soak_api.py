"""Randomised checks of the API-side kernels added in round 3: k_levenshtein2 (both modes, both row / column orientations,
paths of 1..3 bands, targets around the 64-column groups) against the oracle's plain DP; k_path_ks2 against the general
k_path_ks bit for bit (GASM_KS_V=1) and against the oracle's KS; gasm_get_contigs_from_reads against gasm_get_contigs on the
exploded k-mers.  usage: python tools/soak_api.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable, synth  # noqa: E402
from oracle import orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
keys, prob = qtable.keys(), qtable.load_normalised()
uni = ga.qtable.uniform()
os.environ["GASM_LEV_GPU"] = "1"
t0, rounds, n_lev, n_ks, n_gc = time.time(), 0, 0, 0, 0


def mutate(s, n):
    s = list(s)
    for _ in range(n):
        i = int(rng.integers(0, max(1, len(s))))
        r = int(rng.integers(0, 3))
        if r == 0 and s:
            s[i] = "ACGT"[int(rng.integers(0, 4))]
        elif r == 1:
            s.insert(i, "ACGT"[int(rng.integers(0, 4))])
        elif s:
            del s[i]
    return "".join(s)


while time.time() - t0 < budget:
    # ---- Levenshtein
    nt = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 500, 2000, 4095, 4096, 4097, int(rng.integers(1, 7000))]))
    alphabet = str(rng.choice(["ACGT", "ACGT", "AC", "A"]))
    truth = "".join(alphabet[i] for i in rng.integers(0, len(alphabet), nt))
    paths = []
    for _ in range(int(rng.integers(2, 10))):
        kind = int(rng.integers(0, 4))
        if kind == 0:
            n = int(rng.choice([1, 63, 64, 65, 4095, 4096, 4097, 8192, 8193, int(rng.integers(1, 9000))]))
            paths.append("".join(alphabet[i] for i in rng.integers(0, len(alphabet), n)))
        elif kind == 1:
            a = int(rng.integers(0, nt))
            paths.append(mutate(truth[a:a + int(rng.integers(1, nt + 1))], int(rng.integers(0, 20))))
        elif kind == 2:
            paths.append(mutate(truth, int(rng.integers(0, 40))) + truth[:int(rng.integers(0, min(nt, 3000) + 1))])
        else:
            paths.append(truth[::-1][:int(rng.integers(1, nt + 1))])
    paths = [p for p in paths if p] or ["A"]
    reads = [truth[:min(nt, 30)]]
    for variant in ("own", "velvet"):
        m = ga.calc_breakscore(paths, reads, truth, 8, keys, prob, variant=variant, with_lev=True, with_freq=False)
        ref = [orc.levenshtein(p, truth, infix=(variant == "velvet")) for p in paths]
        assert m["lev_device"] == "gpu", "Levenshtein did not run on the GPU"
        assert m["lev_dist_vs_true"].tolist() == ref, ("lev", variant, nt, [len(p) for p in paths], alphabet)
        n_lev += len(paths)
    # ---- KS: k_path_ks2 == k_path_ks, and both the oracle's statistic
    if rounds % 3 == 0:
        L = int(rng.integers(300, 4000))
        g = synth.make_segment(int(rng.integers(1 << 30)), L, planted=bool(rng.integers(0, 2))).tobytes().decode()
        rl = int(rng.integers(12, 90))
        rs = [r.tobytes().decode() for r in synth.simulate_reads(np.frombuffer(g.encode(), dtype=np.uint8), rl, float(rng.uniform(5, 60)), int(rng.integers(1 << 30)))]
        pp = [g[a:a + int(rng.integers(8, L))] for a in rng.integers(0, L - 8, int(rng.integers(1, 12)))] + [g, "ACGTACGTACGTTTTT"]
        table = prob if rng.integers(0, 2) else uni
        got = {}
        for name, env in (("v2", {}), ("v1", {"GASM_KS_V": "1"}), ("bins", {"GASM_DBG_KS_BINS": str(int(rng.integers(2, 6)))})):
            os.environ.update(env)
            try:
                got[name] = ga.calc_breakscore(pp, rs, g, 8, keys, table, with_lev=False, with_freq=(name == "v2"), with_ks=True)
            finally:
                for kk in env:
                    del os.environ[kk]
        a, b, c = (np.asarray(got[n]["stat_test_KS"], dtype=np.float64).view(np.uint64) for n in ("v2", "v1", "bins"))
        assert (a == b).all() and (c == b).all(), ("ks kernels differ", L, rl)
        y = orc.kmer_from_seq(g, 8, keys, table)
        o = orc.calc_breakscore(pp, rs, g, 8, keys, table, with_lev=False, with_freq=True)
        for i in range(len(pp)):
            ref = orc.ks_statistic(o["path_freq"][i], y)
            v = got["v2"]["stat_test_KS"][i]
            assert (np.isnan(ref) and np.isnan(v)) or abs(v - ref) < 1e-9, ("ks vs oracle", i, v, ref)
        n_ks += len(pp)
    # ---- reads -> contigs in one call
    if rounds % 3 == 1:
        L = int(rng.integers(200, 5000))
        k = int(rng.choice([3, 9, 15, 21, 31, 32, 33, 51, 63]))
        g = synth.make_segment(int(rng.integers(1 << 30)), L, planted=bool(rng.integers(0, 2)))
        rl = int(rng.integers(max(2, k - 3), k + 100))
        rr = [r.tobytes().decode() for r in synth.simulate_reads(g, rl, float(rng.uniform(3, 30)), int(rng.integers(1 << 30)))]
        rr = [r[:len(r) - int(rng.integers(0, 6))] for r in rr] + ["", "AC"]
        seed = int(rng.integers(1, 1 << 20))
        a = ga.get_contigs_from_reads(rr, k, seed, matrix_rows=7)
        b = ga.get_contigs(ga.get_kmers_from_reads(rr, k), k, seed, matrix_rows=7)
        assert a.contigs == b.contigs and np.array_equal(a.perm, b.perm) and np.array_equal(a.distinct_keys, b.distinct_keys) and \
            np.array_equal(a.distinct_mult, b.distinct_mult), ("get_contigs_from_reads", L, k, rl)
        n_gc += 1
    rounds += 1
print(f"soak_api ok: {rounds} rounds, {n_lev} Levenshtein distances, {n_ks} KS statistics, {n_gc} reads->contigs calls in {time.time() - t0:.0f} s")
