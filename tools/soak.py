"""Randomised parity soak (GPU box): random (segments, length, read length, coverage, k, hint, planted repeats, alphabet)
batches through build + score + pooled virtual ranks + guided traversal + device scaffolds, every segment against the
oracle.  usage: python tools/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (its HIP runtime first: genomeassembler_dev_amd/_lib.py)

import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import pooled, qtable, synth  # noqa: E402
from oracle import orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
keys, prob = qtable.keys(), qtable.load_normalised()
t0, rounds, segs = time.time(), 0, 0
while time.time() - t0 < budget:
    S = int(rng.integers(1, 14))
    L = int(rng.integers(300, 6000))
    k = int(rng.choice([3, 5, 9, 15, 21, 27, 31, 32, 33, 45, 51, 63]))
    rl = int(rng.integers(k, k + 120))
    cov = float(rng.uniform(3, 40))
    planted = bool(rng.integers(0, 2))
    alphabet = rng.choice(["ACGT", "ACGT", "ACGT", "AC", "ACG", "CCCA", "AAAAAAAC", "ACGTTTTTTT"])     # (the long ones: skewed composition)
    hint = int(rng.choice([0, L, max(1, L // 7)]))
    genomes, parts, off = [], [], [0]
    for s in range(S):
        g = synth.make_segment(int(rng.integers(1 << 30)), L, planted=planted)
        if len(alphabet) > 4:
            g = np.frombuffer(alphabet.encode(), dtype=np.uint8)[rng.integers(0, len(alphabet), L)]
        elif alphabet != "ACGT":
            lut = np.frombuffer(alphabet.encode(), dtype=np.uint8)
            g = lut[np.frombuffer(g.tobytes(), dtype=np.uint8) % len(lut)]
        r = synth.simulate_reads(g, rl, cov, int(rng.integers(1 << 30)))
        genomes.append(g)
        parts.append(r)
        off.append(off[-1] + r.shape[0])
    reads = np.concatenate(parts, axis=0)
    seg_off = np.array(off, dtype=np.uint64)
    tag = f"S={S} L={L} k={k} rl={rl} cov={cov:.1f} planted={planted} alphabet={alphabet} hint={hint}"
    print(f"[{time.time() - t0:6.1f} s] batch {rounds + 1}: {tag} ...", flush=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=hint).score(8, prob)
    contigs, sc = b.contigs(), b.scores()
    for s in range(S):
        rs = [x.tobytes().decode() for x in reads[off[s]:off[s + 1]]]
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        assert contigs[s] == ref["contigs"], (tag, s, "contigs")
        dk, dm = b.distinct_kmers(s)
        assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist(), (tag, s, "counts")
        if len(contigs[s]) * len(rs) <= 1_500_000:          # (the oracle's scorer is contigs x reads x find)
            o = orc.calc_breakscore(contigs[s], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
            a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
            assert sc["kmer_breaks"][a:e].tolist() == o["kmer_breaks"].tolist(), (tag, s, "breaks")
            assert np.abs(sc["bp_score"][a:e] - o["bp_score"]).max(initial=0.0) < 1e-9, (tag, s, "score")
        segs += 1
    # step slots: further steps on the same batch without reading anything in between — another k (everything drains), then
    # the first k three times in flight; the last step's results are the first step's, bit for bit
    if rounds % 2 == 0:
        k2 = int(rng.choice([kk for kk in (3, 9, 21, 31, 33, 51) if kk <= rl and kk != k] or [k]))
        b.build(k2, genome_len_hint=hint).score(8, prob)
        rs0 = [x.tobytes().decode() for x in reads[off[0]:off[1]]]
        assert b.contigs()[0] == orc.get_contigs(orc.kmers_from_reads(rs0, k2), k2, 1, rows=1)["contigs"], (tag, k2, "contigs at the second k")
        for _ in range(3):
            b.build(k, genome_len_hint=hint).score(8, prob)
        sc3 = b.scores()
        assert b.contigs() == contigs, (tag, "contigs after three more steps")
        for name in ("kmer_breaks", "bp_score", "sequence_len", "seg_contig_off"):
            assert np.array_equal(np.asarray(sc3[name]), np.asarray(sc[name])), (tag, name, "after three more steps")
    # pooled virtual ranks = single GPU
    if rounds % 3 == 0:
        world = int(rng.integers(1, 5))
        bbits = int(min(rng.integers(0, 6), 2 * (k - 1)))
        be = {}
        for r in range(world):
            pr = [p[r::world] for p in parts]
            o2 = np.concatenate([[0], np.cumsum([x.shape[0] for x in pr])]).astype(np.uint64)
            be[r] = pooled.GasmBackend(np.concatenate(pr, axis=0), o2, rl)
        try:
            own = pooled.pooled_build(pooled.VirtualComm(world), be, S, k, bbits, kmer=8, table=prob)
            for r in range(world):
                a0, b0 = own[r]
                for s, d in zip(range(a0, b0), be[r].results()):
                    assert d["contigs"] == contigs[s], (tag, "pooled", world, s)
                    ca, ce = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
                    assert d["kmer_breaks"].tolist() == sc["kmer_breaks"][ca:ce].tolist(), (tag, "pooled breaks", world, s)
        except ga.GasmError as e:
            if "GASM_ERR_CAPACITY" not in str(e):
                raise                                      # (too few bucket bits for this input: the documented answer)
        for r in range(world):
            be[r].close()
    # the same through the library's own exchange (gasm_pool_exchange_build, virtual ranks), two steps on the cached plans
    if rounds % 2 == 1:
        world = int(rng.integers(1, 7))
        bbits = int(min(rng.integers(0, 6), 2 * (k - 1)))
        bl = []
        for r in range(world):
            pr = [p[r::world] for p in parts]
            o2 = np.concatenate([[0], np.cumsum([x.shape[0] for x in pr])]).astype(np.uint64)
            bl.append(pooled.GasmBackend(np.concatenate(pr, axis=0), o2, rl))
        comm = pooled.Comm.virtual(ga.default_context(), world)
        try:
            for step in range(2):
                stats, own = pooled.exchange_build(comm, bl, k, bbits, kmer=8, table=prob)
                for r in range(world):
                    a0, b0 = own[r]
                    for s, d in zip(range(a0, b0), bl[r].results()):
                        assert d["contigs"] == contigs[s], (tag, "exchange", world, bbits, step, s)
                        dk, dm = b.distinct_kmers(s)
                        assert d["distinct"] == dk and d["counts"].tolist() == dm.tolist(), (tag, "exchange counts", world, bbits, step, s)
                        ca, ce = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
                        assert d["kmer_breaks"].tolist() == sc["kmer_breaks"][ca:ce].tolist(), (tag, "exchange breaks", world, bbits, step, s)
                        assert np.abs(d["bp_score"] - sc["bp_score"][ca:ce]).max(initial=0.0) < 1e-12, (tag, "exchange score", world, s)
        except ga.GasmError as e:
            if "GASM_ERR_CAPACITY" not in str(e):
                raise                                      # (a bucket no table holds even with all bucket bits: every rank says so together)
        for x in bl:
            x.close()
        comm.close()
    # the reads as files, parsed and packed on the device, against the host reader
    if rounds % 4 == 2:
        import tempfile
        from genomeassembler_dev_amd import seqio
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for s in range(min(S, 3)):
                rs = [x.tobytes().decode() for x in reads[off[s]:off[s + 1]]]
                p = os.path.join(td, f"s{s}.{'fastq' if s % 2 == 0 else 'fa'}")
                with open(p, "w") as f:
                    if s % 2 == 0:
                        f.write("".join(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(rs)))
                    else:
                        w = int(rng.integers(10, 200))
                        f.write("".join(f">r{i}\n" + "".join(r[j:j + w] + "\n" for j in range(0, len(r), w)) for i, r in enumerate(rs)))
                paths.append(p)
            w_h, o_h, s_h, d_h = seqio.read_files(paths)
            w_d, o_d, s_d, d_d, on = seqio.read_files_device(paths)
            assert all(on) and o_d.tolist() == o_h.tolist() and s_d.tolist() == s_h.tolist() and d_d == d_h and np.array_equal(w_d, w_h), (tag, "ingest")
    # device scaffolds of one segment
    if rounds % 4 == 1 and k >= 3:
        s = int(rng.integers(0, S))
        rs = [x.tobytes().decode() for x in reads[off[s]:off[s + 1]]]
        if rs:
            m = ga.get_contigs(ga.get_kmers_from_reads(rs, k), k, 7, matrix_rows=120)
            try:
                ref = orc.assemble_contigs(m.contigs, m.perm, k)
            except IndexError:
                ref = None
            if ref is not None and len(m.contigs) <= 2048:
                dv = ga.assemble_contigs(m, k, on_device=True)
                assert dv.strings() == ref, (tag, "scaffolds", s)
                dv.close()
    # guided traversal against its CPU restatement (small cases: the restatement is Python)
    if rounds % 5 == 2 and reads.shape[0] * sum(len(c) for cs in contigs for c in cs) < 4e7 and rl >= k:
        from oracle import guided_oracle
        table = dict(zip(keys, prob.tolist()))
        fx, shift = b.score_fixed()
        gd = b.guided()
        for s in range(S):
            rs = [x.tobytes().decode() for x in reads[off[s]:off[s + 1]]]
            a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
            ofx = guided_oracle.fixed_sums(contigs[s], rs, table, 8, shift)
            assert ofx == fx[a:e].tolist(), (tag, "guided sums", s)
            want = guided_oracle.guided_paths(contigs[s], ofx, k)
            if [d["sequence"] for d in gd[s]] != want:          # keep the case for a post-mortem
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez("gpurun_out/soak_guided_case.npz", genome=genomes[s], reads=reads[off[s]:off[s + 1]], k=k, fx=np.array(ofx, dtype=np.uint64),
                         shift=shift, contigs=np.array(contigs[s]), got=np.array([d["sequence"] for d in gd[s]]), want=np.array(want))
            assert [d["sequence"] for d in gd[s]] == want, (tag, "guided", s)
    b.close()
    # ragged reads (some shorter than k, some empty) through the general scorer; calc_breakscore with KS on random paths
    if rounds % 5 == 3:
        gs = genomes[0].tobytes().decode()
        rr = [gs[a:a + int(rng.integers(0, 3 * k + 10))] for a in rng.integers(0, max(1, L - 1), 400)]
        bb = ga.SegmentBatch.from_strings([rr, rr[:37]])
        bb.build(k).score(8, prob)
        cs2, sc2 = bb.contigs(), bb.scores()
        for s, rs in enumerate([rr, rr[:37]]):
            ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
            assert cs2[s] == ref["contigs"], (tag, "ragged contigs", s)
            o = orc.calc_breakscore(cs2[s], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
            a, e = int(sc2["seg_contig_off"][s]), int(sc2["seg_contig_off"][s + 1])
            assert sc2["kmer_breaks"][a:e].tolist() == o["kmer_breaks"].tolist(), (tag, "ragged breaks", s)
            assert np.abs(sc2["bp_score"][a:e] - o["bp_score"]).max(initial=0.0) < 1e-9
        bb.close()
        # (reads of >= 4 bases and paths of >= 8: no break window is cut short by the end of a path — there the reference
        # inserts keys into its table and its path_freq is no longer the vector the KS statistic is defined on)
        paths = [p for p in (gs[a:a + int(rng.integers(8, 400))] for a in rng.integers(0, max(1, L - 1), 40)) if len(p) >= 8] + [gs]
        rr = [r for r in rr if len(r) >= 4]
        m = ga.calc_breakscore(paths, rr, gs, 8, keys, prob, with_lev=True, with_freq=True, with_ks=True)
        o = orc.calc_breakscore(paths, rr, gs, 8, keys, prob, with_lev=True, with_freq=True)
        assert m["kmer_breaks"].tolist() == o["kmer_breaks"].tolist() and m["lev_dist_vs_true"].tolist() == o["lev_dist_vs_true"].tolist(), (tag, "api")
        y = orc.kmer_from_seq(gs, 8, keys, prob)
        for i in range(len(paths)):
            r = orc.ks_statistic(o["path_freq"][i], y)
            assert (np.isnan(r) and np.isnan(m["stat_test_KS"][i])) or abs(m["stat_test_KS"][i] - r) < 1e-9, (tag, "ks", i)
    # simulated reads against the oracle's sampler
    if rounds % 5 == 4 and alphabet == "ACGT":
        gl = [g.tobytes().decode() for g in genomes[:3]]
        sd = int(rng.integers(1 << 40))
        wt = prob if rng.integers(0, 2) else None
        sb = ga.SegmentBatch.simulate(gl, rl, cov, sd, kmer=8, table=wt)
        so, st = sb.read_starts()
        for s, g in enumerate(gl):
            ref = orc.simulate_starts(g, s, rl, cov, sd, 8, keys if wt is not None else None, wt)
            assert st[int(so[s]):int(so[s + 1])].tolist() == ref.tolist(), (tag, "sim", s)
        sb.close()
    rounds += 1
    print(f"[{time.time() - t0:6.1f} s] batch {rounds}: {tag} ok", flush=True)
print(f"soak ok: {rounds} batches, {segs} segments against the oracle in {time.time() - t0:.0f} s")
