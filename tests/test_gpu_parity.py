"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact: distinct k-mers + multiplicities, edge lists, contigs, shuffle matrix, scaffolds, kmer_breaks, Levenshtein.
FP64 scores: within 1e-9 absolute (north-star tolerance; the reference sums in hash-iteration order)."""
import json
import os

import numpy as np
import pytest

import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import qtable, synth
from oracle import orc

pytestmark = pytest.mark.gpu
TOL = 1e-9
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_survey_toy.json")))


def _strs(a):
    return [r.tobytes().decode() for r in a]


def _check_scores(mine, ref, with_lev=True):
    assert mine["sequence_len"].tolist() == ref["sequence_len"].tolist()
    assert mine["kmer_breaks"].tolist() == ref["kmer_breaks"].tolist()
    for key in ("bp_score", "bp_score_norm_by_break_freqs", "bp_score_norm_by_len"):
        a, b = np.asarray(mine[key]), np.asarray(ref[key])
        assert np.array_equal(np.isnan(a), np.isnan(b)), key
        ok = ~np.isnan(a)
        assert np.abs(a[ok] - b[ok]).max(initial=0.0) < TOL, key
    if with_lev:
        assert mine["lev_dist_vs_true"].tolist() == ref["lev_dist_vs_true"].tolist()


# ------------------------------------------------------------------------------------------------ golden vector
def test_survey_known_answer_vector(qtable):
    keys, prob = qtable
    g = G["genome"]
    reads = [g[s:s + G["read_len"]] for s in G["read_starts"]]
    km = ga.get_kmers_from_reads(reads, G["dbg_kmer"])
    m = ga.get_contigs(km, G["dbg_kmer"], G["seed"])
    assert m.contigs == G["contigs"]
    assert m.row(0) == G["shuffle_row0"]
    sc = ga.assemble_contigs(m, G["dbg_kmer"])
    assert [len(s) for s in sc] == G["scaffold_lens"] and sc[0] == g
    b = ga.calc_breakscore(sc, reads, g, G["break_kmer"], keys, prob)
    assert b["kmer_breaks"].tolist() == G["kmer_breaks"]
    assert b["lev_dist_vs_true"].tolist() == G["lev_dist_vs_true"]
    assert abs(b["bp_score"][0] - G["bp_score_0"]) < TOL
    assert abs(b["bp_score_norm_by_break_freqs"][0] - G["bp_score_norm_by_break_freqs_0"]) < TOL
    assert abs(b["bp_score_norm_by_len"][0] - G["bp_score_norm_by_len_0"]) < TOL
    assert list(b["path_freq"].shape) == G["path_freq_shape"]
    assert int((b["path_freq"][0] > 0).sum()) == G["path_freq_nonzeros_0"]


# ------------------------------------------------------------------------------------------------ get_contigs
# the reference's own operating points (scripts/02_Real_vs_rand_prob_own.R:21-31): L=1000, coverage 40
@pytest.mark.parametrize("read_len,k", [(12, 9), (14, 9), (16, 13), (18, 15), (20, 15), (25, 15), (40, 15)])
def test_get_contigs_reference_operating_points(read_len, k):
    g = synth.make_segment(1000 + read_len, 1000, planted=False)
    reads = _strs(synth.simulate_reads(g, read_len, 40, 5 + k))
    km = ga.get_kmers_from_reads(reads, k)
    ref = orc.get_contigs(km, k, 1234)
    m = ga.get_contigs(km, k, 1234)
    assert m.contigs == ref["contigs"]
    assert np.array_equal(m.perm, ref["perm"])
    assert m.distinct_kmers() == ref["distinct"]
    assert m.distinct_mult.tolist() == ref["counts"].tolist()
    # distinct k-mers are the distinct edges: (prefix, suffix) lists must agree too
    assert [d[:-1] for d in m.distinct_kmers()] == ref["edge_prefix"]
    assert [d[1:] for d in m.distinct_kmers()] == ref["edge_suffix"]
    # the same from the reads themselves (gasm_get_contigs_from_reads: get_kmers_from_reads + get_contigs in one call)
    m2 = ga.get_contigs_from_reads(reads, k, 1234)
    assert m2.contigs == m.contigs and np.array_equal(m2.perm, m.perm)
    assert np.array_equal(m2.distinct_keys, m.distinct_keys) and np.array_equal(m2.distinct_mult, m.distinct_mult)
    assert ga.assemble_contigs(m, k, ctx=ga.default_context()) == orc.assemble_contigs(ref["contigs"], ref["perm"], k)     # merge on the GPU
    assert ga.assemble_contigs(m, k) == orc.assemble_contigs(ref["contigs"], ref["perm"], k)                                 # host merge (no context)


@pytest.mark.parametrize("k", [2, 3, 5, 11, 21, 31, 32, 33, 41, 51, 63])
def test_get_contigs_k_range_with_repeats(k):
    g = synth.make_segment(7 + k, 6000, n_short=6, short_len=120, n_long=2, long_len=500, tandem_len=200, planted=True)
    reads = _strs(synth.simulate_reads(g, max(k + 9, 40), 15, 3))
    km = ga.get_kmers_from_reads(reads, k)
    ref = orc.get_contigs(km, k, 7, rows=20)
    m = ga.get_contigs(km, k, 7, matrix_rows=20)
    assert m.contigs == ref["contigs"]
    assert np.array_equal(m.perm, ref["perm"])
    assert m.distinct_kmers() == ref["distinct"] and m.distinct_mult.tolist() == ref["counts"].tolist()
    # ragged reads, some shorter than k (no k-mers), through the reads entry
    rr = [r[:len(r) - (i % 7)] for i, r in enumerate(reads)] + ["ACG"[:min(3, k - 1)], ""]
    kr = ga.get_kmers_from_reads(rr, k)
    a, b = ga.get_contigs_from_reads(rr, k, 7, matrix_rows=20), ga.get_contigs(kr, k, 7, matrix_rows=20)
    assert a.contigs == b.contigs and np.array_equal(a.perm, b.perm) and np.array_equal(a.distinct_mult, b.distinct_mult)


def test_get_contigs_edge_cases():
    # isolated cycle: every node has in = out = 1 -> no contig
    cyc = "ACGTTGCA"
    km = [(cyc + cyc)[i:i + 4] for i in range(len(cyc))]
    assert ga.get_contigs(km, 4, 1, matrix_rows=3).contigs == orc.get_contigs(km, 4, 1, rows=3)["contigs"] == []
    # pure linear chain: one contig, source -> sink
    lin = "ACGGTCATTGCAAGTC"
    km = [lin[i:i + 5] for i in range(len(lin) - 4)]
    assert ga.get_contigs(km, 5, 1, matrix_rows=3).contigs == orc.get_contigs(km, 5, 1, rows=3)["contigs"] == [lin]
    # a single k-mer, duplicated
    assert ga.get_contigs(["ACGTA"] * 7, 5, 1, matrix_rows=2).contigs == ["ACGTA"]
    # homopolymer self-loop
    km = ["AAAA", "AAAA", "AAAC"]
    assert ga.get_contigs(km, 4, 1, matrix_rows=2).contigs == orc.get_contigs(km, 4, 1, rows=2)["contigs"]
    # nothing
    m = ga.get_contigs([], 5, 1, matrix_rows=4)
    assert m.contigs == [] and m.perm.shape == (4, 0)
    m = ga.get_contigs_from_reads([], 5, 1, matrix_rows=4)
    assert m.contigs == [] and m.perm.shape == (4, 0)
    assert ga.get_contigs_from_reads(["ACG", "AC"], 5, 1, matrix_rows=2).contigs == []          # only reads shorter than k
    assert ga.get_contigs_from_reads([lin, lin[3:11]], 5, 1, matrix_rows=3).contigs == [lin]
    with pytest.raises(ga.GasmError):
        ga.get_contigs_from_reads(["ACGTNACGT"], 5, 1)


@pytest.mark.parametrize("k,rl", [(15, 50), (41, 90)])
def test_graph_degrees_and_branching_nodes_against_oracle(k, rl):
    """Rows A4-A6 directly (lib/DeNovoAssembler.cpp:125-189), not only through the contigs they lead to: out-degrees (runs of
    edges sharing a source node), in-degree classes (the claim-word trick: bit 1 of the node's first out-edge <=> two or more
    in-edges), the list of branching nodes with out-edges, and one step of the walk (the successor edge, or none where the
    reference's walk stops) against the oracle's degree table and branching list.  64- and 128-bit keys, planted repeats."""
    NONE = 0xFFFFFFFF
    n_seg = 3
    parts, off = [], [0]
    for s in range(n_seg):      # repeats short enough to fit a 6 kb segment: branching nodes, and chains between them
        g = synth.make_segment(4400 + k + s, 6000, n_short=6, short_len=120, n_long=2, long_len=500, tandem_len=200, planted=True)
        parts.append(synth.simulate_reads(g, rl, 18, 77 + s))
        off.append(off[-1] + parts[-1].shape[0])
    reads, seg_off = np.concatenate(parts, axis=0), np.array(off, dtype=np.uint64)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=5000)
    seg = b.distinct()[0]
    flags_all, next_all = b.graph()
    for s in range(n_seg):
        rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        dk = b.distinct_kmers(s)[0]
        a, e = int(seg[s]), int(seg[s + 1])
        fl, nx = flags_all[a:e], next_all[a:e]
        assert [d[:-1] for d in dk] == ref["edge_prefix"] and [d[1:] for d in dk] == ref["edge_suffix"]
        deg = {n: (int(i), int(o)) for n, i, o in zip(ref["node"], ref["node_in"], ref["node_out"])}
        branch = set(ref["branch"])
        assert len(branch) > 2                                                # (the planted repeats branch)
        first, out = {}, {}
        for i, d in enumerate(dk):
            first.setdefault(d[:-1], i)
            out[d[:-1]] = out.get(d[:-1], 0) + 1
        assert sorted({d[:-1] for d, f in zip(dk, fl) if f & 1}) == ref["branch"]
        for i, d in enumerate(dk):
            u, v = d[:-1], d[1:]
            assert bool(fl[i] & 1) == (u in branch)                          # every out-edge of a branching node carries the flag
            assert out[u] == deg[u][1]
            if first[u] == i:
                assert bool(fl[i] & 2) == (deg[u][0] >= 2), (s, u, deg[u])
            if nx[i] == NONE:
                assert v not in first or v in branch                         # the walk stops: no successor, or a branching node
            else:
                j = int(nx[i]) - a
                assert 0 <= j < len(dk) and dk[j][:-1] == v and v not in branch and out[v] == 1 and deg[v][0] == 1
    b.close()


def test_non_acgt_is_rejected():
    with pytest.raises(ga.GasmError) as e:
        ga.get_contigs(["ACGTN"], 5, 1)
    assert "GASM_ERR_NON_ACGT" in str(e.value)
    with pytest.raises(ga.GasmError):
        ga.get_contigs(["acgta"], 5, 1)


# ------------------------------------------------------------------------------------------------ batches
@pytest.mark.parametrize("n_seg,L,rl,cov,k", [(4, 5000, 100, 20, 21), (3, 8000, 150, 25, 31), (5, 1500, 40, 30, 15),
                                              (3, 6000, 250, 30, 51), (2, 4000, 120, 20, 32)])
def test_batch_build_and_score_against_oracle(qtable, n_seg, L, rl, cov, k):
    keys, prob = qtable
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=900 + k, planted=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k).score(8, prob)
    assert b.total_kmers() == reads.shape[0] * (rl - k + 1)
    contigs, sc = b.contigs(), b.scores()
    for s in range(n_seg):
        rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        assert contigs[s] == ref["contigs"]
        dk, dm = b.distinct_kmers(s)
        assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist()
        o = orc.calc_breakscore(contigs[s], rs, genomes[s].tobytes().decode(), 8, keys, prob, with_lev=False, with_freq=False)
        a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        mine = {kk: v[a:e] for kk, v in sc.items() if kk != "seg_contig_off"}
        _check_scores(mine, o, with_lev=False)
    b.close()


def test_overflow_retry_and_skewed_bins():
    """Paths the tuned defaults never take: (a) a far too small genome_len_hint -> one bucket -> table overflow ->
    4096-slot table -> more bucket bits (host retry loop); (b) a two-letter genome -> all keys of a bucket fall into a
    few counting-sort bins -> bitonic fallback of the de-duplication kernel; (c) 128-bit keys through the same."""
    rng = np.random.default_rng(77)
    for k, rl, alphabet in [(21, 60, b"ACGT"), (31, 80, b"AC"), (41, 100, b"AC"), (15, 40, b"CT")]:
        g = np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), 12000)]
        reads = synth.simulate_reads(g, rl, 12, 5)
        seg_off = np.array([0, reads.shape[0]], dtype=np.uint64)
        rs = _strs(reads)
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        for hint in (50, 0, 12000):
            b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
            b.build(k, genome_len_hint=hint)
            assert b.contigs(0) == ref["contigs"], (k, alphabet, hint)
            dk, dm = b.distinct_kmers(0)
            assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist(), (k, alphabet, hint)
            b.close()


def test_step_slots_reproduce_one_step_in_flight(qtable):
    """Consecutive steps of a one-block batch take step slots in turn (capi.hip): whatever the number of slots and the way they
    overlap (GASM_PINGPONG 0 / 1 / 2, GASM_STEP_SLOTS 2..4), every step's results are those of a batch that runs one step
    at a time — with steps queued without a fetch in between, a change of k (all slots drain), a change of the score
    table between steps, a step that needs the retry ladder (far too small hint) in the middle, and fetches that must
    come from the slot of the LAST build."""
    keys, prob = qtable
    uni = ga.qtable.uniform()
    reads, seg_off, _g = synth.make_batch(7, 5000, 90, 18, seed0=4242, planted=True)

    def run(env):
        os.environ.update(env)
        try:
            b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=90)
            out = []
            for k, hint, table, fetch in ((21, 5000, prob, False), (21, 5000, prob, False), (21, 5000, prob, True), (31, 5000, prob, True),
                                          (21, 5000, uni, False), (21, 40, prob, False), (21, 5000, prob, True), (33, 0, uni, True),
                                          (21, 5000, prob, False), (21, 5000, prob, False), (21, 5000, prob, False), (21, 5000, prob, True)):
                b.build(k, genome_len_hint=hint).score(8, table)
                if fetch:
                    sc = b.scores()
                    seg, kk, mm, _w = b.distinct()
                    out.append((b.contigs(), sc["kmer_breaks"].tolist(), sc["bp_score"].tobytes(), np.asarray(seg).tolist(),
                                np.asarray(kk).tobytes(), np.asarray(mm).tobytes()))
            b.close()
            return out
        finally:
            for name in env:
                del os.environ[name]

    ref = run({"GASM_PINGPONG": "0", "GASM_SCORE_LANE": "0"})
    assert ref[0] == ref[2] == ref[4] and ref[0] != ref[1]          # same k and table again: the same bits; another k: not
    for env in ({"GASM_PINGPONG": "0"}, {"GASM_PINGPONG": "1", "GASM_STEP_SLOTS": "2"}, {}, {"GASM_PINGPONG": "1", "GASM_STEP_SLOTS": "4"},
                {"GASM_PINGPONG": "2", "GASM_STEP_SLOTS": "2"}, {"GASM_PINGPONG": "2", "GASM_STEP_SLOTS": "3"}):
        assert run(env) == ref, env
    # and against the oracle once (segment 3 at k = 21)
    rs = _strs(reads[int(seg_off[3]):int(seg_off[4])])
    assert ref[0][0][3] == orc.get_contigs(orc.kmers_from_reads(rs, 21), 21, 1, rows=1)["contigs"]


def test_buckets_no_table_can_hold():
    """Skewed base composition (3 C : 1 A): with all ten bucket bits used, the bucket of the prefix CCCCC still holds more
    distinct k-mers than a table takes (1408 of 128-bit keys, 2816 of 64-bit keys) -> the last rung of the retry ladder,
    k_bucket_dedup_multi (passes over key sub-ranges).  Contigs, distinct k-mers and multiplicities against the oracle."""
    rng = np.random.default_rng(2024)
    for k, rl, L, cov in [(45, 120, 12000, 10), (31, 100, 30000, 8), (63, 150, 9000, 8)]:
        g = np.frombuffer(b"CCCA", dtype=np.uint8)[rng.integers(0, 4, L)]
        reads = synth.simulate_reads(g, rl, cov, 11)
        seg_off = np.array([0, reads.shape[0]], dtype=np.uint64)
        rs = _strs(reads)
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        top = max(np.unique([x[:5] for x in ref["distinct"]], return_counts=True)[1])
        assert top > (2816 if k <= 31 else 1408), (k, top)        # the case really needs the fallback
        b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
        b.build(k, genome_len_hint=L)
        assert b.contigs(0) == ref["contigs"], k
        dk, dm = b.distinct_kmers(0)
        assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist(), k
        b.build(k, genome_len_hint=L)                               # again: the partition that worked is kept
        assert b.contigs(0) == ref["contigs"], k
        b.close()


def test_high_multiplicity_and_homopolymers(qtable):
    """tandem repeats and homopolymer runs: one k-mer thousands of times in a bucket (same-address LDS atomics), nodes with
    self-loops, reads that are all the same string"""
    keys, prob = qtable
    unit = "ACGGTC"
    g = "TTGACCA" * 30 + unit * 400 + "A" * 300 + "GATTACA" * 50 + "C" * 200 + "ACGT" * 100
    rl, k = 70, 25
    reads = [g[i:i + rl] for i in range(0, len(g) - rl + 1, 3)] + [g[500:500 + rl]] * 500
    b = ga.SegmentBatch.from_strings([reads])
    b.build(k).score(8, prob)
    ref = orc.get_contigs(orc.kmers_from_reads(reads, k), k, 1, rows=1)
    assert b.contigs(0) == ref["contigs"]
    dk, dm = b.distinct_kmers(0)
    assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist()
    o = orc.calc_breakscore(ref["contigs"], reads, g, 8, keys, prob, with_lev=False, with_freq=False)
    sc = b.scores()
    _check_scores({kk: v for kk, v in sc.items() if kk != "seg_contig_off"}, o, with_lev=False)
    b.close()


def test_batch_ragged_reads_and_empty_segment():
    rng = np.random.default_rng(3)
    g = _strs(synth.make_segment(5, 3000, planted=False)[None, :])[0]
    seg0 = [g[a:a + int(rng.integers(10, 90))] for a in rng.integers(0, 2900, 400)]
    segs = [seg0, [], ["ACGTACGTTGCA", "ACG"], seg0[:50]]
    b = ga.SegmentBatch.from_strings(segs)
    b.build(11)
    contigs = b.contigs()
    for s, rs in enumerate(segs):
        ref = orc.get_contigs(orc.kmers_from_reads(rs, 11), 11, 1, rows=1)
        assert contigs[s] == ref["contigs"]
        dk, dm = b.distinct_kmers(s)
        assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist()
    b.close()


def test_large_segment_long_reads_and_cycles(qtable):
    """(a) a segment with more than 65534 distinct k-mers: list ranking takes the whole-GPU doubling path instead of the
    LDS one, and the bucket count goes past 64; (b) reads of more than 8192 k-mers: several offset rounds (tiles) per
    read; (c) a circular sequence: an isolated cycle of edges, which the reference's walk never enters (no contig)."""
    keys, prob = qtable
    # (a)
    g = synth.make_segment(91, 90000, planted=False)
    reads = synth.simulate_reads(g, 120, 12, 92)
    rs = _strs(reads)
    k = 27
    ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
    b = ga.SegmentBatch(reads.reshape(-1), np.array([0, reads.shape[0]], dtype=np.uint64), fixed_len=120)
    b.build(k, genome_len_hint=90000).score(8, prob)
    assert len(ref["distinct"]) > 65534
    assert b.contigs(0) == ref["contigs"]
    dk, dm = b.distinct_kmers(0)
    assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist()
    o = orc.calc_breakscore(ref["contigs"], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
    _check_scores({kk: v for kk, v in b.scores().items() if kk != "seg_contig_off"}, o, with_lev=False)
    b.close()
    # (a') a segment that needs all ten bucket bits (1024 buckets: the tile kernels' many-buckets-per-thread paths)
    g = synth.make_segment(95, 700000, planted=False)
    reads = synth.simulate_reads(g, 100, 6, 96)
    rs = _strs(reads)
    k = 25
    ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
    b = ga.SegmentBatch(reads.reshape(-1), np.array([0, reads.shape[0]], dtype=np.uint64), fixed_len=100)
    b.build(k, genome_len_hint=700000)
    assert len(ref["distinct"]) > 512 * 900
    assert b.contigs(0) == ref["contigs"]
    dk, dm = b.distinct_kmers(0)
    assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist()
    b.close()
    # (b) + (c): two very long reads, short reads, and a 400-base circle read round and round
    g2 = _strs(synth.make_segment(93, 30000, planted=False)[None, :])[0]
    circ = _strs(synth.make_segment(94, 400, planted=False)[None, :])[0]
    long_reads = [g2[100:100 + 9000], g2[4000:4000 + 17001], (circ * 3)[:1000]]
    short = [g2[i:i + 80] for i in range(0, 29900, 37)]
    for k in (21, 33):
        segs = [long_reads + short, [(circ * 4)[i:i + 150] for i in range(0, 800, 7)]]
        b = ga.SegmentBatch.from_strings(segs)
        b.build(k)
        for s, rs2 in enumerate(segs):
            ref = orc.get_contigs(orc.kmers_from_reads(rs2, k), k, 1, rows=1)
            assert b.contigs(s) == ref["contigs"], (k, s)
            dk, dm = b.distinct_kmers(s)
            assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist(), (k, s)
        b.close()


def test_calc_breakscore_scaffold_like_paths(qtable):
    """many paths that repeat the same stretches of one genome (what assemble_contigs hands to calc_breakscore), duplicate
    reads, a read that occurs twice inside a path (first occurrence counts), paths shorter than a read, an empty path"""
    keys, prob = qtable
    rng = np.random.default_rng(21)
    g = _strs(synth.make_segment(301, 1500, planted=False)[None, :])[0]
    g = g[:700] + g[100:400] + g[700:]                      # a 300-base repeat: reads inside it occur twice in long paths
    paths = [g[a:a + int(rng.integers(30, 1200))] for a in rng.integers(0, 900, 120)] + [g, g[:20], "", g[50:1700], g]
    reads = [g[a:a + 36] for a in rng.integers(0, len(g) - 36, 900)]
    reads += reads[:100] + [g[150:150 + 36]] * 7
    for variant in ("own", "velvet"):
        m = ga.calc_breakscore(paths, reads, g, 8, keys, prob, variant=variant, with_lev=False, with_freq=(variant == "own"))
        o = orc.calc_breakscore(paths, reads, g, 8, keys, prob, velvet=(variant == "velvet"), with_lev=False,
                                with_freq=(variant == "own"))
        _check_scores(m, o, with_lev=False)
    # the same through the sliced form of the first-occurrence table (a budget of 20 paths' worth of entries at a time)
    os.environ["GASM_DBG_FIRST_BUDGET"] = str(20 * len(reads))
    try:
        m2 = ga.calc_breakscore(paths, reads, g, 8, keys, prob, variant="own", with_lev=False, with_freq=False)
    finally:
        del os.environ["GASM_DBG_FIRST_BUDGET"]
    o = orc.calc_breakscore(paths, reads, g, 8, keys, prob, with_lev=False, with_freq=False)
    _check_scores(m2, o, with_lev=False)


def test_levenshtein_kernel_against_oracle(qtable):
    """k_levenshtein (one wave per path, Myers blocks passed lane to lane): global distance for the own-assembler variant,
    infix for the velvet one; path lengths around the 64-row block and the 4096-row band boundaries, mutated copies of
    pieces of the truth, a path that is the truth, one longer than the truth, and an empty path."""
    keys, prob = qtable
    rng = np.random.default_rng(11)
    truth = _strs(synth.make_segment(201, 6001, planted=False)[None, :])[0]

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

    paths = [truth, mutate(truth, 40), truth[100:101], truth[:63], truth[5:69], mutate(truth[900:965], 3), truth[1000:5095],
             mutate(truth[1000:5096], 25), mutate(truth[200:4297], 60), mutate(truth, 300) + truth[:700], "", "ACGT" * 30]
    reads = [truth[i:i + 40] for i in range(0, 5900, 50)]
    os.environ["GASM_LEV_GPU"] = "1"          # (a dozen paths would go to the host routine otherwise: see gasm_calc_breakscore)
    try:
        for version in ("2", "1"):            # k_levenshtein2 (the default) and the first form of the kernel (GASM_LEV_V=1)
            os.environ["GASM_LEV_V"] = version
            for variant in ("own", "velvet"):
                m = ga.calc_breakscore(paths, reads, truth, 8, keys, prob, variant=variant, with_lev=True, with_freq=False)
                ref = [orc.levenshtein(p, truth, infix=(variant == "velvet")) for p in paths]
                assert m["lev_dist_vs_true"].tolist() == ref, (variant, version)
    finally:
        del os.environ["GASM_LEV_GPU"]
        os.environ.pop("GASM_LEV_V", None)
    # a target with a byte outside ACGT goes through the host routine
    t2 = truth[:500] + "N" + truth[500:900]
    m = ga.calc_breakscore(paths[2:6], reads, t2, 8, keys, prob, variant="own", with_lev=True, with_freq=False)
    assert m["lev_dist_vs_true"].tolist() == [orc.levenshtein(p, t2) for p in paths[2:6]]


def test_levenshtein_band_and_group_boundaries(qtable):
    """k_levenshtein2's seams: paths of 1 .. 3 bands (4096 rows each) against targets whose length sits around the 64-column
    groups in which a band hands its deltas to the next (1, 63, 64, 65, 127, 128, 129, 191, 4100 columns), both modes;
    random sequences and mutated copies, so distances are both near the maximum and small."""
    keys, prob = qtable
    rng = np.random.default_rng(77)
    base = _strs(synth.make_segment(909, 9300, planted=False)[None, :])[0]
    rnd = "".join("ACGT"[i] for i in rng.integers(0, 4, 9000))
    os.environ["GASM_LEV_GPU"] = "1"
    try:
        for nt in (1, 63, 64, 65, 127, 128, 129, 191, 4100):
            truth = base[37:37 + nt]
            paths = [base[37:37 + n] for n in (1, 64, 65, 4096, 4097, 8192, 8193, 9000)] + [rnd[:n] for n in (4097, 8200)] + [base[100:100 + 4500][::-1]]
            reads = [base[:60]]
            for variant in ("own", "velvet"):
                m = ga.calc_breakscore(paths, reads, truth, 8, keys, prob, variant=variant, with_lev=True, with_freq=False)
                ref = [orc.levenshtein(p, truth, infix=(variant == "velvet")) for p in paths]
                assert m["lev_dist_vs_true"].tolist() == ref, (nt, variant)
                assert m["lev_device"] == "gpu"
    finally:
        del os.environ["GASM_LEV_GPU"]


def test_batch_from_fastq_files(tmp_path, qtable):
    """FASTQ in (one file per segment, one of them gzipped FASTA, a read with an N dropped) -> the same contigs and
    scores as the same reads handed over as strings"""
    import gzip
    keys, prob = qtable
    segs, paths = [], []
    for s in range(2):
        g = synth.make_segment(500 + s, 3000, planted=True)
        rs = _strs(synth.simulate_reads(g, 60, 25, 600 + s))
        segs.append(rs)
        if s == 0:
            p = tmp_path / "s0.fastq"
            p.write_text("".join(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(rs)) + "@bad\nACGTNACGT\n+\nIIIIIIIII\n")
        else:
            p = tmp_path / "s1.fa.gz"
            with gzip.open(p, "wb") as f:
                f.write("".join(f">r{i}\n{r[:30]}\n{r[30:]}\n" for i, r in enumerate(rs)).encode())
        paths.append(p)
    from genomeassembler_dev_amd import seqio
    from oracle import seq_oracle
    a = ga.SegmentBatch.from_fastq(paths)
    assert a.dropped_reads == 1 and a.n_reads == sum(len(x) for x in segs)
    a.build(21).score(8, prob)
    # the checker: the oracle's own reading of the same files, then the oracle's graph and scores
    o_reads, o_off, o_seg, o_dropped = seq_oracle.segments_from_files(paths)
    assert o_dropped == 1
    cs, sc = a.contigs(), a.scores()
    for s in range(2):
        rs = [o_reads[int(o_off[r]):int(o_off[r + 1])].tobytes().decode() for r in range(int(o_seg[s]), int(o_seg[s + 1]))]
        assert rs == segs[s]
        ref = orc.get_contigs(orc.kmers_from_reads(rs, 21), 21, 1, rows=1)
        assert cs[s] == ref["contigs"]
        dk, dm = a.distinct_kmers(s)
        assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist()
        o = orc.calc_breakscore(cs[s], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
        ca, ce = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        _check_scores({kk: v[ca:ce] for kk, v in sc.items() if kk != "seg_contig_off"}, o, with_lev=False)
    a.close()
    # packed reads handed over directly (ragged and fixed length)
    words, off, seg, _ = seqio.read_files(paths)
    b = ga.SegmentBatch.from_packed(words, seg, read_off=off)
    b.build(21)
    assert b.contigs() == cs
    b.close()
    b = ga.SegmentBatch.from_packed(words, seg, fixed_len=60)
    b.build(21)
    assert b.contigs() == cs
    b.close()
    with pytest.raises(ga.GasmError):
        ga.SegmentBatch.from_fastq(paths, non_acgt="error")


def test_batch_score_without_the_graph_shortcut(qtable):
    """ragged reads, some shorter than k and some empty: the batch scorer cannot use "a read starts with a k-mer of the
    graph" and takes the general path (reads indexed, paths scanned, first occurrence per (path, read)) with several
    segments at once"""
    keys, prob = qtable
    rng = np.random.default_rng(8)
    k = 15
    segs, genomes = [], []
    for s in range(4):
        g = _strs(synth.make_segment(400 + s, 2500, planted=(s % 2 == 0))[None, :])[0]
        rs = [g[a:a + int(rng.integers(6, 70))] for a in rng.integers(0, 2430, 700)]
        rs += ["", rs[0], rs[1]] if s != 2 else []
        segs.append(rs)
        genomes.append(g)
    b = ga.SegmentBatch.from_strings(segs)
    b.build(k).score(8, prob)
    contigs, sc = b.contigs(), b.scores()
    for s, rs in enumerate(segs):
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        assert contigs[s] == ref["contigs"]
        o = orc.calc_breakscore(contigs[s], rs, genomes[s], 8, keys, prob, with_lev=False, with_freq=False)
        a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        _check_scores({kk: v[a:e] for kk, v in sc.items() if kk != "seg_contig_off"}, o, with_lev=False)
    b.close()


# ------------------------------------------------------------------------------------------------ calc_breakscore
def _score_case(seed, L=1200, rl=20, cov=40, k=15):
    g = synth.make_segment(seed, L, n_short=3, short_len=60, n_long=1, long_len=150, tandem_len=60, planted=True)
    reads = _strs(synth.simulate_reads(g, rl, cov, seed + 1))
    km = ga.get_kmers_from_reads(reads, k)
    m = ga.get_contigs(km, k, 1234, matrix_rows=200)
    return g.tobytes().decode(), reads, ga.assemble_contigs(m, k, ctx=ga.default_context())


def test_calc_breakscore_own_variant(qtable):
    keys, prob = qtable
    for seed in (21, 22):
        truth, reads, paths = _score_case(seed)
        mine = ga.calc_breakscore(paths, reads, truth, 8, keys, prob)
        ref = orc.calc_breakscore(paths, reads, truth, 8, keys, prob)
        _check_scores(mine, ref)
        for i in range(len(paths)):
            a, b = mine["path_freq"][i], ref["path_freq_by_input"][i]
            assert np.array_equal(np.isnan(a), np.isnan(b))
            assert np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max() == 0.0
            # the reference's own order is hash order: as a multiset it must agree as well
            assert np.array_equal(np.sort(np.nan_to_num(a)), np.sort(np.nan_to_num(ref["path_freq"][i])))


def test_calc_breakscore_velvet_variant(qtable):
    keys, prob = qtable
    truth, reads, paths = _score_case(31)
    paths = paths + [truth[100:400], truth[:50]]
    mine = ga.calc_breakscore(paths, reads, truth, 8, keys, prob, variant="velvet")
    ref = orc.calc_breakscore(paths, reads, truth, 8, keys, prob, velvet=True)
    _check_scores(mine, ref)
    assert mine["path_prob_dist_startpos"].tolist() == ref["path_prob_dist_startpos"].tolist()
    for a, b in zip(mine["path_prob_dist"], ref["path_prob_dist"]):
        assert np.array_equal(a, b)


def test_calc_breakscore_window_edges_and_duplicates(qtable):
    """reads matching at path positions 0..5 (window widths 8,2,4,6,8,8), duplicate reads, a read occurring twice in
    a path (first occurrence only), reads matching nothing, random ('uniform') table."""
    keys, _ = qtable
    prob = qtable_uniform = ga.qtable.uniform()
    path = "ACGTTGCATGCAAGTCCGATAGGCTTACGATCGGATCCGTA"
    rep = "GGATTACAGGATTACATTTT"
    paths = [path, rep + "CC" + rep, path[3:30]]
    reads = [path[i:i + 9] for i in range(6)] + [path[2:11]] * 3 + ["GATTACA", "TTTTTTTTT", path[-9:]]
    mine = ga.calc_breakscore(paths, reads, path, 8, keys, prob)
    ref = orc.calc_breakscore(paths, reads, path, 8, keys, prob)
    _check_scores(mine, ref)
    assert mine["kmer_breaks"][0] == 6 + 3 + 1
    del qtable_uniform
    # nothing matches anywhere: total 0 -> NaN frequencies like the reference's 0/0
    mine = ga.calc_breakscore(["ACGTACGTACGT"], ["GGGGGGGGG"], "ACGT", 8, keys, prob)
    ref = orc.calc_breakscore(["ACGTACGTACGT"], ["GGGGGGGGG"], "ACGT", 8, keys, prob)
    _check_scores(mine, ref)
    assert np.isnan(mine["path_freq"]).all()


# ------------------------------------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("k,rl,cov", [(31, 100, 50), (51, 250, 100)])
def test_full_size_segment_properties(qtable, k, rl, cov):
    """BASELINE configs[1] (50 kb, 100 bp, 50x, k=31) and configs[4] (250 bp, 100x, k=51: 128-bit keys) shapes without the
    oracle: size-independent properties."""
    _, prob = qtable
    reads, seg_off, genomes = synth.make_batch(2, 50000, rl, cov, seed0=4242, planted=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=50000).score(8, prob)
    seg, keys_, mult, w = b.distinct()
    sc = b.scores()
    for s in range(2):
        a, e = int(seg[s]), int(seg[s + 1])
        ks = keys_[a * w:e * w].reshape(-1, w)
        big = [int(r[0]) if w == 1 else (int(r[0]) << 64) | int(r[1]) for r in ks]
        assert all(x < y for x, y in zip(big, big[1:]))                     # sorted, distinct
        assert int(mult[a:e].sum()) == (int(seg_off[s + 1]) - int(seg_off[s])) * (rl - k + 1)   # every k-mer counted once
        cs = b.contigs(s)
        assert cs == sorted(set(cs))                                       # canonical order
        gs = genomes[s].tobytes().decode()
        # every contig is a walk in the graph of the reads: all its k-mers are distinct k-mers of the segment
        dk = set(b.distinct_kmers(s)[0])
        tot = 0
        for c in cs:
            assert len(c) >= k
            for i in range(len(c) - k + 1):
                assert c[i:i + k] in dk
            tot += len(c) - k + 1
        assert tot <= len(dk)                                              # edge-disjoint chains
        # error-free reads: each read lies inside the genome, so at full coverage the contigs tile it
        assert sum(1 for c in cs if c in gs) >= 1
        ca, ce = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        assert (sc["kmer_breaks"][ca:ce] >= 0).all() and sc["kmer_breaks"][ca:ce].sum() > 0
        assert np.allclose(sc["bp_score_norm_by_len"][ca:ce], sc["bp_score"][ca:ce] / sc["sequence_len"][ca:ce], rtol=0, atol=1e-18)
    # idempotence: a second build+score of the same batch gives bit-identical scores (fixed summation order)
    b.build(k, genome_len_hint=50000).score(8, prob)
    sc2 = b.scores()
    for key in ("bp_score", "bp_score_norm_by_break_freqs", "kmer_breaks"):
        assert np.array_equal(sc[key], sc2[key])
    b.close()


# ------------------------------------------------------------------------------------------------ headline shapes
def _check_segments_vs_oracle(b, reads, seg_off, genomes, segments, k, keys, prob, contigs=None, sc=None):
    """full oracle comparison (contigs, distinct k-mers + multiplicities, kmer_breaks exact, scores <= 1e-9) of the listed
    segments of a built + scored batch"""
    contigs = b.contigs() if contigs is None else contigs
    sc = b.scores() if sc is None else sc
    for s in segments:
        rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        assert contigs[s] == ref["contigs"], f"segment {s}: contigs"
        dk, dm = b.distinct_kmers(s)
        assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist(), f"segment {s}: k-mer counts"
        o = orc.calc_breakscore(contigs[s], rs, genomes[s].tobytes().decode(), 8, keys, prob, with_lev=False, with_freq=False)
        a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        _check_scores({kk: v[a:e] for kk, v in sc.items() if kk != "seg_contig_off"}, o, with_lev=False)


def _check_batch_properties(b, seg_off, rl, k, n_seg):
    """size-independent properties of every segment of a built + scored batch (numeric, no string decoding)"""
    seg, keys_, mult, w = b.distinct()
    sc = b.scores()
    contigs = b.contigs()
    for s in range(n_seg):
        a, e = int(seg[s]), int(seg[s + 1])
        ks = keys_[a * w:e * w].reshape(-1, w)
        if w == 1:
            assert (ks[1:, 0] > ks[:-1, 0]).all(), f"segment {s}: distinct k-mers not strictly sorted"
        else:
            gt = (ks[1:, 0] > ks[:-1, 0]) | ((ks[1:, 0] == ks[:-1, 0]) & (ks[1:, 1] > ks[:-1, 1]))
            assert gt.all(), f"segment {s}: distinct k-mers not strictly sorted"
        assert int(mult[a:e].astype(np.int64).sum()) == (int(seg_off[s + 1]) - int(seg_off[s])) * (rl - k + 1), f"segment {s}"
        cs = contigs[s]
        assert cs == sorted(set(cs)) and all(len(c) >= k for c in cs), f"segment {s}: contig order"
        assert sum(len(c) - k + 1 for c in cs) <= e - a, f"segment {s}: chains are not edge-disjoint"
        ca, ce = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        assert ce - ca == len(cs)
        assert sc["sequence_len"][ca:ce].tolist() == [len(c) for c in cs]
        assert (sc["kmer_breaks"][ca:ce] >= 0).all()
        assert int(sc["kmer_breaks"][ca:ce].sum()) <= int(seg_off[s + 1]) - int(seg_off[s])     # a read occurs in at most one contig
    return contigs, sc


def test_bench_prints_one_json_line_with_the_contract_keys():
    """bench.py on the smallest workload: file descriptor 1 carries exactly ONE line, a JSON object with the keys of the
    contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline /
    dtype / data / config.workload) plus `roofline` (bound, achieved, peak, unit, frac, traffic; the kernel's own duration and the
    timed region's residence times) — whatever libraries print on the way (RCCL and Gloo banners) goes to stderr."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.split("\n") if x.strip()]
    assert len(lines) == 1, p.stdout[:500]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["higher_is_better"] is True and j["data"] == "synthetic"
    assert "workload" in j["config"] and j["value"] > 0 and abs(j["value"] - j["config"]["kmers_per_step"] / (j["ms_per_step"] * 1e-3)) < 1e-3 * j["value"]
    r = j["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "measured", "timed_region", "step"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_more_segments_than_cus(qtable):
    """A batch with more segments than the chip has CUs (configs[3] on fewer GPUs than eight looks like this): the other side
    of several launch decisions — the LDS list of the ranking sized by the estimate, rulers at every second edge, more than
    one round of the segment-major grids and of the 64-segment directory scan.  Properties on all 600 segments, the oracle on a
    sample; two steps in flight on the way."""
    keys, prob = qtable
    n_seg, L, rl, cov, k = 600, 900, 50, 22, 21
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=6100, planted=False)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=L).score(8, prob)
    b.build(k, genome_len_hint=L).score(8, prob)
    contigs, sc = _check_batch_properties(b, seg_off, rl, k, n_seg)
    _check_segments_vs_oracle(b, reads, seg_off, genomes, [0, 1, 63, 64, 255, 256, 257, 511, 512, 598, 599], k, keys, prob, contigs, sc)
    b.close()


@pytest.mark.parametrize("n_seg,L,rl,cov,k", [(20, 3000, 60, 25, 21), (80, 1500, 40, 30, 15), (70, 1200, 90, 20, 33)])
def test_many_segments_against_oracle(qtable, n_seg, L, rl, cov, k):
    """more than 8 and more than 64 segments, all against the oracle: the segment-major grids (seg_chunk), the 64-segment
    rounds of k_seg_offsets, one workgroup per segment in k_tile_scan / k_rank_lds / k_contig_scan, and — from
    n_cu / 4 = 64 segments on — rulers at every second edge in the LDS list ranking (the headline bench's configuration)"""
    keys, prob = qtable
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=7000 + k, planted=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=L).score(8, prob)
    contigs, sc = _check_batch_properties(b, seg_off, rl, k, n_seg)
    _check_segments_vs_oracle(b, reads, seg_off, genomes, range(n_seg), k, keys, prob, contigs, sc)
    b.close()


@pytest.mark.parametrize("mode", ["two_pass", "region_overflow", "tight_regions"])
@pytest.mark.parametrize("n_seg,L,rl,cov,k", [(12, 6000, 70, 30, 25), (9, 5000, 120, 25, 45)])
def test_partition_paths_against_oracle(qtable, monkeypatch, mode, n_seg, L, rl, cov, k):
    """The partition of the k-mers into (segment, bucket) ranges has two forms: one pass into regions of fixed capacity
    (k_bucket_partition, the default every other test runs) and count + scan + scatter (k_tile_hist, k_tile_scan,
    k_bucket_scatter: the exact layout).  Here the second one from the start (GASM_SINGLE_PASS=0), regions far too small
    for their buckets (every run overflows: the build must come back through the two-pass kernels), and regions without
    room to spare where only some buckets overflow — all against the oracle, 64- and 128-bit keys, built twice."""
    keys, prob = qtable
    if mode == "two_pass":
        monkeypatch.setenv("GASM_SINGLE_PASS", "0")
    elif mode == "region_overflow":
        monkeypatch.setenv("GASM_DBG_PART_CAP", "16")
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=9100 + k, planted=True)
    if mode == "tight_regions":      # a segment's k-mers per bucket at the five bucket bits these shapes get: about half the regions too small
        monkeypatch.setenv("GASM_DBG_PART_CAP", str(int(seg_off[1] - seg_off[0]) * (rl - k + 1) // 32))
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    for _ in range(2):
        b.build(k, genome_len_hint=L).score(8, prob)
        contigs, sc = _check_batch_properties(b, seg_off, rl, k, n_seg)
        _check_segments_vs_oracle(b, reads, seg_off, genomes, range(n_seg), k, keys, prob, contigs, sc)
    b.close()


def test_headline_shape_configs2(qtable):
    """BASELINE configs[2] as bench.py runs it: 100 x 50 kb segments, 150 bp reads at 50x, k=31, scoring on all contigs.
    Properties on all 100 segments; the full oracle comparison on segments either side of the 8-segment (XCD) and
    64-segment (scan round) boundaries and at both ends."""
    keys, prob = qtable
    n_seg, L, rl, cov, k = 100, 50000, 150, 50, 31
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=1234, planted=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=L).score(8, prob)
    assert b.total_kmers() == reads.shape[0] * (rl - k + 1)
    contigs, sc = _check_batch_properties(b, seg_off, rl, k, n_seg)
    _check_segments_vs_oracle(b, reads, seg_off, genomes, [0, 7, 8, 63, 64, 99], k, keys, prob, contigs, sc)
    b.close()


def test_headline_shape_configs4_per_gpu(qtable):
    """BASELINE configs[4]'s shape per GPU with more than 8 segments: 50 kb, 250 bp reads at 100x, k=51 (128-bit keys).
    Properties on all segments, the oracle on the first, the ninth and the last."""
    keys, prob = qtable
    n_seg, L, rl, cov, k = 10, 50000, 250, 100, 51
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=5150, planted=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=L).score(8, prob)
    contigs, sc = _check_batch_properties(b, seg_off, rl, k, n_seg)
    _check_segments_vs_oracle(b, reads, seg_off, genomes, [0, 8, 9], k, keys, prob, contigs, sc)
    b.close()


def test_alternating_batch_shapes_on_one_context(qtable):
    """batches of 5, 1, 3, 12 and again 1 segment(s) built one after the other on the same Context: the reports the
    kernels write into the context's pinned area change layout with the number of segments, stale words must never be
    taken for a fresh report"""
    keys, prob = qtable
    ctx = ga.Context(0)
    for rep, (n_seg, k) in enumerate([(5, 15), (1, 15), (3, 21), (12, 15), (1, 33), (5, 15)]):
        rl = 50
        reads, seg_off, genomes = synth.make_batch(n_seg, 1200 + 100 * rep, rl, 20, seed0=300 + 17 * rep, planted=True)
        b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl, ctx=ctx)
        for _ in range(2):
            b.build(k).score(8, prob)
            _check_segments_vs_oracle(b, reads, seg_off, genomes, range(n_seg), k, keys, prob)
        b.close()
    ctx.close()


def test_sub_batches_on_lanes(qtable, monkeypatch):
    """GASM_SUBBATCHES=3: a batch runs as three blocks of segments on three streams (lanes of the context), the blocks'
    builds chained by stream events; fetched results are the concatenation, identical to the oracle's"""
    keys, prob = qtable
    monkeypatch.setenv("GASM_SUBBATCHES", "3")
    n_seg, L, rl, cov, k = 11, 2500, 70, 20, 25
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=8100, planted=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    for _ in range(2):
        b.build(k, genome_len_hint=L).score(8, prob)
    assert b.total_kmers() == reads.shape[0] * (rl - k + 1)
    contigs, sc = _check_batch_properties(b, seg_off, rl, k, n_seg)
    _check_segments_vs_oracle(b, reads, seg_off, genomes, range(n_seg), k, keys, prob, contigs, sc)
    b.close()
    # ragged reads (general scorer) through the same split
    rng = np.random.default_rng(5)
    segs = []
    for s in range(5):
        g = _strs(synth.make_segment(8200 + s, 1500, planted=False)[None, :])[0]
        segs.append([g[a:a + int(rng.integers(8, 60))] for a in rng.integers(0, 1440, 300)])
    b = ga.SegmentBatch.from_strings(segs)
    b.build(13).score(8, prob)
    cs, sc = b.contigs(), b.scores()
    for s, rs in enumerate(segs):
        ref = orc.get_contigs(orc.kmers_from_reads(rs, 13), 13, 1, rows=1)
        assert cs[s] == ref["contigs"]
        o = orc.calc_breakscore(cs[s], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
        a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        _check_scores({kk: v[a:e] for kk, v in sc.items() if kk != "seg_contig_off"}, o, with_lev=False)
    b.close()


def test_read_simulator_against_oracle(qtable):
    """gasm_batch_simulate (reads made on the device: weights -> scan -> draws -> kept starts -> packed reads) against the
    oracle's restatement of lib/GenerateReads.R:235-313 with the same pinned random stream: the kept starts bit-exact, per
    segment, weighted (the reference's ultrasonication model) and uniform; then the batch built from those reads equals
    the batch built from the same reads handed over as text.  Statistical sanity: weighted starts follow the weights."""
    keys, prob = qtable
    genomes = [synth.make_segment(9000 + s, L, planted=True) for s, L in enumerate((6000, 2500, 9, 4000, 7))]
    gs = [g.tobytes().decode() for g in genomes]
    for table, rl, cov, seed in ((prob, 40, 12.0, 5), (None, 75, 7.5, 2**40 + 3), (ga.qtable.uniform(), 33, 3.0, 0)):
        b = ga.SegmentBatch.simulate(gs, rl, cov, seed, kmer=8, table=table)
        seg, starts = b.read_starts()
        all_reads = []
        for s, g in enumerate(gs):
            ref = orc.simulate_starts(g, s, rl, cov, seed, 8, keys if table is not None else None, table)
            assert starts[int(seg[s]):int(seg[s + 1])].tolist() == ref.tolist(), (s, rl)
            assert all(int(p) + rl <= len(g) for p in ref)
            all_reads.append([g[int(p):int(p) + rl] for p in ref])
        assert len(all_reads[2]) == 0 and len(all_reads[4]) == 0          # genomes shorter than a read
        k = 17
        b.build(k)
        t = ga.SegmentBatch.from_strings(all_reads)
        t.build(k)
        assert b.contigs() == t.contigs()
        for s in (0, 1, 3):
            ref = orc.get_contigs(orc.kmers_from_reads(all_reads[s], k), k, 1, rows=1)
            assert b.contigs(s) == ref["contigs"]
            dk, dm = b.distinct_kmers(s)
            assert dk == ref["distinct"] and dm.tolist() == ref["counts"].tolist()
        b.close()
        t.close()
    # kmer = 2 on a 100 kb genome: with a shift of 52 the segment's running sum would pass 2^64 (p ~ 1/16 each, 10^5 of them)
    big = synth.make_segment(9100, 100000, planted=True).tobytes().decode()
    # (a table whose 2-mer rows sum to one: the caller's table need not be normalised over all 69 904 rows as the standard one is)
    t2 = np.asarray(prob, dtype=np.float64) / float(np.sum(prob[:16]))
    sh = orc.sim_weight_shift([len(big), len(gs[1])], 2, keys, t2)
    assert sh < 52
    b = ga.SegmentBatch.simulate([big, gs[1]], 50, 2.0, 9, kmer=2, table=t2)
    seg, starts = b.read_starts()
    for s, g in enumerate((big, gs[1])):
        ref = orc.simulate_starts(g, s, 50, 2.0, 9, 2, keys, t2, weight_shift=sh)
        assert starts[int(seg[s]):int(seg[s + 1])].tolist() == ref.tolist(), s
    assert len(set(starts[:int(seg[1])].tolist())) > 3000            # (a wrapped CDF sent most draws to a few positions)
    b.close()
    # weights steer the draws: two 8-mers with very different probabilities, counted over many draws
    g = gs[0]
    b = ga.SegmentBatch.simulate([g], 20, 400.0, 11, kmer=8, table=prob)
    _, starts = b.read_starts()
    y = orc.kmer_from_seq(g, 8, keys, prob)
    hi, lo = int(np.argmax(y[:5000])), int(np.argmin(y[:5000]))
    cnt = np.bincount(starts, minlength=len(g))
    expect_ratio = y[hi] / y[lo]
    assert cnt[hi] > cnt[lo] and 0.5 * expect_ratio < (cnt[hi] + 1) / (cnt[lo] + 1) < 2.0 * expect_ratio
    b.close()


def test_device_resident_scaffolds(qtable, monkeypatch):
    """assemble_contigs with the result left on the GPU (chains expanded to 2-bit, ordered and de-duplicated as strings
    there) and calc_breakscore fed from the handle: the scaffold list, its order and every score equal the oracle's; the
    string form of the same call goes through the same device route when it is given a context"""
    keys, prob = qtable
    rng = np.random.default_rng(13)
    for seed, L, rl, cov, k, rows in ((61, 1500, 24, 40, 15, 400), (62, 2500, 30, 35, 17, 900), (63, 800, 20, 30, 9, 150)):
        g = synth.make_segment(seed, L, n_short=4, short_len=50, n_long=2, long_len=120, tandem_len=40, planted=True)
        truth = g.tobytes().decode()
        reads = _strs(synth.simulate_reads(g, rl, cov, seed + 1))
        km = ga.get_kmers_from_reads(reads, k)
        m = ga.get_contigs(km, k, 1234, matrix_rows=rows)
        ref = orc.assemble_contigs(m.contigs, m.perm, k)
        sc = ga.assemble_contigs(m, k, on_device=True)
        assert len(sc) == len(ref) and sc.lengths.tolist() == [len(s) for s in ref]
        assert sc.strings() == ref
        assert ga.assemble_contigs(m, k, ctx=ga.default_context()) == ref
        monkeypatch.setenv("GASM_ASM_HOST_MERGE", "1")          # same scaffolds with the merge on host threads
        assert ga.assemble_contigs(m, k, ctx=ga.default_context()) == ref
        monkeypatch.delenv("GASM_ASM_HOST_MERGE")
        for variant in ("own", "velvet"):
            mine = ga.calc_breakscore(sc, reads, truth, 8, keys, prob, variant=variant, with_lev=True, with_freq=False, with_ks=(variant == "own"))
            o = orc.calc_breakscore(ref, reads, truth, 8, keys, prob, velvet=(variant == "velvet"), with_lev=True, with_freq=(variant == "own"))
            _check_scores(mine, o, with_lev=True)
            if variant == "velvet":
                assert mine["path_prob_dist_startpos"].tolist() == o["path_prob_dist_startpos"].tolist()
            else:
                y = orc.kmer_from_seq(truth, 8, keys, prob)
                for i in range(0, len(ref), max(1, len(ref) // 25)):
                    r = orc.ks_statistic(o["path_freq"][i], y)
                    assert (np.isnan(r) and np.isnan(mine["stat_test_KS"][i])) or abs(mine["stat_test_KS"][i] - r) < 1e-9
        sc.close()
    # the velvet entry (its own 20 000 shuffles), contigs that are not contigs of one graph
    contigs = sorted(set(truth[a:a + int(rng.integers(20, 90))] for a in rng.integers(0, len(truth) - 90, 14)))
    ref = orc.assemble_contigs_velvet(contigs, 11, 5, rows=300)
    sc = ga.assemble_contigs_velvet(contigs, 11, 5, rows=300, on_device=True)
    assert sc.strings() == ref
    sc.close()
    # a contig shorter than k-1: the string form behind the same handle
    short = ["ACGTACGTAA", "ACG", "GTAAACCCGGGT", "TTTTTTTTTT"]
    try:
        ref = orc.assemble_contigs_velvet(short, 4, 2, rows=40)
    except IndexError:
        ref = None
    if ref is not None:
        sc = ga.assemble_contigs_velvet(short, 4, 2, rows=40, on_device=True)
        assert sc.strings() == ref
        sc.close()


def test_guided_traversal_with_large_fixed_point_sums(qtable):
    """The guided traversal compares contig scores as exact rationals (128-bit cross products).  Round 2's kernel chose wrong
    seeds on small, tie-rich segments: the code generated for `if (better(c, b)) b = c;` left b.len at the lane's first
    candidate's length (NOTES_r3.md; found by tools/soak.py seed 91; tests/golden/guided_case_seed91.npz keeps that segment's
    reads: 2-letter genome, 367 contigs, 358 of them tied at score 0).  The kept case plus a few dozen random small batches of
    the same kind (few reads: fixed-point shift >= 63, sums of 50+ bits)."""
    from oracle import guided_oracle
    keys, prob = qtable
    table = dict(zip(keys, prob.tolist()))

    def check(reads, off, k, tag):
        b = ga.SegmentBatch(reads.reshape(-1), np.array(off, dtype=np.uint64), fixed_len=reads.shape[1])
        b.build(k).score(8, prob)
        contigs, sc = b.contigs(), b.scores()
        fx, shift = b.score_fixed()
        g = b.guided()
        for s in range(len(off) - 1):
            rs = _strs(reads[off[s]:off[s + 1]])
            a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
            ofx = guided_oracle.fixed_sums(contigs[s], rs, table, 8, shift)
            assert ofx == fx[a:e].tolist(), tag
            assert [d["sequence"] for d in g[s]] == guided_oracle.guided_paths(contigs[s], ofx, k), (tag, s, shift)
        b.close()
        return shift

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "guided_case_seed91.npz"))
    assert check(d["reads"], [0, d["reads"].shape[0]], int(d["k"]), "kept case") >= 63
    rng = np.random.default_rng(4321)
    big = 0
    for it in range(40):
        L, k, rl = int(rng.integers(800, 3500)), int(rng.choice([9, 21, 27])), int(rng.integers(40, 110))
        lut = np.frombuffer(str(rng.choice(["AC", "ACGT", "CT"])).encode(), dtype=np.uint8)
        parts, off = [], [0]
        for s in range(int(rng.integers(1, 4))):
            g = synth.make_segment(int(rng.integers(1 << 30)), L, planted=False)
            g = lut[np.frombuffer(g.tobytes(), dtype=np.uint8) % len(lut)]
            r = synth.simulate_reads(g, rl, float(rng.uniform(5, 20)), int(rng.integers(1 << 30)))
            parts.append(r)
            off.append(off[-1] + r.shape[0])
        big += check(np.concatenate(parts, axis=0), off, k, it) >= 63
    assert big >= 10


def test_guided_traversal_against_own_restatement(qtable):
    """SURVEY §8 row A16 (configs[4]'s "combined" mode).  Not in the reference: gasm_batch_guided is checked against
    oracle/guided_oracle.py, the CPU restatement of this project's own specification — the fixed-point sums it steers by
    are recomputed there read by read, the guided scaffolds must be the same strings in the same order, and their scores
    equal the oracle scorer's on those strings.  64- and 128-bit keys, several segments, a segment without branching."""
    from oracle import guided_oracle
    keys, prob = qtable
    table = dict(zip(keys, prob.tolist()))
    for k, rl, n_seg, L in ((15, 30, 4, 2500), (41, 70, 3, 4000)):
        reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, 30, seed0=9500 + k, planted=True)
        plain = synth.make_segment(77, 1500, planted=False)                      # a segment that is one contig
        rp = synth.simulate_reads(plain, rl, 30, 78)
        reads = np.concatenate([reads, rp], axis=0)
        seg_off = np.concatenate([seg_off, [seg_off[-1] + rp.shape[0]]]).astype(np.uint64)
        genomes = genomes + [plain]
        b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
        b.build(k, genome_len_hint=L).score(8, prob)
        contigs = b.contigs()
        fx, shift = b.score_fixed()
        sc = b.scores()
        assert np.array_equal(sc["bp_score"], fx.astype(np.float64) * 2.0 ** -shift)
        g = b.guided()
        for s in range(n_seg + 1):
            rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
            a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
            ofx = guided_oracle.fixed_sums(contigs[s], rs, table, 8, shift)
            assert ofx == fx[a:e].tolist(), (k, s)                               # the steering quantity, exactly
            want = guided_oracle.guided_paths(contigs[s], ofx, k)
            got = [d["sequence"] for d in g[s]]
            assert got == want, (k, s)
            assert sum(len(x) for x in got) <= sum(len(c) for c in contigs[s]) and len(got) <= len(contigs[s])
            o = orc.calc_breakscore(got, rs, "", 8, keys, prob, with_lev=False, with_freq=False)
            assert [d["kmer_breaks"] for d in g[s]] == o["kmer_breaks"].tolist()
            assert np.abs(np.array([d["bp_score"] for d in g[s]]) - o["bp_score"]).max(initial=0.0) < TOL
        assert len(g[n_seg]) == len(contigs[n_seg])                              # nothing to chain without branching nodes... or all chained
        b.close()


def test_graph_scorer_needs_no_base_comparison(qtable, monkeypatch):
    """The batch scorer takes a read's place from the graph (first k-mer -> edge -> contig, offset) and, since round 3, does not
    compare the rest of the read with the contig: inside a contig every walk is forced, so a read that fits IS the text there.
    GASM_SCORE_VERIFY=1 makes the comparison again and refuses the scores on any mismatch: both ways must give the same numbers
    (and the oracle's) — planted repeats, reads that cross branching nodes, 64- and 128-bit keys, ragged lengths."""
    keys, prob = qtable
    for k, rl, n_seg, L in ((21, 60, 4, 9000), (41, 100, 3, 8000)):
        reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, 25, seed0=9800 + k, planted=True)
        out = []
        for verify in ("0", "1"):
            monkeypatch.setenv("GASM_SCORE_VERIFY", verify)
            b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
            b.build(k, genome_len_hint=L).score(8, prob)
            out.append((b.contigs(), b.scores()))
            b.close()
        (c0, s0), (c1, s1) = out
        assert c0 == c1
        for kk in ("kmer_breaks", "sequence_len", "bp_score", "bp_score_norm_by_break_freqs", "bp_score_norm_by_len"):
            assert np.array_equal(s0[kk], s1[kk]), (k, kk)
        assert int(s0["kmer_breaks"].sum()) < int(seg_off[-1])          # some reads cross a branching node and occur in no contig
        for s in range(n_seg):
            rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
            o = orc.calc_breakscore(c0[s], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
            a, e = int(s0["seg_contig_off"][s]), int(s0["seg_contig_off"][s + 1])
            assert s0["kmer_breaks"][a:e].tolist() == o["kmer_breaks"].tolist(), (k, s)
            assert np.abs(s0["bp_score"][a:e] - o["bp_score"]).max(initial=0.0) < TOL


@pytest.mark.timeout(900)
def test_guided_traversal_at_configs4_size(qtable):
    """Row A16 at the size BASELINE configs[4] names: 50 kb segments, 250 bp reads at 100x, k = 51 (128-bit keys) — two
    segments, the steering sums recomputed read by read by the CPU restatement, the guided scaffolds equal as strings and in
    order, their scores equal to the oracle scorer's.  (Parity unpinned by the reference: the mode is this project's own.)"""
    from oracle import guided_oracle
    keys, prob = qtable
    table = dict(zip(keys, prob.tolist()))
    k, rl, n_seg, L = 51, 250, 2, 50000
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, 100, seed0=1234, planted=True)
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    b.build(k, genome_len_hint=L).score(8, prob)
    contigs = b.contigs()
    fx, shift = b.score_fixed()
    sc = b.scores()
    g = b.guided()
    for s in range(n_seg):
        rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
        a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
        assert len(contigs[s]) > 20                                            # the planted repeats branch the graph
        ofx = guided_oracle.fixed_sums(contigs[s], rs, table, 8, shift)
        assert ofx == fx[a:e].tolist(), s
        want = guided_oracle.guided_paths(contigs[s], ofx, k)
        got = [d["sequence"] for d in g[s]]
        assert got == want, s
        assert len(got) < len(contigs[s])                                      # something was chained
        o = orc.calc_breakscore(got, rs, "", 8, keys, prob, with_lev=False, with_freq=False)
        assert [d["kmer_breaks"] for d in g[s]] == o["kmer_breaks"].tolist()
        assert np.abs(np.array([d["bp_score"] for d in g[s]]) - o["bp_score"]).max(initial=0.0) < TOL
    b.close()


@pytest.mark.timeout(180)
def test_failed_attempt_is_safe_to_run_ahead(qtable):
    """Found by tools/soak.py: a build whose hint is far too small overflows its buckets; the graph kernels of that attempt
    are queued before the host reads the report, so they run on the failed attempt's leftovers — which must be empty but
    searchable (an unwritten fine directory once sent graph_lower_bound into an endless bisection).  128-bit keys, a
    three-letter genome, hint = L / 7; and the same with 64-bit keys."""
    keys, prob = qtable
    for k, rl, L, hint, alphabet in ((63, 152, 5368, 766, b"GCGA"), (27, 90, 6000, 100, b"ACGT"), (41, 120, 3000, 60, b"AACC")):
        lut = np.frombuffer(alphabet, dtype=np.uint8)
        parts, off, gens = [], [0], []
        for s in range(2):
            g = lut[np.random.default_rng(40 + s).integers(0, 4, L)]
            r = synth.simulate_reads(g, rl, 30, 50 + s)
            parts.append(r); off.append(off[-1] + r.shape[0]); gens.append(g)
        reads = np.concatenate(parts, axis=0)
        b = ga.SegmentBatch(reads.reshape(-1), np.array(off, dtype=np.uint64), fixed_len=rl)
        for _ in range(2):
            b.build(k, genome_len_hint=hint).score(8, prob)
            b.ctx.sync()                                  # the queued attempt (and its successors) must come to an end
        _check_segments_vs_oracle(b, reads, np.array(off, dtype=np.uint64), gens, range(2), k, keys, prob)
        b.close()
