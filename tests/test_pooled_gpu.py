"""Pooled (multi-GPU) build on ONE GPU: N virtual ranks in this process run the same protocol code as N ranks over RCCL
(genomeassembler_dev_amd/pooled.py), buffers swapped in memory.  Results must equal the single-GPU path's and the
oracle's whatever N is."""
import numpy as np
import pytest

import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import pooled, synth
from oracle import orc

pytestmark = pytest.mark.gpu


def _strs(a):
    return [r.tobytes().decode() for r in a]


def _shard_reads(reads, seg_off, rank, world):
    """rank's share of the reads of every segment (every world-th read), with its own seg_read_off"""
    parts, off = [], [0]
    for s in range(len(seg_off) - 1):
        r = reads[int(seg_off[s]):int(seg_off[s + 1])][rank::world]
        parts.append(r)
        off.append(off[-1] + r.shape[0])
    return np.concatenate(parts, axis=0), np.array(off, dtype=np.uint64)


@pytest.mark.parametrize("n_seg,L,rl,cov,k,bbits,worlds", [(7, 3000, 60, 24, 21, 3, (1, 2, 3, 4)), (5, 2500, 90, 20, 41, 2, (1, 3)),
                                                         (3, 9000, 100, 30, 31, 5, (2, 8))])
def test_pooled_virtual_ranks_equal_single_gpu(qtable, n_seg, L, rl, cov, k, bbits, worlds):
    keys, prob = qtable
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=6100 + k, planted=True)
    single = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    single.build(k, genome_len_hint=L).score(8, prob)
    s_contigs, s_sc = single.contigs(), single.scores()
    s_dist = [single.distinct_kmers(s) for s in range(n_seg)]
    for world in worlds:
        comm = pooled.VirtualComm(world)
        be = {}
        for r in range(world):
            rr, so = _shard_reads(reads, seg_off, r, world)
            be[r] = pooled.GasmBackend(rr, so, rl)
        stats = {}
        own = pooled.pooled_build(comm, be, n_seg, k, bbits, kmer=8, table=prob, stats=stats)
        seen = 0
        for r in range(world):
            a, b = own[r]
            res = be[r].results()
            assert len(res) == b - a
            for s in range(a, b):
                d = res[s - a]
                assert d["contigs"] == s_contigs[s], (world, s)
                assert d["distinct"] == s_dist[s][0] and d["counts"].tolist() == s_dist[s][1].tolist(), (world, s)
                ca, ce = int(s_sc["seg_contig_off"][s]), int(s_sc["seg_contig_off"][s + 1])
                assert d["kmer_breaks"].tolist() == s_sc["kmer_breaks"][ca:ce].tolist(), (world, s)
                assert d["sequence_len"].tolist() == s_sc["sequence_len"][ca:ce].tolist()
                for kk in ("bp_score", "bp_score_norm_by_break_freqs", "bp_score_norm_by_len"):
                    assert np.abs(d[kk] - s_sc[kk][ca:ce]).max(initial=0.0) < 1e-12, (world, s, kk)
                seen += 1
        assert seen == n_seg
        assert all(len(v) == 3 for v in stats["bytes_sent"].values())
        for r in range(world):
            be[r].close()
    # ... and against the oracle (first and last segment)
    for s in (0, n_seg - 1):
        rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
        ref = orc.get_contigs(orc.kmers_from_reads(rs, k), k, 1, rows=1)
        assert s_contigs[s] == ref["contigs"]
    single.close()


def test_pooled_ownership_functions():
    for world in (1, 2, 3, 8):
        o1 = pooled.bucket_owner(100, 6, world)
        assert o1.min() >= 0 and o1.max() < world and len(o1) == 6400
        if world > 1:
            cnt = np.bincount(o1, minlength=world)
            assert cnt.min() > 0.8 * 6400 / world and cnt.max() < 1.2 * 6400 / world      # even enough to balance the merge
        o2 = pooled.segment_owner(100, world)
        assert (np.diff(o2) >= 0).all() and set(o2.tolist()) == set(range(min(world, 100)))


def test_pooled_with_empty_ranks(qtable):
    """more ranks than there is work: ranks whose buckets receive no record, a rank without reads, a segment without reads"""
    keys, prob = qtable
    k, rl = 9, 20
    g = synth.make_segment(5, 200, planted=False)
    r0 = synth.simulate_reads(g, rl, 6, 6)
    reads = r0
    seg_off = np.array([0, r0.shape[0], r0.shape[0]], dtype=np.uint64)          # second segment: no reads at all
    single = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    single.build(k).score(8, prob)
    want = single.contigs()
    for world, bbits in ((4, 0), (8, 2), (3, 1)):
        be = {}
        for r in range(world):
            rr, so = _shard_reads(reads, seg_off, r, world)
            if r == world - 1:
                rr, so = rr[:0], np.zeros(3, dtype=np.uint64)                   # a rank that holds nothing
            be[r] = pooled.GasmBackend(rr, so, rl)
        own = pooled.pooled_build(pooled.VirtualComm(world), be, 2, k, bbits, kmer=8, table=prob)
        got = {}
        for r in range(world):
            for s, d in zip(range(*own[r]), be[r].results()):
                got[s] = d["contigs"]
            be[r].close()
        # the last rank's reads were dropped, so compare with a single-GPU build of the same reduced read set
        kept = np.concatenate([_shard_reads(reads, seg_off, r, world)[0] for r in range(world - 1)], axis=0)
        ref = ga.SegmentBatch(kept.reshape(-1), np.array([0, kept.shape[0], kept.shape[0]], dtype=np.uint64), fixed_len=rl)
        ref.build(k)
        assert [got[0], got[1]] == ref.contigs(), (world, bbits)
        ref.close()
    assert want[1] == []
    single.close()


# ---------------------------------------------------------------------------------------------------------------------
# The same step through the library's own exchange (gasm_pool_exchange_build, csrc/exchange.hip): plans on the device, the
# three all-to-alls inside libgasm — device copies between virtual ranks here, ncclSend / ncclRecv between processes.
# ---------------------------------------------------------------------------------------------------------------------
def _check_against_single(be, own, single, n_seg, tag):
    s_contigs, s_sc = single.contigs(), single.scores()
    seen = 0
    for r, (a, b) in own.items():
        res = be[r].results()
        assert len(res) == b - a, tag
        for s in range(a, b):
            d = res[s - a]
            sk, sm = single.distinct_kmers(s)
            assert d["contigs"] == s_contigs[s], (tag, s)
            assert d["distinct"] == sk and d["counts"].tolist() == sm.tolist(), (tag, s)
            ca, ce = int(s_sc["seg_contig_off"][s]), int(s_sc["seg_contig_off"][s + 1])
            assert d["kmer_breaks"].tolist() == s_sc["kmer_breaks"][ca:ce].tolist(), (tag, s)
            assert d["sequence_len"].tolist() == s_sc["sequence_len"][ca:ce].tolist(), (tag, s)
            for kk in ("bp_score", "bp_score_norm_by_break_freqs", "bp_score_norm_by_len"):
                assert np.abs(d[kk] - s_sc[kk][ca:ce]).max(initial=0.0) < 1e-12, (tag, s, kk)
            seen += 1
    assert seen == n_seg, tag


@pytest.mark.parametrize("n_seg,L,rl,cov,k,bbits,worlds", [(7, 3000, 60, 24, 21, 3, (1, 2, 3, 4)), (5, 2500, 90, 20, 51, 2, (1, 3)),
                                                         (3, 9000, 100, 30, 31, 5, (2, 8)), (9, 2000, 120, 16, 63, 4, (5,))])
def test_exchange_build_virtual_ranks_equal_single_gpu(qtable, n_seg, L, rl, cov, k, bbits, worlds):
    keys, prob = qtable
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=7100 + k, planted=True)
    single = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    single.build(k, genome_len_hint=L).score(8, prob)
    ctx = ga.default_context()
    for world in worlds:
        comm = pooled.Comm.virtual(ctx, world)
        be = [pooled.GasmBackend(*_shard_reads(reads, seg_off, r, world), rl) for r in range(world)]
        for step in range(2):          # the second step runs on the cached plans (ownership, reads) of the first
            stats, own = pooled.exchange_build(comm, be, k, bbits, kmer=8, table=prob)
            assert stats["bbits"] == bbits and stats["attempts"] == 1
            _check_against_single(dict(enumerate(be)), own, single, n_seg, (world, step))
        if world > 1:
            assert stats["bytes_sent_remote"][0] > 0 and stats["bytes_sent_remote"][0] < stats["bytes_sent"][0]
        for b in be:
            b.close()
        comm.close()
    single.close()


def test_exchange_build_with_empty_ranks(qtable):
    """more ranks than there is work: ranks whose buckets receive no record, a rank without reads, a segment without reads"""
    keys, prob = qtable
    k, rl = 9, 20
    g = synth.make_segment(5, 200, planted=False)
    r0 = synth.simulate_reads(g, rl, 6, 6)
    seg_off = np.array([0, r0.shape[0], r0.shape[0]], dtype=np.uint64)          # second segment: no reads at all
    ctx = ga.default_context()
    for world, bbits in ((4, 0), (8, 2), (3, 1)):
        be = []
        for r in range(world):
            rr, so = _shard_reads(r0, seg_off, r, world)
            if r == world - 1:
                rr, so = rr[:0], np.zeros(3, dtype=np.uint64)                   # a rank that holds nothing
            be.append(pooled.GasmBackend(rr, so, rl))
        comm = pooled.Comm.virtual(ctx, world)
        stats, own = pooled.exchange_build(comm, be, k, bbits, kmer=8, table=prob)
        kept = np.concatenate([_shard_reads(r0, seg_off, r, world)[0] for r in range(world - 1)], axis=0)
        ref = ga.SegmentBatch(kept.reshape(-1), np.array([0, kept.shape[0], kept.shape[0]], dtype=np.uint64), fixed_len=rl)
        ref.build(k).score(8, prob)
        _check_against_single(dict(enumerate(be)), own, ref, 2, (world, bbits))
        ref.close()
        for b in be:
            b.close()
        comm.close()


def test_exchange_build_overflow_is_collective(qtable):
    """Capacity failures are decided by all ranks together.  Two segments at 1x coverage per rank, one bucket per segment: the
    larger segment's local runs outgrow the small tables (every rank: first rung, larger tables), then fit — but their union
    outgrows the merge table at the ONE rank that owns the bucket.  Round 2's protocol raised on that rank while the others
    walked into the next all-to-all; here the flag travels with the length tables, the whole group moves to more bucket bits
    in the same call, and the result equals the single-GPU build."""
    keys, prob = qtable
    k, rl, world = 21, 60, 3
    g_a, g_b = synth.make_segment(41, 1500, planted=False), synth.make_segment(42, 4000, planted=False)
    r_a, r_b = synth.simulate_reads(g_a, rl, 3, 43), synth.simulate_reads(g_b, rl, 3, 44)
    reads = np.concatenate([r_a, r_b], axis=0)
    seg_off = np.array([0, r_a.shape[0], r_a.shape[0] + r_b.shape[0]], dtype=np.uint64)
    single = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    single.build(k).score(8, prob)
    n_b = len(single.distinct_kmers(1)[0])
    assert n_b > 2816, n_b                         # the union cannot fit one merge table ...
    ctx = ga.default_context()
    be = [pooled.GasmBackend(*_shard_reads(reads, seg_off, r, world), rl) for r in range(world)]
    comm = pooled.Comm.virtual(ctx, world)
    stats, own = pooled.exchange_build(comm, be, k, 0, kmer=8, table=prob)
    assert stats["attempts"] >= 2 and stats["bbits"] >= 2, stats
    _check_against_single(dict(enumerate(be)), own, single, 2, "overflow")
    for b in be:
        b.close()
    comm.close()
    single.close()


def test_exchange_build_over_rccl_world_of_one(qtable):
    """The RCCL communicator itself (ncclCommInitRank, all-gather, all-reduce, the grouped send / receive path with a rank's own
    part) with the one rank a one-GPU box can have; N > 1 is what the driver's 8-GPU run executes."""
    keys, prob = qtable
    n_seg, L, rl, cov, k, bbits = 4, 3000, 80, 20, 31, 3
    reads, seg_off, _ = synth.make_batch(n_seg, L, rl, cov, seed0=8100, planted=True)
    single = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    single.build(k, genome_len_hint=L).score(8, prob)
    ctx = ga.default_context()
    comm = pooled.Comm.rccl(ctx, pooled.Comm.unique_id(), 0, 1)
    be = pooled.GasmBackend(reads, seg_off, rl)
    for _ in range(2):
        stats, own = pooled.exchange_build(comm, be, k, bbits, kmer=8, table=prob)
        _check_against_single({0: be}, own, single, n_seg, "rccl-1")
    be.close()
    comm.close()
    single.close()


def test_exchange_build_headline_shape(qtable):
    """configs[2]'s batch (100 x 50 kb, 150 bp at 50x, k = 31) through 8 virtual ranks: every segment's contigs, multiplicities
    and scores equal the single-GPU build's; the oracle on segments 0, 63, 64, 99."""
    keys, prob = qtable
    n_seg, L, rl, cov, k, bbits, world = 100, 50000, 150, 50, 31, 6, 8
    reads, seg_off, genomes = synth.make_batch(n_seg, L, rl, cov, seed0=1234, planted=True)
    single = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    single.build(k, genome_len_hint=L).score(8, prob)
    ctx = ga.default_context()
    comm = pooled.Comm.virtual(ctx, world)
    be = [pooled.GasmBackend(*_shard_reads(reads, seg_off, r, world), rl) for r in range(world)]
    stats, own = pooled.exchange_build(comm, be, k, bbits, kmer=8, table=prob)
    assert stats["attempts"] == 1
    _check_against_single(dict(enumerate(be)), own, single, n_seg, "headline")
    s_contigs, s_sc = single.contigs(), single.scores()
    for s in (0, 63, 64, 99):
        rs = _strs(reads[int(seg_off[s]):int(seg_off[s + 1])])
        o = orc.build_score(rs, k, 8, keys, prob)
        assert s_contigs[s] == o["contigs"], s
        ca, ce = int(s_sc["seg_contig_off"][s]), int(s_sc["seg_contig_off"][s + 1])
        assert s_sc["kmer_breaks"][ca:ce].tolist() == o["kmer_breaks"].tolist(), s
        assert np.abs(s_sc["bp_score"][ca:ce] - o["bp_score"]).max(initial=0.0) < 1e-9, s
    for b in be:
        b.close()
    comm.close()
    single.close()
