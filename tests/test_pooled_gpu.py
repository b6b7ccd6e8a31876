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
