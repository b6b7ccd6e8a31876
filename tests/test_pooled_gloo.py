"""The pooled-build protocol (genomeassembler_dev_amd/pooled.py) over torch.distributed with the gloo backend on the
CPU, world sizes 2 and 4: the three all-to-alls, run directories and offsets are the code that runs over RCCL on the
GPUs; the per-rank device work is stood in for by tests/pooled_oracle_backend.py.  What every rank ends up with must
equal a single-process oracle run, whatever the world size (and the in-process virtual ranks must agree with both)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, BBITS, RL = 15, 3, 40


def _data():
    from genomeassembler_dev_amd import synth
    return synth.make_batch(5, 1200, RL, 14, seed0=4100, planted=True)


def _shard(reads, seg_off, rank, world):
    parts, off = [], [0]
    for s in range(len(seg_off) - 1):
        r = reads[int(seg_off[s]):int(seg_off[s + 1])][rank::world]
        parts.append(r)
        off.append(off[-1] + r.shape[0])
    return np.concatenate(parts, axis=0), np.array(off, dtype=np.uint64)


def _table():
    import itertools

    from oracle import orc
    raw = np.fromfile(os.path.join(ROOT, "genomeassembler_dev_amd", "data", "querytable_raw_f64.bin"), dtype="<f8")
    prob = orc.normalise_tables(raw, [16, 256, 4096, 65536])
    keys = ["".join(t) for k in (2, 4, 6, 8) for t in itertools.product("ACGT", repeat=k)]
    return keys, prob


def _reference():
    from oracle import orc
    reads, seg_off, _ = _data()
    keys, prob = _table()
    out = []
    for s in range(len(seg_off) - 1):
        rs = [r.tobytes().decode() for r in reads[int(seg_off[s]):int(seg_off[s + 1])]]
        ref = orc.get_contigs(orc.kmers_from_reads(rs, K), K, 1, rows=1)
        o = orc.calc_breakscore(ref["contigs"], rs, "", 8, keys, prob, with_lev=False, with_freq=False)
        out.append((ref, o))
    return out


def _check(res, a, b, ref):
    for s in range(a, b):
        d, (g, o) = res[s - a], ref[s]
        assert d["contigs"] == g["contigs"], s
        assert d["distinct"] == g["distinct"] and list(d["counts"]) == g["counts"].tolist(), s
        assert list(d["kmer_breaks"]) == o["kmer_breaks"].tolist(), s
        assert np.abs(np.asarray(d["bp_score"]) - o["bp_score"]).max(initial=0.0) < 1e-12, s


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch.distributed as dist

        from genomeassembler_dev_amd import pooled
        from pooled_oracle_backend import OracleBackend
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        reads, seg_off, _ = _data()
        keys, prob = _table()
        rr, so = _shard(reads, seg_off, rank, world)
        be = OracleBackend(rr, so, RL, keys)
        comm = pooled.DistComm()
        own = pooled.pooled_build(comm, {rank: be}, len(seg_off) - 1, K, BBITS, kmer=8, table=prob)
        a, b = own[rank]
        _check(be.results(), a, b, _reference())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", (a, b)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail: " + repr(e) + "\n" + traceback.format_exc(), None))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_pooled_protocol_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(g[1] == "ok" for g in got), got
    covered = sorted(g[2] for g in got)
    assert covered[0][0] == 0 and covered[-1][1] == 5 and all(x[1] == y[0] for x, y in zip(covered, covered[1:]))


@pytest.mark.parametrize("world", [1, 3])
def test_pooled_protocol_virtual_ranks_cpu(world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from genomeassembler_dev_amd import pooled
    from pooled_oracle_backend import OracleBackend
    reads, seg_off, _ = _data()
    keys, prob = _table()
    be = {}
    for r in range(world):
        rr, so = _shard(reads, seg_off, r, world)
        be[r] = OracleBackend(rr, so, RL, keys)
    own = pooled.pooled_build(pooled.VirtualComm(world), be, len(seg_off) - 1, K, BBITS, kmer=8, table=prob)
    ref = _reference()
    for r in range(world):
        _check(be[r].results(), own[r][0], own[r][1], ref)
