"""Pooled builds over several GPUs: SURVEY §8(e) mode 2 — reads of every segment spread evenly over the ranks, k-mers
bucketed by hash of (segment, k-mer prefix), one all-to-all to bring every bucket's records together, the global
edge-list merge, and the graph / contigs / scores at each segment's owner.  (The reference has no counterpart: it loops
over segments on one thread, scripts/02_Real_vs_rand_prob_own.R:33-53.  Mode 1 — whole segments per rank, no exchange —
is `parallel.py` and stays the default for configs 3-5: it moves nothing.)

What is here is host logic only: who owns what, which runs go where, and where they lie in the buffers that come back.
The device work is libgasm's (gasm_pool_* in include/gasm.h, reached through `GasmBackend`); the exchange itself is
`torch.distributed.all_to_all_single` — RCCL's grouped send/recv when the backend is "nccl": on a fully connected xGMI
node every pair of ranks has its own link, so the all-to-all is link-parallel (no ring) — or, for N virtual ranks inside
one process, a plain swap of the buffers (`VirtualComm`: the same protocol code, testable on one GPU).

Three exchanges per build:
  1. records: rank r's sorted distinct (key, count) run of bucket (segment, prefix) -> bucket_owner(segment, prefix)
     12 B per record for k <= 31 (8 B key + 4 B count), 20 B for k <= 63; the run lengths travel first (4 B per bucket);
  2. merged records (the global distinct edge list) -> segment_owner(segment), which needs the whole graph of a segment;
  3. the segment's reads, 2-bit packed, -> segment_owner(segment) for scoring.
Outputs are independent of the number of ranks: ownership only decides where a bucket is merged, not what the merge gives.
"""
import numpy as np

from .parallel import shard_bounds


# ------------------------------------------------------------------------------------------------- ownership
def bucket_owner(n_segments, bbits, world):
    """owner rank of every bucket (index = segment << bbits | prefix): a multiplicative hash of (segment, prefix)"""
    nb = 1 << bbits
    gb = np.arange(n_segments * nb, dtype=np.uint64)
    seg, pre = gb >> np.uint64(bbits), gb & np.uint64(nb - 1)
    h = (seg * np.uint64(0x9E3779B97F4A7C15) + pre * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(0x165667B19E3779F9))
    h ^= h >> np.uint64(29)
    h = h * np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(32)
    return (h % np.uint64(world)).astype(np.int64)


def segment_owner(n_segments, world):
    """owner rank of every segment: contiguous blocks (parallel.shard_bounds), so a rank's segments are a range"""
    own = np.zeros(n_segments, dtype=np.int64)
    for r, (a, b) in enumerate(shard_bounds(n_segments, world)):
        own[a:b] = r
    return own


# ------------------------------------------------------------------------------------------------- communicators
class VirtualComm:
    """`world` virtual ranks inside this process: an all-to-all is a transposition of the send lists"""

    def __init__(self, world):
        self.world = world
        self.local_ranks = list(range(world))

    def all_to_all(self, sends, recv_sizes=None):
        return {r: [sends[s][r] for s in range(self.world)] for r in self.local_ranks}

    def barrier(self):
        pass


class DistComm:
    """one rank per process over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" on the CPU)"""

    def __init__(self, group=None, stage_on_host=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local_ranks = [self.rank]
        # gloo moves host memory: device buffers are staged through the host (rehearsals of the multi-rank flow on a box
        # without RCCL peers); with "nccl" the device buffers go as they are
        self.stage = (dist.get_backend(group) == "gloo") if stage_on_host is None else stage_on_host

    def all_to_all(self, sends, recv_sizes):
        """sends[rank] = list of `world` 1-D tensors; recv_sizes[rank] = list of `world` element counts"""
        import torch
        send = sends[self.rank]
        sizes_in = [int(t.numel()) for t in send]
        sizes_out = [int(n) for n in recv_sizes[self.rank]]
        ref = send[0]
        inp = torch.cat([t.reshape(-1) for t in send]) if sum(sizes_in) else ref.new_empty(0)
        if self.stage and inp.is_cuda:
            h_in = inp.cpu()
            h_out = h_in.new_empty(sum(sizes_out))
            self.dist.all_to_all_single(h_out, h_in, output_split_sizes=sizes_out, input_split_sizes=sizes_in, group=self.group)
            out = h_out.to(ref.device)
        else:
            out = ref.new_empty(sum(sizes_out))
            self.dist.all_to_all_single(out, inp, output_split_sizes=sizes_out, input_split_sizes=sizes_in, group=self.group)
        return {self.rank: list(torch.split(out, sizes_out))}

    def barrier(self):
        self.dist.barrier(self.group)


# ------------------------------------------------------------------------------------------------- the libgasm backend
class GasmBackend:
    """One rank's device work through the gasm_pool_* C ABI; exchange buffers are torch tensors on the rank's GPU."""

    def __init__(self, reads, seg_read_off, fixed_len, ctx=None, device=None):
        import ctypes as C

        from ._lib import check, default_context, lib
        self.C, self.lib, self.check = C, lib(), check
        self.ctx = ctx or default_context()
        self._device = device
        reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
        self.seg_read_off = np.ascontiguousarray(seg_read_off, dtype=np.uint64)
        self.n_segments = len(self.seg_read_off) - 1
        self.fixed_len = int(fixed_len)
        h = C.c_void_p()
        check(self.lib.gasm_pool_create(self.ctx.h, reads.ctypes.data_as(C.c_void_p), int(self.seg_read_off[-1]), self.fixed_len,
                                        self.seg_read_off.ctypes.data_as(C.c_void_p), self.n_segments, C.byref(h)))
        self.h = h
        self.words = 1
        self.k = None

    # torch is needed only by the stage-by-stage protocol below (exchange buffers as tensors); the library's own exchange
    # (`exchange_build`) never touches it
    @property
    def torch(self):
        import torch
        return torch

    @property
    def device(self):
        if self._device is None:
            self._device = self.torch.device("cuda", self.ctx.device)
        return self._device

    def _sync_torch(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def _u32(self, ptr, n):
        return np.ctypeslib.as_array(self.C.cast(ptr, self.C.POINTER(self.C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def local_runs(self, k, bbits):
        p = self.C.c_void_p()
        self.check(self.lib.gasm_pool_local_runs(self.h, int(k), int(bbits), self.C.byref(p)))
        self.k, self.bbits = int(k), int(bbits)
        self.words = self.lib.gasm_pool_key_words(self.h)
        return self._u32(p, self.n_segments << bbits)

    def pack_runs(self, bucket_ix, n_records):
        """-> (keys int64[n_records * words], counts int32[n_records]) on the device"""
        t = self.torch
        keys = t.empty(max(n_records * self.words, 0), dtype=t.int64, device=self.device)
        cnt = t.empty(max(n_records, 0), dtype=t.int32, device=self.device)
        ix = np.ascontiguousarray(bucket_ix, dtype=np.uint32)
        if ix.size:
            self._sync_torch()
            self.check(self.lib.gasm_pool_pack_runs(self.h, ix.ctypes.data_as(self.C.c_void_p), ix.size, self.C.c_void_p(keys.data_ptr()),
                                                    self.C.c_void_p(cnt.data_ptr())))
        return keys, cnt

    def merge_runs(self, n_out, n_src, run_off, run_len, keys, counts):
        ro = np.ascontiguousarray(run_off, dtype=np.uint64)
        rl = np.ascontiguousarray(run_len, dtype=np.uint32)
        p = self.C.c_void_p()
        self._sync_torch()
        self.check(self.lib.gasm_pool_merge_runs(self.h, int(n_out), int(n_src), ro.ctypes.data_as(self.C.c_void_p), rl.ctypes.data_as(self.C.c_void_p),
                                                 self.C.c_void_p(keys.data_ptr()), self.C.c_void_p(counts.data_ptr()), self.C.byref(p)))
        return self._u32(p, n_out)

    def graph(self, n_local):
        self.n_local = int(n_local)
        self.check(self.lib.gasm_pool_graph(self.h, self.n_local))

    def reads_per_segment(self):
        return np.diff(self.seg_read_off).astype(np.int64)

    def pack_reads(self, seg_lo, seg_hi):
        """-> int64 words of the pieces of segments [seg_lo, seg_hi), back to back"""
        n = seg_hi - seg_lo
        nw = np.zeros(max(n, 1), dtype=np.uint64)
        if n:
            self.check(self.lib.gasm_pool_piece_words(self.h, seg_lo, seg_hi, nw.ctypes.data_as(self.C.c_void_p)))
        total = int(nw[:n].sum())
        w = self.torch.empty(total, dtype=self.torch.int64, device=self.device)
        if total:
            self._sync_torch()
            self.check(self.lib.gasm_pool_pack_reads(self.h, seg_lo, seg_hi, self.C.c_void_p(w.data_ptr())))
        return w

    def set_reads(self, words, piece_seg, piece_reads, piece_word_off):
        ps = np.ascontiguousarray(piece_seg, dtype=np.uint32)
        pr = np.ascontiguousarray(piece_reads, dtype=np.uint64)
        po = np.ascontiguousarray(piece_word_off, dtype=np.uint64)
        self._sync_torch()
        self.check(self.lib.gasm_pool_set_reads(self.h, self.C.c_void_p(words.data_ptr()), int(words.numel()), ps.size, ps.ctypes.data_as(self.C.c_void_p),
                                                pr.ctypes.data_as(self.C.c_void_p), po.ctypes.data_as(self.C.c_void_p)))

    def score(self, kmer, table):
        t = np.ascontiguousarray(table, dtype=np.float64)
        self._table = t
        self.check(self.lib.gasm_pool_score(self.h, int(kmer), t.ctypes.data_as(self.C.c_void_p)))

    # results of the rank's own segments (as SegmentBatch)
    def results(self, with_scores=True):
        from .api import unpack_kmers
        C = self.C
        so, ks, ms, w = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        self.check(self.lib.gasm_pool_fetch_distinct(self.h, C.byref(so), C.byref(ks), C.byref(ms), C.byref(w)))
        S = self.n_local
        seg = np.ctypeslib.as_array(C.cast(so, C.POINTER(C.c_uint64)), shape=(S + 1,)).copy()
        n = int(seg[-1])
        keys = np.ctypeslib.as_array(C.cast(ks, C.POINTER(C.c_uint64)), shape=(n * w.value,)).copy() if n else np.zeros(0, np.uint64)
        mult = np.ctypeslib.as_array(C.cast(ms, C.POINTER(C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        co, off, data = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.check(self.lib.gasm_pool_fetch_contigs(self.h, C.byref(co), C.byref(off), C.byref(data)))
        cseg = np.ctypeslib.as_array(C.cast(co, C.POINTER(C.c_uint64)), shape=(S + 1,)).copy()
        nc = int(cseg[-1])
        o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(nc + 1,)).copy()
        raw = C.string_at(data, int(o[-1])) if nc and o[-1] else b""
        out = []
        sc = None
        if with_scores:
            ps = [C.c_void_p() for _ in range(5)]
            self.check(self.lib.gasm_pool_fetch_scores(self.h, *[C.byref(p) for p in ps]))
            arr = lambda p, ct: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(nc,)).copy() if nc else np.zeros(0, ct)
            sc = dict(bp_score=arr(ps[0], C.c_double), bp_score_norm_by_break_freqs=arr(ps[1], C.c_double),
                      bp_score_norm_by_len=arr(ps[2], C.c_double), kmer_breaks=arr(ps[3], C.c_int32), sequence_len=arr(ps[4], C.c_int32))
        for s in range(S):
            a, b = int(seg[s]), int(seg[s + 1])
            ca, cb = int(cseg[s]), int(cseg[s + 1])
            d = dict(distinct=unpack_kmers(keys[a * w.value:b * w.value], self.k, w.value), counts=mult[a:b].astype(np.int64),
                     contigs=[raw[int(o[c]):int(o[c + 1])].decode() for c in range(ca, cb)])
            if sc is not None:
                d.update({kk: v[ca:cb] for kk, v in sc.items()})
            out.append(d)
        return out

    def close(self):
        if self.h:
            self.lib.gasm_pool_free(self.h)
            self.h = None


# ------------------------------------------------------------------------------------------------- the library's own exchange
class Comm:
    """A communicator of libgasm (include/gasm.h, csrc/exchange.hip): RCCL — one rank per process, created from a 128-byte id
    that rank 0 makes (`Comm.unique_id()`) and the caller's bootstrap hands round — or `world` virtual ranks of this process
    on one GPU (the exchanges are device copies: the same plans and kernels, testable on a one-GPU box)."""

    def __init__(self, h, ctx, world, rank):
        self.h, self.ctx, self.world, self.rank = h, ctx, world, rank

    @staticmethod
    def unique_id():
        import ctypes as C

        from ._lib import check, lib
        buf = C.create_string_buffer(128)
        check(lib().gasm_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx, uid, rank, world):
        import ctypes as C

        from ._lib import check, lib
        h = C.c_void_p()
        check(lib().gasm_comm_create(ctx.h, C.create_string_buffer(bytes(uid), 128), int(rank), int(world), C.byref(h)))
        return cls(h, ctx, int(world), int(rank))

    @classmethod
    def virtual(cls, ctx, world):
        import ctypes as C

        from ._lib import check, lib
        h = C.c_void_p()
        check(lib().gasm_comm_create_virtual(ctx.h, int(world), C.byref(h)))
        return cls(h, ctx, int(world), -1)

    def stage(self):
        from ._lib import lib
        return lib().gasm_comm_stage(self.h)

    def close(self):
        from ._lib import lib
        if self.h:
            lib().gasm_comm_destroy(self.h)
            self.h = None


def owners(n_segments, bbits, world):
    """(bucket owner per bucket, first segment of every rank) as libgasm computes them (gasm_pool_bucket_owner / _segment_bounds)"""
    import ctypes as C

    from ._lib import check, lib
    own = np.zeros(n_segments << bbits, dtype=np.uint32)
    first = np.zeros(world + 1, dtype=np.uint32)
    check(lib().gasm_pool_bucket_owner(n_segments, bbits, world, own.ctypes.data_as(C.c_void_p)))
    check(lib().gasm_pool_segment_bounds(n_segments, world, first.ctypes.data_as(C.c_void_p)))
    return own, first


def exchange_build(comm, backends, k, bbits, kmer=8, table=None):
    """One pooled step through gasm_pool_exchange_build: `backends` = this rank's GasmBackend (RCCL) or the list of all virtual
    ranks' backends in rank order.  Returns (stats dict, {rank: (first own segment, one past the last)}); per-segment results
    are the backends' (`results()`)."""
    import ctypes as C

    from ._lib import check, lib
    bl = list(backends) if isinstance(backends, (list, tuple)) else [backends]
    arr = (C.c_void_p * len(bl))(*[b.h for b in bl])
    st = (C.c_uint64 * 8)()
    t = None
    if table is not None:
        t = np.ascontiguousarray(table, dtype=np.float64)
    check(lib().gasm_pool_exchange_build(comm.h, arr, len(bl), int(k), int(bbits), int(kmer), t.ctypes.data_as(C.c_void_p) if t is not None else None, st))
    n_segments = bl[0].n_segments
    _own, first = owners(n_segments, int(st[7]), comm.world)
    ranks = range(comm.world) if comm.rank < 0 else [comm.rank]
    own = {}
    for b, r in zip(bl, ranks):
        b.k, b.bbits, b.n_local = int(k), int(st[7]), int(first[r + 1] - first[r])
        b.words = lib().gasm_pool_key_words(b.h)
        own[r] = (int(first[r]), int(first[r + 1]))
    stats = {"bytes_sent": [int(st[0]), int(st[1]), int(st[2])], "bytes_sent_remote": [int(st[3]), int(st[4]), int(st[5])],
             "attempts": int(st[6]), "bbits": int(st[7])}
    return stats, own


# ------------------------------------------------------------------------------------------------- the protocol
def _as_i64(torch, a, like):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64), device=like.device)


def pooled_build(comm, backends, n_segments, k, bbits, kmer=8, table=None, stats=None):
    """One pooled build + scoring.  backends: {rank: backend} for the ranks this process hosts (comm.local_ranks).
    Returns {rank: (first own segment, one past the last)}; the per-segment results are the backends' (`results()`).
    stats (optional dict) receives the bytes every local rank sends per exchange."""
    import torch
    W, nb = comm.world, 1 << bbits
    own1 = bucket_owner(n_segments, bbits, W)            # bucket -> rank that merges it
    own2 = segment_owner(n_segments, W)                  # segment -> rank that builds its graph and scores it
    seg_of = np.arange(n_segments * nb) >> bbits
    mine1 = [np.nonzero(own1 == d)[0] for d in range(W)]           # buckets merged at rank d, increasing
    bounds = shard_bounds(n_segments, W)
    lr = comm.local_ranks
    sent = {r: [0, 0, 0] for r in lr}

    # ---- stage 1: local runs
    lens = {r: backends[r].local_runs(k, bbits).astype(np.int64) for r in lr}
    words = backends[lr[0]].words

    def exchange_runs(run_lists, run_lens, which):
        """run_lists[r][d]: bucket indices (of r's current runs) bound for d, run_lens[r]: r's current run lengths.
        -> {r: (keys, counts, lens_from[src] arrays)} as received"""
        keys_c, cnt_c, len_s, n_recs = {}, {}, {}, {}
        for r in lr:
            order = np.concatenate(run_lists[r]) if W else np.zeros(0, np.int64)
            per_dst = [int(run_lens[r][ix].sum()) for ix in run_lists[r]]
            keys, cnt = backends[r].pack_runs(order, sum(per_dst))
            n_recs[r] = per_dst
            keys_c[r] = list(torch.split(keys, [n * words for n in per_dst]))
            cnt_c[r] = list(torch.split(cnt, per_dst))
            len_s[r] = [_as_i64(torch, run_lens[r][ix], keys) for ix in run_lists[r]]
            sent[r][which] += sum(per_dst) * (8 * words + 4) + 4 * len(order)
        return keys_c, cnt_c, len_s

    def recv_all(keys_c, cnt_c, len_s, n_lens_from):
        """n_lens_from[r][src]: how many run lengths r gets from src (known from the ownership functions)"""
        lens_in = comm.all_to_all(len_s, n_lens_from)
        rec_from = {r: [int(t.sum().item()) for t in lens_in[r]] for r in lr}
        keys_in = comm.all_to_all(keys_c, {r: [n * words for n in rec_from[r]] for r in lr})
        cnt_in = comm.all_to_all(cnt_c, rec_from)
        return lens_in, rec_from, keys_in, cnt_in

    # ---- exchange 1: every bucket's runs to the bucket's owner
    keys_c, cnt_c, len_s = exchange_runs({r: mine1 for r in lr}, lens, 0)
    lens_in, rec_from, keys_in, cnt_in = recv_all(keys_c, cnt_c, len_s, {r: [len(mine1[r])] * W for r in lr})
    merged = {}
    for r in lr:
        n_out = len(mine1[r])
        L = np.stack([t.cpu().numpy() for t in lens_in[r]], axis=1) if n_out else np.zeros((0, W), np.int64)   # [bucket, src]
        base = np.concatenate([[0], np.cumsum(rec_from[r])[:-1]])
        off = base[None, :] + (np.cumsum(L, axis=0) - L)
        merged[r] = backends[r].merge_runs(n_out, W, off, L, torch.cat(keys_in[r]), torch.cat(cnt_in[r])).astype(np.int64)

    # ---- exchange 2: the merged runs (global distinct edge list) to the segment's owner
    lists2 = {r: [np.nonzero(own2[seg_of[mine1[r]]] == d)[0] for d in range(W)] for r in lr}       # indices into r's merged runs
    keys_c, cnt_c, len_s = exchange_runs(lists2, merged, 1)
    # rank r gets from src the buckets of r's segments that src merged, increasing
    def from_src(r, src):
        a, b = bounds[r]
        g = np.arange(a * nb, b * nb)
        return g[own1[g] == src]
    lens_in, rec_from, keys_in, cnt_in = recv_all(keys_c, cnt_c, len_s, {r: [len(from_src(r, s)) for s in range(W)] for r in lr})
    for r in lr:
        a, b = bounds[r]
        n_out = (b - a) * nb
        L = np.zeros((n_out, W), np.int64)
        off = np.zeros((n_out, W), np.int64)
        base = np.concatenate([[0], np.cumsum(rec_from[r])[:-1]])
        for s in range(W):
            g = from_src(r, s) - a * nb
            l = lens_in[r][s].cpu().numpy()
            L[g, s] = l
            off[g, s] = base[s] + (np.cumsum(l) - l)
        backends[r].merge_runs(n_out, W, off, L, torch.cat(keys_in[r]), torch.cat(cnt_in[r]))
        backends[r].graph(b - a)

    # ---- exchange 3: the reads of a segment to the segment's owner
    if table is not None:
        flen = backends[lr[0]].fixed_len
        nreads_s, words_s = {}, {}
        for r in lr:
            per_seg = backends[r].reads_per_segment()
            w_all = [backends[r].pack_reads(a, b) for (a, b) in bounds]
            nreads_s[r] = [_as_i64(torch, per_seg[a:b], w_all[0]) for (a, b) in bounds]
            words_s[r] = w_all
            sent[r][2] += sum(int(t.numel()) for t in w_all) * 8
        nreads_in = comm.all_to_all(nreads_s, {r: [bounds[r][1] - bounds[r][0]] * W for r in lr})
        nw = lambda n: (n * flen + 31) // 32
        wsize = {r: [int(nw(t.cpu().numpy()).sum()) for t in nreads_in[r]] for r in lr}
        words_in = comm.all_to_all(words_s, wsize)
        for r in lr:
            a, b = bounds[r]
            N = np.stack([t.cpu().numpy() for t in nreads_in[r]], axis=1) if b > a else np.zeros((0, W), np.int64)    # [segment, src]
            Wd = nw(N)
            base = np.concatenate([[0], np.cumsum(wsize[r])[:-1]])
            woff = base[None, :] + (np.cumsum(Wd, axis=0) - Wd)
            seg_ix = np.repeat(np.arange(b - a), W)
            backends[r].set_reads(torch.cat(words_in[r]), seg_ix, N.reshape(-1), woff.reshape(-1))
            backends[r].score(kmer, table)
    if stats is not None:
        stats["bytes_sent"] = sent
    return {r: bounds[r] for r in lr}
