#!/usr/bin/env python3
"""bench.py — k-mers built+scored per second on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one resident batch of synthetic segments:
    reads (2-bit, in HBM) -> k-mers -> distinct k-mers + multiplicities -> (k-1)-mer graph -> contigs
    -> breakage scoring of every contig against its segment's reads.
Workload at N=1 = BASELINE.json configs[2]: 100 x 50 kb segments, 150 bp reads at 50x, k=31, breakage scoring on all
contigs, one MI355X (the largest single-GPU configuration; configs[1] — one 50 kb segment — is the same path at 1/100
of the work and is launch-latency-bound: `--workload cfg1` runs it).

Multi-GPU (`--gpus N`, one process per GPU under torch.distributed.run): two partitions of the same global batch of
100 x N segments (segment g comes from seed 1234 + g whatever N is):
  --mode segments (default, `value`)  every rank owns a contiguous block of 100 segments: weak scaling, no data-path
                 collective (SURVEY §8(e) mode 1 — what the reference's loop over independent segments shards into);
  --mode pooled  every rank holds every N-th read of ALL segments; k-mer records are bucketed by hash of (segment, k-mer
                 prefix) and meet at the bucket's owner through an RCCL all-to-all, the merged edge list goes on to the
                 segment's owner, so do the segment's reads (SURVEY §8(e) mode 2, genomeassembler_dev_amd/pooled.py).
With N > 1 and the default mode a few pooled steps are timed as well and reported under "pooled" in the same JSON line
(guarded by an alarm: if the exchange stalls, the line is printed without them).

Synthetic input recipe (genomeassembler_dev_amd/synth.py): per segment a 50 000-base ACGT string from
numpy MT19937(seed = 1234 + global segment id) with 20 copies of one 300-bp block, 5 copies of one 2-kb block and one
1-kb tandem repeat of a 6-bp unit planted at seeded offsets; reads = ceil(cov*L/rl) uniform starts from
MT19937(10000019 + seed), starts whose read would run past the end dropped, forward strand, no errors.

Prints ONE JSON line (rank 0).  roofline: the dominant kernel's algorithmic bytes per launch / its mean launch time
from HIP events recorded on the library's own stream during the timed steps.  cpu_baseline: the oracle (a
std::string/hash-map restatement of the reference, oracle/) on a bounded sample of the same workload — one thread (the
reference is single-threaded) and segment-parallel on all host cores.  "verified": the GPU results of the sampled
segments (contigs, kmer_breaks bit-exact, bp_score within 1e-9) equal the oracle's; a mismatch fails the run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (segments per GPU, segment length, read length, coverage, k)
    "cfg2": (100, 50000, 150, 50, 31),   # BASELINE.json configs[2]
    "cfg1": (1, 50000, 100, 50, 31),     # BASELINE.json configs[1]
    "cfg0": (1, 50000, 100, 20, 21),     # BASELINE.json configs[0]
    "cfg4": (100, 50000, 250, 100, 51),  # BASELINE.json configs[4] shape per GPU (128-bit keys; its "guided traversal" has no reference)
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # (a step is ~1.2 ms: 20 of them were over before the clocks had settled)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="segments", choices=["segments", "pooled"])
    ap.add_argument("--segments-per-gpu", type=int, default=0)
    ap.add_argument("--bbits", type=int, default=6, help="bucket bits of the pooled mode (all ranks alike)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pooled", action="store_true", help="N > 1: skip the additional pooled steps")
    ap.add_argument("--cpu-sample-segments", type=int, default=3)
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel time table to stderr")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # GASM_BENCH_BACKEND=gloo + GASM_BENCH_ONE_GPU=1: rehearsal of the multi-rank flow on a one-GPU box (all ranks share
    # GPU 0, exchanges staged through the host) — never what the driver runs
    backend = os.environ.get("GASM_BENCH_BACKEND", "nccl")
    if os.environ.get("GASM_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import genomeassembler_dev_amd as ga
    from genomeassembler_dev_amd import parallel, pooled, qtable, synth

    nseg, L, rl, cov, k = WORKLOADS[args.workload]
    if args.segments_per_gpu:
        nseg = args.segments_per_gpu
    ctx = ga.Context(local_rank)
    table = qtable.load_normalised()
    n_global = nseg * world
    seg_lo, seg_hi = parallel.shard_bounds(n_global, world)[rank]
    assert seg_hi - seg_lo == nseg

    def sync_all():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(step, steps, warmup, profile=None):
        for _ in range(warmup):
            step()
        sync_all()
        if profile:
            ctx.profile(True, only=profile)
            ctx.profile_reset()
            sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync_all()
        dt = time.perf_counter() - t0
        prof = None
        if profile:
            prof = ctx.profile_read()
            ctx.profile(False)
        return parallel.max_over_ranks(dt, device="cuda" if backend == "nccl" else "cpu"), prof

    # ---- the pooled partition of the global batch: this rank's every world-th read of every segment
    def pooled_setup():
        parts, off = [], [0]
        for g in range(n_global):
            gen = synth.make_segment(1234 + g, L, planted=True)
            r = synth.simulate_reads(gen, rl, cov, 10_000_019 + 1234 + g)[rank::world]
            parts.append(r)
            off.append(off[-1] + r.shape[0])
        rr = np.concatenate(parts, axis=0)
        be = pooled.GasmBackend(rr, np.array(off, dtype=np.uint64), rl, ctx=ctx)
        comm = pooled.DistComm() if world > 1 else pooled.VirtualComm(1)
        stats = {}
        n_km = int(off[-1]) * (rl - k + 1)

        def step():
            pooled.pooled_build(comm, {rank: be}, n_global, k, args.bbits, kmer=8, table=table, stats=stats)
        return be, step, stats, n_km, int(off[-1])

    dominant = ("k_bucket_partition", "k_bucket_scatter", "k_bucket_dedup")   # (partition: one pass; scatter: its two-pass form)
    profiled = dominant + ("k_score_reads_graph",)
    reads = seg_off = batch = None
    if args.mode == "segments":
        # rank r owns the contiguous block of global segments parallel.shard_bounds(nseg * world, world)[r]
        reads, seg_off, genomes = synth.make_batch(nseg, L, rl, cov, seed0=1234 + seg_lo, planted=True)
        batch = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl, ctx=ctx)   # upload + 2-bit packing: not timed
        n_reads = int(seg_off[-1])
        n_kmers = n_reads * (rl - k + 1)

        def step():
            batch.build(k, genome_len_hint=L)
            batch.score(8, table)
        # one untimed build whose report is read before anything is timed: a batch whose buckets outgrow the single-pass
        # partition's regions (or its tables) settles on the configuration that works here, not inside the timed steps
        step()
        batch.distinct()
        dt, prof = timed(step, args.steps, args.warmup, profile=profiled)
        seg, keys, mult, _w = batch.distinct()
        n_distinct = int(seg[-1])
        sc_all = batch.scores()
        # SURVEY §8(d) bytes of the scoring: packed reads + packed paths + 16 B per hit + 32 B of results per path
        score_bytes = (n_reads * rl / 4.0 + float(np.sum(sc_all["sequence_len"])) / 4.0 + 16.0 * float(np.sum(sc_all["kmer_breaks"]))
                       + 32.0 * len(sc_all["kmer_breaks"]))
    else:
        be, step, pstats, n_kmers, n_reads = pooled_setup()
        dt, prof = timed(step, args.steps, args.warmup, profile=profiled)
        score_bytes = None
        res = be.results(with_scores=False)
        n_distinct = sum(len(d["counts"]) for d in res)

    # ---- roofline of the dominant kernel (HIP events on the library's stream, timed steps only)
    key_bytes = 8.0 if k <= 31 else 16.0      # W of SURVEY §8(d)
    alg_bytes = {
        # read packed bases 0.25*rl/(rl-k+1) B per k-mer + write the W-byte key to its bucket
        "k_bucket_scatter": n_kmers * (key_bytes + 0.25 * rl / (rl - k + 1)),
        "k_bucket_partition": n_kmers * (key_bytes + 0.25 * rl / (rl - k + 1)),
        # read the key back (W bytes per k-mer) + write (key, multiplicity) per distinct k-mer
        "k_bucket_dedup": n_kmers * key_bytes + n_distinct * (key_bytes + 4.0),
    }
    dom = max(dominant, key=lambda n: prof.get(n, (0.0, 0))[0])
    ms, launches = prof.get(dom, (0.0, 0))
    roofline = None
    if launches:
        avg_s = ms / launches / 1e3
        ach = alg_bytes[dom] / avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(ms / launches, 4),
                    "algorithmic_bytes_per_launch": int(alg_bytes[dom]),
                    "other": {n: {"avg_launch_ms": round(prof[n][0] / prof[n][1], 4),
                                  "achieved_GBs": round(alg_bytes[n] / (prof[n][0] / prof[n][1] / 1e3) / 1e9, 1)}
                              for n in dominant if n != dom and n in prof and prof[n][1]}}

    if roofline and score_bytes and prof.get("k_score_reads_graph", (0.0, 0))[1]:
        # (low by construction: the scorer's work is index look-ups and compares, not bytes — see the compares/s figure)
        sms = prof["k_score_reads_graph"][0] / prof["k_score_reads_graph"][1]
        roofline["other"]["k_score_reads_graph"] = {"avg_launch_ms": round(sms, 4), "algorithmic_bytes_per_launch": int(score_bytes),
                                                    "achieved_GBs": round(score_bytes / (sms / 1e3) / 1e9, 1)}
    # the whole step against the same peak: SURVEY §8(d)'s per-unit bytes of every stage (k-mer pass: packed bases in, key
    # out, key in again = 2W + 0.25*rl/(rl-k+1) per k-mer; graph + contigs + scoring: 33 B per distinct k-mer; L bases per segment)
    if roofline and args.mode == "segments":
        step_bytes = n_kmers * (2 * key_bytes + 0.25 * rl / (rl - k + 1)) + n_distinct * 33.0 + float(nseg) * L    # this rank's
        step_gbs = step_bytes / (dt / args.steps) / 1e9
        roofline["step"] = {"algorithmic_bytes": int(step_bytes), "achieved": round(step_gbs, 1), "frac": round(step_gbs / HBM_PEAK_GBS, 4)}

    # HBM traffic of the dominant kernel from the committed PMC passes of this workload (rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950).  The file
    # names the commit it was measured at: a later change of the kernel makes the figure stale, which the line says.
    if roofline and args.workload == "cfg2" and args.mode == "segments" and not args.segments_per_gpu:
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic_cfg2_v*.json")),
                       key=lambda f: (os.path.basename(os.path.dirname(f)), int(os.path.basename(f).split("_v")[-1].split(".")[0])))
        if found:
            tp = found[-1]      # the most recent committed pass (tools/profile_round.sh + tools/pmc_traffic.py)
            tj = json.load(open(tp))
            for name, v in tj.items():
                if name.startswith(dom) and isinstance(v, dict):
                    roofline["traffic"] = int(v["FETCH_x2_bytes"] + v["WRITE_SIZE_bytes"])
                    roofline["traffic_source"] = os.path.relpath(tp, ROOT) + " (separate rocprofv3 --pmc passes, FETCH_SIZE x2)"
                    if "_measured_at" in tj:
                        roofline["traffic_measured_at"] = tj["_measured_at"]

    breakdown = None
    if args.breakdown:
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            step()
        ctx.sync()
        breakdown = {n: round(v[0] / 3, 4) for n, v in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][0])}
        ctx.profile(False)

    # ---- additional pooled steps at N > 1 (default mode), guarded: a stalled exchange must not cost the main line
    pooled_info = None
    if world > 1 and args.mode == "segments" and not args.no_pooled:
        psteps = max(1, min(5, args.steps))
        # watchdog thread (a signal handler would not run while the main thread sits in a blocking runtime call): after
        # 240 s rank 0 prints the line without the pooled figures and every rank leaves
        import threading

        def give_up():
            if rank == 0:
                _print_line(args, world, nseg, L, rl, cov, k, n_kmers, n_reads, n_distinct, dt, roofline, None, breakdown,
                            {"error": "the pooled steps did not finish within 240 s"}, None)
            os._exit(0)
        dog = threading.Timer(240.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            be, pstep, pstats, p_kmers, _ = pooled_setup()
            pdt, _ = timed(pstep, psteps, 1)
            sent = pstats["bytes_sent"][rank]
            pooled_info = {"ms_per_step": round(pdt / psteps * 1e3, 4), "value": round(p_kmers * world * psteps / pdt, 1), "unit": "k-mers/s",
                           "steps": psteps, "bbits": args.bbits,
                           "bytes_sent_rank0_per_step": {"records": sent[0], "merged_records": sent[1], "reads": sent[2]},
                           "note": "hash-bucket all-to-all over RCCL + global merge + reads to the segment's owner; mode 1 (value) moves nothing"}
            be.close()
        except Exception as e:  # noqa: BLE001
            pooled_info = {"error": repr(e)[:300]}
        finally:
            dog.cancel()

    # ---- CPU baseline + self-check: the oracle on a bounded sample (rank 0, N=1 only)
    cpu, verified = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "segments":
        from concurrent.futures import ThreadPoolExecutor

        from oracle import orc  # checker / baseline only
        keys_s = qtable.keys()
        contigs, sc = batch.contigs(), batch.scores()

        def seg_reads(s):
            return [r.tobytes().decode() for r in reads[int(seg_off[s]):int(seg_off[s + 1])]]

        def check(s, o):
            a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
            if contigs[s] != o["contigs"]:
                raise SystemExit(f"bench self-check FAILED: contigs of segment {s} differ from the oracle's")
            if sc["kmer_breaks"][a:e].tolist() != o["kmer_breaks"].tolist():
                raise SystemExit(f"bench self-check FAILED: kmer_breaks of segment {s} differ from the oracle's")
            if np.abs(sc["bp_score"][a:e] - o["bp_score"]).max(initial=0.0) >= 1e-9:
                raise SystemExit(f"bench self-check FAILED: bp_score of segment {s} differs from the oracle's by >= 1e-9")

        ns = min(args.cpu_sample_segments, nseg)
        nk1, t1 = 0, 0.0
        for s in range(ns):
            o = orc.build_score(seg_reads(s), k, 8, keys_s, table)
            check(s, o)
            nk1 += o["n_kmers"]
            t1 += o["seconds"]
        # all host cores: the segments are independent, one oracle call per core at a time (ctypes releases the GIL)
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:  # noqa: BLE001
            pass
        host_cores = cores
        cores = min(cores, 32)      # (one process: the oracle's string allocations stop scaling long before a 256-core host is full)
        picks = [nseg - 1 - i for i in range(min(cores, nseg))]       # from the far end of the batch: other segments than above
        inputs = [seg_reads(s) for s in picks]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            outs = list(ex.map(lambda rs: orc.build_score(rs, k, 8, keys_s, table), inputs))
        tall = time.perf_counter() - t0
        for s, o in zip(picks, outs):
            check(s, o)
        nk_all = sum(o["n_kmers"] for o in outs)
        verified = sorted(set(range(ns)) | set(picks))
        cpu = {"value": round(nk_all / tall, 1), "unit": "k-mers/s", "cores": len(picks), "kind": "port",
               "single_thread_value": round(nk1 / t1, 1),
               "sample": f"oracle/gasm_oracle.cpp (k-mer extraction + contigs + scoring): {len(picks)} segments at once on {len(picks)} of "
                         f"{host_cores} host cores ({nk_all} k-mers, {tall:.1f} s wall); single thread (as the reference runs): first {ns} segments "
                         f"({nk1} k-mers, {t1:.1f} s)"}

    if rank == 0:
        _print_line(args, world, nseg, L, rl, cov, k, n_kmers, n_reads, n_distinct, dt, roofline, cpu, breakdown, pooled_info, verified)
    if batch is not None:
        batch.close()
    if world > 1:
        dist.destroy_process_group()


def _print_line(args, world, nseg, L, rl, cov, k, n_kmers, n_reads, n_distinct, dt, roofline, cpu, breakdown, pooled_info, verified):
    total_kmers = n_kmers * world
    par = (f"segments sharded over {world} GPU(s), no collective" if args.mode == "segments" else
           f"reads of all segments dealt over {world} GPU(s); hash-bucket all-to-all (RCCL) + global merge, bbits={args.bbits}")
    out = {
        "metric": "k-mers built+scored/sec",
        "value": round(total_kmers * args.steps / dt, 1),
        "unit": "k-mers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("u64" if k <= 31 else "u128") + " keys / u32 counts / f64 scores",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {nseg} x {L} bp segments per GPU, {rl} bp reads at {cov}x, k={k}, "
                               "build + breakage scoring of all contigs", "mode": args.mode, "segments_total": nseg * world,
                   "kmers_per_step": total_kmers, "reads_per_step": n_reads * world,
                   "distinct_kmers_rank0": n_distinct, "parallelism": par},
        "reads_scored_per_sec": round(n_reads * world * args.steps / dt, 1),
        # what a brute-force scorer would do for the same result — every read against every base position of
        # its segment's contigs (~ distinct k-mers per segment); the graph-indexed scorer does one lookup + one compare per read
        "read_position_compares_equiv_per_sec": round(float(n_reads) * (n_distinct / max(nseg, 1)) * world * args.steps / dt, 1),
        "roofline": roofline,
        "cpu_baseline": cpu,
        "verified": (None if verified is None else {"ok": True, "segments": verified,
                                                    "what": "contigs + kmer_breaks bit-exact, bp_score < 1e-9 vs oracle"}),
    }
    if pooled_info is not None:
        out["pooled"] = pooled_info
    if breakdown:
        out["kernel_ms_per_step"] = breakdown
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
