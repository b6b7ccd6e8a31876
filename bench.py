#!/usr/bin/env python3
"""bench.py — k-mers built+scored per second on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one resident batch of synthetic segments:
    reads (2-bit, in HBM) -> k-mers -> distinct k-mers + multiplicities -> (k-1)-mer graph -> contigs
    -> breakage scoring of every contig against its segment's reads.
Workload at N=1 = BASELINE.json configs[2]: 100 x 50 kb segments, 150 bp reads at 50x, k=31, breakage scoring on all
contigs, one MI355X (the largest single-GPU configuration; configs[1] — one 50 kb segment — is the same path at 1/100
of the work and is launch-latency-bound: `--workload cfg1` runs it).

Multi-GPU: `python bench.py --gpus N` with no RANK in the environment starts its own N ranks (python -m
torch.distributed.run, before anything touches a GPU) and relays rank 0's line; under a launcher (RANK set) it is one
rank.  Per GPU the workload is configs[3]'s share — 1000 segments on 8 GPUs = 125 segments per GPU, the same share at N = 2
and 4 (segment g comes from seed 1234 + g whatever N is).  Two partitions of that global batch:
  --mode segments (default, `value`)  every rank owns a contiguous block of segments: weak scaling, no data-path
                 collective (SURVEY §8(e) mode 1 — what the reference's loop over independent segments shards into);
  --mode pooled  every rank holds every N-th read of ALL segments; k-mer records are bucketed by hash of (segment, k-mer
                 prefix) and meet at the bucket's owner through an RCCL all-to-all inside libgasm (gasm_pool_exchange_build:
                 grouped ncclSend / ncclRecv on the library's stream), the merged edge list goes on to the segment's owner,
                 so do the segment's reads (SURVEY §8(e) mode 2).
With N > 1 and the default mode a few pooled steps are timed as well and reported under "pooled" in the same JSON line
(a watchdog prints the line without them and exits with status 3 if an exchange stalls, naming the stage and rank).
torch.distributed is bootstrap only (gloo: barrier, max over ranks, the 128-byte RCCL id); no tensor of the data path is torch's.

Synthetic input recipe (genomeassembler_dev_amd/synth.py): per segment a 50 000-base ACGT string from
numpy MT19937(seed = 1234 + global segment id) with 20 copies of one 300-bp block, 5 copies of one 2-kb block and one
1-kb tandem repeat of a 6-bp unit planted at seeded offsets; reads = ceil(cov*L/rl) uniform starts from
MT19937(10000019 + seed), starts whose read would run past the end dropped, forward strand, no errors.

Prints ONE JSON line (rank 0).  roofline: the dominant kernel's algorithmic bytes per launch / its mean launch time
from HIP events recorded on the library's own streams.  Consecutive steps of the timed region overlap (step slots), which
makes a launch's duration THERE a residence time on a shared chip; the kernel's own duration comes from a short pass with
one step in flight right after the timed region (roofline.measured says so; roofline.timed_region keeps the other figures;
roofline.step is the whole step's bytes over ms_per_step).  cpu_baseline: the oracle (a
std::string/hash-map restatement of the reference, oracle/) on a bounded sample of the same workload — one thread (the
reference is single-threaded) and one oracle PROCESS per host core, all cores at once.  "verified": the GPU results of the
sampled segments (contigs, kmer_breaks bit-exact, bp_score within 1e-9) equal the oracle's; a mismatch fails the run.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (segments per GPU at N = 1, segment length, read length, coverage, k)
    "cfg2": (100, 50000, 150, 50, 31),   # BASELINE.json configs[2]; at N > 1: configs[3], 125 segments per GPU
    "cfg1": (1, 50000, 100, 50, 31),     # BASELINE.json configs[1]
    "cfg0": (1, 50000, 100, 20, 21),     # BASELINE.json configs[0]
    "cfg4": (100, 50000, 250, 100, 51),  # BASELINE.json configs[4] shape per GPU (128-bit keys); --guided adds its traversal
}
SEGMENTS_PER_GPU_MULTI = 125             # configs[3] / configs[4]: 1000 segments on 8 GPUs
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--alone-steps", type=int, default=10, help="untimed extra steps with ONE step in flight, for the kernels' own durations (0 = skip)")
    ap.add_argument("--steps", type=int, default=100)      # (a step is ~1 ms: 20 of them were over before the clocks had settled)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="segments", choices=["segments", "pooled"])
    ap.add_argument("--segments-per-gpu", type=int, default=0)
    ap.add_argument("--bbits", type=int, default=6, help="bucket bits of the pooled mode (all ranks alike)")
    ap.add_argument("--guided", action="store_true", help="cfg4: time the breakage-score-guided traversal with the step (configs[4]'s combined mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pooled", action="store_true", help="N > 1: skip the additional pooled steps")
    ap.add_argument("--cpu-sample-segments", type=int, default=3)
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel time table to stderr")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: N fresh ranks under torch.distributed.run — started before this process has imported
    torch or touched a GPU (a process that has must never be replaced by another) — and this process is only their parent."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd[1:])}", file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env)            # rank 0's JSON line goes straight to our stdout
    return p.returncode


def host_cores():
    """cores this process may use: the affinity mask, cut down to the cgroup's CPU quota where there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    return n


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)          # stdout proper is kept for the JSON line; fd 1 is stderr from here on (all ranks)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # GASM_BENCH_ONE_GPU=1: rehearsal of the multi-rank flow on a one-GPU box (all ranks share GPU 0; RCCL refuses two ranks
    # on one device, so the pooled steps are skipped) — never what the driver runs
    one_gpu = bool(os.environ.get("GASM_BENCH_ONE_GPU"))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")       # bootstrap only: barrier, max over ranks, the RCCL id; the data path's RCCL is libgasm's

    import genomeassembler_dev_amd as ga
    from genomeassembler_dev_amd import parallel, pooled, qtable, synth

    nseg, L, rl, cov, k = WORKLOADS[args.workload]
    if world > 1 and args.workload in ("cfg2", "cfg4"):
        nseg = SEGMENTS_PER_GPU_MULTI
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
        return parallel.max_over_ranks(dt, device="cpu"), prof

    # ---- the pooled partition of the global batch: this rank's every world-th read of every segment; the exchange is libgasm's
    def pooled_setup():
        parts, off = [], [0]
        for g in range(n_global):
            gen = synth.make_segment(1234 + g, L, planted=True)
            r = synth.simulate_reads(gen, rl, cov, 10_000_019 + 1234 + g)[rank::world]
            parts.append(r)
            off.append(off[-1] + r.shape[0])
        rr = np.concatenate(parts, axis=0)
        be = pooled.GasmBackend(rr, np.array(off, dtype=np.uint64), rl, ctx=ctx)
        uid = [pooled.Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        comm = pooled.Comm.rccl(ctx, uid[0], rank, world)       # ncclCommInitRank: also at N = 1, so the one-GPU box runs the RCCL path
        state = {}
        n_km = int(off[-1]) * (rl - k + 1)

        def step():
            state["stats"], state["own"] = pooled.exchange_build(comm, be, k, args.bbits, kmer=8, table=table)
        return be, comm, step, state, n_km, int(off[-1])

    def oracle_check(s_global, contigs_s, breaks_s, bp_s, o):
        if contigs_s != o["contigs"]:
            raise SystemExit(f"bench self-check FAILED: contigs of segment {s_global} differ from the oracle's")
        if list(breaks_s) != o["kmer_breaks"].tolist():
            raise SystemExit(f"bench self-check FAILED: kmer_breaks of segment {s_global} differ from the oracle's")
        if np.abs(np.asarray(bp_s) - o["bp_score"]).max(initial=0.0) >= 1e-9:
            raise SystemExit(f"bench self-check FAILED: bp_score of segment {s_global} differs from the oracle's by >= 1e-9")

    dominant = ("k_bucket_partition", "k_bucket_scatter", "k_bucket_dedup")   # (partition: one pass; scatter: its two-pass form)
    profiled = dominant + ("k_score_reads_graph",)
    reads = seg_off = batch = None
    prof_alone = dt_alone = None
    pcie = None
    guided_info = None
    pooled_main = None
    if args.mode == "segments":
        # rank r owns the contiguous block of global segments parallel.shard_bounds(nseg * world, world)[r]
        reads, seg_off, genomes = synth.make_batch(nseg, L, rl, cov, seed0=1234 + seg_lo, planted=True)
        # the batch goes up 2-bit packed (gasm_batch_create_packed): a quarter of the bytes over PCIe; upload is outside the
        # clock of `value` (inputs resident in HBM) and timed separately for the PCIe-inclusive figure
        words = synth.pack_2bit(reads)
        t_up = time.perf_counter()
        batch = ga.SegmentBatch.from_packed(words, seg_off, fixed_len=rl, ctx=ctx)
        ctx.sync()
        t_up = time.perf_counter() - t_up
        n_reads = int(seg_off[-1])
        n_kmers = n_reads * (rl - k + 1)

        def step():
            batch.build(k, genome_len_hint=L)
            batch.score(8, table)
        # untimed builds whose reports are read before anything is timed: a batch whose buckets outgrow the single-pass
        # partition's regions (or its tables) settles on the configuration that works here, not inside the timed steps (four:
        # consecutive steps of a one-block batch take its step slots in turn, up to four of them)
        for _ in range(4):
            step()
            batch.distinct()
        dt, prof = timed(step, args.steps, args.warmup, profile=profiled)
        seg, keys, mult, _w = batch.distinct()
        n_distinct = int(seg[-1])
        sc_all = batch.scores()
        # Consecutive steps of a one-block batch are in flight together (step slots, capi.hip): inside the timed region a
        # kernel shares the chip with the other steps' kernels and its launch duration says how long it was resident, not
        # what it moves per second when it has the chip.  A short untimed pass with one step in flight (GASM_PINGPONG=0,
        # read by every gasm_batch_build) gives the kernels' own durations; both go into `roofline`.
        prof_alone = dt_alone = None
        if args.alone_steps > 0 and os.environ.get("GASM_PINGPONG", "1") != "0" and nseg > 0:
            keep = os.environ.get("GASM_SCORE_LANE")
            os.environ["GASM_PINGPONG"] = os.environ["GASM_SCORE_LANE"] = "0"       # (nor the last step's scoring beside the partition)
            dt_alone, prof_alone = timed(step, args.alone_steps, 3, profile=profiled)
            del os.environ["GASM_PINGPONG"], os.environ["GASM_SCORE_LANE"]
            if keep is not None:
                os.environ["GASM_SCORE_LANE"] = keep
        # a second upload, timed warm (the first one pays the allocations): what a caller that hands over host buffers sees
        t_up2 = time.perf_counter()
        b2 = ga.SegmentBatch.from_packed(words, seg_off, fixed_len=rl, ctx=ctx)
        ctx.sync()
        t_up2 = time.perf_counter() - t_up2
        b2.close()
        step_s = dt / args.steps
        pcie = {"upload_ms": round(min(t_up, t_up2) * 1e3, 3), "bytes": int(words.nbytes), "what": "2-bit packed reads, pageable host memory",
                "value_incl_upload": round(n_kmers * world / (step_s + min(t_up, t_up2)), 1), "unit": "k-mers/s"}
        # SURVEY §8(d) bytes of the scoring: packed reads + packed paths + 16 B per hit + 32 B of results per path
        score_bytes = (n_reads * rl / 4.0 + float(np.sum(sc_all["sequence_len"])) / 4.0 + 16.0 * float(np.sum(sc_all["kmer_breaks"]))
                       + 32.0 * len(sc_all["kmer_breaks"]))
        if args.guided:
            # configs[4]'s combined mode: build + score + breakage-score-guided traversal (row A16; not in the reference)
            def gstep():
                batch.build(k, genome_len_hint=L)
                batch.score(8, table)
                ga._lib.check(ga._lib.lib().gasm_batch_guided(batch.h))
            gdt, _ = timed(gstep, max(1, args.steps // 5), 2)
            gsteps = max(1, args.steps // 5)
            guided_info = {"ms_per_step": round(gdt / gsteps * 1e3, 4), "value": round(n_kmers * world * gsteps / gdt, 1), "unit": "k-mers/s", "steps": gsteps,
                           "what": "build + score + guided traversal (gasm_batch_guided waits for the scores it steers by: one host round trip per step)"}
    else:
        be, comm, step, pstate, n_kmers, n_reads = pooled_setup()
        dt, prof = timed(step, args.steps, args.warmup, profile=profiled)
        score_bytes = None
        res = be.results(with_scores=True)
        n_distinct = sum(len(d["counts"]) for d in res)
        pooled_main = (be, comm, pstate, res)

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

    if roofline and args.mode == "segments" and prof_alone and any(prof_alone.get(n, (0.0, 0))[1] for n in dominant):
        # Steps overlapped in the timed region: what the HIP events bracket there is how long a kernel was RESIDENT on a chip
        # it shared with the other steps' kernels — not a throughput of the kernel, and not what rocprofv3 reports for the same
        # command either (the tracer spaces the dispatches out).  The roofline of the kernel is taken from the pass with one
        # step in flight (same process, same batch, right after the timed region); the timed region's figures stay beside it.
        a_ms = {n: prof_alone[n][0] / prof_alone[n][1] for n in dominant if n in prof_alone and prof_alone[n][1]}
        adom = max(a_ms, key=a_ms.get)
        timed_region = {k2: roofline[k2] for k2 in ("kernel", "achieved", "frac", "avg_launch_ms", "other")}
        timed_region["what"] = ("HIP events over the K timed steps; consecutive steps overlap (step slots), so these are residence times on a "
                                "chip shared with the other steps' kernels")
        ach = alg_bytes[adom] / (a_ms[adom] / 1e3) / 1e9
        dom = adom
        roofline.update({
            "kernel": adom, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(a_ms[adom], 4),
            "algorithmic_bytes_per_launch": int(alg_bytes[adom]),
            "other": {n: {"avg_launch_ms": round(v, 4), "achieved_GBs": round(alg_bytes[n] / (v / 1e3) / 1e9, 1)} for n, v in a_ms.items() if n != adom},
            "measured": (f"HIP events on the library's streams over {args.alone_steps} steps with ONE step in flight and every kernel alone on the chip "
                         "(GASM_PINGPONG=0, GASM_SCORE_LANE=0), in this process right after the timed region"),
            "one_step_in_flight_ms_per_step": round(dt_alone / args.alone_steps * 1e3, 4),
            "timed_region": timed_region})
    if roofline and score_bytes and (prof_alone or prof).get("k_score_reads_graph", (0.0, 0))[1]:
        # (low by construction: the scorer's work is index look-ups and compares, not bytes)
        sp = (prof_alone or prof)["k_score_reads_graph"]
        sms = sp[0] / sp[1]
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
    if roofline and args.workload == "cfg2" and args.mode == "segments" and not args.segments_per_gpu and world == 1:
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

    def pooled_report(pdt, psteps, p_kmers, stats):
        sent, rem = stats["bytes_sent"], stats["bytes_sent_remote"]
        links = max(1, world - 1)
        return {"ms_per_step": round(pdt / psteps * 1e3, 4), "value": round(p_kmers * world * psteps / pdt, 1), "unit": "k-mers/s",
                "steps": psteps, "bbits": stats["bbits"], "attempts_last_step": stats["attempts"],
                "bytes_sent_rank0_per_step": {"records": sent[0], "merged_records": sent[1], "reads": sent[2]},
                "bytes_per_xgmi_link_rank0_per_step": {"records": rem[0] // links, "merged_records": rem[1] // links, "reads": rem[2] // links},
                "host_waits_per_step": 2,
                "note": "hash-bucket all-to-all inside libgasm (grouped ncclSend/ncclRecv over RCCL) + global merge + reads to the segment's owner; "
                        "mode 1 (value) moves nothing"}

    # ---- additional pooled steps at N > 1 (default mode), guarded: a stalled exchange must not cost the main line
    pooled_info = None
    if world > 1 and args.mode == "segments" and not args.no_pooled and not one_gpu:
        psteps = max(1, min(5, args.steps))
        # watchdog thread (a signal handler would not run while the main thread sits in a blocking runtime call): after
        # 180 s rank 0 prints the line without the pooled figures and every rank leaves with status 3
        import threading
        pending = {"comm": None}

        def give_up():
            st = pending["comm"].stage() if pending["comm"] is not None else -1
            print(f"[bench] rank {rank}: the pooled steps did not finish within 180 s (libgasm exchange stage {st}: 10 local runs, 11-13 exchange 1, "
                  "21-23 exchange 2, 31 reads, 32 scoring, -1 set-up)", file=sys.stderr, flush=True)
            if rank == 0:
                _print_line(args, world, nseg, L, rl, cov, k, n_kmers, n_reads, n_distinct, dt, roofline, None, breakdown,
                            {"error": f"the pooled steps did not finish within 180 s (rank 0 at exchange stage {st})"}, None, pcie, guided_info)
            os._exit(3)
        dog = threading.Timer(180.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            be, comm, pstep, pstate, p_kmers, _ = pooled_setup()
            pending["comm"] = comm
            pdt, _ = timed(pstep, psteps, 1)
            pooled_info = pooled_report(pdt, psteps, p_kmers, pstate["stats"])
            be.close()
            comm.close()
        except Exception as e:  # noqa: BLE001
            pooled_info = {"error": repr(e)[:300]}
        finally:
            dog.cancel()

    # ---- CPU baseline + self-check: the oracle on a bounded sample (rank 0, N=1 only)
    cpu, verified = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc  # checker / baseline only
        keys_s = qtable.keys()
        if args.mode == "segments":
            contigs, sc = batch.contigs(), batch.scores()

            def gpu_segment(s):
                a, e = int(sc["seg_contig_off"][s]), int(sc["seg_contig_off"][s + 1])
                return contigs[s], sc["kmer_breaks"][a:e], sc["bp_score"][a:e]
        else:
            res = pooled_main[3]

            def gpu_segment(s):
                return res[s]["contigs"], res[s]["kmer_breaks"], res[s]["bp_score"]

        def seg_reads(s):
            gen = synth.make_segment(1234 + s, L, planted=True)
            return [r.tobytes().decode() for r in synth.simulate_reads(gen, rl, cov, 10_000_019 + 1234 + s)]

        # single thread, as the reference runs
        ns = min(args.cpu_sample_segments, nseg)
        nk1, t1 = 0, 0.0
        for s in range(ns):
            o = orc.build_score(seg_reads(s), k, 8, keys_s, table)
            oracle_check(s, *gpu_segment(s), o)
            nk1 += o["n_kmers"]
            t1 += o["seconds"]
        # all host cores: one oracle PROCESS per core, all at once, each on a segment of its own (tools/cpu_oracle_worker.py:
        # children that never load libgasm; one process with a thread per core stops scaling at its allocator).  Segments
        # beyond the batch (seed 1234 + g, g >= nseg) are timed only; those inside are also checked against the GPU's results
        cores = host_cores()
        nwork = max(1, min(cores, 256))
        tmp = tempfile.mkdtemp(prefix="gasm_cpu_")
        picks = [nseg - 1 - i if i < nseg else i for i in range(nwork)]       # from the far end of the batch first: other segments than above
        procs = []
        t_spawn = time.time()
        for i, g in enumerate(picks):
            out = os.path.join(tmp, f"w{i}.npz")
            procs.append((g, out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_oracle_worker.py"), str(g), str(L), str(rl), str(cov), str(k), out],
                                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        t0s, t1s, nk_all, checked = [], [], 0, set(range(ns))
        for g, out, p in procs:
            _, err = p.communicate()
            if p.returncode != 0:
                raise SystemExit(f"cpu_baseline worker for segment {g} failed: {err.decode(errors='replace')[-500:]}")
            z = np.load(out)
            t0s.append(float(z["t0"])); t1s.append(float(z["t1"])); nk_all += int(z["n_kmers"])
            if g < nseg:
                txt = str(z["contigs"])
                o = {"contigs": txt.split("\n") if txt else [], "kmer_breaks": z["kmer_breaks"], "bp_score": z["bp_score"]}
                oracle_check(g, *gpu_segment(g), o)
                checked.add(g)
            os.remove(out)
        os.rmdir(tmp)
        tall = max(t1s) - min(t0s)          # first oracle call started .. last one finished (process start-up and input synthesis not counted)
        verified = sorted(checked)
        cpu = {"value": round(nk_all / tall, 1), "unit": "k-mers/s", "cores": nwork, "kind": "port",
               "single_thread_value": round(nk1 / t1, 1),
               "sample": f"oracle/gasm_oracle.cpp (k-mer extraction + contigs + scoring): {nwork} processes at once, one segment each, on the "
                         f"{cores} host cores this process may use ({nk_all} k-mers, {tall:.1f} s from the first oracle call to the last return, "
                         f"{time.time() - t_spawn:.1f} s with process start-up); single thread (as the reference runs): first {ns} segments "
                         f"({nk1} k-mers, {t1:.1f} s)"}

    if args.mode == "pooled" and rank == 0:
        pooled_info = pooled_report(dt, args.steps, n_kmers, pooled_main[2]["stats"])
    if rank == 0:
        _print_line(args, world, nseg, L, rl, cov, k, n_kmers, n_reads, n_distinct, dt, roofline, cpu, breakdown, pooled_info, verified, pcie, guided_info)
    if batch is not None:
        batch.close()
    if pooled_main is not None:
        pooled_main[0].close()
        pooled_main[1].close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _print_line(args, world, nseg, L, rl, cov, k, n_kmers, n_reads, n_distinct, dt, roofline, cpu, breakdown, pooled_info, verified, pcie, guided_info):
    total_kmers = n_kmers * world
    par = (f"segments sharded over {world} GPU(s), no collective" if args.mode == "segments" else
           f"reads of all segments dealt over {world} GPU(s); hash-bucket all-to-all (RCCL, inside libgasm) + global merge, bbits={args.bbits}")
    cfg_name = {"cfg2": "configs[2]" if world == 1 else "configs[3] share", "cfg4": "configs[4] share", "cfg1": "configs[1]", "cfg0": "configs[0] shape"}[args.workload]
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
        "config": {"workload": f"{args.workload} ({cfg_name}): {nseg} x {L} bp segments per GPU, {rl} bp reads at {cov}x, k={k}, "
                               "build + breakage scoring of all contigs", "mode": args.mode, "segments_total": nseg * world,
                   "kmers_per_step": total_kmers, "reads_per_step": n_reads * world,
                   "distinct_kmers_rank0": n_distinct, "parallelism": par},
        "reads_scored_per_sec": round(n_reads * world * args.steps / dt, 1),
        "roofline": roofline,
        "cpu_baseline": cpu,
        "verified": (None if verified is None else {"ok": True, "segments": verified,
                                                    "what": "contigs + kmer_breaks bit-exact, bp_score < 1e-9 vs oracle"}),
    }
    if pcie is not None:
        out["pcie_inclusive"] = pcie
    if guided_info is not None:
        out["guided"] = guided_info
    if pooled_info is not None:
        out["pooled"] = pooled_info
    if breakdown:
        out["kernel_ms_per_step"] = breakdown
    # the ONE line goes to the stdout this process was started with; everything else that writes to file descriptor 1
    # (Gloo's "connected to N peer ranks", RCCL's version banner, whatever a library prints) has been sent to stderr
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (json.dumps(out) + "\n").encode())


_REAL_STDOUT = None


if __name__ == "__main__":
    main()
