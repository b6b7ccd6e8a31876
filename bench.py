#!/usr/bin/env python3
"""bench.py — k-mers built+scored per second on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one resident batch of synthetic segments:
    reads (2-bit, in HBM) -> k-mers -> distinct k-mers + multiplicities -> (k-1)-mer graph -> contigs
    -> breakage scoring of every contig against its segment's reads.
Workload at N=1 = BASELINE.json configs[2]: 100 x 50 kb segments, 150 bp reads at 50x, k=31, breakage scoring on all
contigs, one MI355X (the largest single-GPU configuration; configs[1] — one 50 kb segment — is the same path at 1/100
of the work and is launch-latency-bound: `--workload cfg1` runs it).  With N GPUs every rank owns its own 100
segments (weak scaling; segments are independent, no data-path collective: SURVEY §8(e) mode 1).

Synthetic input recipe (genomeassembler_dev_amd/synth.py): per segment a 50 000-base ACGT string from
numpy MT19937(seed = 1234 + global segment id) with 20 copies of one 300-bp block, 5 copies of one 2-kb block and one
1-kb tandem repeat of a 6-bp unit planted at seeded offsets; reads = ceil(cov*L/rl) uniform starts from
MT19937(10000019 + seed), starts whose read would run past the end dropped, forward strand, no errors.

Prints ONE JSON line (rank 0).  roofline: the dominant kernel's algorithmic bytes per launch / its mean launch time
from HIP events recorded on the library's own stream during the timed steps.  cpu_baseline: the oracle (a
single-threaded std::string/hash-map restatement of the reference, oracle/) on a bounded sample of the same workload.
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--segments-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import genomeassembler_dev_amd as ga
    from genomeassembler_dev_amd import parallel, qtable, synth

    nseg, L, rl, cov, k = WORKLOADS[args.workload]
    if args.segments_per_gpu:
        nseg = args.segments_per_gpu
    ctx = ga.Context(local_rank)
    table = qtable.load_normalised()
    # rank r owns the contiguous block of global segments parallel.shard_bounds(nseg * world, world)[r]; segment g is
    # generated from seed 1234 + g, so the union over ranks is the same batch whatever the number of GPUs per segment count
    seg_lo, seg_hi = parallel.shard_bounds(nseg * world, world)[rank]
    assert seg_hi - seg_lo == nseg
    reads, seg_off, genomes = synth.make_batch(nseg, L, rl, cov, seed0=1234 + seg_lo, planted=True)
    batch = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl, ctx=ctx)   # upload + 2-bit packing: not timed
    n_reads = int(seg_off[-1])
    n_kmers = n_reads * (rl - k + 1)

    def step():
        batch.build(k, genome_len_hint=L)
        batch.score(8, table)

    def sync_all():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    dominant = ("k_bucket_scatter", "k_bucket_dedup")
    ctx.profile(True, only=dominant)
    ctx.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    dt = parallel.max_over_ranks(dt, device="cuda")

    # ---- roofline of the dominant kernel (HIP events on the library's stream, timed steps only)
    seg, keys, mult, _w = batch.distinct()
    n_distinct = int(seg[-1])
    key_bytes = 8.0 if k <= 31 else 16.0      # W of SURVEY §8(d)
    alg_bytes = {
        # read packed bases 0.25*rl/(rl-k+1) B per k-mer + write the W-byte key to its bucket
        "k_bucket_scatter": n_kmers * (key_bytes + 0.25 * rl / (rl - k + 1)),
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

    # HBM traffic of the dominant kernel from the committed PMC passes of this workload (rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950)
    if roofline and args.workload == "cfg2" and not args.segments_per_gpu:
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic_cfg2_v*.json")),
                       key=lambda f: (os.path.basename(os.path.dirname(f)), int(os.path.basename(f).split("_v")[-1].split(".")[0])))
        if found:
            tp = found[-1]      # the most recent committed pass (tools/profile_round.sh + tools/pmc_traffic.py)
            for name, v in json.load(open(tp)).items():
                if name.startswith(dom):
                    roofline["traffic"] = int(v["FETCH_x2_bytes"] + v["WRITE_SIZE_bytes"])
                    roofline["traffic_source"] = os.path.relpath(tp, ROOT) + " (separate rocprofv3 --pmc passes, FETCH_SIZE x2)"

    breakdown = None
    if args.breakdown:
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            step()
        ctx.sync()
        breakdown = {n: round(v[0] / 3, 4) for n, v in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][0])}
        ctx.profile(False)

    # ---- CPU baseline: the oracle on a bounded sample (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc  # checker / baseline only
        keys_s = qtable.keys()
        nk_tot, t_tot = 0, 0.0
        ns = min(args.cpu_sample_segments, nseg)
        for s in range(ns):
            rs = [r.tobytes().decode() for r in reads[int(seg_off[s]):int(seg_off[s + 1])]]
            nk_s, t_s = orc.time_build_score(rs, k, 8, keys_s, table)
            nk_tot += nk_s
            t_tot += t_s
        cpu = {"value": round(nk_tot / t_tot, 1), "unit": "k-mers/s", "cores": 1, "kind": "port",
               "sample": f"first {ns} of the {nseg} segments of this workload ({nk_tot} k-mers, {t_tot:.1f} s): k-mer "
                         "extraction + contigs + scoring by oracle/gasm_oracle.cpp, single thread as the reference"}

    if rank == 0:
        total_kmers = n_kmers * world
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
                                   "build + breakage scoring of all contigs", "segments_total": nseg * world,
                       "kmers_per_step": total_kmers, "reads_per_step": n_reads * world,
                       "distinct_kmers_rank0": n_distinct, "parallelism": f"segments sharded over {world} GPU(s), no collective"},
            "reads_scored_per_sec": round(n_reads * world * args.steps / dt, 1),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if breakdown:
            out["kernel_ms_per_step"] = breakdown
        print(json.dumps(out), flush=True)
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
