"""Post-mortem of a guided-traversal case kept by tools/soak.py (gpurun_out/soak_guided_case.npz).  usage: python tools/guided_case.py [exercise]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomeassembler_dev_amd as ga  # noqa: E402
from genomeassembler_dev_amd import qtable  # noqa: E402
from oracle import guided_oracle  # noqa: E402

d = np.load(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".npz") else "gpurun_out/soak_guided_case.npz")
reads, k = d["reads"], int(d["k"])
want = [str(x) for x in d["want"]]
keys, prob = qtable.keys(), qtable.load_normalised()
rl = reads.shape[1]
seg_off = np.array([0, reads.shape[0]], dtype=np.uint64)
os.environ["GASM_DBG_GUIDED"] = "gpurun_out/guided_dump.bin"
os.makedirs("gpurun_out", exist_ok=True)
for mode in ("plain",):
    b = ga.SegmentBatch(reads.reshape(-1), seg_off, fixed_len=rl)
    if mode == "another k first":
        b.build(9).score(8, prob)
        b.contigs()
    if mode == "three steps first":
        for _ in range(3):
            b.build(k).score(8, prob)
    b.build(k).score(8, prob)
    contigs = b.contigs()[0]
    fx, shift = b.score_fixed()
    gd = b.guided()
    got = [x["sequence"] for x in gd[0]]
    mine = guided_oracle.guided_paths(contigs, [int(v) for v in fx], k)
    print(f"{mode:20s}: contigs {len(contigs)}, shift {shift}, paths {len(got)}; equals the restatement: {got == mine}; equals the kept expectation: {got == want}; "
          f"longest {max(map(len, got))} / {max(map(len, mine))}")
    b.close()
