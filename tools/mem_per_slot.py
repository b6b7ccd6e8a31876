"""Device memory a cfg2 batch takes with 1, 2, 3 step slots (hipMemGetInfo around the first steps).  usage: python tools/mem_per_slot.py"""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import genomeassembler_dev_amd as ga
    from genomeassembler_dev_amd import qtable, synth
    reads, seg_off, _g = synth.make_batch(100, 50000, 150, 50, seed0=1234, planted=True)
    ctx = ga.default_context()
    free0 = torch.cuda.mem_get_info()[0]
    b = ga.SegmentBatch.from_packed(synth.pack_2bit(reads), seg_off, fixed_len=150, ctx=ctx)
    table = qtable.load_normalised()
    for _ in range(6):
        b.build(31, genome_len_hint=50000).score(8, table)
    b.distinct()
    ctx.sync()
    print(f"slots={sys.argv[1]} pingpong={os.environ.get('GASM_PINGPONG', '1')}: {(free0 - torch.cuda.mem_get_info()[0]) / 2**30:.2f} GiB")
else:
    for pp, n in (("0", "1"), ("1", "2"), ("1", "3")):
        env = dict(os.environ, GASM_PINGPONG=pp, GASM_STEP_SLOTS=n)
        subprocess.run([sys.executable, __file__, n], env=env, check=True)
