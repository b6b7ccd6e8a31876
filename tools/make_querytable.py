"""Converts the reference's breakage-probability tables (data, not code:
/root/reference/data/QueryTable/QueryTable_kmer-{2,4,6,8}.csv, header `kmer,prob`) into the flat binary the package
ships: genomeassembler_dev_amd/data/querytable_raw_f64.bin = 16+256+4096+65536 raw (un-normalised) little-endian
doubles in file order.  Rows are verified to be in lexicographic ACGT order, so the row index of a k-mer is its base-4
value and no key column needs to be stored.  'NA' becomes NaN (lib/GenerateReads.R:153-184 replaces it later).
Run once in the build container; the reference tree does not travel to the GPU box."""
import itertools
import os
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/QueryTable"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "genomeassembler_dev_amd", "data",
                   "querytable_raw_f64.bin")
vals = []
for k in (2, 4, 6, 8):
    with open(os.path.join(src, f"QueryTable_kmer-{k}.csv")) as f:
        rows = [l.rstrip("\r\n").split(",") for l in f]
    assert rows[0] == ["kmer", "prob"], rows[0]
    rows = rows[1:]
    assert len(rows) == 4 ** k
    for row, key in zip(rows, itertools.product("ACGT", repeat=k)):
        assert row[0].strip('"') == "".join(key), (row, key)
        vals.append(float("nan") if row[1] in ("NA", "") else float(row[1]))
a = np.array(vals, dtype="<f8")
assert a.size == 69904
a.tofile(dst)
print("rows", a.size, "nan", int(np.isnan(a).sum()), "sum", repr(float(a.sum())), "->", os.path.normpath(dst))
