"""The k-meric breakage probability table.

Data: data/querytable_raw_f64.bin = the reference's data/QueryTable/QueryTable_kmer-{2,4,6,8}.csv `prob` columns, raw,
in file order (tools/make_querytable.py); rows of each table are in lexicographic ACGT order, so no key column is kept.
`load_normalised()` applies what lib/GenerateReads.R:153-184 (get_prob_values) does: NA -> minimum of its own table,
then every value divided by the sum over all 69 904 rows (R's sum() accumulates in long double)."""
import itertools
import os

import numpy as np

SIZES = (16, 256, 4096, 65536)
ROWS = sum(SIZES)
_RAW = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "querytable_raw_f64.bin")


def load_raw(path=_RAW):
    a = np.fromfile(path, dtype="<f8")
    if a.size != ROWS:
        raise ValueError(f"{path}: expected {ROWS} doubles, found {a.size}")
    return a


def load_csv_dir(dirname):
    """raw values from a directory holding QueryTable_kmer-{2,4,6,8}.csv (header kmer,prob)"""
    out = []
    for k in (2, 4, 6, 8):
        with open(os.path.join(dirname, f"QueryTable_kmer-{k}.csv")) as f:
            rows = [l.rstrip("\r\n").split(",") for l in f][1:]
        exp = ["".join(t) for t in itertools.product("ACGT", repeat=k)]
        if [r[0].strip('"') for r in rows] != exp:
            raise ValueError(f"QueryTable_kmer-{k}.csv is not in lexicographic ACGT order")
        out.extend(float("nan") if r[1] in ("NA", "") else float(r[1]) for r in rows)
    return np.array(out, dtype=np.float64)


def normalise(raw):
    p = np.array(raw, dtype=np.float64)
    off = 0
    for n in SIZES:
        t = p[off:off + n]
        bad = np.isnan(t)
        if bad.any():
            t[bad] = np.nanmin(t)
        off += n
    total = np.float64(np.sum(p.astype(np.longdouble)))
    return p / total


def load_normalised():
    return normalise(load_raw())


def uniform():
    """the reference's 'random' control: 1/69904 for every row (lib/DeNovoAssembler.R:326-330)"""
    return np.full(ROWS, 1.0 / ROWS)


def keys():
    return ["".join(t) for k in (2, 4, 6, 8) for t in itertools.product("ACGT", repeat=k)]
