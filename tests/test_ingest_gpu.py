"""Row F3, the device half: FASTQ / FASTA parsed and 2-bit packed on the GPU (csrc/ingest.hip; the host only inflates) against
the host reader (csrc/seqio.cpp) word for word, and against oracle/seq_oracle.py read by read.  The reference has no reader on
this path (it writes FASTA itself, lib/GenerateReads.R:405-433): the grammar is this project's, parity = the three agree."""
import gzip
import random

import numpy as np
import pytest

import genomeassembler_dev_amd as ga
from genomeassembler_dev_amd import seqio
from oracle import seq_oracle

pytestmark = pytest.mark.gpu


def _same(paths, expect_device, non_acgt="drop", oracle=True):
    w_h, o_h, s_h, d_h = seqio.read_files(paths, non_acgt)
    w_d, o_d, s_d, d_d, on = seqio.read_files_device(paths, non_acgt)
    assert on == expect_device, (paths, on)
    assert o_d.tolist() == o_h.tolist() and s_d.tolist() == s_h.tolist() and d_d == d_h
    assert np.array_equal(w_d, w_h)
    if not oracle:
        return seqio.unpack_reads(w_d, o_d)
    reads, off, seg, dropped = seq_oracle.segments_from_files(paths, non_acgt)
    assert off.tolist() == o_d.tolist() and seg.tolist() == s_d.tolist() and dropped == d_d
    got = seqio.unpack_reads(w_d, o_d)
    assert got == [reads[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]
    return got


def test_device_ingest_edge_cases(tmp_path):
    fq = tmp_path / "a.fastq"
    fq.write_text("@r1\nACGTAC\n+\nIIIIII\n@r2\nacgtn\n+\nIIIII\n@r3\nTTTT\n+\n@@@@")          # N dropped; last record without newline
    crlf = tmp_path / "b.fastq"
    crlf.write_bytes(b"@r1\r\nACGTACGTAC\r\n+\r\nIIIIIIIIII\r\n@r2\r\nGGGGCCCC\r\n+\r\nIIIIIIII\r\n\r\n\r\n")   # CRLF, blank lines at the end
    noqual = tmp_path / "c.fastq"
    noqual.write_text("@r1\nACGT\n+\nIIII\n@r2\nGATTACA\n+\n")                                  # the last record lacks its quality line
    fa = tmp_path / "d.fa.gz"
    with gzip.open(fa, "wb") as f:
        f.write(b">c1\nACG\nTAC\n>c2 empty\n>c3\nGGGG\n\n  acgt  \n>c4\nACGNT\n>c5\n" + b"ACGT" * 5000 + b"\n")   # multi-line, an empty record, blanks, case, N
    fa_crlf = tmp_path / "e.fa"
    fa_crlf.write_bytes(b">x\r\nACGTACGT\r\nTTTT\r\n>y\r\nCC")                                   # CRLF; the last line without newline
    got = _same([fq, crlf, noqual, fa, fa_crlf], [True] * 5)
    assert got[:2] == [b"ACGTAC", b"TTTT"] and b"" in got and got[-1] == b"CC" and b"GGGGACGT" in got
    # irregular texts: the host reader's grammar decides (blank lines between FASTQ records), its errors are the errors
    gaps = tmp_path / "f.fastq"
    gaps.write_text("@r1\nACGT\n+\nIIII\n\n@r2\nGGCC\n+\nIIII\n")
    got = _same([gaps, fq], [False, True], oracle=False)      # (the Python restatement reads strict four-line records only)
    assert got[:2] == [b"ACGT", b"GGCC"]
    empty = tmp_path / "g.fastq"
    empty.write_text("")
    _same([empty, fa], [True, True])
    bad = tmp_path / "h.txt"
    bad.write_text("not a sequence file\n")
    with pytest.raises(ga.GasmError):
        seqio.read_files_device([bad])
    broken = tmp_path / "i.fastq"
    broken.write_text("@r1\nACGT\n+\nIIII\n@r2\nACGT\n")
    with pytest.raises(ga.GasmError):
        seqio.read_files_device([broken])
    with pytest.raises(ga.GasmError):
        seqio.read_files_device([fq], non_acgt="error")
    cut = tmp_path / "j.fa.gz"
    whole = gzip.compress(b"".join(b">r%d\n%s\n" % (i, b"ACGT" * 20) for i in range(4000)))
    cut.write_bytes(whole[:len(whole) // 2])
    with pytest.raises(ga.GasmError):
        seqio.read_files_device([cut])


def test_device_ingest_random_files(tmp_path):
    rnd = random.Random(7)
    paths = []
    for f in range(4):
        n = rnd.randrange(2000, 30000)
        recs = []
        for i in range(n):
            L = rnd.choice((0, 1, 31, 32, 33, 100, 150, 151, rnd.randrange(1, 400)))
            s = "".join(rnd.choice("ACGTacgt" if rnd.random() < 0.1 else "ACGT") for _ in range(L))
            if rnd.random() < 0.01 and L:
                s = s[:L // 2] + "N" + s[L // 2 + 1:]
            recs.append(s)
        if f % 2 == 0:
            txt = "".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(recs))
        else:
            w = rnd.choice((60, 70, 1000))
            txt = "".join(f">r{i}\n" + "".join(s[j:j + w] + "\n" for j in range(0, len(s), w)) for i, s in enumerate(recs))
        p = tmp_path / (f"s{f}." + ("fastq" if f % 2 == 0 else "fa") + (".gz" if f >= 2 else ""))
        if f >= 2:
            with gzip.open(p, "wb") as g:
                g.write(txt.encode())
        else:
            p.write_text(txt)
        paths.append(p)
    _same(paths, [True] * 4)
