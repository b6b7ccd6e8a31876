"""ORACLE (test infrastructure): a plain Python restatement of what libgasm's sequence-file reader accepts and yields
(genomeassembler_dev_amd/csrc/seqio.cpp: FASTQ with four-line records, FASTA with one- or multi-line sequences, plain or
gzip, case folded, reads with a byte outside ACGT dropped or refused) — the checker of tests/, never the product.  The
reference itself has no reader on this path (it simulates reads in R and writes FASTA, lib/GenerateReads.R:405-433), so
the accepted grammar is this project's definition; parity = C++ reader and this restatement agree record by record."""
import gzip

import numpy as np

_ACGT = np.zeros(256, dtype=bool)
_ACGT[[ord(c) for c in "ACGT"]] = True


def _open(path):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


def read_sequences(path):
    """Sequences of a FASTQ or FASTA file (plain or .gz), upper-cased, as a list of bytes.  FASTQ records are the
    four-line kind sequencers write; FASTA sequences may span lines."""
    out = []
    with _open(path) as f:
        first = f.read(1)
        if not first:
            return out
        rest = f.read()
    data = first + rest
    lines = data.split(b"\n")
    if first == b"@":
        for i in range(0, len(lines) - 1, 4):
            if not lines[i].startswith(b"@"):
                if lines[i].strip() == b"":
                    continue
                raise ValueError(f"{path}: record {i // 4} does not start with '@'")
            if i + 2 >= len(lines) or not lines[i + 2].startswith(b"+"):
                raise ValueError(f"{path}: record {i // 4} has no '+' line (multi-line FASTQ is not supported)")
            out.append(lines[i + 1].strip().upper())
    elif first == b">":
        cur = None
        for ln in lines:
            if ln.startswith(b">"):
                if cur is not None:
                    out.append(b"".join(cur).upper())
                cur = []
            elif cur is not None:
                cur.append(ln.strip())
        if cur is not None:
            out.append(b"".join(cur).upper())
    else:
        raise ValueError(f"{path}: neither FASTQ ('@') nor FASTA ('>')")
    return out


def segments_from_files(paths, non_acgt="drop"):
    """One file per segment -> (reads uint8, read_off uint64[n+1], seg_read_off uint64[S+1], dropped).
    non_acgt: 'drop' removes reads holding a base outside ACGT (N, IUPAC codes) and counts them in `dropped`;
    'error' raises instead (the packing kernel accepts ACGT only, as the reference's k-mer tables do)."""
    seg = np.zeros(len(paths) + 1, dtype=np.uint64)
    chunks, lens, dropped = [], [], 0
    for s, p in enumerate(paths):
        kept = 0
        for r in read_sequences(p):
            a = np.frombuffer(r, dtype=np.uint8)
            if a.size and not _ACGT[a].all():
                if non_acgt == "error":
                    raise ValueError(f"{p}: read with a base outside ACGT")
                dropped += 1
                continue
            chunks.append(a)
            lens.append(a.size)
            kept += 1
        seg[s + 1] = seg[s] + kept
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    if lens:
        off[1:] = np.cumsum(np.array(lens, dtype=np.uint64))
    reads = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return reads, off, seg, dropped
