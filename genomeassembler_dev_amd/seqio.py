"""FASTQ / FASTA in (the north star's "FASTQ-in" surface — the reference itself simulates its reads in
R, lib/GenerateReads.R, and has no file reader on this path).  The reader is libgasm's (csrc/seqio.cpp, C++ with zlib):
it parses and packs 2-bit in one go; this module is its ctypes face."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def _paths(paths):
    arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
    return arr


def read_files(paths, non_acgt="drop"):
    """one file per segment -> (words uint64 (2-bit, 32 bases per word, first base most significant, reads back to back),
    read_off uint64[n+1] (base offsets), seg_read_off uint64[S+1], dropped reads)"""
    h = C.c_void_p()
    check(lib().gasm_read_files(_paths(paths), len(paths), 1 if non_acgt == "error" else 0, C.byref(h)))
    L = lib()
    try:
        n, S = L.gasm_packed_n_reads(h), L.gasm_packed_n_segments(h)
        off = np.ctypeslib.as_array(C.cast(L.gasm_packed_read_off(h), C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        seg = np.ctypeslib.as_array(C.cast(L.gasm_packed_seg_read_off(h), C.POINTER(C.c_uint64)), shape=(S + 1,)).copy()
        nw = (int(off[-1]) + 31) // 32
        words = np.ctypeslib.as_array(C.cast(L.gasm_packed_words(h), C.POINTER(C.c_uint64)), shape=(nw,)).copy() if nw else np.zeros(0, np.uint64)
        return words, off, seg, int(L.gasm_packed_dropped(h))
    finally:
        L.gasm_packed_free(h)


def read_files_device(paths, non_acgt="drop", ctx=None):
    """the same through the device path (gasm_read_files_device, csrc/ingest.hip: the host inflates, the GPU finds the records
    and packs them) -> (words, read_off, seg_read_off, dropped, parsed_on_device[bool per file])"""
    from ._lib import default_context
    ctx = ctx or default_context()
    h = C.c_void_p()
    check(lib().gasm_read_files_device(ctx.h, _paths(paths), len(paths), 1 if non_acgt == "error" else 0, C.byref(h)))
    L = lib()
    try:
        n, S = L.gasm_packed_n_reads(h), L.gasm_packed_n_segments(h)
        off = np.ctypeslib.as_array(C.cast(L.gasm_packed_read_off(h), C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        seg = np.ctypeslib.as_array(C.cast(L.gasm_packed_seg_read_off(h), C.POINTER(C.c_uint64)), shape=(S + 1,)).copy()
        nw = (int(off[-1]) + 31) // 32
        words = np.ctypeslib.as_array(C.cast(L.gasm_packed_words(h), C.POINTER(C.c_uint64)), shape=(nw,)).copy() if nw else np.zeros(0, np.uint64)
        return words, off, seg, int(L.gasm_packed_dropped(h)), [bool(L.gasm_packed_parsed_on_device(h, f)) for f in range(len(paths))]
    finally:
        L.gasm_packed_free(h)


def unpack_reads(words, read_off):
    """packed reads -> list of bytes (for inspection and tests)"""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = int(read_off[-1])
    w = np.asarray(words, dtype=np.uint64)
    idx = np.arange(n)
    bases = lut[((w[idx >> 5] >> (62 - 2 * (idx & 31)).astype(np.uint64)) & np.uint64(3)).astype(np.int64)] if n else np.zeros(0, np.uint8)
    return [bases[int(read_off[i]):int(read_off[i + 1])].tobytes() for i in range(len(read_off) - 1)]
