"""ctypes binding of libgasm.so (include/gasm.h).  There is no fallback: if the library is missing or no gfx950 GPU is
usable, calls raise — the product never computes on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GASM_LIBGASM") or os.path.join(_HERE, "libgasm.so")     # (GASM_LIBGASM: a build variant, experiments only)

GASM_OK = 0
STATUS = {0: "GASM_OK", -1: "GASM_ERR_INVALID", -2: "GASM_ERR_NON_ACGT", -3: "GASM_ERR_NO_DEVICE", -4: "GASM_ERR_HIP",
          -5: "GASM_ERR_CAPACITY", -6: "GASM_ERR_RANGE", -7: "GASM_ERR_STATE"}
TABLE_ROWS = 69904
SCORE_OWN, SCORE_VELVET = 0, 1
WANT_LEV, WANT_FREQ, WANT_KS = 1, 2, 4


class GasmError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"{STATUS.get(status, status)}: {text}")
        self.status = status


_vp, _u64, _u32, _i32, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int
_PP = C.POINTER(C.c_void_p)

# every symbol include/gasm.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "gasm_last_error": (C.c_char_p, []),
    "gasm_version": (C.c_char_p, []),
    "gasm_ctx_create": (_int, [_int, _PP]),
    "gasm_ctx_destroy": (None, [_vp]),
    "gasm_ctx_sync": (_int, [_vp]),
    "gasm_ctx_stream": (_vp, [_vp]),
    "gasm_get_contigs": (_int, [_vp, _vp, _u64, _int, _int, _int, _PP]),
    "gasm_get_contigs_from_reads": (_int, [_vp, _vp, _vp, _u64, _int, _int, _int, _PP]),
    "gasm_contigs_count": (_u64, [_vp]),
    "gasm_contigs_data": (_vp, [_vp]),
    "gasm_contigs_offsets": (_vp, [_vp]),
    "gasm_contigs_rows": (_u64, [_vp]),
    "gasm_contigs_perm": (_vp, [_vp]),
    "gasm_contigs_distinct_count": (_u64, [_vp]),
    "gasm_contigs_key_words": (_int, [_vp]),
    "gasm_contigs_distinct_keys": (_vp, [_vp]),
    "gasm_contigs_distinct_mult": (_vp, [_vp]),
    "gasm_contigs_free": (None, [_vp]),
    "gasm_assemble_contigs": (_int, [_vp, _vp, _vp, _u64, _vp, _u64, _u64, _int, _PP]),
    "gasm_assemble_contigs_velvet": (_int, [_vp, _vp, _vp, _u64, _int, _int, _int, _PP]),
    "gasm_assemble_contigs_dev": (_int, [_vp, _vp, _vp, _u64, _vp, _u64, _u64, _int, _PP]),
    "gasm_assemble_contigs_velvet_dev": (_int, [_vp, _vp, _vp, _u64, _int, _int, _int, _PP]),
    "gasm_scaffolds_count": (_u64, [_vp]),
    "gasm_scaffolds_offsets": (_vp, [_vp]),
    "gasm_scaffolds_merge_device": (_int, [_vp, C.POINTER(_u64)]),
    "gasm_scaffolds_fetch": (_int, [_vp, _PP]),
    "gasm_scaffolds_free": (None, [_vp]),
    "gasm_calc_breakscore_dev": (_int, [_vp, _vp, _vp, _vp, _u64, _vp, _u64, _int, _vp, _vp, _u64, _vp, _int, _int, _PP]),
    "gasm_strlist_count": (_u64, [_vp]),
    "gasm_strlist_data": (_vp, [_vp]),
    "gasm_strlist_offsets": (_vp, [_vp]),
    "gasm_strlist_free": (None, [_vp]),
    "gasm_calc_breakscore": (_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _u64, _int, _vp, _vp, _u64, _vp, _int, _int,
                                    _PP]),
    "gasm_scores_count": (_u64, [_vp]),
    "gasm_scores_sequence_len": (_vp, [_vp]),
    "gasm_scores_bp_score": (_vp, [_vp]),
    "gasm_scores_norm_by_break_freqs": (_vp, [_vp]),
    "gasm_scores_norm_by_len": (_vp, [_vp]),
    "gasm_scores_kmer_breaks": (_vp, [_vp]),
    "gasm_scores_lev_dist": (_vp, [_vp]),
    "gasm_scores_path_freq": (_vp, [_vp]),
    "gasm_scores_startpos": (_vp, [_vp]),
    "gasm_scores_prob_dist": (_vp, [_vp]),
    "gasm_scores_prob_dist_offsets": (_vp, [_vp]),
    "gasm_scores_ks": (_vp, [_vp]),
    "gasm_scores_lev_device": (_int, [_vp]),
    "gasm_coverage_percent": (_int, [_vp, _vp, _vp, _u64, C.c_int64, _vp]),
    "gasm_scores_free": (None, [_vp]),
    "gasm_levenshtein": (_int, [_vp, _u64, _vp, _u64, _int, C.POINTER(_i32)]),
    "gasm_batch_create": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _u32, _PP]),
    "gasm_batch_create_packed": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _u32, _PP]),
    "gasm_batch_from_files": (_int, [_vp, _vp, _u32, _int, _PP, C.POINTER(_u64)]),
    "gasm_batch_guided": (_int, [_vp]),
    "gasm_batch_fetch_guided": (_int, [_vp, _PP, _PP, _PP, _PP, _PP, _PP]),
    "gasm_batch_fetch_score_fixed": (_int, [_vp, _PP, C.POINTER(_int)]),
    "gasm_batch_simulate": (_int, [_vp, _vp, _vp, _u32, _u32, C.c_double, _u64, _int, _vp, _PP]),
    "gasm_batch_fetch_read_starts": (_int, [_vp, _PP, _PP]),
    "gasm_read_files": (_int, [_vp, _u32, _int, _PP]),
    "gasm_read_files_device": (_int, [_vp, _vp, _u32, _int, _PP]),
    "gasm_packed_parsed_on_device": (_int, [_vp, _u32]),
    "gasm_packed_n_reads": (_u64, [_vp]),
    "gasm_packed_n_segments": (_u32, [_vp]),
    "gasm_packed_words": (_vp, [_vp]),
    "gasm_packed_read_off": (_vp, [_vp]),
    "gasm_packed_seg_read_off": (_vp, [_vp]),
    "gasm_packed_dropped": (_u64, [_vp]),
    "gasm_packed_free": (None, [_vp]),
    "gasm_batch_free": (None, [_vp]),
    "gasm_batch_build": (_int, [_vp, _int, _u64]),
    "gasm_batch_score": (_int, [_vp, _int, _vp]),
    "gasm_batch_total_kmers": (_u64, [_vp]),
    "gasm_batch_total_reads": (_u64, [_vp]),
    "gasm_batch_fetch_distinct": (_int, [_vp, _PP, _PP, _PP, C.POINTER(_int)]),
    "gasm_batch_fetch_contigs": (_int, [_vp, _PP, _PP, _PP]),
    "gasm_batch_fetch_graph": (_int, [_vp, _PP, _PP]),
    "gasm_batch_fetch_scores": (_int, [_vp, _PP, _PP, _PP, _PP, _PP]),
    "gasm_pool_create": (_int, [_vp, _vp, _u64, _u32, _vp, _u32, _PP]),
    "gasm_pool_free": (None, [_vp]),
    "gasm_pool_key_words": (_int, [_vp]),
    "gasm_pool_local_runs": (_int, [_vp, _int, _int, _PP]),
    "gasm_pool_pack_runs": (_int, [_vp, _vp, _u64, _vp, _vp]),
    "gasm_pool_merge_runs": (_int, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _PP]),
    "gasm_pool_graph": (_int, [_vp, _u32]),
    "gasm_pool_piece_words": (_int, [_vp, _u32, _u32, _vp]),
    "gasm_pool_pack_reads": (_int, [_vp, _u32, _u32, _vp]),
    "gasm_pool_set_reads": (_int, [_vp, _vp, _u64, _u32, _vp, _vp, _vp]),
    "gasm_pool_score": (_int, [_vp, _int, _vp]),
    "gasm_pool_fetch_distinct": (_int, [_vp, _PP, _PP, _PP, C.POINTER(_int)]),
    "gasm_pool_fetch_contigs": (_int, [_vp, _PP, _PP, _PP]),
    "gasm_pool_fetch_scores": (_int, [_vp, _PP, _PP, _PP, _PP, _PP]),
    "gasm_comm_unique_id": (_int, [_vp]),
    "gasm_comm_create": (_int, [_vp, _vp, _int, _int, _PP]),
    "gasm_comm_create_virtual": (_int, [_vp, _int, _PP]),
    "gasm_comm_destroy": (None, [_vp]),
    "gasm_comm_world": (_int, [_vp]),
    "gasm_comm_rank": (_int, [_vp]),
    "gasm_comm_stage": (_int, [_vp]),
    "gasm_pool_bucket_owner": (_int, [_u32, _int, _u32, _vp]),
    "gasm_pool_segment_bounds": (_int, [_u32, _u32, _vp]),
    "gasm_pool_exchange_build": (_int, [_vp, _vp, _u32, _int, _int, _int, _vp, _vp]),
    "gasm_profile_enable": (_int, [_vp, _int]),
    "gasm_profile_filter": (_int, [_vp, C.c_char_p]),
    "gasm_profile_reset": (_int, [_vp]),
    "gasm_profile_read": (_int, [_vp, C.POINTER(_int), _PP, _PP, _PP]),
}

_lib = None


def _one_hip_runtime():
    """One HIP runtime per process, whatever the import order.  PyTorch-ROCm ships its own libamdhip64.so (soname
    libamdhip64.so.7, the soname libgasm.so asks for) and its own librccl.so.  If libgasm were loaded first, the loader would
    take ROCm's copy for it and later a second one (torch's, by path) for torch — two runtimes, and the one that initialises
    the GPU second finds no device (round 2: "No HIP GPUs are available" after 55 tests).  So where torch is installed its
    copy is mapped BEFORE libgasm, imported or not: libgasm's NEEDED entry then resolves to it by soname, and a later
    `import torch` finds its own file already mapped.  Without torch on the machine ROCm's copy is the only one there is."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        base = os.path.dirname(sys.modules["torch"].__file__)
    else:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin:
            return
        base = os.path.dirname(spec.origin)
    for name in ("libamdhip64.so", "librccl.so"):
        path = os.path.join(base, "lib", name)
        if os.path.exists(path):
            try:
                C.CDLL(path)       # (RTLD_LOCAL: the loader matches NEEDED entries by soname in any case; RTLD_GLOBAL on librccl ends in a double free at exit)
            except OSError as e:       # a torch build without ROCm libraries: nothing to share
                if name == "libamdhip64.so":
                    raise RuntimeError(f"PyTorch's HIP runtime at {path} could not be mapped ({e}); libgasm would bring a "
                                       "second runtime into a process that later imports torch") from e


def hip_runtimes_mapped():
    """paths of the libamdhip64 copies mapped into this process (the guard's check: there must be exactly one)"""
    out = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                out.add(line.split()[-1])
    return sorted(out)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc, gfx950).  genomeassembler_dev_amd has no CPU fallback.")
        _one_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != GASM_OK:
        raise GasmError(status, (lib().gasm_last_error() or b"").decode(errors="replace"))


_ctxs = {}


class Context:
    """One GPU, one HIP stream (gasm_ctx)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        check(lib().gasm_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)

    def sync(self):
        check(lib().gasm_ctx_sync(self.h))

    def stream(self):
        return lib().gasm_ctx_stream(self.h)

    def profile(self, on=True, only=None):
        """HIP-event timing per kernel launch; `only` = iterable of kernel names to restrict it to"""
        check(lib().gasm_profile_filter(self.h, ",".join(only).encode() if only else None))
        check(lib().gasm_profile_enable(self.h, int(on)))

    def profile_reset(self):
        check(lib().gasm_profile_reset(self.h))

    def profile_read(self):
        """{kernel name: (total ms, launches)} since the last reset"""
        n = C.c_int()
        names, ms, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().gasm_profile_read(self.h, C.byref(n), C.byref(names), C.byref(ms), C.byref(cnt)))
        if n.value == 0:
            return {}
        nm = C.cast(names, C.POINTER(C.c_char_p))
        m = C.cast(ms, C.POINTER(C.c_double))
        c = C.cast(cnt, C.POINTER(C.c_uint64))
        return {nm[i].decode(): (m[i], int(c[i])) for i in range(n.value)}

    def close(self):
        if self.h:
            lib().gasm_ctx_destroy(self.h)
            self.h = None


def default_context(device=0):
    if device not in _ctxs:
        _ctxs[device] = Context(device)
    return _ctxs[device]
