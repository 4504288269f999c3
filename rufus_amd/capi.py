"""ctypes binding of ``librufus_hip.so`` (C-ABI declared in ``include/rufus_hip.h``).

Plumbing only: every compute call goes through the C-ABI into hand-written HIP kernels.  There is no
CPU fallback -- importing works without a GPU (so the symbol table and the host-only helpers can be
tested), but ``Context()`` raises ``RufusError`` when no gfx950 device is visible, and a missing
shared library is an ``ImportError``.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RFX_LIB") or os.path.join(_HERE, "librufus_hip.so")  # (RFX_LIB: an experiment's build, scratch/build_variants.sh)

HISTO_BINS = 10002
PACK_COUNT, PACK_FILTER = 1, 2
COUNT_AUTO, COUNT_TABLE, COUNT_P2L, COUNT_MSP = 0, 1, 2, 3
E_FULL, E_RANGE, E_MIXEDCASE = -4, -7, -6

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)

# name -> (restype, argtypes); the unit tests check that every one of these is exported.
SIGNATURES = {
    "rfx_version": (C.c_char_p, []),
    "rfx_strerror": (C.c_char_p, [C.c_int]),
    "rfx_last_error": (C.c_char_p, []),
    "rfx_jf_matrix": (C.c_int, [C.c_int, C.c_int, u64p]),
    "rfx_jf_pos": (C.c_uint64, [u64p, C.c_int, C.c_int, C.c_uint64]),
    "rfx_pack_words": (C.c_uint64, [u64p, C.c_uint32]),
    "rfx_pack_reads": (C.c_int, [C.c_char_p, C.c_char_p, u64p, C.c_uint32, C.c_int, C.c_int, u64p, u32p, u32p, u32p,
                                 u32p]),
    "rfx_pack_spans": (C.c_int, [C.c_void_p, u64p, u32p, u64p, C.c_uint32, C.c_int, C.c_int, u64p, u32p, u32p, u32p, u32p]),
    "rfx_hashlist_keys": (C.c_long, [C.c_char_p, C.c_size_t, C.c_int, C.c_int, u64p, C.c_size_t]),
    "rfx_jhash_header": (C.c_long, [C.c_int, C.c_int, u64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                    C.c_char_p, C.c_size_t]),
    "rfx_open": (C.c_void_p, [C.c_int, C.c_size_t]),
    "rfx_close": (None, [C.c_void_p]),
    "rfx_sync": (C.c_int, [C.c_void_p]),
    "rfx_host_alloc": (C.c_void_p, [C.c_size_t]),
    "rfx_host_alloc_lazy": (C.c_void_p, [C.c_size_t]),
    "rfx_host_pin": (C.c_int, [C.c_void_p]),
    "rfx_host_cpus": (C.c_uint, []),
    "rfx_host_free": (None, [C.c_void_p]),
    "rfx_stream": (C.c_void_p, [C.c_void_p]),
    "rfx_mem_stats": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "rfx_memcpy_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rfx_prof_filter": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rfx_prof_reset": (C.c_int, [C.c_void_p]),
    "rfx_prof_query": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), u64p]),
    "rfx_prof_names": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "rfx_reads_device_bytes": (C.c_uint64, [C.c_void_p]),
    "rfx_records_subtract": (C.c_void_p, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32]),
    "rfx_reads_upload": (C.c_void_p, [C.c_void_p, u64p, u32p, u32p, u32p, u32p, C.c_uint32]),
    "rfx_reads_free": (None, [C.c_void_p]),
    "rfx_reads_count": (C.c_uint32, [C.c_void_p]),
    "rfx_reads_bases": (C.c_uint64, [C.c_void_p]),
    "rfx_reads_words": (C.c_uint64, [C.c_void_p]),
    "rfx_reads_get": (C.c_int, [C.c_void_p, u64p, u32p, u32p, u32p, u32p]),
    "rfx_synth_reads": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int]),
    "rfx_text_open": (C.c_void_p, [C.c_void_p, C.c_uint64]),
    "rfx_text_close": (None, [C.c_void_p]),
    "rfx_text_room": (C.c_uint64, [C.c_void_p]),
    "rfx_text_bytes": (C.c_uint64, [C.c_void_p]),
    "rfx_text_append": (C.c_long, [C.c_void_p, C.c_char_p, C.c_uint64]),
    "rfx_text_copied": (C.c_int, [C.c_void_p, C.c_long]),
    "rfx_text_wait": (C.c_int, [C.c_void_p, C.c_long]),
    "rfx_text_parse": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rfx_text_fetch": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rfx_text_reset": (None, [C.c_void_p]),
    "rfx_synth_text": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rfx_synth_snv": (C.c_int, [C.c_void_p, C.c_uint32, u64p, C.c_char_p, C.c_char_p]),
    "rfx_synth_genome": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "rfx_count_begin": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]),
    "rfx_count_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "rfx_count_set_passes": (C.c_int, [C.c_void_p, C.c_int]),
    "rfx_count_add": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rfx_count_add_pairs_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rfx_count_stats": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "rfx_count_finish_begin": (C.c_void_p, [C.c_void_p, C.c_uint64, C.c_uint64, u64p]),
    "rfx_count_finish_end": (C.c_void_p, [C.c_void_p]),
    "rfx_count_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rfx_count_set_early": (C.c_int, [C.c_void_p, C.c_int]),
    "rfx_count_early_segments": (C.c_int, [C.c_void_p]),
    "rfx_count_adopt_early": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rfx_mem_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rfx_runmaps_create": (C.c_void_p, [C.c_void_p, C.c_uint64]),
    "rfx_runmaps_create_pooled": (C.c_void_p, [C.c_void_p, C.c_uint64]),
    "rfx_runmaps_free": (None, [C.c_void_p]),
    "rfx_runmaps_bytes": (C.c_uint64, [C.c_void_p]),
    "rfx_runmaps_blocks": (C.c_int, [C.c_void_p]),
    "rfx_runmaps_drop": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rfx_runmaps_clear": (C.c_int, [C.c_void_p]),
    "rfx_count_set_runmaps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rfx_count_prepare_maps": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    "rfx_count_prefetch_maps": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    "rfx_count_replayed": (C.c_uint64, [C.c_void_p]),
    "rfx_count_segments": (C.c_int, [C.c_void_p]),
    "rfx_count_segment_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_uint32), u64p]),
    "rfx_count_add_records_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]),
    "rfx_count_segment_ext": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "rfx_count_add_records_ext_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]),
    "rfx_count_free": (None, [C.c_void_p]),
    "rfx_count_finish": (C.c_void_p, [C.c_void_p, C.c_uint64, C.c_uint64, u64p]),
    "rfx_records_size": (C.c_uint64, [C.c_void_p]),
    "rfx_records_k": (C.c_int, [C.c_void_p]),
    "rfx_records_lsize": (C.c_int, [C.c_void_p]),
    "rfx_records_payload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "rfx_records_payload_range": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]),
    "rfx_records_get": (C.c_int, [C.c_void_p, u64p, u32p, u64p]),
    "rfx_records_load": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, u64p, C.c_void_p, C.c_uint64, C.c_int]),
    "rfx_records_load_fd": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, u64p, C.c_int, C.c_uint64, C.c_uint64, C.c_int]),
    "rfx_records_from_dev": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, u64p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rfx_records_dev_keys": (C.c_void_p, [C.c_void_p]),
    "rfx_records_dev_counts": (C.c_void_p, [C.c_void_p]),
    "rfx_records_dev_pos": (C.c_void_p, [C.c_void_p]),
    "rfx_records_histo": (C.c_int, [C.c_void_p, u64p]),
    "rfx_records_verify": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u64p]),
    "rfx_records_checksum": (C.c_int, [C.c_void_p, u64p]),
    "rfx_count_adopt_records_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]),
    "rfx_ctx_allow_peers": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "rfx_peers_create": (C.c_void_p, [C.c_int]),
    "rfx_peers_free": (None, [C.c_void_p]),
    "rfx_count_set_peers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rfx_records_free": (None, [C.c_void_p]),
    "rfx_merge_unique": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_uint32, u64p, u32p, C.c_uint64,
                                   u64p]),
    "rfx_query": (C.c_int, [C.c_void_p, u64p, C.c_uint64, u32p]),
    "rfx_unique_to_subject": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_uint32,
                                        C.c_uint32, C.c_uint32, u64p, u32p, C.c_uint64, u64p]),
    "rfx_set_build": (C.c_void_p, [C.c_void_p, u64p, C.c_uint64, C.c_int]),
    "rfx_set_size": (C.c_uint64, [C.c_void_p]),
    "rfx_set_free": (None, [C.c_void_p]),
    "rfx_filter_many": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(u64p), u64p]),
    "rfx_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, u32p, u64p, u64p]),
    "rfx_overlap_score": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                    C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rfx_annotate": (C.c_int, [C.c_void_p, C.c_void_p, u32p]),
    "rfx_ovl_pool_create": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int]),
    "rfx_ovl_pool_set": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "rfx_ovl_pool_score": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float,
                                     C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rfx_ovl_pool_free": (None, [C.c_void_p]),
    "rfx_model_residuals": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_double)]),
    "rfx_model_tables": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                                   C.c_size_t, C.POINTER(C.c_double)]),
}


class RufusError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library; a missing build is a hard error, never a silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or `make -C rufus_amd/csrc`)")
        # PyTorch-ROCm ships its own copy of the HIP runtime.  If it is loaded AFTER librufus_hip.so has
        # pulled in /opt/rocm's, the process ends up with two runtimes and torch sees no GPU; loaded first,
        # ours binds to the runtime that is already there.  torch is optional for this module.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        L = lib()
        raise RufusError(f"{what}: {L.rfx_strerror(rc).decode()} ({rc}) {L.rfx_last_error().decode()}")


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def ceil_log2(x: int) -> int:
    return max(0, (int(x) - 1).bit_length())


# ---------------------------------------------------------------------------------------------------
# host-only helpers
# ---------------------------------------------------------------------------------------------------
def jf_matrix(lsize: int, k: int) -> np.ndarray:
    cols = np.zeros(2 * k, dtype=np.uint64)
    _check(lib().rfx_jf_matrix(lsize, k, _p(cols, u64p)), "rfx_jf_matrix")
    return cols


def jf_pos(cols: np.ndarray, k: int, lsize: int, key: int) -> int:
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    return int(lib().rfx_jf_pos(_p(cols, u64p), k, lsize, int(key)))


def hashlist_keys(text: bytes, k: int, single_end: bool = False) -> np.ndarray:
    n = lib().rfx_hashlist_keys(text, len(text), k, int(single_end), None, 0)
    if n < 0:
        _check(int(n), "rfx_hashlist_keys")
    out = np.zeros(max(n, 1), dtype=np.uint64)
    n2 = lib().rfx_hashlist_keys(text, len(text), k, int(single_end), _p(out, u64p), len(out))
    assert n2 == n
    return out[:n]


def jhash_header(k: int, lsize: int, cols: np.ndarray, canonical: bool = True, counter_len: int = 4, argv=()) -> bytes:
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    arr = (C.c_char_p * max(1, len(argv)))(*[a.encode() if isinstance(a, str) else a for a in argv])
    buf = C.create_string_buffer(1 << 16)
    n = lib().rfx_jhash_header(k, lsize, _p(cols, u64p), int(canonical), counter_len, len(argv), arr, buf, len(buf))
    if n < 0:
        _check(int(n), "rfx_jhash_header")
    return buf.raw[:n]


class Synth(C.Structure):
    """``rfx_synth``: parameters of one sample of the synthetic trio workload."""
    _fields_ = [("genome_len", C.c_uint64), ("genome_seed", C.c_uint64), ("snv_seed", C.c_uint64),
                ("read_seed", C.c_uint64), ("n_snv", C.c_uint32), ("read_len", C.c_uint32),
                ("insert_lo", C.c_uint32), ("insert_span", C.c_uint32), ("err_1024", C.c_uint32),
                ("lowq_256", C.c_uint32), ("n_1024", C.c_uint32), ("carrier", C.c_uint32)]

    @classmethod
    def sample(cls, genome_len: int, which: int, n_snv: int = 20, seed: int = 12345, read_len: int = 150):
        """Sample `which` of a trio on one genome: 0 = child (carrier of the SNVs), 1 / 2 = parents."""
        return cls(genome_len=genome_len, genome_seed=seed, snv_seed=seed + 7, read_seed=seed * 1000 + which,
                   n_snv=n_snv, read_len=read_len, insert_lo=250, insert_span=151, err_1024=5, lowq_256=5, n_1024=1,
                   carrier=1 if which == 0 else 0)

    def text(self, first_pair: int, n_pairs: int):
        """(seq, qual) uint8 matrices of shape (2*n_pairs, read_len): read 2p = mate 1, 2p+1 = mate 2."""
        seq = np.zeros((2 * n_pairs, self.read_len), dtype=np.uint8)
        qual = np.zeros_like(seq)
        _check(lib().rfx_synth_text(C.byref(self), first_pair, n_pairs, seq.ctypes.data, qual.ctypes.data),
               "rfx_synth_text")
        return seq, qual

    def snvs(self):
        out = []
        for i in range(self.n_snv):
            pos, ref, alt = C.c_uint64(0), C.create_string_buffer(1), C.create_string_buffer(1)
            _check(lib().rfx_synth_snv(C.byref(self), i, C.byref(pos), ref, alt), "rfx_synth_snv")
            out.append((pos.value, ref.raw, alt.raw))
        return out

    def genome(self, first: int, n: int) -> bytes:
        buf = np.zeros(max(n, 1), dtype=np.uint8)
        _check(lib().rfx_synth_genome(C.byref(self), first, n, buf.ctypes.data), "rfx_synth_genome")
        return buf[:n].tobytes()


class PackedReads:
    """Host-side packed block: 32 bases per 64-bit code word, one mask bit per base."""

    def __init__(self, seq: bytes, off: np.ndarray, qual: bytes | None = None, min_q: int = 0,
                 flags: int = PACK_COUNT):
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        L = lib()
        nw = int(L.rfx_pack_words(_p(off, u64p), n))
        self.n = n
        self.codes = np.zeros(max(nw, 1), dtype=np.uint64)
        self.acgt = np.zeros(max(nw, 1), dtype=np.uint32) if flags & PACK_COUNT else None
        self.good = np.zeros(max(nw, 1), dtype=np.uint32) if flags & PACK_FILTER else None
        self.word_off = np.zeros(n + 1, dtype=np.uint32)
        self.len = np.zeros(max(n, 1), dtype=np.uint32)
        _check(L.rfx_pack_reads(seq, qual, _p(off, u64p), n, min_q, flags, _p(self.codes, u64p), _p(self.acgt, u32p),
                                _p(self.good, u32p), _p(self.word_off, u32p), _p(self.len, u32p)), "rfx_pack_reads")

    @classmethod
    def from_reads(cls, seqs, quals=None, min_q: int = 0, flags: int = PACK_COUNT):
        lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        q = None
        if quals is not None:
            # a quality string shorter than its read reads as '\0' (bad) past its end
            q = b"".join((qq + b"\0" * (len(s) - len(qq)))[:len(s)] for s, qq in zip(seqs, quals))
        return cls(b"".join(seqs), off, q, min_q, flags)


# ---------------------------------------------------------------------------------------------------
# device objects
# ---------------------------------------------------------------------------------------------------
class Context:
    def __init__(self, device: int = 0, hbm_budget: int = 0):
        self._h = lib().rfx_open(device, hbm_budget)
        if not self._h:
            raise RufusError("rfx_open failed (no CPU fallback): " + lib().rfx_last_error().decode())

    def close(self):
        if self._h:
            if getattr(self, "_pin_ptr", None):
                self._pin_arr = None
                lib().rfx_host_free(self._pin_ptr)
                self._pin_ptr = None
            lib().rfx_close(self._h)
            self._h = None

    def sync(self):
        _check(lib().rfx_sync(self._h), "rfx_sync")

    def mem_stats(self) -> dict:
        u, pk, m = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().rfx_mem_stats(self._h, C.byref(u), C.byref(pk), C.byref(m)), "rfx_mem_stats")
        return {"used": u.value, "peak": pk.value, "mapped": m.value}

    def memcpy_dev(self, dst: int, src: int, nbytes: int):
        _check(lib().rfx_memcpy_dev(self._h, dst, src, nbytes), "rfx_memcpy_dev")

    def prof(self, on: bool):
        _check(lib().rfx_prof_enable(self._h, int(on)), "rfx_prof_enable")

    def prof_filter(self, names=()):
        _check(lib().rfx_prof_filter(self._h, ",".join(names).encode()), "rfx_prof_filter")

    def prof_reset(self):
        _check(lib().rfx_prof_reset(self._h), "rfx_prof_reset")

    def prof_dict(self) -> dict:
        buf = C.create_string_buffer(8192)
        _check(lib().rfx_prof_names(self._h, buf, len(buf)), "rfx_prof_names")
        out = {}
        for name in buf.value.decode().split("\n"):
            if not name:
                continue
            ms, n = C.c_double(0), C.c_uint64(0)
            _check(lib().rfx_prof_query(self._h, name.encode(), C.byref(ms), C.byref(n)), "rfx_prof_query")
            out[name] = (ms.value, n.value)
        return out

    def upload(self, p: PackedReads) -> "ReadBlock":
        return ReadBlock(self, p)

    def synth_reads(self, sy: Synth, first_pair: int, n_pairs: int, min_q: int = 15, want_good: bool = True,
                    compact: bool = False):
        """Pairs [first_pair, first_pair + n_pairs) of a synthetic sample, generated on the device.  compact: no
        offset / length arrays, ACGT mask only for the reads with an N (43 instead of 68 B per 150 bp read)."""
        h = lib().rfx_synth_reads(self._h, C.byref(sy), first_pair, n_pairs, min_q, int(want_good) | (2 if compact else 0))
        if not h:
            raise RufusError("rfx_synth_reads failed: " + lib().rfx_last_error().decode())
        return ReadBlock.from_handle(self, h)

    def pinned_u64(self, n_words: int) -> np.ndarray:
        """A page-locked uint64 array of at least n_words (rfx_host_alloc), kept by the ctx and reused by the next call."""
        have = getattr(self, "_pin_arr", None)
        if have is None or len(have) < n_words:
            old_p = getattr(self, "_pin_ptr", None)
            p = lib().rfx_host_alloc(max(n_words, 1) * 8)
            if not p:
                raise RufusError("rfx_host_alloc failed")
            self._pin_ptr = p
            self._pin_arr = np.ctypeslib.as_array((C.c_uint64 * max(n_words, 1)).from_address(p))
            if old_p:
                lib().rfx_host_free(old_p)
        return self._pin_arr[:max(n_words, 1)]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class TextArena:
    """Strict 4-line FASTQ text -> a read block, parsed and packed on the device (rfx_text_*: SURVEY section 2, K1)."""

    def __init__(self, ctx: Context, cap_bytes: int):
        self.ctx = ctx
        self._h = lib().rfx_text_open(ctx._h, cap_bytes)
        if not self._h:
            raise RufusError("rfx_text_open failed: " + lib().rfx_last_error().decode())

    def append(self, data: bytes) -> int:
        t = lib().rfx_text_append(self._h, data, len(data))
        if t < 0:
            _check(int(t), "rfx_text_append")
        _check(lib().rfx_text_wait(self._h, t), "rfx_text_wait")     # (a bytes object is not pinned and may go away)
        return int(t)

    def parse(self, flags: int = PACK_COUNT, min_q: int = 0):
        """The block, or None when the text is not strict 4-line FASTQ."""
        strict = C.c_int(1)
        h = lib().rfx_text_parse(self._h, flags, min_q, C.byref(strict))
        if not h:
            if strict.value == 0:
                return None
            raise RufusError("rfx_text_parse failed: " + lib().rfx_last_error().decode())
        return ReadBlock.from_handle(self.ctx, h)

    def fetch(self) -> bytes:
        n = int(lib().rfx_text_bytes(self._h))
        buf = C.create_string_buffer(max(n, 1))
        _check(lib().rfx_text_fetch(self._h, buf), "rfx_text_fetch")
        return buf.raw[:n]

    def reset(self):
        lib().rfx_text_reset(self._h)

    def close(self):
        if self._h:
            lib().rfx_text_close(self._h)
            self._h = None


class ReadBlock:
    def __init__(self, ctx: Context, p: PackedReads):
        self.ctx = ctx
        self._h = lib().rfx_reads_upload(ctx._h, _p(p.codes, u64p), _p(p.acgt, u32p), _p(p.good, u32p),
                                         _p(p.word_off, u32p), _p(p.len, u32p), p.n)
        if not self._h:
            raise RufusError("rfx_reads_upload failed: " + lib().rfx_last_error().decode())
        self.n = p.n

    @classmethod
    def from_handle(cls, ctx: Context, handle):
        self = cls.__new__(cls)
        self.ctx, self._h = ctx, handle
        self.n = int(lib().rfx_reads_count(handle))
        return self

    @property
    def bases(self) -> int:
        return int(lib().rfx_reads_bases(self._h))

    @property
    def device_bytes(self) -> int:
        return int(lib().rfx_reads_device_bytes(self._h))

    def get(self, want_good: bool = True, want_acgt: bool = True):
        """Download the packed arrays: dict codes / acgt / good / word_off / len."""
        nw = int(lib().rfx_reads_words(self._h))
        out = {"codes": np.zeros(max(nw, 1), np.uint64), "acgt": np.zeros(max(nw, 1), np.uint32) if want_acgt else None,
               "good": np.zeros(max(nw, 1), np.uint32) if want_good else None,
               "word_off": np.zeros(self.n + 1, np.uint32), "len": np.zeros(max(self.n, 1), np.uint32)}
        _check(lib().rfx_reads_get(self._h, _p(out["codes"], u64p), _p(out["acgt"], u32p), _p(out["good"], u32p),
                                   _p(out["word_off"], u32p), _p(out["len"], u32p)), "rfx_reads_get")
        for k_ in ("codes", "acgt", "good"):
            if out[k_] is not None:
                out[k_] = out[k_][:nw]
        out["len"] = out["len"][:self.n]
        return out

    def free(self):
        if self._h:
            lib().rfx_reads_free(self._h)
            self._h = None


class Records:
    def __init__(self, ctx: Context, handle):
        if not handle:
            raise RufusError("records: " + lib().rfx_last_error().decode())
        self.ctx, self._h = ctx, handle

    def __len__(self):
        return int(lib().rfx_records_size(self._h))

    @property
    def k(self):
        return lib().rfx_records_k(self._h)

    @property
    def lsize(self):
        return lib().rfx_records_lsize(self._h)

    def payload(self, counter_len: int = 4) -> bytes:
        n = len(self) * ((2 * self.k + 7) // 8 + counter_len)
        buf = np.zeros(max(n, 1), dtype=np.uint8)
        _check(lib().rfx_records_payload(self._h, buf.ctypes.data, n, counter_len), "rfx_records_payload")
        return buf[:n].tobytes()

    def payload_range(self, first: int, n: int, counter_len: int = 4) -> bytes:
        nb = n * ((2 * self.k + 7) // 8 + counter_len)
        buf = np.zeros(max(nb, 1), dtype=np.uint8)
        _check(lib().rfx_records_payload_range(self._h, first, n, buf.ctypes.data, nb, counter_len),
               "rfx_records_payload_range")
        return buf[:nb].tobytes()

    def get(self):
        n = len(self)
        keys, counts, pos = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint64)
        _check(lib().rfx_records_get(self._h, _p(keys, u64p), _p(counts, u32p), _p(pos, u64p)), "rfx_records_get")
        return keys, counts, pos

    def histo(self) -> np.ndarray:
        h = np.zeros(HISTO_BINS, dtype=np.uint64)
        _check(lib().rfx_records_histo(self._h, _p(h, u64p)), "rfx_records_histo")
        return h

    def verify(self, min_count: int = 0, max_count: int = 0xFFFFFFFF) -> dict:
        """Device-side invariants: strict (pos,key) order, pos == M * key, counts in range (all 0 when correct)."""
        o = np.zeros(4, dtype=np.uint64)
        _check(lib().rfx_records_verify(self._h, min_count, max_count, _p(o, u64p)), "rfx_records_verify")
        return {"bad_order": int(o[0]), "bad_pos": int(o[1]), "bad_count": int(o[2]), "sum_counts": int(o[3])}

    def checksum(self) -> tuple:
        """(sum of mix(key) * count, sum of mix(key)) mod 2^64 -- rfx_records_checksum: adds up over shards / slices."""
        o = np.zeros(2, dtype=np.uint64)
        _check(lib().rfx_records_checksum(self._h, _p(o, u64p)), "rfx_records_checksum")
        return int(o[0]), int(o[1])

    def query(self, keys: np.ndarray) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=np.uint32)
        _check(lib().rfx_query(self._h, _p(keys, u64p), len(keys), _p(out, u32p)), "rfx_query")
        return out

    def dev_ptrs(self):
        L = lib()
        return L.rfx_records_dev_keys(self._h), L.rfx_records_dev_counts(self._h), L.rfx_records_dev_pos(self._h)

    @classmethod
    def load(cls, ctx: Context, k: int, lsize: int, cols: np.ndarray, payload: bytes, counter_len: int = 4):
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        rl = (2 * k + 7) // 8 + counter_len
        n = len(payload) // rl
        buf = np.frombuffer(payload, dtype=np.uint8)
        h = lib().rfx_records_load(ctx._h, k, lsize, _p(cols, u64p), buf.ctypes.data if n else None, n, counter_len)
        return cls(ctx, h)

    @classmethod
    def load_fd(cls, ctx: Context, k: int, lsize: int, cols: np.ndarray, fd: int, offset: int, n: int,
                counter_len: int = 4):
        """n records at byte `offset` of an open .Jhash file, streamed (no whole copy of the payload anywhere)."""
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        h = lib().rfx_records_load_fd(ctx._h, k, lsize, _p(cols, u64p), fd, offset, n, counter_len)
        if not h:
            raise RufusError("rfx_records_load_fd: " + lib().rfx_last_error().decode())
        return cls(ctx, h)

    @classmethod
    def from_dev(cls, ctx: Context, k: int, lsize: int, cols: np.ndarray, d_keys: int, d_counts: int, n: int):
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        return cls(ctx, lib().rfx_records_from_dev(ctx._h, k, lsize, _p(cols, u64p), d_keys, d_counts, n))

    def free(self):
        if self._h:
            lib().rfx_records_free(self._h)
            self._h = None


class RunMaps:
    """Run maps of one sample's read blocks (rfx_runmaps_*): written by the first shard pass that adds a big block,
    replayed by the later ones instead of hashing the block again.  Free it before the blocks."""

    def __init__(self, ctx: Context, budget_bytes: int = 0, pooled: bool = False):
        """pooled: the budget is allocated now, in one piece, and the maps are cut out of it."""
        self.ctx = ctx
        self._h = (lib().rfx_runmaps_create_pooled if pooled else lib().rfx_runmaps_create)(ctx._h, int(budget_bytes))
        if not self._h:
            raise RufusError("rfx_runmaps_create failed: " + lib().rfx_last_error().decode())

    def bytes(self) -> int:
        return int(lib().rfx_runmaps_bytes(self._h)) if self._h else 0

    def blocks(self) -> int:
        return int(lib().rfx_runmaps_blocks(self._h)) if self._h else 0

    def drop(self, reads: "ReadBlock"):
        _check(lib().rfx_runmaps_drop(self._h, reads._h), "rfx_runmaps_drop")

    def clear(self):
        _check(lib().rfx_runmaps_clear(self._h), "rfx_runmaps_clear")

    def free(self):
        if self._h:
            lib().rfx_runmaps_free(self._h)
            self._h = None


class CountTable:
    """``jellyfish count`` state: exact canonical k-mer counts in an HBM hash table."""

    def __init__(self, ctx: Context, k: int, size: int, canonical: bool = True, capacity: int = 0, pos_lo: int = 0,
                 pos_hi: int = 0, mode: int = COUNT_AUTO):
        self.ctx, self.k, self.lsize, self.canonical = ctx, k, ceil_log2(size), canonical
        self._h = lib().rfx_count_begin(ctx._h, k, int(canonical), self.lsize, capacity, pos_lo, pos_hi)
        if not self._h:
            raise RufusError("rfx_count_begin failed: " + lib().rfx_last_error().decode())
        if mode != COUNT_AUTO:
            _check(lib().rfx_count_set_mode(self._h, mode), "rfx_count_set_mode")

    def add(self, reads: ReadBlock):
        _check(lib().rfx_count_add(self._h, reads._h), "rfx_count_add")

    def add_pairs_dev(self, d_keys: int, d_counts: int, n: int):
        _check(lib().rfx_count_add_pairs_dev(self._h, d_keys, d_counts, n), "rfx_count_add_pairs_dev")

    def set_passes(self, passes: int = 0):
        """Defer the adds: finish() runs `passes` minimizer-shard passes over the (still alive) blocks (0 = plan)."""
        _check(lib().rfx_count_set_passes(self._h, passes), "rfx_count_set_passes")

    def set_peers(self, peers: int, index: int):
        """Table `index` of a group of tables on several devices (rfx_peers_create); after set_passes, before add."""
        _check(lib().rfx_count_set_peers(self._h, peers, index), "rfx_count_set_peers")

    def set_shard(self, shard: int, n_shards: int):
        """Count only the k-mers of minimizer shard `shard` of `n_shards` (before the first add)."""
        _check(lib().rfx_count_set_shard(self._h, shard, n_shards), "rfx_count_set_shard")

    def set_early(self, on: bool = True):
        """Big blocks added from now on are also cut for shard s + 1 (rfx_count_set_early): one hashing pass for two shards."""
        _check(lib().rfx_count_set_early(self._h, int(on)), "rfx_count_set_early")

    def early_segments(self) -> int:
        return int(lib().rfx_count_early_segments(self._h))

    def adopt_early(self, other: "CountTable"):
        """Take the early segments `other` (the table of the shard before this one) holds for this shard."""
        _check(lib().rfx_count_adopt_early(self._h, other._h), "rfx_count_adopt_early")

    def set_runmaps(self, store: "RunMaps"):
        """Share a store of run maps with the tables of the sample's other shard passes (rfx_count_set_runmaps)."""
        _check(lib().rfx_count_set_runmaps(self._h, store._h if store is not None else None), "rfx_count_set_runmaps")

    def prepare_maps(self, blocks) -> None:
        """The run maps of `blocks` (Reads the table is about to add) with one wait for the device (rfx_count_prepare_maps)."""
        arr = (C.c_void_p * max(1, len(blocks)))(*[b._h for b in blocks])
        _check(lib().rfx_count_prepare_maps(self._h, arr, len(blocks)), "rfx_count_prepare_maps")

    def prefetch_maps(self, blocks) -> int:
        """Queue the run maps of `blocks` (another sample's, counted next) on the ctx's second stream; returns at once
        with the number of launches queued (rfx_count_prefetch_maps)."""
        arr = (C.c_void_p * max(1, len(blocks)))(*[b._h for b in blocks])
        n = int(lib().rfx_count_prefetch_maps(self._h, arr, len(blocks)))
        if n < 0:
            _check(n, "rfx_count_prefetch_maps")
        return n

    def replayed(self) -> int:
        return int(lib().rfx_count_replayed(self._h))

    def segments(self):
        """[(d_records, d_bin_start, bins, n_records)] of the MSP record segments held (device pointers)."""
        n = lib().rfx_count_segments(self._h)
        if n < 0:
            _check(n, "rfx_count_segments")
        out = []
        for i in range(n):
            dr, db, bins, nr = C.c_void_p(0), C.c_void_p(0), C.c_uint32(0), C.c_uint64(0)
            _check(lib().rfx_count_segment_get(self._h, i, C.byref(dr), C.byref(db), C.byref(bins), C.byref(nr)),
                   "rfx_count_segment_get")
            out.append((dr.value or 0, db.value or 0, bins.value, nr.value))
        return out

    def n_segments(self) -> int:
        """Segments held so far (no synchronisation)."""
        n = lib().rfx_count_segments(self._h)
        if n < 0:
            _check(n, "rfx_count_segments")
        return n

    def segment(self, i: int):
        """(d_records, d_bin_start, bins, n_records) of segment i; waits for the adds queued so far."""
        dr, db, bins, nr = C.c_void_p(0), C.c_void_p(0), C.c_uint32(0), C.c_uint64(0)
        _check(lib().rfx_count_segment_get(self._h, i, C.byref(dr), C.byref(db), C.byref(bins), C.byref(nr)),
               "rfx_count_segment_get")
        return (dr.value or 0, db.value or 0, bins.value, nr.value)

    def add_records_dev(self, d_records: int, n_records: int, d_bin_start: int, bins: int, d_ext: int = 0):
        """d_ext: the 32-bit plane of the records (k = 26 .. 31), grouped like them."""
        _check(lib().rfx_count_add_records_ext_dev(self._h, d_records, d_ext or None, n_records, d_bin_start, bins),
               "rfx_count_add_records_dev")

    def adopt_records_dev(self, d_records: int, n_records: int, d_bin_start: int, bins: int, d_ext: int = 0):
        """add_records_dev without the copy: the arrays must stay alive until finish() / free()."""
        _check(lib().rfx_count_adopt_records_dev(self._h, d_records, d_ext or None, n_records, d_bin_start, bins),
               "rfx_count_adopt_records_dev")

    def segment_ext(self, i: int) -> int:
        """Device pointer of the 32-bit plane of segment i (0 for k <= 25)."""
        p = C.c_void_p(0)
        _check(lib().rfx_count_segment_ext(self._h, i, C.byref(p)), "rfx_count_segment_ext")
        return p.value or 0

    def stats(self):
        d, c, m = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().rfx_count_stats(self._h, C.byref(d), C.byref(c), C.byref(m)), "rfx_count_stats")
        return {"distinct": d.value, "capacity": c.value, "max_displacement": m.value}

    def finish(self, lower: int = 0, upper: int = 2**64 - 1, want_histo: bool = False):
        h = np.zeros(HISTO_BINS, dtype=np.uint64) if want_histo else None
        rec = Records(self.ctx, lib().rfx_count_finish(self._h, lower, upper, _p(h, u64p)))
        return (rec, h) if want_histo else rec

    def finish_begin(self, lower: int = 0, upper: int = 2**64 - 1, want_histo: bool = False):
        """Queue the finish; returns a handle for finish_end().  Several tables can be queued before the
        first finish_end() waits."""
        h = np.zeros(HISTO_BINS, dtype=np.uint64) if want_histo else None
        return (lib().rfx_count_finish_begin(self._h, lower, upper, _p(h, u64p)), h)

    def finish_end(self, handle):
        f, h = handle
        rec = Records(self.ctx, lib().rfx_count_finish_end(f))
        return (rec, h) if h is not None else rec

    def free(self):
        if self._h:
            lib().rfx_count_free(self._h)
            self._h = None


def merge_unique(ctx: Context, files, min_count: int = 5):
    arr = (C.c_void_p * len(files))(*[f._h for f in files])
    n = C.c_uint64(0)
    rc = lib().rfx_merge_unique(ctx._h, arr, len(files), min_count, None, None, 0, C.byref(n))
    if rc not in (0, E_RANGE):
        _check(rc, "rfx_merge_unique")
    keys, counts = np.zeros(max(n.value, 1), np.uint64), np.zeros(max(n.value, 1), np.uint32)
    _check(lib().rfx_merge_unique(ctx._h, arr, len(files), min_count, _p(keys, u64p), _p(counts, u32p), len(keys),
                                  C.byref(n)), "rfx_merge_unique")
    return keys[:n.value], counts[:n.value]


_uniq_buf = {}


def unique_to_subject(ctx: Context, subject: Records, others, min_cov: int, max_cov: int, min_count: int = 5):
    arr = (C.c_void_p * max(1, len(others)))(*[f._h for f in others])
    # mutant lists are tiny next to the record arrays: start from a reused 64 K buffer, enlarge on demand
    cap = _uniq_buf.get("cap", 1 << 16)
    while True:
        if _uniq_buf.get("cap") != cap or "k" not in _uniq_buf:
            _uniq_buf.update(cap=cap, k=np.zeros(cap, np.uint64), c=np.zeros(cap, np.uint32))
        keys, counts = _uniq_buf["k"], _uniq_buf["c"]
        n = C.c_uint64(0)
        rc = lib().rfx_unique_to_subject(ctx._h, subject._h, arr, len(others), min_count, min_cov, max_cov,
                                         _p(keys, u64p), _p(counts, u32p), cap, C.byref(n))
        if rc == E_RANGE and n.value > cap:
            cap = int(n.value)
            continue
        _check(rc, "rfx_unique_to_subject")
        return keys[:n.value].copy(), counts[:n.value].copy()


def records_subtract(ctx: Context, a: Records, others, min_count: int = 0, max_count: int = 0xFFFFFFFF) -> Records:
    """Records of `a` with min_count <= count <= max_count that occur in none of `others` (device resident)."""
    arr = (C.c_void_p * max(1, len(others)))(*[f._h for f in others])
    return Records(ctx, lib().rfx_records_subtract(ctx._h, a._h, arr, len(others), min_count, max_count))


OVL_SAM, OVL_CONTIG, OVL_REGION = 0, 1, 2


def overlap_score(ctx: Context, a: bytes, cands, min_pct: float, min_ovl: int, variant: int = OVL_SAM) -> np.ndarray:
    """rows of (phase-1 score, overlap, perfect, full score, overlap) per candidate."""
    nb = len(cands)
    arr = (C.c_char_p * max(nb, 1))(*cands)
    lens = (C.c_int * max(nb, 1))(*[len(x) for x in cands])
    out = np.zeros((max(nb, 1), 5), dtype=np.int32)
    _check(lib().rfx_overlap_score(ctx._h, a, len(a), arr, lens, nb, min_pct, min_ovl, variant,
                                   out.ctypes.data_as(C.POINTER(C.c_int))), "rfx_overlap_score")
    return out[:nb]


def model_residuals(ctx: Context, histo, cands, log_resid: bool, inflection: int, max_copy: int = 5) -> np.ndarray:
    """Residuals of candidate coverage models (rows of (SC, stdev, factor, skew, power)) against a histogram:
    testModelLog / testModel of the reference's ModelDist (src/ModelDist.cpp:72-318), all candidates in one pass."""
    histo = np.ascontiguousarray(histo, dtype=np.int64)
    cands = np.ascontiguousarray(cands, dtype=np.float64).reshape(-1, 5)
    out = np.zeros(len(cands), dtype=np.float64)
    _check(lib().rfx_model_residuals(ctx._h, _p(histo, C.POINTER(C.c_int64)), len(histo), cands.ctypes.data, len(cands),
                                     int(bool(log_resid)), inflection, max_copy, _p(out, C.POINTER(C.c_double))),
           "rfx_model_residuals")
    return out


def model_tables(ctx: Context, n: int, model):
    """(dist[n][cols + 1], rowtot[n]) of one model as ModelDist's main() tabulates it (src/ModelDist.cpp:716-772)."""
    model = np.ascontiguousarray(model, dtype=np.float64).reshape(5)
    cols = C.c_uint32(0)
    rc = lib().rfx_model_tables(ctx._h, n, model.ctypes.data, C.byref(cols), None, 0, None)
    if rc != -7:  # RFX_E_RANGE: the size query
        _check(rc, "rfx_model_tables")
    dist = np.zeros((n, cols.value + 1), dtype=np.float64)
    rowtot = np.zeros(n, dtype=np.float64)
    _check(lib().rfx_model_tables(ctx._h, n, model.ctypes.data, C.byref(cols), _p(dist, C.POINTER(C.c_double)), dist.size,
                                  _p(rowtot, C.POINTER(C.c_double))), "rfx_model_tables")
    return dist, rowtot


class MutantSet:
    def __init__(self, ctx: Context, fwd_keys: np.ndarray, k: int):
        fwd_keys = np.ascontiguousarray(fwd_keys, dtype=np.uint64)
        self.ctx, self.k = ctx, k
        self._h = lib().rfx_set_build(ctx._h, _p(fwd_keys, u64p), len(fwd_keys), k)
        if not self._h:
            raise RufusError("rfx_set_build failed: " + lib().rfx_last_error().decode())

    def filter(self, reads: ReadBlock, thresh: int = 1, last_base_skipped: bool = True, want_hits: bool = True,
               want_mask: bool = True):
        hits = np.zeros(max(reads.n, 1), np.uint32) if want_hits else None
        mask = np.zeros((reads.n + 63) // 64 or 1, np.uint64) if want_mask else None
        n = C.c_uint64(0)
        _check(lib().rfx_filter(self._h, reads._h, thresh, int(last_base_skipped), _p(hits, u32p), _p(mask, u64p),
                                C.byref(n)), "rfx_filter")
        return (hits[:reads.n] if want_hits else None), mask, n.value

    def filter_many(self, blocks, thresh: int = 1, last_base_skipped: bool = True):
        """The hit masks of several blocks with one wait for the device (rfx_filter_many): [(mask, n_hit_reads)]."""
        # The masks land in page-locked host memory (one buffer per ctx, kept and grown on demand): a read-back into
        # pageable memory is staged by the runtime and waits for the kernel before it -- the device would idle between two
        # blocks after all.  The arrays returned are views of that buffer: valid until the next filter_many of this ctx.
        words = [(b.n + 63) // 64 or 1 for b in blocks]
        buf = self.ctx.pinned_u64(sum(words))
        masks, at = [], 0
        for w in words:
            masks.append(buf[at:at + w])
            at += w
        arr = (C.c_void_p * max(1, len(blocks)))(*[b._h for b in blocks])
        mp = (u64p * max(1, len(blocks)))(*[m.ctypes.data_as(u64p) for m in masks])
        nh = np.zeros(max(1, len(blocks)), np.uint64)
        _check(lib().rfx_filter_many(self._h, arr, len(blocks), thresh, int(last_base_skipped), mp, _p(nh, u64p)), "rfx_filter_many")
        return [(m, int(n)) for m, n in zip(masks, nh)]

    def annotate(self, reads: ReadBlock) -> np.ndarray:
        cov = np.zeros(max(reads.bases, 1), dtype=np.uint32)
        _check(lib().rfx_annotate(self._h, reads._h, _p(cov, u32p)), "rfx_annotate")
        return cov[:reads.bases]

    def free(self):
        if self._h:
            lib().rfx_set_free(self._h)
            self._h = None
