"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_VARIANT = ""  # "" = liboracle.so (-O2 -mpopcnt), "avx2" = liboracle_avx2.so (-O3 -mavx2): same source, same results

AMINO, DNA, RNA = 1, 2, 3


class OrcIndex(C.Structure):
    _fields_ = [
        ("alphabet", C.c_uint8),
        ("saRatio", C.c_uint8),
        ("seedK", C.c_uint8),
        ("saWidth", C.c_uint8),
        ("ownsArrays", C.c_uint8),
        ("blockBytes", C.c_uint32),
        ("bwtLength", C.c_uint64),
        ("numBlocks", C.c_uint64),
        ("blocks", C.POINTER(C.c_uint8)),
        ("prefixSums", C.c_uint64 * 24),
        ("seedLen", C.c_uint64),
        ("seedTable", C.POINTER(C.c_uint64)),
        ("saBytes", C.c_uint64),
        ("sa", C.POINTER(C.c_uint8)),
        ("fullSa", C.POINTER(C.c_uint64)),
    ]


class OrcTally(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("queries", "seeded", "steps", "blocks", "hits", "lfSteps", "chars")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _so_name():
    return "liboracle_avx2.so" if _VARIANT == "avx2" else "liboracle.so"


def build(force=False):
    so = os.path.join(_HERE, _so_name())
    src = [os.path.join(_HERE, f) for f in ("awfm_oracle.c", "awfm_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", _so_name()])
    return so


def set_variant(name):
    """switch the build the module calls into ("" or "avx2"); indices made before the switch stay usable (plain C
    structs), the next call goes to the other library"""
    global _LIB, _VARIANT
    if name not in ("", "avx2"):
        raise ValueError(name)
    if name != _VARIANT:
        _VARIANT = name
        _LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, _so_name())
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    u8p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    P = C.POINTER(OrcIndex)
    sig = {
        "orc_nuc_ascii_to_index": (C.c_uint8, [C.c_uint8]),
        "orc_amino_ascii_to_index": (C.c_uint8, [C.c_uint8]),
        "orc_nuc_sanitize": (C.c_uint8, [C.c_uint8]),
        "orc_amino_sanitize": (C.c_uint8, [C.c_uint8]),
        "orc_nuc_index_to_code": (C.c_uint8, [C.c_uint8]),
        "orc_amino_index_to_code": (C.c_uint8, [C.c_uint8]),
        "orc_nuc_code_to_index": (C.c_uint8, [C.c_uint8]),
        "orc_amino_code_to_index": (C.c_uint8, [C.c_uint8]),
        "orc_letter_is_ambiguous": (C.c_int, [C.c_uint8, C.c_uint8]),
        "orc_masked_popcount": (C.c_uint32, [u8p, C.c_uint8]),
        "orc_sa_width": (C.c_uint8, [C.c_uint64]),
        "orc_sa_num_samples": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "orc_sa_packed_bytes": (C.c_uint64, [C.c_uint64, C.c_uint8]),
        "orc_sa_pack": (None, [u64p, C.c_uint64, C.c_uint8, u8p]),
        "orc_sa_get": (C.c_uint64, [u8p, C.c_uint8, C.c_uint64]),
        "orc_build": (P, [u8p, C.c_uint64, C.c_uint8, C.c_uint8, C.c_uint8]),
        "orc_wrap": (P, [C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "orc_free": (None, [P]),
        "orc_suffix_array": (None, [u8p, C.c_uint64, u64p]),
        "orc_occ": (C.c_uint64, [P, C.c_uint8, C.c_uint64]),
        "orc_occ_vector": (None, [P, C.c_uint64, C.c_uint8, u8p]),
        "orc_step": (None, [P, u64p, u64p, C.c_uint8]),
        "orc_letter_at": (C.c_uint8, [P, C.c_uint64]),
        "orc_lf": (C.c_uint64, [P, C.c_uint64]),
        "orc_range_for_string": (None, [P, C.c_char_p, C.c_uint64, u64p, u64p]),
        "orc_range_length": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "orc_locate_one": (C.c_uint64, [P, C.c_uint64, u64p]),
        "orc_batch_search": (None, [P, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(OrcTally), C.c_int]),
        "orc_batch_locate": (None, [P, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.POINTER(OrcTally), C.c_int]),
        "orc_fnv1a": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint64]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def pack_queries(queries):
    """list of bytes -> (chars uint8[sum], offsets uint64[n+1])"""
    lens = np.fromiter((len(q) for q in queries), dtype=np.uint64, count=len(queries))
    offsets = np.zeros(len(queries) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    chars = np.frombuffer(b"".join(queries), dtype=np.uint8).copy() if len(queries) else np.zeros(0, np.uint8)
    if chars.size == 0:
        chars = np.zeros(1, np.uint8)
    return chars, offsets


class Index:
    """Oracle index: either built here from text, or wrapping external arrays."""

    def __init__(self, ptr, keep=None):
        self.ptr = ptr
        self._keep = keep

    @classmethod
    def from_text(cls, text, alphabet, sa_ratio, seed_k):
        t = np.frombuffer(bytes(text), dtype=np.uint8).copy()
        if t.size == 0:
            t = np.zeros(1, np.uint8)
            n = 0
        else:
            n = t.size
        p = lib().orc_build(_u8(t), n, alphabet, sa_ratio, seed_k)
        return cls(p)

    @classmethod
    def wrap(cls, alphabet, sa_ratio, seed_k, bwt_length, blocks, prefix_sums, seed_table, sa):
        """blocks/seed_table/sa: numpy arrays or raw addresses (ints) in reference layout."""
        def addr(x):
            return x if isinstance(x, int) else x.ctypes.data
        ps = np.ascontiguousarray(prefix_sums, dtype=np.uint64)
        p = lib().orc_wrap(alphabet, sa_ratio, seed_k, bwt_length, addr(blocks), ps.ctypes.data, addr(seed_table),
                           addr(sa))
        return cls(p, keep=(blocks, ps, seed_table, sa))

    def __del__(self):
        try:
            if self.ptr:
                lib().orc_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    # --- array views (reference layout) ---
    @property
    def c(self):
        return self.ptr.contents

    @property
    def bwt_length(self):
        return int(self.c.bwtLength)

    def blocks(self):
        c = self.c
        return np.ctypeslib.as_array(c.blocks, shape=(int(c.numBlocks) * int(c.blockBytes),)).copy()

    def prefix_sums(self):
        n = (20 if self.c.alphabet == AMINO else 4) + 2
        return np.array(list(self.c.prefixSums)[:n], dtype=np.uint64)

    def seed_table(self):
        c = self.c
        return np.ctypeslib.as_array(c.seedTable, shape=(int(c.seedLen), 2)).copy()

    def packed_sa(self):
        c = self.c
        return np.ctypeslib.as_array(c.sa, shape=(int(c.saBytes),)).copy()

    def full_sa(self):
        c = self.c
        return np.ctypeslib.as_array(c.fullSa, shape=(int(c.bwtLength),)).copy()

    # --- primitives ---
    def occ(self, letter, q):
        return int(lib().orc_occ(self.ptr, letter, q))

    def step(self, sp, ep, letter):
        a, b = C.c_uint64(sp), C.c_uint64(ep)
        lib().orc_step(self.ptr, C.byref(a), C.byref(b), letter)
        return int(a.value), int(b.value)

    def letter_at(self, p):
        return int(lib().orc_letter_at(self.ptr, p))

    def lf(self, p):
        return int(lib().orc_lf(self.ptr, p))

    def range_for_string(self, kmer):
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().orc_range_for_string(self.ptr, bytes(kmer), len(kmer), C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def locate_one(self, p):
        return int(lib().orc_locate_one(self.ptr, p, None))

    # --- batch (awFmParallelSearchCount / Locate semantics) ---
    def batch_search(self, chars, offsets, threads=1):
        n = len(offsets) - 1
        chars = np.ascontiguousarray(chars, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sp = np.zeros(n, np.uint64)
        ep = np.zeros(n, np.uint64)
        cnt = np.zeros(n, np.uint32)
        t = OrcTally()
        lib().orc_batch_search(self.ptr, chars.ctypes.data, offsets.ctypes.data, n, sp.ctypes.data, ep.ctypes.data,
                               cnt.ctypes.data, C.byref(t), threads)
        return sp, ep, cnt, t.as_dict()

    def batch_locate(self, sp, ep, threads=1):
        n = len(sp)
        lens = np.where(sp <= ep, ep - sp + np.uint64(1), np.uint64(0)).astype(np.uint64)
        hit_off = np.zeros(n + 1, np.uint64)
        np.cumsum(lens, out=hit_off[1:])
        pos = np.zeros(max(int(hit_off[-1]), 1), np.uint64)
        t = OrcTally()
        lib().orc_batch_locate(self.ptr, sp.ctypes.data, ep.ctypes.data, n, hit_off.ctypes.data, pos.ctypes.data,
                               C.byref(t), threads)
        return hit_off, pos[: int(hit_off[-1])], t.as_dict()

    def search_list(self, queries, threads=1):
        chars, offsets = pack_queries(queries)
        return self.batch_search(chars, offsets, threads)


def fnv1a(arr, seed=0):
    a = np.ascontiguousarray(arr)
    return int(lib().orc_fnv1a(a.ctypes.data, a.nbytes, seed))
