"""Python mirror of the AwFmIndex.h interface, calling libawfmindex_amd.so through its C ABI.

Names and argument meaning follow the reference API (src/AwFmIndex.h) so tests read
like the reference's own: create_index -> awFmCreateIndex, KmerSearchList ->
awFmCreateKmerSearchList, parallel_search_count / parallel_search_locate ->
awFmParallelSearchCount / awFmParallelSearchLocate.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (AwFmAlphabetAmino, AwFmAlphabetDna, AwFmAlphabetRna, AwFmFileReadOkay, AwFmFileWriteOkay,  # noqa: F401
                   AwFmIllegalPositionError, AwFmSuccess)


class AwFmError(RuntimeError):
    def __init__(self, what, rc):
        msg = _lib.lib().awfmGpuLastError().decode(errors="replace")
        super().__init__(f"{what} failed with AwFmReturnCode {rc}" + (f": {msg}" if msg else ""))
        self.rc = rc


def _check(what, rc, ok=(AwFmSuccess, AwFmFileReadOkay, AwFmFileWriteOkay)):
    if rc not in ok:
        raise AwFmError(what, rc)
    return rc


class Index:
    """struct AwFmIndex* owner"""

    def __init__(self, ptr):
        self.ptr = ptr

    @property
    def c(self):
        return self.ptr.contents

    @property
    def bwt_length(self):
        return int(self.c.bwtLength)

    @property
    def sa_width(self):
        """bits per sampled suffix-array value (ref src/AwFmSuffixArray.c:12-18)"""
        return int(self.c.suffixArray.valueBitWidth)

    @property
    def is_amino(self):
        return self.c.config.alphabetType == AwFmAlphabetAmino

    @property
    def num_blocks(self):
        return 1 + (self.bwt_length - 1) // 256

    # host arrays in reference layout (views, valid while the index lives)
    def blocks(self):
        nbytes = self.num_blocks * (352 if self.is_amino else 160)
        return np.ctypeslib.as_array(C.cast(self.c.bwtBlockList, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def prefix_sums(self):
        return np.ctypeslib.as_array(self.c.prefixSums, shape=((20 if self.is_amino else 4) + 2,))

    def seed_table(self):
        n = (20 if self.is_amino else 4) ** int(self.c.config.kmerLengthInSeedTable)
        return np.ctypeslib.as_array(C.cast(self.c.kmerSeedTable, C.POINTER(C.c_uint64)), shape=(n, 2))

    def packed_sa(self):
        sa = self.c.suffixArray
        if not sa.values:
            return None
        return np.ctypeslib.as_array(sa.values, shape=(int(sa.compressedByteLength),))

    def find_search_range_for_string(self, kmer):
        r = _lib.lib().awFmFindSearchRangeForString(self.ptr, bytes(kmer), len(kmer))
        return int(r.startPtr), int(r.endPtr)

    def local_position(self, global_position):
        """awFmGetLocalSequencePositionFromIndexPosition -> (sequence number, position in it)"""
        seq, loc = C.c_size_t(0), C.c_size_t(0)
        rc = _lib.lib().awFmGetLocalSequencePositionFromIndexPosition(self.ptr, global_position, C.byref(seq), C.byref(loc))
        _check("awFmGetLocalSequencePositionFromIndexPosition", rc)
        return int(seq.value), int(loc.value)

    def header(self, sequence_number):
        """awFmGetHeaderStringFromSequenceNumber"""
        buf, n = C.c_char_p(), C.c_size_t(0)
        rc = _lib.lib().awFmGetHeaderStringFromSequenceNumber(self.ptr, sequence_number, C.byref(buf), C.byref(n))
        _check("awFmGetHeaderStringFromSequenceNumber", rc)
        return C.string_at(buf, n.value)

    def dealloc(self):
        if self.ptr:
            _lib.lib().awFmDeallocIndex(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.dealloc()
        except Exception:
            pass


def create_index(sequence, alphabet=AwFmAlphabetDna, sa_ratio=8, seed_k=8, keep_sa_in_memory=True,
                 store_sequence=False, file_src=None):
    """awFmCreateIndex (ref src/AwFmIndex.h:164-169)"""
    L = _lib.lib()
    cfg = _lib.AwFmIndexConfiguration(sa_ratio, seed_k, alphabet, keep_sa_in_memory, store_sequence)
    seq = np.frombuffer(bytes(sequence), dtype=np.uint8) if not isinstance(sequence, np.ndarray) else sequence
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    holder = seq if seq.size else np.zeros(1, np.uint8)
    if file_src is None:
        file_src = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"awfm_{os.getpid()}_{id(seq):x}.awfmi")
    out = C.POINTER(_lib.AwFmIndex)()
    rc = L.awFmCreateIndex(C.byref(out), C.byref(cfg), holder.ctypes.data, seq.size, file_src.encode())
    _check("awFmCreateIndex", rc, ok=(AwFmFileWriteOkay,))
    ix = Index(out)
    ix.file_src = file_src
    return ix


def gpu_create_index(sequence, alphabet=AwFmAlphabetDna, sa_ratio=8, seed_k=8, keep_sa_in_memory=True,
                     store_sequence=False, file_src=None, device=-1, on_device_length=None):
    """awfmGpuCreateIndex: same arrays as create_index, built on the GPU.  `sequence` is bytes / a numpy
    array, or a device address (int) together with on_device_length."""
    L = _lib.lib()
    cfg = _lib.AwFmIndexConfiguration(sa_ratio, seed_k, alphabet, keep_sa_in_memory, store_sequence)
    out = C.POINTER(_lib.AwFmIndex)()
    if on_device_length is not None:
        rc = L.awfmGpuCreateIndex(C.byref(out), C.byref(cfg), int(sequence), on_device_length, 1,
                                  file_src.encode() if file_src else None, device)
    else:
        seq = np.frombuffer(bytes(sequence), dtype=np.uint8) if not isinstance(sequence, np.ndarray) else sequence
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        holder = seq if seq.size else np.zeros(1, np.uint8)
        rc = L.awfmGpuCreateIndex(C.byref(out), C.byref(cfg), holder.ctypes.data, seq.size, 0,
                                  file_src.encode() if file_src else None, device)
    _check("awfmGpuCreateIndex", rc, ok=(AwFmFileWriteOkay,))
    ix = Index(out)
    ix.file_src = file_src
    return ix


def create_index_from_fasta(fasta_src, alphabet=AwFmAlphabetDna, sa_ratio=8, seed_k=8, keep_sa_in_memory=True,
                            store_sequence=False, file_src=None):
    """awFmCreateIndexFromFasta (ref src/AwFmIndex.h:196-200)"""
    cfg = _lib.AwFmIndexConfiguration(sa_ratio, seed_k, alphabet, keep_sa_in_memory, store_sequence)
    if file_src is None:
        file_src = fasta_src + ".awfmi"
    out = C.POINTER(_lib.AwFmIndex)()
    rc = _lib.lib().awFmCreateIndexFromFasta(C.byref(out), C.byref(cfg), fasta_src.encode(), file_src.encode())
    _check("awFmCreateIndexFromFasta", rc, ok=(AwFmFileWriteOkay,))
    ix = Index(out)
    ix.file_src = file_src
    return ix


def read_index_from_file(file_src, keep_sa_in_memory=True):
    """awFmReadIndexFromFile (ref src/AwFmIndex.h:260-262)"""
    out = C.POINTER(_lib.AwFmIndex)()
    rc = _lib.lib().awFmReadIndexFromFile(C.byref(out), file_src.encode(), keep_sa_in_memory)
    _check("awFmReadIndexFromFile", rc, ok=(AwFmFileReadOkay,))
    ix = Index(out)
    ix.file_src = file_src
    return ix


class KmerSearchList:
    """struct AwFmKmerSearchList* owner; fill() sets kmerString/kmerLength like the reference tests do"""

    def __init__(self, capacity):
        self.ptr = _lib.lib().awFmCreateKmerSearchList(capacity)
        if not self.ptr:
            raise MemoryError("awFmCreateKmerSearchList")
        self._keep = None

    def fill(self, kmers):
        lst = self.ptr.contents
        assert len(kmers) <= lst.capacity
        bufs = [C.create_string_buffer(bytes(k), len(k)) if len(k) else C.create_string_buffer(1) for k in kmers]
        for i, (k, b) in enumerate(zip(kmers, bufs)):
            lst.kmerSearchData[i].kmerString = C.addressof(b)
            lst.kmerSearchData[i].kmerLength = len(k)
        lst.count = len(kmers)
        self._keep = bufs

    def counts(self):
        lst = self.ptr.contents
        return np.array([lst.kmerSearchData[i].count for i in range(lst.count)], dtype=np.uint32)

    def capacities(self):
        lst = self.ptr.contents
        return np.array([lst.kmerSearchData[i].capacity for i in range(lst.count)], dtype=np.uint32)

    def positions(self, i):
        d = self.ptr.contents.kmerSearchData[i]
        if not d.count:
            return np.zeros(0, np.uint64)
        return np.ctypeslib.as_array(d.positionList, shape=(d.count,)).astype(np.uint64)

    def dealloc(self):
        if self.ptr:
            _lib.lib().awFmDeallocKmerSearchList(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.dealloc()
        except Exception:
            pass


def parallel_search_count(index, search_list, num_threads=4):
    """awFmParallelSearchCount (ref src/AwFmIndex.h:400-403); returns nothing, like the reference"""
    _lib.lib().awFmParallelSearchCount(index.ptr, search_list.ptr, num_threads)


def parallel_search_locate(index, search_list, num_threads=4):
    """awFmParallelSearchLocate (ref src/AwFmIndex.h:364-367); returns the AwFmReturnCode"""
    return _lib.lib().awFmParallelSearchLocate(index.ptr, search_list.ptr, num_threads)


def pack_kmers(kmers, alphabet=AwFmAlphabetDna):
    """awfmPackKmers: uint8[n, L] fixed-length ASCII k-mers -> uint64[n] packed words (2 bits per nucleotide, 5 bits
    per amino acid, first character most significant).  Raises on a k-mer the packing cannot express."""
    kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
    n, length = kmers.shape
    out = np.zeros(n, np.uint64)
    bad = C.c_uint64(0)
    holder = kmers if kmers.size else np.zeros(1, np.uint8)
    rc = _lib.lib().awfmPackKmers(alphabet, holder.ctypes.data, length, n, out.ctypes.data, C.byref(bad))
    if rc != AwFmSuccess:
        raise ValueError(f"awfmPackKmers: k-mer {bad.value} cannot be packed (rc {rc})")
    return out


# enum AwFmGpuKernel (include/awfm_gpu.h)
AWFM_GPU_KERNEL_AUTO, AWFM_GPU_KERNEL_GROUP8, AWFM_GPU_KERNEL_GROUP4, AWFM_GPU_KERNEL_GROUP2, AWFM_GPU_KERNEL_GROUP1 = range(5)


class GpuIndex:
    """AwFmGpuIndex* owner: the device image plus the flat batch API of include/awfm_gpu.h"""

    def __init__(self, index, device=-1, acquire=False):
        """acquire=True reuses (or lazily creates) the image registered for `index` -- the one
        awFmParallelSearch* uses, e.g. the image a GPU-built index already has; it is then owned by the index"""
        L = _lib.lib()
        if L.awfmGpuDeviceCount() <= 0:
            raise RuntimeError("no HIP device: the search path is GPU only (no CPU fallback)")
        self.owned = not acquire
        if acquire:
            h = L.awfmGpuIndexAcquire(index.ptr)
            if not h:
                raise AwFmError("awfmGpuIndexAcquire", -1)
            self.handle = C.c_void_p(h)
        else:
            h = C.c_void_p()
            _check("awfmGpuIndexCreate", L.awfmGpuIndexCreate(index.ptr, device, C.byref(h)))
            self.handle = h
        self.index = index

    @property
    def device_bytes(self):
        return int(_lib.lib().awfmGpuIndexDeviceBytes(self.handle))

    def set_deep_seed(self, deep_k):
        """device-only deeper seed table (nucleotide); 0 drops it"""
        _check("awfmGpuIndexSetDeepSeed", _lib.lib().awfmGpuIndexSetDeepSeed(self.handle, deep_k))

    @property
    def deep_seed_k(self):
        """depth of the device-only deeper seed table of this image (0: none)"""
        return int(_lib.lib().awfmGpuIndexDeepSeedK(self.handle))

    @property
    def deep_seed_build(self):
        """(wall seconds, transient device bytes) of the construction of that table, whoever started it"""
        L = _lib.lib()
        return float(L.awfmGpuIndexDeepSeedBuildSeconds(self.handle)), int(L.awfmGpuIndexDeepSeedTransientBytes(self.handle))

    @property
    def deep_seed_alloc_s(self):
        """seconds of that construction spent inside hipMalloc"""
        return float(_lib.lib().awfmGpuIndexDeepSeedAllocSeconds(self.handle))

    def set_dense_sa(self, enable=True):
        """device-only full suffix array (32-bit entries) so that a locate is a single gather"""
        _check("awfmGpuIndexSetDenseSa", _lib.lib().awfmGpuIndexSetDenseSa(self.handle, int(bool(enable))))

    @property
    def has_dense_sa(self):
        return bool(_lib.lib().awfmGpuIndexHasDenseSa(self.handle))

    @property
    def dense_sa_build_s(self):
        return float(_lib.lib().awfmGpuIndexDenseSaBuildSeconds(self.handle))

    @property
    def length_tables(self):
        """(bytes, build seconds) of the device-only tables per k-mer length a mixed-length batch builds on first use; (0, 0.0): none"""
        return (int(_lib.lib().awfmGpuIndexLengthTableBytes(self.handle)), float(_lib.lib().awfmGpuIndexLengthTableBuildSeconds(self.handle)))

    def mixed_lookup_line_tally(self, d_chars, d_offsets, n):
        """what the lookup-first kernel of mixed-length batches has to read for this batch (awfmGpuMixedLookupLineTally)"""
        out = (C.c_uint64 * 8)()
        _check("awfmGpuMixedLookupLineTally", _lib.lib().awfmGpuMixedLookupLineTally(self.handle, d_chars, d_offsets, n, C.byref(out)))
        keys = ("length_table_lines", "deep_table_lines", "pair_level_lines", "nuc_level_lines", "kmers_alive_after_the_table",
                "kmers_with_hits", "general_kmers", "block_reads_executed")
        return {k: int(v) for k, v in zip(keys, out)}

    def set_pair_image(self, enable=True):
        """device-only pair image (two steps per block read); built by default with nucleotide images"""
        _check("awfmGpuIndexSetPairImage", _lib.lib().awfmGpuIndexSetPairImage(self.handle, int(bool(enable))))

    @property
    def has_pair_image(self):
        return bool(_lib.lib().awfmGpuIndexHasPairImage(self.handle))

    def set_kernel(self, kernel):
        """enum AwFmGpuKernel (include/awfm_gpu.h): AWFM_GPU_KERNEL_AUTO, or a fixed number of lanes per k-mer for the general
        kernel and the walk -- anything but AUTO / GROUP4 keeps a search away from the pair image and the device-only tables,
        i.e. runs the reference's letter-by-letter algorithm (what the differential fuzz compares everything else with)"""
        _lib.lib().awfmGpuIndexSetKernel(self.handle, kernel)

    def set_wide(self, wide=True):
        """64-bit BWT positions in the kernels although bwtLength < 2^32 (testing)"""
        _lib.lib().awfmGpuIndexSetWide(self.handle, int(bool(wide)))

    @property
    def is_wide(self):
        return bool(_lib.lib().awfmGpuIndexIsWide(self.handle))

    # host-buffer calls -------------------------------------------------
    def count_host(self, chars, offsets=None, fixed_length=0):
        chars = np.ascontiguousarray(chars, dtype=np.uint8)
        n = (len(offsets) - 1) if offsets is not None else chars.size // fixed_length
        ranges = np.zeros((n, 2), np.uint64)
        counts = np.zeros(n, np.uint32)
        off = np.ascontiguousarray(offsets, dtype=np.uint64) if offsets is not None else None
        holder = chars if chars.size else np.zeros(1, np.uint8)
        rc = _lib.lib().awfmGpuCountHost(self.handle, holder.ctypes.data, off.ctypes.data if off is not None else None,
                                         fixed_length, n, ranges.ctypes.data, counts.ctypes.data)
        _check("awfmGpuCountHost", rc)
        return ranges, counts

    def locate_host(self, chars, offsets=None, fixed_length=0):
        L = _lib.lib()
        chars = np.ascontiguousarray(chars, dtype=np.uint8)
        n = (len(offsets) - 1) if offsets is not None else chars.size // fixed_length
        ranges = np.zeros((n, 2), np.uint64)
        hit_off = np.zeros(n + 1, np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.uint64) if offsets is not None else None
        holder = chars if chars.size else np.zeros(1, np.uint8)
        pos_ptr = C.POINTER(C.c_uint64)()
        rc = L.awfmGpuLocateHost(self.handle, holder.ctypes.data, off.ctypes.data if off is not None else None,
                                 fixed_length, n, ranges.ctypes.data, hit_off.ctypes.data, C.byref(pos_ptr))
        _check("awfmGpuLocateHost", rc)
        total = int(hit_off[n])
        pos = np.ctypeslib.as_array(pos_ptr, shape=(total,)).copy() if total else np.zeros(0, np.uint64)
        L.free(C.cast(pos_ptr, C.c_void_p))
        return ranges, hit_off, pos

    def locate_host_windows(self, chars, offsets, sink):
        """awfmGpuLocateHostWindows: sink(user, query_begin, query_end, hit_begin, hit_end, positions) per window of the
        flat hit list; returns the hit offsets (uint64[n+1])"""
        chars = np.ascontiguousarray(chars, dtype=np.uint8)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(off) - 1
        hit_off = np.zeros(n + 1, np.uint64)
        holder = chars if chars.size else np.zeros(1, np.uint8)
        cb = _lib.HIT_WINDOW_SINK(sink)
        _check("awfmGpuLocateHostWindows", _lib.lib().awfmGpuLocateHostWindows(
            self.handle, holder.ctypes.data, off.ctypes.data, 0, n, None, hit_off.ctypes.data, cb, None))
        return hit_off

    # chunked pipeline on host buffers (awfm_gpu_stream.hip) ---------------------------
    def stream(self, kmers, kmer_length, locate=True, chunk=0, packed=True, threads=4, sink=None):
        """awfmGpuStreamPacked / awfmGpuStreamChars over a host array (numpy uint64[n] packed words, or uint8[n*L]
        ASCII; or an (address, n) pair for page-locked memory).  Without a sink the chunks are gathered:
        returns (counts uint32[n], positions uint64[total] or None)."""
        L = _lib.lib()
        if isinstance(kmers, tuple):
            address, n = kmers
        else:
            kmers = np.ascontiguousarray(kmers, dtype=np.uint64 if packed else np.uint8)
            n = kmers.size if packed else kmers.size // kmer_length
            address = kmers.ctypes.data if kmers.size else None
        counts = np.zeros(n, np.uint32)
        parts = []

        def gather(user, first, m, c, p, total):
            counts[first:first + m] = np.ctypeslib.as_array(c, shape=(m,))
            if locate and total:
                parts.append(np.ctypeslib.as_array(p, shape=(total,)).copy())
            return 0

        cb = _lib.CHUNK_SINK(sink or gather)
        fn = L.awfmGpuStreamPacked if packed else L.awfmGpuStreamChars
        _check(fn.__name__, fn(self.handle, address, kmer_length, n, chunk, int(bool(locate)), threads, cb, None))
        if sink is not None:
            return None
        return counts, (np.concatenate(parts) if parts else np.zeros(0, np.uint64)) if locate else None

    def stream_sparse(self, kmers, kmer_length, locate=True, chunk=0, packed=True, threads=4, sink=None):
        """awfmGpuStreamPackedSparse / awfmGpuStreamCharsSparse.  Without a sink the chunks are gathered: returns
        (hit_kmers uint64[m] batch-wide numbers, hit_offsets uint64[m+1], positions uint64[total] or None)."""
        L = _lib.lib()
        if isinstance(kmers, tuple):
            address, n = kmers
        else:
            kmers = np.ascontiguousarray(kmers, dtype=np.uint64 if packed else np.uint8)
            n = kmers.size if packed else kmers.size // kmer_length
            address = kmers.ctypes.data if kmers.size else None
        ids, lens, parts = [], [], []

        def gather(user, first, m, num, hit_kmers, hit_offsets, p, total):
            if num:
                ids.append(np.ctypeslib.as_array(hit_kmers, shape=(num,)).astype(np.uint64) + np.uint64(first))
                lens.append(np.diff(np.ctypeslib.as_array(hit_offsets, shape=(num + 1,))))
            if locate and total:
                parts.append(np.ctypeslib.as_array(p, shape=(total,)).copy())
            return 0

        cb = _lib.SPARSE_CHUNK_SINK(sink or gather)
        fn = L.awfmGpuStreamPackedSparse if packed else L.awfmGpuStreamCharsSparse
        _check(fn.__name__, fn(self.handle, address, kmer_length, n, chunk, int(bool(locate)), threads, cb, None))
        if sink is not None:
            return None
        hit_kmers = np.concatenate(ids) if ids else np.zeros(0, np.uint64)
        offsets = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.uint64) if lens else np.zeros(1, np.uint64)
        return hit_kmers, offsets, (np.concatenate(parts) if parts else np.zeros(0, np.uint64)) if locate else None

    def count_packed_host(self, packed, kmer_length):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        counts = np.zeros(packed.size, np.uint32)
        _check("awfmGpuCountPackedHost", _lib.lib().awfmGpuCountPackedHost(
            self.handle, packed.ctypes.data if packed.size else None, kmer_length, packed.size, counts.ctypes.data))
        return counts

    def locate_packed_host(self, packed, kmer_length):
        L = _lib.lib()
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        counts = np.zeros(packed.size, np.uint32)
        pos_ptr, total = C.POINTER(C.c_uint64)(), C.c_uint64(0)
        _check("awfmGpuLocatePackedHost", L.awfmGpuLocatePackedHost(
            self.handle, packed.ctypes.data if packed.size else None, kmer_length, packed.size, counts.ctypes.data,
            C.byref(pos_ptr), C.byref(total)))
        pos = np.ctypeslib.as_array(pos_ptr, shape=(total.value,)).copy() if total.value else np.zeros(0, np.uint64)
        L.free(C.cast(pos_ptr, C.c_void_p))
        return counts, pos

    def pack_device(self, d_chars, kmer_length, n, d_packed, stream=0):
        """awfmGpuPackKmers on device buffers; returns how many k-mers could not be expressed (the library call itself
        reports AwFmIllegalPositionError then: the packed words of such a batch must not be searched)"""
        bad = C.c_uint64(0)
        rc = _lib.lib().awfmGpuPackKmers(self.handle, d_chars, kmer_length, n, d_packed, C.byref(bad), stream or None)
        if rc == AwFmIllegalPositionError and bad.value:
            return int(bad.value)
        _check("awfmGpuPackKmers", rc)
        return int(bad.value)

    def unpack_device(self, d_packed, kmer_length, n, d_chars, stream=0):
        _check("awfmGpuUnpackKmers", _lib.lib().awfmGpuUnpackKmers(self.handle, d_packed, kmer_length, n, d_chars, stream or None))

    # device-pointer calls (addresses as ints, e.g. torch tensor.data_ptr()) -------------
    def search(self, d_chars, d_offsets, fixed_length, n, d_ranges, d_counts, stream=0):
        _check("awfmGpuSearch", _lib.lib().awfmGpuSearch(self.handle, d_chars, d_offsets or None, fixed_length, n,
                                                         d_ranges or None, d_counts or None, stream or None))

    def search_hits(self, d_chars, d_offsets, fixed_length, n, d_ranges, d_counts, stream=0):
        """awfmGpuSearchHits: like search(), but a query without hits only gets count 0 and some empty range"""
        _check("awfmGpuSearchHits", _lib.lib().awfmGpuSearchHits(self.handle, d_chars, d_offsets or None, fixed_length,
                                                                 n, d_ranges or None, d_counts or None, stream or None))

    def search_hits_sparse(self, d_chars, d_offsets, fixed_length, n, d_ranges, d_counts, stream=0):
        """awfmGpuSearchHitsSparse: counts for every query, ranges only for the queries with hits (the others' may stay
        as passed); hit offsets then come from the counts"""
        _check("awfmGpuSearchHitsSparse", _lib.lib().awfmGpuSearchHitsSparse(self.handle, d_chars, d_offsets or None, fixed_length,
                                                                            n, d_ranges or None, d_counts, stream or None))

    def search_hits_packed(self, d_packed, kmer_length, n, d_ranges, d_counts, d_chars_scratch=0, stream=0):
        """awfmGpuSearchHitsPacked: hits-only search of bit-packed k-mers resident on the device"""
        _check("awfmGpuSearchHitsPacked", _lib.lib().awfmGpuSearchHitsPacked(
            self.handle, d_packed, kmer_length, n, d_ranges or None, d_counts or None, d_chars_scratch or None, stream or None))

    def search_hits_compact(self, d_chars, d_offsets, fixed_length, n, d_hit_kmers, d_hit_ranges, capacity, d_num_hits,
                            packed=False, stream=0):
        """awfmGpuSearchHitsCompact: the k-mers with hits appended to a list (seed-order path only)"""
        _check("awfmGpuSearchHitsCompact", _lib.lib().awfmGpuSearchHitsCompact(
            self.handle, d_chars, d_offsets or None, fixed_length, n, int(bool(packed)), d_hit_kmers, d_hit_ranges, capacity,
            d_num_hits, stream or None))

    # seed-bucket sharding (include/awfm_gpu.h) ---------------------------
    def order_buckets(self, fixed_length, total_queries):
        """buckets of the seed order of such a batch on this image (0: not a batch for 8-byte records)"""
        return int(_lib.lib().awfmGpuOrderBuckets(self.handle, fixed_length, total_queries))

    def order_kmers(self, d_chars, fixed_length, n, first_number, total_queries, d_records, d_bucket_start, stream=0):
        """awfmGpuOrderKmers: the shard's records {rest of the code string, number in the whole batch} in bucket order"""
        _check("awfmGpuOrderKmers", _lib.lib().awfmGpuOrderKmers(self.handle, d_chars, fixed_length, n, first_number, total_queries,
                                                               d_records, d_bucket_start, stream or None))

    def search_ordered_records(self, d_records, d_bucket_start, first_bucket, end_bucket, fixed_length, total_queries, d_order_kmers,
                               d_order_ranges, stream=0, d_order_counts=0):
        """awfmGpuSearchOrderedRecords[Counts]: the buckets [first, end) of a record array, results in that order (d_order_counts:
        the 32-bit counts in that order as well)"""
        _check("awfmGpuSearchOrderedRecordsCounts", _lib.lib().awfmGpuSearchOrderedRecordsCounts(
            self.handle, d_records, d_bucket_start, first_bucket, end_bucket, fixed_length, total_queries, d_order_kmers, d_order_ranges,
            d_order_counts or None, stream or None))

    def merge_bucket_runs(self, d_received, d_slice_at, d_slice_starts, num_slices, first_bucket, end_bucket, buckets, d_records,
                          d_bucket_start, stream=0):
        """awfmGpuMergeBucketRuns: the slices a rank received in the exchange of the seed-bucket sharding, put in bucket order
        (one launch), with the bucket starts search_ordered_records wants"""
        _check("awfmGpuMergeBucketRuns", _lib.lib().awfmGpuMergeBucketRuns(
            self.handle, d_received, d_slice_at, d_slice_starts, num_slices, first_bucket, end_bucket, buckets, d_records, d_bucket_start,
            stream or None))

    def search_general_records(self, d_chars, fixed_length, n, first_number, total_queries, d_records, d_bucket_start, d_order_kmers,
                               d_order_ranges, stream=0):
        """awfmGpuSearchGeneralRecords: the tail of a shard's own records (k-mers with ambiguity characters) through the general kernel"""
        _check("awfmGpuSearchGeneralRecords", _lib.lib().awfmGpuSearchGeneralRecords(
            self.handle, d_chars, fixed_length, n, first_number, total_queries, d_records, d_bucket_start, d_order_kmers, d_order_ranges,
            stream or None))

    def search_hits_in_order(self, d_chars, d_offsets, fixed_length, n, d_order_kmers, d_order_ranges, packed=False, stream=0,
                             d_order_counts=0):
        """awfmGpuSearchHitsInOrder[Counts]: {k-mer number, range} for every k-mer, in the order the seed-order search took them;
        d_order_counts: the 32-bit counts in that order as well"""
        _check("awfmGpuSearchHitsInOrderCounts", _lib.lib().awfmGpuSearchHitsInOrderCounts(
            self.handle, d_chars, d_offsets or None, fixed_length, n, int(bool(packed)), d_order_kmers, d_order_ranges,
            d_order_counts or None, stream or None))

    def compact_hits(self, d_counts, d_ranges, n, d_flag_offsets, d_scratch, d_hit_kmers, d_hit_ranges, capacity, d_num_hits,
                     stream=0):
        _check("awfmGpuCompactHits", _lib.lib().awfmGpuCompactHits(self.handle, d_counts, d_ranges, n, d_flag_offsets, d_scratch,
                                                                   d_hit_kmers, d_hit_ranges, capacity, d_num_hits, stream or None))

    def sort_hits(self, d_hit_kmers, d_hit_ranges, num_entries, stream=0):
        _check("awfmGpuSortHits", _lib.lib().awfmGpuSortHits(self.handle, d_hit_kmers, d_hit_ranges, num_entries, stream or None))

    def sort_hits_on_device(self, d_hit_kmers, d_hit_ranges, capacity, d_num_hits, n, stream=0):
        """awfmGpuSortHitsOnDevice: the list in k-mer order, its length read on the device (no host wait)"""
        _check("awfmGpuSortHitsOnDevice", _lib.lib().awfmGpuSortHitsOnDevice(self.handle, d_hit_kmers, d_hit_ranges, capacity,
                                                                             d_num_hits, n, stream or None))

    def describe(self):
        """awfmGpuIndexDescribe: one line about what the image holds and which accelerators it did not get"""
        import ctypes as C
        buf = C.create_string_buffer(2048)
        _lib.lib().awfmGpuIndexDescribe(self.handle, buf, 2048)
        return buf.value.decode()

    def stream_retire(self, stream):
        """awfmGpuStreamRetire: call before destroying a stream that has searched on this image"""
        _lib.lib().awfmGpuStreamRetire(self.handle, stream or None)

    def last_lookup_front(self):
        """awfmGpuLastLookupFront: 0 both front ends, 1 the lookup kernel only, 2 the ordered kernels only, -1 none yet"""
        return int(_lib.lib().awfmGpuLastLookupFront(self.handle))

    def last_search_was_exact_lookup(self):
        """awfmGpuLastSearchWasExactLookup: the last awfmGpuSearch took exactLookupSearchKernel"""
        return bool(_lib.lib().awfmGpuLastSearchWasExactLookup(self.handle))

    def list_locate_on_device(self, d_hit_kmers, d_hit_ranges, capacity, d_num_hits, n, d_sorted_kmers, d_sorted_ranges, d_hit_offsets,
                              capacity_hits, d_positions, stream=0):
        """awfmGpuListLocateOnDevice: the appended list -> the list in k-mer order, its hit offsets and positions, in one launch"""
        _check("awfmGpuListLocateOnDevice", _lib.lib().awfmGpuListLocateOnDevice(
            self.handle, d_hit_kmers, d_hit_ranges, capacity, d_num_hits, n, d_sorted_kmers, d_sorted_ranges, d_hit_offsets,
            capacity_hits, d_positions or None, stream or None))

    def hit_offsets_on_device(self, d_counts, d_ranges, n, d_hit_offsets, d_scratch, stream=0):
        """awfmGpuHitOffsetsOnDevice: the scan; the total stays in d_hit_offsets[n]"""
        _check("awfmGpuHitOffsetsOnDevice", _lib.lib().awfmGpuHitOffsetsOnDevice(self.handle, d_counts or None, d_ranges or None, n,
                                                                                 d_hit_offsets, d_scratch, stream or None))

    def locate_on_device(self, d_ranges, d_hit_offsets, n, capacity_hits, d_positions, stream=0):
        """awfmGpuLocateOnDevice: the locate with the number of hits read on the device, at most capacity_hits of them"""
        _check("awfmGpuLocateOnDevice", _lib.lib().awfmGpuLocateOnDevice(self.handle, d_ranges, d_hit_offsets, n, capacity_hits,
                                                                         d_positions, stream or None))

    def ordered_kernel_log(self, max_entries=1024):
        """awfmGpuOrderedKernelLog: [(encodeLookupKernel ms or -1, orderedSearchKernel ms or -1)] of the searches timed since
        the last call ($AWFM_GPU_TIME_ORDERED), oldest first"""
        front = (C.c_double * max_entries)()
        kern = (C.c_double * max_entries)()
        n = _lib.lib().awfmGpuOrderedKernelLog(self.handle, front, kern, max_entries)
        return [(front[i], kern[i]) for i in range(n)]

    def last_ordered_search_kernel_ms(self):
        return float(_lib.lib().awfmGpuLastOrderedSearchKernelMs(self.handle))

    def search_hits_is_ordered(self, has_offsets, fixed_length, n):
        return bool(_lib.lib().awfmGpuSearchHitsIsOrdered(self.handle, int(bool(has_offsets)), fixed_length, n))

    def last_ordered_kernel_ms(self):
        return float(_lib.lib().awfmGpuLastOrderedKernelMs(self.handle))

    def last_ordered_kept(self):
        """k-mers the last seed-order search with 8-byte records ordered and searched (awfmGpuLastOrderedKept)"""
        return int(_lib.lib().awfmGpuLastOrderedKept(self.handle))

    def last_ordered_kernel_is_lookup(self):
        """the kernel last_ordered_kernel_ms() timed was encodeLookupKernel ("lookup first", include/awfm_gpu.h)"""
        return bool(_lib.lib().awfmGpuLastOrderedKernelIsLookup(self.handle))

    def set_ordered(self, mode):
        """-1 automatic, 0 never, 1 always: search_hits() of fixed-length nucleotide batches in seed order"""
        _lib.lib().awfmGpuIndexSetOrdered(self.handle, mode)

    def search_tally(self, d_chars, d_offsets, fixed_length, n):
        """{seeded, steps, blocks, chars} of the instrumented search kernel"""
        out = (C.c_uint64 * 4)()
        _check("awfmGpuSearchTally", _lib.lib().awfmGpuSearchTally(self.handle, d_chars, d_offsets or None, fixed_length,
                                                                   n, C.byref(out)))
        return {"seeded": int(out[0]), "steps": int(out[1]), "blocks": int(out[2]), "chars": int(out[3])}

    def search_hits_line_tally(self, d_chars, d_offsets, fixed_length, n):
        """compulsory traffic of the seed-order search of this batch (awfmGpuSearchHitsLineTally): distinct 128-B lines
        per search level, records read, results stored"""
        out = (C.c_uint64 * 8)()
        _check("awfmGpuSearchHitsLineTally", _lib.lib().awfmGpuSearchHitsLineTally(self.handle, d_chars, d_offsets or None,
                                                                                   fixed_length, n, C.byref(out)))
        keys = ("seed_table_lines", "deep_table_lines", "pair_level_lines", "nuc_level_lines", "ordered_kmers",
                "record_bytes_per_kmer", "kmers_with_hits", "general_kmers")
        return {k: int(v) for k, v in zip(keys, out)}

    def hit_offsets(self, d_ranges, n, d_hit_offsets, d_scratch, stream=0):
        total = C.c_uint64(0)
        _check("awfmGpuHitOffsets", _lib.lib().awfmGpuHitOffsets(self.handle, d_ranges, n, d_hit_offsets, d_scratch,
                                                                 C.byref(total), stream or None))
        return int(total.value)

    def hit_offsets_from_counts(self, d_counts, n, d_hit_offsets, d_scratch, stream=0):
        total = C.c_uint64(0)
        _check("awfmGpuHitOffsetsFromCounts", _lib.lib().awfmGpuHitOffsetsFromCounts(
            self.handle, d_counts, n, d_hit_offsets, d_scratch, C.byref(total), stream or None))
        return int(total.value)

    def locate(self, d_ranges, d_hit_offsets, n, total_hits, d_positions, stream=0):
        _check("awfmGpuLocate", _lib.lib().awfmGpuLocate(self.handle, d_ranges, d_hit_offsets, n, total_hits,
                                                         d_positions, stream or None))

    def locate_window(self, d_ranges, d_hit_offsets, query_begin, query_end, hit_begin, hit_end, d_positions, stream=0):
        """awfmGpuLocateWindow: the hits numbered hit_begin .. hit_end-1 of the batch's flat hit list"""
        _check("awfmGpuLocateWindow", _lib.lib().awfmGpuLocateWindow(self.handle, d_ranges, d_hit_offsets, query_begin, query_end,
                                                                     hit_begin, hit_end, d_positions, d_positions, stream or None))

    @staticmethod
    def scan_scratch_bytes(n):
        return int(_lib.lib().awfmGpuScanScratchBytes(n))

    def destroy(self):
        if self.handle:
            if self.owned:
                _lib.lib().awfmGpuIndexDestroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
