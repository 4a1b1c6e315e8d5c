"""ctypes loader and struct mirrors for libawfmindex_amd.so (include/AwFmIndex.h, include/awfm_gpu.h)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AWFM_LIB_PATH") or os.path.join(_HERE, "libawfmindex_amd.so")  # override: A/B builds
_LIB = None

AwFmAlphabetAmino, AwFmAlphabetDna, AwFmAlphabetRna = 1, 2, 3
AwFmSuccess, AwFmFileReadOkay, AwFmFileWriteOkay = 1, 2, 3
AwFmGeneralFailure = -1
AwFmIllegalPositionError = -6
AwFmFileReadFail = -11

# every symbol the two public headers declare
API_SYMBOLS = [
    "awFmCreateIndex", "awFmCreateIndexFromFasta", "awFmDeallocIndex", "awFmWriteIndexToFile",
    "awFmReadIndexFromFile", "awFmCreateKmerSearchList", "awFmDeallocKmerSearchList", "awFmParallelSearchLocate",
    "awFmParallelSearchCount", "awFmFindSearchRangeForString", "awFmReadSequenceFromFile",
    "awFmCreateInitialQueryRange", "awFmCreateInitialQueryRangeFromChar",
    "awFmNucleotideIterativeStepBackwardSearch", "awFmAminoIterativeStepBackwardSearch",
    "awFmFindDatabaseHitPositions", "awFmFindDatabaseHitPositionSingle",
    "awFmGetLocalSequencePositionFromIndexPosition", "awFmNucleotideBacktraceReturnPreviousLetterIndex",
    "awFmAminoBacktraceReturnPreviousLetterIndex", "awFmGetHeaderStringFromSequenceNumber", "awFmSearchRangeLength",
    "awFmReturnCodeIsFailure", "awFmReturnCodeIsSuccess", "awFmGetNumSequences",
]
GPU_SYMBOLS = [
    "awfmGpuDeviceCount", "awfmGpuLastError", "awfmGpuIndexCreate", "awfmGpuIndexDestroy", "awfmGpuIndexAcquire", "awfmGpuIndexAcquireAll",
    "awfmGpuIndexRelease", "awfmGpuIndexDeviceBytes", "awfmGpuIndexDevice", "awfmGpuIndexSetKernel", "awfmGpuIndexSetWide", "awfmGpuIndexIsWide", "awfmGpuLastBatchStatus", "awfmGpuIndexSetDeepSeed", "awfmGpuIndexSetDenseSa", "awfmGpuPinnedBuffer", "awfmGpuLocateHostWindows", "awfmGpuLocateWindow", "awfmGpuAosLock",
    "awfmGpuAosUnlock", "awfmGpuSearch", "awfmGpuSearchHits", "awfmGpuSearchHitsSparse", "awfmGpuIndexSetOrdered", "awfmGpuSearchHitsIsOrdered", "awfmGpuLastOrderedKernelMs", "awfmGpuLastOrderedKernelIsLookup", "awfmGpuLastOrderedKept",
    "awfmGpuScanScratchBytes", "awfmGpuHitOffsets", "awfmGpuHitOffsetsFromCounts", "awfmGpuLocate", "awfmGpuCountHost", "awfmGpuLocateHost",
    "awfmGpuCreateIndex", "awfmGpuSearchTally", "awfmGpuSynthText", "awfmGpuSynthRandomQueries", "awfmGpuSynthPlantedQueries",
    "awfmGpuSynthMixedLengths", "awfmGpuSynthMixedQueries", "awfmGpuSynthGenomeText", "awfmGpuSynthPlantedQueriesClean",
    "awfmPackKmers", "awfmGpuPackKmers", "awfmGpuUnpackKmers", "awfmGpuHostAlloc", "awfmGpuHostFree", "awfmGpuStreamPacked",
    "awfmGpuStreamChars", "awfmGpuCountPackedHost", "awfmGpuLocatePackedHost", "awfmGpuIndexSetPairImage", "awfmGpuIndexHasPairImage",
    "awfmGpuSearchHitsPacked", "awfmGpuLocateTo", "awfmGpuSearchHitsLineTally", "awfmGpuIndexDeepSeedK", "awfmGpuSearchHitsCompact", "awfmGpuCompactHits", "awfmGpuSortHits",
    "awfmGpuStreamPackedSparse", "awfmGpuStreamCharsSparse", "awfmGpuSearchHitsInOrder", "awfmGpuSearchHitsInOrderCounts",
    "awfmGpuSortHitsOnDevice", "awfmGpuHitOffsetsOnDevice", "awfmGpuLocateOnDevice", "awfmGpuLastOrderedSearchKernelMs", "awfmGpuOrderedKernelLog", "awfmGpuIndexDeepSeedBuildSeconds", "awfmGpuIndexDeepSeedTransientBytes", "awfmGpuIndexHasDenseSa", "awfmGpuIndexDenseSaBuildSeconds", "awfmGpuIndexLengthTableBytes", "awfmGpuIndexLengthTableBuildSeconds", "awfmGpuMixedLookupLineTally",
    "awfmGpuListLocateOnDevice", "awfmGpuLastLookupFront", "awfmGpuLastSearchWasExactLookup", "awfmGpuSynthPlantedQueriesUnique", "awfmGpuStreamRetire", "awfmGpuIndexDescribe", "awfmGpuIndexDeepSeedAllocSeconds", "awfmGpuAosLastStages", "awfmHostCopyGBs",
    "awfmGpuOrderBuckets", "awfmGpuOrderKmers", "awfmGpuSearchOrderedRecords", "awfmGpuSearchOrderedRecordsCounts", "awfmGpuSearchGeneralRecords", "awfmGpuMergeBucketRuns",
]
# int sink(void *user, uint64 firstKmer, uint64 numKmers, const uint32 *counts, const uint64 *positions, uint64 numPositions)
CHUNK_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_uint64)


# int sink(void *user, uint64 firstKmer, uint64 numKmers, uint64 numHitKmers, const uint32 *hitKmers, const uint64 *hitOffsets,
#          const uint64 *positions, uint64 numPositions)
SPARSE_CHUNK_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint64), C.c_uint64)
# int sink(void *user, uint64 queryBegin, uint64 queryEnd, uint64 hitBegin, uint64 hitEnd, const uint64 *positions)
HIT_WINDOW_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64))


class AwFmIndexConfiguration(C.Structure):
    _fields_ = [("suffixArrayCompressionRatio", C.c_uint8), ("kmerLengthInSeedTable", C.c_uint8),
                ("alphabetType", C.c_int), ("keepSuffixArrayInMemory", C.c_bool), ("storeOriginalSequence", C.c_bool)]


class AwFmCompressedSuffixArray(C.Structure):
    _fields_ = [("valueBitWidth", C.c_uint8), ("values", C.POINTER(C.c_uint8)), ("compressedByteLength", C.c_uint64)]


class AwFmSearchRange(C.Structure):
    _fields_ = [("startPtr", C.c_uint64), ("endPtr", C.c_uint64)]


class AwFmIndex(C.Structure):
    _fields_ = [("versionNumber", C.c_uint32), ("featureFlags", C.c_uint32), ("bwtLength", C.c_uint64),
                ("bwtBlockList", C.c_void_p), ("prefixSums", C.POINTER(C.c_uint64)),
                ("kmerSeedTable", C.POINTER(AwFmSearchRange)), ("fileHandle", C.c_void_p),
                ("config", AwFmIndexConfiguration), ("fileDescriptor", C.c_int), ("suffixArrayFileOffset", C.c_size_t),
                ("sequenceFileOffset", C.c_size_t), ("fastaVector", C.c_void_p),
                ("suffixArray", AwFmCompressedSuffixArray)]


class AwFmKmerSearchData(C.Structure):
    _fields_ = [("kmerString", C.c_void_p), ("kmerLength", C.c_uint64), ("positionList", C.POINTER(C.c_uint64)),
                ("count", C.c_uint32), ("capacity", C.c_uint32)]


class AwFmKmerSearchList(C.Structure):
    _fields_ = [("capacity", C.c_size_t), ("count", C.c_size_t), ("kmerSearchData", C.POINTER(AwFmKmerSearchData))]


def build(force=False):
    """compile libawfmindex_amd.so in-tree (hipcc --offload-arch=gfx950 + gcc)"""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s", "-j4"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libawfmindex_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
            "there is no fallback implementation")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    IP, LP = C.POINTER(AwFmIndex), C.POINTER(AwFmKmerSearchList)
    RP = C.POINTER(AwFmSearchRange)
    vp, u64 = C.c_void_p, C.c_uint64
    sig = {
        "awFmCreateIndex": (C.c_int, [C.POINTER(IP), C.POINTER(AwFmIndexConfiguration), vp, C.c_size_t, C.c_char_p]),
        "awFmCreateIndexFromFasta": (C.c_int, [C.POINTER(IP), C.POINTER(AwFmIndexConfiguration), C.c_char_p, C.c_char_p]),
        "awFmDeallocIndex": (None, [IP]),
        "awFmWriteIndexToFile": (C.c_int, [IP, vp, u64, C.c_char_p]),
        "awFmReadIndexFromFile": (C.c_int, [C.POINTER(IP), C.c_char_p, C.c_bool]),
        "awFmCreateKmerSearchList": (LP, [C.c_size_t]),
        "awFmDeallocKmerSearchList": (None, [LP]),
        "awFmParallelSearchLocate": (C.c_int, [IP, LP, C.c_uint32]),
        "awFmParallelSearchCount": (None, [IP, LP, C.c_uint32]),
        "awFmFindSearchRangeForString": (AwFmSearchRange, [IP, C.c_char_p, C.c_size_t]),
        "awFmReadSequenceFromFile": (C.c_int, [IP, C.c_size_t, C.c_size_t, C.c_char_p]),
        "awFmCreateInitialQueryRange": (AwFmSearchRange, [IP, C.c_char_p, u64]),
        "awFmCreateInitialQueryRangeFromChar": (AwFmSearchRange, [IP, C.c_char]),
        "awFmNucleotideIterativeStepBackwardSearch": (None, [IP, RP, C.c_uint8]),
        "awFmAminoIterativeStepBackwardSearch": (None, [IP, RP, C.c_uint8]),
        "awFmFindDatabaseHitPositions": (C.POINTER(u64), [IP, RP, C.POINTER(C.c_int)]),
        "awFmFindDatabaseHitPositionSingle": (u64, [IP, u64, C.POINTER(C.c_int)]),
        "awFmNucleotideBacktraceReturnPreviousLetterIndex": (C.c_uint8, [IP, C.POINTER(u64)]),
        "awFmAminoBacktraceReturnPreviousLetterIndex": (C.c_uint8, [IP, C.POINTER(u64)]),
        "awFmGetLocalSequencePositionFromIndexPosition": (C.c_int, [IP, C.c_size_t, C.POINTER(C.c_size_t),
                                                                   C.POINTER(C.c_size_t)]),
        "awFmGetHeaderStringFromSequenceNumber": (C.c_int, [IP, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]),
        "awFmSearchRangeLength": (C.c_size_t, [RP]),
        "awFmReturnCodeIsFailure": (C.c_bool, [C.c_int]),
        "awFmReturnCodeIsSuccess": (C.c_bool, [C.c_int]),
        "awFmGetNumSequences": (C.c_uint32, [IP]),
        "awfmGpuDeviceCount": (C.c_int, []),
        "awfmGpuLastError": (C.c_char_p, []),
        "awfmGpuIndexCreate": (C.c_int, [IP, C.c_int, C.POINTER(vp)]),
        "awfmGpuIndexDestroy": (None, [vp]),
        "awfmGpuIndexAcquire": (vp, [IP]),
        "awfmGpuIndexAcquireAll": (C.c_int, [IP, C.POINTER(vp), C.c_int]),
        "awfmGpuIndexRelease": (None, [IP]),
        "awfmGpuIndexDeviceBytes": (u64, [vp]),
        "awfmGpuIndexDevice": (C.c_int, [vp]),
        "awfmGpuIndexSetKernel": (None, [vp, C.c_int]),
        "awfmGpuIndexSetWide": (None, [vp, C.c_int]),
        "awfmGpuIndexIsWide": (C.c_int, [vp]),
        "awfmGpuLastBatchStatus": (C.c_int, []),
        "awfmGpuIndexSetOrdered": (None, [vp, C.c_int]),
        "awfmGpuSearchHitsIsOrdered": (C.c_int, [vp, C.c_int, C.c_uint32, u64]),
        "awfmGpuLastOrderedKernelMs": (C.c_double, [vp]),
        "awfmGpuLastOrderedKernelIsLookup": (C.c_int, [vp]),
        "awfmGpuLastOrderedKept": (C.c_uint64, [vp]),
        "awfmGpuIndexSetDeepSeed": (C.c_int, [vp, C.c_uint]),
        "awfmGpuIndexSetDenseSa": (C.c_int, [vp, C.c_int]),
        "awfmGpuIndexSetPairImage": (C.c_int, [vp, C.c_int]),
        "awfmGpuIndexHasPairImage": (C.c_int, [vp]),
        "awfmGpuIndexDeepSeedK": (C.c_uint, [vp]),
        "awfmGpuIndexDeepSeedBuildSeconds": (C.c_double, [vp]),
        "awfmGpuIndexDeepSeedTransientBytes": (u64, [vp]),
        "awfmGpuIndexHasDenseSa": (C.c_int, [vp]),
        "awfmGpuIndexDenseSaBuildSeconds": (C.c_double, [vp]),
        "awfmGpuIndexLengthTableBytes": (u64, [vp]),
        "awfmGpuIndexLengthTableBuildSeconds": (C.c_double, [vp]),
        "awfmGpuMixedLookupLineTally": (C.c_int, [vp, vp, vp, u64, C.POINTER(u64 * 8)]),
        "awfmGpuSearch": (C.c_int, [vp, vp, vp, C.c_uint32, u64, vp, vp, vp]),
        "awfmGpuSearchHits": (C.c_int, [vp, vp, vp, C.c_uint32, u64, vp, vp, vp]),
        "awfmGpuSearchHitsSparse": (C.c_int, [vp, vp, vp, C.c_uint32, u64, vp, vp, vp]),
        "awfmGpuSearchHitsCompact": (C.c_int, [vp, vp, vp, C.c_uint32, u64, C.c_int, vp, vp, C.c_uint32, vp, vp]),
        "awfmGpuCompactHits": (C.c_int, [vp, vp, vp, u64, vp, vp, vp, vp, C.c_uint32, vp, vp]),
        "awfmGpuSearchHitsInOrder": (C.c_int, [vp, vp, vp, C.c_uint32, u64, C.c_int, vp, vp, vp]),
        "awfmGpuSearchHitsInOrderCounts": (C.c_int, [vp, vp, vp, C.c_uint32, u64, C.c_int, vp, vp, vp, vp]),
        "awfmGpuOrderBuckets": (C.c_uint32, [vp, C.c_uint32, u64]),
        "awfmGpuOrderKmers": (C.c_int, [vp, vp, C.c_uint32, u64, u64, u64, vp, vp, vp]),
        "awfmGpuSearchOrderedRecords": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, u64, vp, vp, vp]),
        "awfmGpuSearchOrderedRecordsCounts": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, u64, vp, vp, vp, vp]),
        "awfmGpuMergeBucketRuns": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]),
        "awfmGpuSearchGeneralRecords": (C.c_int, [vp, vp, C.c_uint32, u64, u64, u64, vp, vp, vp, vp, vp]),
        "awfmGpuSortHits": (C.c_int, [vp, vp, vp, C.c_uint32, vp]),
        "awfmGpuSortHitsOnDevice": (C.c_int, [vp, vp, vp, C.c_uint32, vp, u64, vp]),
        "awfmGpuListLocateOnDevice": (C.c_int, [vp, vp, vp, C.c_uint32, vp, u64, vp, vp, vp, u64, vp, vp]),
        "awfmGpuLastLookupFront": (C.c_int, [vp]),
        "awfmGpuLastSearchWasExactLookup": (C.c_int, [vp]),
        "awfmGpuStreamRetire": (None, [vp, vp]),
        "awfmGpuIndexDescribe": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "awfmGpuIndexDeepSeedAllocSeconds": (C.c_double, [vp]),
        "awfmGpuAosLastStages": (None, [C.POINTER(C.c_double * 10)]),
        "awfmHostCopyGBs": (C.c_double, [C.c_uint32, u64]),
        "awfmGpuHitOffsetsOnDevice": (C.c_int, [vp, vp, vp, u64, vp, vp, vp]),
        "awfmGpuLocateOnDevice": (C.c_int, [vp, vp, vp, u64, u64, vp, vp]),
        "awfmGpuLastOrderedSearchKernelMs": (C.c_double, [vp]),
        "awfmGpuOrderedKernelLog": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
        "awfmGpuScanScratchBytes": (u64, [u64]),
        "awfmGpuHitOffsets": (C.c_int, [vp, vp, u64, vp, vp, C.POINTER(u64), vp]),
        "awfmGpuHitOffsetsFromCounts": (C.c_int, [vp, vp, u64, vp, vp, C.POINTER(u64), vp]),
        "awfmGpuLocate": (C.c_int, [vp, vp, vp, u64, u64, vp, vp]),
        "awfmGpuLocateTo": (C.c_int, [vp, vp, vp, u64, u64, vp, vp, vp]),
        "awfmGpuLocateWindow": (C.c_int, [vp, vp, vp, u64, u64, u64, u64, vp, vp, vp]),
        "awfmGpuLocateHostWindows": (C.c_int, [vp, vp, vp, C.c_uint32, u64, vp, vp, HIT_WINDOW_SINK, vp]),
        "awfmGpuCountHost": (C.c_int, [vp, vp, vp, C.c_uint32, u64, vp, vp]),
        "awfmGpuLocateHost": (C.c_int, [vp, vp, vp, C.c_uint32, u64, vp, vp, C.POINTER(C.POINTER(u64))]),
        "awfmGpuCreateIndex": (C.c_int, [C.POINTER(IP), C.POINTER(AwFmIndexConfiguration), vp, u64, C.c_int,
                                         C.c_char_p, C.c_int]),
        "awfmGpuSearchTally": (C.c_int, [vp, vp, vp, C.c_uint32, u64, C.POINTER(u64 * 4)]),
        "awfmGpuSearchHitsLineTally": (C.c_int, [vp, vp, vp, C.c_uint32, u64, C.POINTER(u64 * 8)]),
        "awfmGpuSynthText": (C.c_int, [vp, u64, u64, u64, C.c_int, vp]),
        "awfmGpuSynthRandomQueries": (C.c_int, [vp, u64, u64, C.c_uint32, u64, C.c_int, vp]),
        "awfmGpuSynthPlantedQueries": (C.c_int, [vp, u64, u64, C.c_uint32, u64, vp, u64, vp]),
        "awfmGpuSynthPlantedQueriesClean": (C.c_int, [vp, u64, u64, C.c_uint32, u64, vp, u64, vp]),
        "awfmGpuSynthPlantedQueriesUnique": (C.c_int, [vp, u64, u64, C.c_uint32, u64, vp, u64, u64, vp, vp]),
        "awfmGpuSynthGenomeText": (C.c_int, [vp, u64, u64, vp]),
        "awfmGpuSynthMixedLengths": (C.c_int, [vp, u64, u64, C.c_uint32, C.c_uint32, u64, vp]),
        "awfmGpuSynthMixedQueries": (C.c_int, [vp, vp, u64, u64, u64, vp, u64, C.c_int, vp]),
        "awfmPackKmers": (C.c_int, [C.c_int, vp, C.c_uint32, u64, vp, C.POINTER(u64)]),
        "awfmGpuPackKmers": (C.c_int, [vp, vp, C.c_uint32, u64, vp, C.POINTER(u64), vp]),
        "awfmGpuUnpackKmers": (C.c_int, [vp, vp, C.c_uint32, u64, vp, vp]),
        "awfmGpuSearchHitsPacked": (C.c_int, [vp, vp, C.c_uint32, u64, vp, vp, vp, vp]),
        "awfmGpuHostAlloc": (vp, [u64]),
        "awfmGpuHostFree": (None, [vp]),
        "awfmGpuStreamPacked": (C.c_int, [vp, vp, C.c_uint32, u64, u64, C.c_int, C.c_uint, CHUNK_SINK, vp]),
        "awfmGpuStreamChars": (C.c_int, [vp, vp, C.c_uint32, u64, u64, C.c_int, C.c_uint, CHUNK_SINK, vp]),
        "awfmGpuStreamPackedSparse": (C.c_int, [vp, vp, C.c_uint32, u64, u64, C.c_int, C.c_uint, SPARSE_CHUNK_SINK, vp]),
        "awfmGpuStreamCharsSparse": (C.c_int, [vp, vp, C.c_uint32, u64, u64, C.c_int, C.c_uint, SPARSE_CHUNK_SINK, vp]),
        "awfmGpuCountPackedHost": (C.c_int, [vp, vp, C.c_uint32, u64, vp]),
        "awfmGpuLocatePackedHost": (C.c_int, [vp, vp, C.c_uint32, u64, vp, C.POINTER(C.POINTER(u64)), C.POINTER(u64)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    L.free = C.CDLL(None).free
    L.free.argtypes = [vp]
    L.free.restype = None
    _LIB = L
    return L
