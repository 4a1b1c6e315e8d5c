/*
 * awfm_gpu_stream.hip -- the flat host-buffer batch API as a chunked, overlapped pipeline, with bit-packed k-mers.
 *
 * The reference hands a batch over as an array of structs (one kmerString pointer and one malloc'ed positionList
 * per k-mer, ref src/AwFmParallelSearch.c:36-93, :367-387); packing that into something a device can read and
 * scattering the answers back costs more host time than the search costs device time.  This entry point takes the
 * batch flat instead: one 64-bit word per k-mer (2 bits per nucleotide, 5 bits per amino acid), or fixed-length
 * ASCII, in one host array; results come back chunk by chunk through a callback, as 32-bit counts and one flat
 * array of positions (the hits of k-mer i of the chunk follow those of k-mer i-1, in BWT order).
 *
 * Pipeline: the batch is cut into chunks of chunkKmers; three slots of device buffers and page-locked staging, each with
 * its own stream.  For chunk t the calling thread
 *   A(t)   enqueues the upload of the k-mers, search -> hit-offset scan -> total to the host,
 *   B(t-1) waits for that total, enqueues expand + LF walk + sample read and the copies back,
 *   C(t-2) waits for the copies and calls the sink,
 * in that order in one loop, so that the upload of chunk t, the kernels of t-1 and the download of t-2 are all in flight
 * and the sink runs on the host while the device works on the next two chunks.  What overlaps in practice (rocprofv3,
 * MI355X, ROCm 7.2): host-to-device copies run on the DMA engines beside the kernels; device-to-host copies are executed
 * as a shader copy (__amd_rocclr_copyBuffer) whatever the API variant or size, and a shader copy does not get onto the
 * chip while a persistent search or walk kernel holds every CU -- so the downloads mostly alternate with the kernels:
 * a batch costs about kernels + downloads (10^8 planted 21-mers: 26 ms + 1.2 GB / 55 GB/s), uploads and the host side
 * are hidden.
 *
 * Device side of a chunk: awfmGpuSearchHits (ordered path for large nucleotide chunks) -> scan -> awfmGpuLocate,
 * i.e. exactly the kernels of the device-buffer API; results are those of ref src/AwFmParallelSearch.c:95-365.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "awfm_device.h"

namespace {

typedef unsigned long long u64;

__constant__ unsigned char kUnpackDna[4] = {'a', 'c', 'g', 't'};
/* letter index -> ASCII (ref src/AwFmLetter.c:55-67 maps the other way); 20.. = ambiguity */
__constant__ unsigned char kUnpackAmino[32] = {'a', 'c', 'd', 'e', 'f', 'g', 'h', 'i', 'k', 'l', 'm', 'n', 'p', 'q', 'r', 's',
                                               't', 'v', 'w', 'y', 'x', 'x', 'x', 'x', 'x', 'x', 'x', 'x', 'x', 'x', 'x', 'x'};

/* packed words -> ASCII k-mers, one thread per character (coalesced byte stores; the word is read once per wave
 * and a few times per cache line) */
__global__ void __launch_bounds__(256)
    unpackKmersKernel(const u64 *__restrict__ packed, unsigned len, u64 n, int amino, unsigned char *__restrict__ chars) {
  /* a workgroup takes tiles of 256 k-mers (256 * len contiguous bytes); inside a tile the indices are small, so the
   * division by the runtime length is a 32-bit multiply-high (exact for values below 2^13 and len <= 32) */
  const unsigned bits = amino ? 5u : 2u;
  const unsigned magic = len > 1u ? (unsigned)((0x100000000ull + len - 1u) / len) : 0u;
  const u64 tiles = (n + 255ull) / 256ull;
  for (u64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const u64 first = tile * 256ull;
    const unsigned inTile = n - first < 256ull ? (unsigned)(n - first) : 256u;
    for (unsigned i = threadIdx.x; i < inTile * len; i += 256u) {
      const unsigned j = len > 1u ? __umulhi(i, magic) : i, c = i - j * len;
      const unsigned code = (unsigned)(packed[first + j] >> (bits * (len - 1u - c))) & (amino ? 31u : 3u);
      chars[first * len + i] = amino ? kUnpackAmino[code] : kUnpackDna[code];
    }
  }
}

/* ASCII k-mers -> packed words, one thread per k-mer; bad[0] counts k-mers with a character the packing cannot
 * express (those get the all-ones word) */
__global__ void __launch_bounds__(256)
    packKmersKernel(const unsigned char *__restrict__ chars, unsigned len, u64 n, int amino, u64 *__restrict__ packed,
                    u64 *__restrict__ bad) {
  for (u64 j = (u64)blockIdx.x * 256ull + threadIdx.x; j < n; j += (u64)gridDim.x * 256ull) {
    u64 w = 0;
    bool ok = true;
    for (unsigned c = 0; c < len; c++) {
      const unsigned ch = chars[j * len + c];
      if (amino) {
        const unsigned a = ch == '$' ? 21u : (unsigned)kAminoTables.letterOfAscii[ch & 31u]; /* as the ASCII API maps it */
        ok &= a < 20u;
        w = (w << 5) | (a & 31u);
      } else {
        ok &= (nucIsAcgtu(ch) & 1u) != 0u;
        w = (w << 2) | (nucLetterIndex(ch) & 3u);
      }
    }
    packed[j] = ok ? w : ~0ull;
    if (!ok) atomicAdd(bad, 1ull);
  }
}

struct StreamSlot {
  hipEvent_t uploaded = nullptr, searched = nullptr, located = nullptr, done = nullptr;
  /* device */
  void *dIn = nullptr;       /* packed words or ASCII as uploaded */
  void *dChars = nullptr;    /* ASCII the search reads (packed input only) */
  void *dRanges = nullptr, *dCounts = nullptr, *dHitOffsets = nullptr, *dScratch = nullptr;
  size_t capKmers = 0, capChars = 0, capInBytes = 0; /* capInBytes: what dIn holds -- its width per k-mer differs between batches */
  bool ready = false;                                /* events and hTotal all exist */
  void *dPositions = nullptr;
  size_t capPositions = 0;
  /* page-locked host */
  void *hIn = nullptr;
  size_t hInBytes = 0;
  uint32_t *hCounts = nullptr;
  u64 *hTotal = nullptr;
  u64 *hPositions = nullptr;
  size_t hCapPositions = 0;
  /* sparse results (awfmGpuStream*Sparse): the k-mers with hits as a list */
  void *dHitKmers = nullptr, *dHitRanges = nullptr, *dListOffsets = nullptr, *dNumHits = nullptr, *dFlagOffsets = nullptr;
  size_t capList = 0, capFlags = 0;
  uint32_t *hHitKmers = nullptr;
  u64 *hListOffsets = nullptr;
  size_t hCapList = 0;
  u64 numHitKmers = 0;
  uint32_t capUsed = 0; /* entries of the list buffers the chunk in the slot was searched with */
  u64 *hOffsets = nullptr; /* hit offsets of a chunk whose hits exceed the device's hit budget (taken in windows) */
  size_t hCapOffsets = 0;
  /* the chunk in the slot */
  u64 first = 0, n = 0, total = 0;
  bool windowed = false; /* its hit list is located window by window when the chunk is handed over */
};

}  // namespace

constexpr int kStreamSlots = 3; /* chunk t uploads and searches, t-1 walks and downloads, t-2 is with the caller */

struct AwFmGpuStreamState {
  /* one stream per slot: three chunks' copies and kernels are in flight at a time, so the downloads of neighbouring chunks
   * overlap (a single download stream moved device-to-host bytes at well under half the rate the link gives several:
   * 10^8 planted 21-mers, 1.2 GB back, 69 ms against 36 ms) */
  hipStream_t slotStream[kStreamSlots] = {}; /* everything of a chunk -- upload, kernels, download -- on its slot's stream */
  bool streamsReady = false;
  StreamSlot slot[kStreamSlots];
};

namespace {

void freeSlot(StreamSlot &s) {
  void *dev[] = {s.dIn, s.dChars, s.dRanges, s.dCounts, s.dHitOffsets, s.dScratch, s.dPositions,
                 s.dHitKmers, s.dHitRanges, s.dListOffsets, s.dNumHits, s.dFlagOffsets};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  void *host[] = {s.hIn, s.hCounts, s.hTotal, s.hPositions, s.hOffsets, s.hHitKmers, s.hListOffsets};
  for (void *p : host)
    if (p) (void)hipHostFree(p);
  hipEvent_t events[] = {s.uploaded, s.searched, s.located, s.done};
  for (hipEvent_t e : events)
    if (e) (void)hipEventDestroy(e);
  s = StreamSlot();
}

#define STREAM_TRY(call)                   \
  do {                                     \
    hipError_t e__ = (call);               \
    if (e__ != hipSuccess) {               \
      setError(#call, e__);                \
      return AwFmGeneralFailure;           \
    }                                      \
  } while (0)

/* (re)allocations happen while nothing of this slot is in flight: its previous chunk was handed to the sink */
enum AwFmReturnCode ensureSlot(StreamSlot &s, size_t kmers, size_t inBytesPerKmer, size_t charsPerKmer, bool locate,
                               bool stageInput) {
  if (!s.ready) {
    /* published only when every create succeeded: a slot half set up is torn down, not run on null events */
    hipEvent_t *events[] = {&s.uploaded, &s.searched, &s.located, &s.done};
    hipError_t e = hipSuccess;
    for (hipEvent_t *ev : events)
      if (e == hipSuccess && !*ev) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e == hipSuccess && !s.hTotal) e = hipHostMalloc((void **)&s.hTotal, 64, hipHostMallocDefault);
    if (e != hipSuccess) {
      setError("awfmGpuStream: creating the events / staging of a pipeline slot failed", e);
      freeSlot(s);
      return AwFmGeneralFailure;
    }
    s.ready = true;
  }
  const size_t chars = kmers * charsPerKmer;
  const size_t inBytes = kmers * (inBytesPerKmer > 8 ? inBytesPerKmer : 8);
  if (kmers > s.capKmers || chars > s.capChars || inBytes > s.capInBytes) {
    void **dev[] = {&s.dIn, &s.dChars, &s.dRanges, &s.dCounts, &s.dHitOffsets, &s.dScratch};
    for (void **p : dev) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    if (s.hCounts) (void)hipHostFree(s.hCounts);
    s.hCounts = nullptr;
    s.capKmers = s.capChars = s.capInBytes = 0;
    STREAM_TRY(hipMalloc(&s.dIn, inBytes + 256));
    STREAM_TRY(hipMalloc(&s.dChars, chars + 256));
    STREAM_TRY(hipMalloc(&s.dRanges, kmers * 16 + 256));
    STREAM_TRY(hipMalloc(&s.dCounts, kmers * 4 + 256));
    STREAM_TRY(hipMalloc(&s.dHitOffsets, (kmers + 1) * 8 + 256));
    STREAM_TRY(hipMalloc(&s.dScratch, awfmGpuScanScratchBytes(kmers) + 256));
    STREAM_TRY(hipHostMalloc((void **)&s.hCounts, kmers * 4 + 256, hipHostMallocDefault));
    s.capKmers = kmers;
    s.capChars = chars;
    s.capInBytes = inBytes;
  }
  (void)locate;
  if (stageInput && kmers * inBytesPerKmer > s.hInBytes) {
    if (s.hIn) (void)hipHostFree(s.hIn);
    s.hIn = nullptr;
    s.hInBytes = 0;
    STREAM_TRY(hipHostMalloc(&s.hIn, kmers * inBytesPerKmer + 256, hipHostMallocDefault));
    s.hInBytes = kmers * inBytesPerKmer;
  }
  return AwFmSuccess;
}

enum AwFmReturnCode ensurePositions(StreamSlot &s, u64 total) {
  if (total > s.capPositions) {
    if (s.dPositions) (void)hipFree(s.dPositions);
    s.dPositions = nullptr;
    s.capPositions = 0;
    const size_t want = total + total / 4 + 1024;
    hipError_t e = hipMalloc(&s.dPositions, want * 8);
    if (e != hipSuccess) {
      setError("awfmGpuStream: hipMalloc of the positions of a chunk failed (use smaller chunks)", e);
      return AwFmAllocationFailure;
    }
    s.capPositions = want;
  }
  if (total > s.hCapPositions) {
    if (s.hPositions) (void)hipHostFree(s.hPositions);
    s.hPositions = nullptr;
    s.hCapPositions = 0;
    const size_t want = total + total / 4 + 1024;
    hipError_t e = hipHostMalloc((void **)&s.hPositions, want * 8, hipHostMallocDefault);
    if (e != hipSuccess) {
      setError("awfmGpuStream: hipHostMalloc of the positions of a chunk failed (use smaller chunks)", e);
      return AwFmAllocationFailure;
    }
    s.hCapPositions = want;
  }
  return AwFmSuccess;
}

/* the list buffers of a slot for `entries` k-mers with hits (device + page-locked host); flags: the scan work space of
 * the dense -> list conversion (awfmGpuCompactHits) for a chunk of `kmers` */
enum AwFmReturnCode ensureList(StreamSlot &s, size_t entries, size_t flagsForKmers) {
  if (entries > s.capList) {
    void **dev[] = {&s.dHitKmers, &s.dHitRanges, &s.dListOffsets};
    for (void **p : dev) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    s.capList = 0;
    STREAM_TRY(hipMalloc(&s.dHitKmers, entries * 4 + 256));
    STREAM_TRY(hipMalloc(&s.dHitRanges, entries * 16 + 256));
    STREAM_TRY(hipMalloc(&s.dListOffsets, (entries + 1) * 8 + 256));
    if (!s.dNumHits) STREAM_TRY(hipMalloc(&s.dNumHits, 256));
    s.capList = entries;
  }
  if (entries > s.hCapList) {
    if (s.hHitKmers) (void)hipHostFree(s.hHitKmers);
    if (s.hListOffsets) (void)hipHostFree(s.hListOffsets);
    s.hHitKmers = nullptr;
    s.hListOffsets = nullptr;
    s.hCapList = 0;
    STREAM_TRY(hipHostMalloc((void **)&s.hHitKmers, entries * 4 + 256, hipHostMallocDefault));
    STREAM_TRY(hipHostMalloc((void **)&s.hListOffsets, (entries + 1) * 8 + 256, hipHostMallocDefault));
    s.hCapList = entries;
  }
  if (flagsForKmers > s.capFlags) {
    if (s.dFlagOffsets) (void)hipFree(s.dFlagOffsets);
    s.dFlagOffsets = nullptr;
    s.capFlags = 0;
    STREAM_TRY(hipMalloc(&s.dFlagOffsets, (flagsForKmers + 1) * 8 + 256));
    s.capFlags = flagsForKmers;
  }
  return AwFmSuccess;
}

struct CopyCtx {
  const uint8_t *src;
  uint8_t *dst;
};
void copyRange(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  const CopyCtx *c = (const CopyCtx *)p;
  memcpy(c->dst + begin, c->src + begin, end - begin);
}

/* is this host pointer page-locked memory HIP knows (hipHostMalloc / hipHostRegister)?  Then the DMA engine reads
 * it directly; anything else goes through the slot's staging buffer first. */
bool isPinned(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

}  // namespace

void awfmGpuStreamStateFree(AwFmGpuIndex *g) {
  if (!g || !g->streamState) return;
  AwFmGpuStreamState &state = *g->streamState;
  hipStream_t streams[] = {state.slotStream[0], state.slotStream[1], state.slotStream[2]};
  for (hipStream_t s : streams)
    if (s) {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
  for (StreamSlot &s : g->streamState->slot) freeSlot(s);
  delete g->streamState;
  g->streamState = nullptr;
}

extern "C" {

void *awfmGpuHostAlloc(uint64_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    setError("awfmGpuHostAlloc: hipHostMalloc failed");
    return nullptr;
  }
  return p;
}
void awfmGpuHostFree(void *p) {
  if (p) (void)hipHostFree(p);
}

enum AwFmReturnCode awfmGpuUnpackKmers(AwFmGpuIndex *g, const uint64_t *dPacked, uint32_t kmerLength, uint64_t numKmers,
                                       uint8_t *dChars, void *stream) {
  if (!g || !dPacked || !dChars) {
    setError("awfmGpuUnpackKmers: null argument");
    return AwFmNullPtrError;
  }
  if (kmerLength == 0 || kmerLength > (g->amino ? 12u : 32u)) {
    setError("awfmGpuUnpackKmers: a packed k-mer holds 1..32 nucleotides or 1..12 amino acids");
    return AwFmIllegalPositionError;
  }
  if (numKmers == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  hipLaunchKernelGGL(unpackKmersKernel, dim3((unsigned)g->numCUs * 16u), dim3(256), 0, (hipStream_t)stream, (const u64 *)dPacked,
                     kmerLength, (u64)numKmers, g->amino ? 1 : 0, dChars);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}

enum AwFmReturnCode awfmGpuPackKmers(AwFmGpuIndex *g, const uint8_t *dChars, uint32_t kmerLength, uint64_t numKmers,
                                     uint64_t *dPacked, uint64_t *numUnpackable, void *stream) {
  if (!g || !dPacked || !dChars) {
    setError("awfmGpuPackKmers: null argument");
    return AwFmNullPtrError;
  }
  if (kmerLength == 0 || kmerLength > (g->amino ? 12u : 32u)) {
    setError("awfmGpuPackKmers: a packed k-mer holds 1..32 nucleotides or 1..12 amino acids");
    return AwFmIllegalPositionError;
  }
  if (numUnpackable) *numUnpackable = 0;
  if (numKmers == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  u64 *dBad = nullptr;
  AWFM_HIP_TRY(hipMalloc((void **)&dBad, 8), AwFmAllocationFailure);
  hipError_t e = hipMemsetAsync(dBad, 0, 8, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(packKmersKernel, dim3((unsigned)((numKmers + 255) / 256 < (1ull << 22) ? (numKmers + 255) / 256 : (1ull << 22))), dim3(256), 0, s, dChars, kmerLength,
                       (u64)numKmers, g->amino ? 1 : 0, (u64 *)dPacked, dBad);
    e = hipGetLastError();
  }
  u64 bad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, dBad, 8, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(dBad);
  if (e != hipSuccess) {
    setError("awfmGpuPackKmers", e);
    return AwFmGeneralFailure;
  }
  if (numUnpackable) *numUnpackable = bad;
  if (bad) {
    /* the all-ones word is itself a k-mer ('t' x 32): the output must not be searched */
    setError("awfmGpuPackKmers: the batch holds k-mers with characters a packed word cannot express; search it as ASCII");
    return AwFmIllegalPositionError;
  }
  return AwFmSuccess;
}

/* Hits-only search of bit-packed k-mers resident on the device: nucleotide batches the seed-order path takes are
 * searched straight from the packed words (they are its record format); anything else is unpacked into dCharsScratch
 * (kmerLength bytes per k-mer) and searched as ASCII. */
static enum AwFmReturnCode searchHitsPacked(AwFmGpuIndex *g, const uint64_t *dPacked, uint32_t kmerLength, uint64_t numKmers,
                                            struct AwFmSearchRange *dRanges, uint32_t *dCounts, uint8_t *dCharsScratch,
                                            void *stream, bool rangesOfHitsOnly);

enum AwFmReturnCode awfmGpuSearchHitsPacked(AwFmGpuIndex *g, const uint64_t *dPacked, uint32_t kmerLength, uint64_t numKmers,
                                            struct AwFmSearchRange *dRanges, uint32_t *dCounts, uint8_t *dCharsScratch,
                                            void *stream) {
  return searchHitsPacked(g, dPacked, kmerLength, numKmers, dRanges, dCounts, dCharsScratch, stream, false);
}

/* rangesOfHitsOnly: the contract of awfmGpuSearchHitsSparse (the pipeline reads the ranges through the counts) */
static enum AwFmReturnCode searchHitsPacked(AwFmGpuIndex *g, const uint64_t *dPacked, uint32_t kmerLength, uint64_t numKmers,
                                            struct AwFmSearchRange *dRanges, uint32_t *dCounts, uint8_t *dCharsScratch,
                                            void *stream, bool rangesOfHitsOnly) {
  if (!g || (!dPacked && numKmers)) {
    setError("awfmGpuSearchHitsPacked: null argument");
    return AwFmNullPtrError;
  }
  if (kmerLength == 0 || kmerLength > (g->amino ? 12u : 32u)) {
    setError("awfmGpuSearchHitsPacked: a packed k-mer holds 1..32 nucleotides or 1..12 amino acids");
    return AwFmIllegalPositionError;
  }
  if (numKmers == 0) return AwFmSuccess;
  if (!g->amino && (g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4)) {
    DeviceGuard guard(g->device);
    const int ordered = awfmGpuOrderedSearch(g, (hipStream_t)stream, (const uint8_t *)dPacked, nullptr, kmerLength, numKmers,
                                             (ulonglong2 *)dRanges, dCounts, true, rangesOfHitsOnly && dCounts);
    if (ordered < 0 && ordered != -(int)AwFmAllocationFailure) return (enum AwFmReturnCode)(-ordered);
    if (ordered > 0) return AwFmSuccess;
  }
  if (!dCharsScratch) {
    setError("awfmGpuSearchHitsPacked: this batch is searched as ASCII and needs dCharsScratch");
    return AwFmNullPtrError;
  }
  const enum AwFmReturnCode rc = awfmGpuUnpackKmers(g, dPacked, kmerLength, numKmers, dCharsScratch, stream);
  if (rc != AwFmSuccess) return rc;
  return (rangesOfHitsOnly && dCounts ? awfmGpuSearchHitsSparse : awfmGpuSearchHits)(g, dCharsScratch, nullptr, kmerLength, numKmers,
                                                                                    dRanges, dCounts, stream);
}

/* dense search of the chunk in the slot, then the list of its k-mers with hits (in k-mer order) made from the counts */
static enum AwFmReturnCode denseToList(AwFmGpuIndex *g, StreamSlot &s, int packed, uint32_t kmerLength, hipStream_t comp) {
  enum AwFmReturnCode rc;
  if (packed)
    rc = searchHitsPacked(g, (const uint64_t *)s.dIn, kmerLength, s.n, (struct AwFmSearchRange *)s.dRanges, (uint32_t *)s.dCounts,
                          (uint8_t *)s.dChars, comp, false);
  else
    rc = awfmGpuSearchHits(g, (const uint8_t *)s.dIn, nullptr, kmerLength, s.n, (struct AwFmSearchRange *)s.dRanges,
                           (uint32_t *)s.dCounts, comp);
  if (rc != AwFmSuccess) return rc;
  return awfmGpuCompactHits(g, (const uint32_t *)s.dCounts, (const struct AwFmSearchRange *)s.dRanges, s.n, (uint64_t *)s.dFlagOffsets,
                            s.dScratch, (uint32_t *)s.dHitKmers, (struct AwFmSearchRange *)s.dHitRanges, (uint32_t)s.capList,
                            (uint32_t *)s.dNumHits, comp);
}

/* the pipeline; packed != 0: `input` is one 64-bit word per k-mer, else kmerLength ASCII characters per k-mer */
static enum AwFmReturnCode streamBatch(AwFmGpuIndex *g, const void *input, int packed, uint32_t kmerLength,
                                       uint64_t numKmers, uint64_t chunkKmers, int locate, unsigned hostThreads,
                                       AwFmGpuChunkSink sink, void *user, AwFmGpuSparseChunkSink sparseSink = nullptr) {
  const bool sparse = sparseSink != nullptr;
  if (!g || (!input && numKmers) || (!sink && !sparseSink)) {
    setError("awfmGpuStream: null argument");
    return AwFmNullPtrError;
  }
  if (kmerLength == 0 || (packed && kmerLength > (g->amino ? 12u : 32u))) {
    setError("awfmGpuStream: a packed k-mer holds 1..32 nucleotides or 1..12 amino acids");
    return AwFmIllegalPositionError;
  }
  if (numKmers == 0) return AwFmSuccess;
  if (g->shares) g = g->shares; /* lanes have no pipeline of their own */
  if (chunkKmers == 0) chunkKmers = 1ull << 24;
  if (chunkKmers > numKmers) chunkKmers = numKmers;
  if (chunkKmers >= 0xFFFFFFFFull) chunkKmers = 0xFFFFFFFEull;
  if (hostThreads == 0) hostThreads = 4;
  DeviceGuard guard(g->device);
  std::lock_guard<std::mutex> lock(g->streamMutex);
  if (!g->streamState) g->streamState = new AwFmGpuStreamState();
  AwFmGpuStreamState &st = *g->streamState;
  StreamSlot *slots = st.slot;
  if (!st.streamsReady) {
    /* all three streams or none: a later failure must not leave a guard satisfied with null streams behind it */
    hipStream_t *streams[] = {&st.slotStream[0], &st.slotStream[1], &st.slotStream[2]};
    hipError_t e = hipSuccess;
    for (hipStream_t *p : streams)
      if (e == hipSuccess && !*p) e = hipStreamCreateWithFlags(p, hipStreamNonBlocking);
    if (e != hipSuccess) {
      setError("awfmGpuStream: hipStreamCreate failed", e);
      for (hipStream_t *p : streams) {
        if (*p) (void)hipStreamDestroy(*p);
        *p = nullptr;
      }
      return AwFmGeneralFailure;
    }
    st.streamsReady = true;
  }
  /* Everything of a chunk runs on its slot's own stream (rounds 3-5 also carried a mode with one upload stream, one kernel
   * stream and download streams: 10^8 planted 21-mers located in 60 ms against 43-58 ms this way, random ones 22.8 against
   * 23-25; retired in round 6) */
  const size_t inBytesPerKmer = packed ? 8 : kmerLength;
  const bool stage = !isPinned(input);
  const bool narrowCounts = g->dev.bwtLength < (1ull << 32);
  const u64 hitBudget = awfmGpuHitBudget(g);
  bool denseList = false; /* sparse results: a chunk's list overflowed, so the batch is not sparse: lists out of dense results from now on */
  const u64 numChunks = (numKmers + chunkKmers - 1) / chunkKmers;
  enum AwFmReturnCode rc = AwFmSuccess;
  auto drain = [&]() {
    for (int i = 0; i < kStreamSlots; i++) (void)hipStreamSynchronize(st.slotStream[i]);
  };
#define STEP_TRY(call)                    \
  do {                                    \
    hipError_t e__ = (call);              \
    if (e__ != hipSuccess) {              \
      setError(#call, e__);               \
      drain();                            \
      return AwFmGeneralFailure;          \
    }                                     \
  } while (0)
#define STEP_RC(call)          \
  do {                         \
    rc = (call);               \
    if (rc != AwFmSuccess) {   \
      drain();                 \
      return rc;               \
    }                          \
  } while (0)

  const bool trace = awfmGpuDiag("stream_trace") != nullptr; /* host-side timeline of the loop on stderr */
  struct timespec ts0;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  auto now = [&]() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (ts.tv_sec - ts0.tv_sec) * 1e3 + (ts.tv_nsec - ts0.tv_nsec) * 1e-6;
  };
  const u64 lag = 2; /* a deeper loop (four slots, the sink three chunks behind) measured no faster */
  for (u64 t = 0; t < numChunks + lag; t++) {
    if (trace) fprintf(stderr, "[stream] t=%llu start %.2f ms\n", (unsigned long long)t, now());
    if (t < numChunks) { /* ---- A(t): upload; search, scan ---- */
      StreamSlot &s = slots[t % kStreamSlots];
      s.first = t * chunkKmers;
      s.n = numKmers - s.first < chunkKmers ? numKmers - s.first : chunkKmers;
      s.total = 0;
      STEP_RC(ensureSlot(s, chunkKmers, inBytesPerKmer, packed ? kmerLength : 0u /* ASCII needs no unpack scratch */, locate != 0, stage));
      const uint8_t *src = (const uint8_t *)input + s.first * inBytesPerKmer;
      const size_t bytes = s.n * inBytesPerKmer;
      if (stage) {
        CopyCtx ctx = {src, (uint8_t *)s.hIn};
        awfmParallelFor(hostThreads, bytes, copyRange, &ctx);
        src = (const uint8_t *)s.hIn;
      }
      hipStream_t in = st.slotStream[t % kStreamSlots], comp = in;
      STEP_TRY(hipMemcpyAsync(s.dIn, src, bytes, hipMemcpyHostToDevice, in));
      STEP_TRY(hipEventRecord(s.uploaded, in));
      STEP_TRY(hipStreamWaitEvent(comp, s.uploaded, 0));
      /* the hit offsets of an image below 2^32 positions are scanned from the counts, and the locate reads a range only
       * when its k-mer has hits: the ranges of the others need not be written */
      if (sparse) {
        /* the k-mers with hits as a list: appended by the seed-order search itself when the chunk takes that path (and
         * few of its k-mers occur: capacity n / 64), otherwise made from the dense results */
        const bool fused = awfmGpuSearchHitsIsOrdered(g, 0, kmerLength, s.n) != 0 && !denseList;
        const uint32_t cap = (uint32_t)(fused ? (s.n / 64 > 1024 ? s.n / 64 : 1024) : s.n);
        STEP_RC(ensureList(s, cap, fused ? 0 : chunkKmers));
        uint32_t capUsed = cap;
        if (fused) {
          rc = awfmGpuSearchHitsCompact(g, (const uint8_t *)s.dIn, nullptr, kmerLength, s.n, packed, (uint32_t *)s.dHitKmers,
                                        (struct AwFmSearchRange *)s.dHitRanges, cap, (uint32_t *)s.dNumHits, comp);
          if (rc == AwFmAllocationFailure) {
            /* no memory for the seed-order scratch: the list out of dense results (general kernel), for this chunk and the rest */
            denseList = true;
            capUsed = (uint32_t)s.n;
            STEP_RC(ensureList(s, capUsed, chunkKmers));
            STEP_RC(denseToList(g, s, packed, kmerLength, comp));
          } else {
            STEP_RC(rc);
            STEP_RC(awfmGpuSortHits(g, (uint32_t *)s.dHitKmers, (struct AwFmSearchRange *)s.dHitRanges, cap, comp));
          }
        } else {
          STEP_RC(denseToList(g, s, packed, kmerLength, comp));
        }
        s.capUsed = capUsed;
        STEP_RC(awfmGpuHitOffsetsAsync(g, nullptr, (const struct AwFmSearchRange *)s.dHitRanges, capUsed, (uint64_t *)s.dListOffsets,
                                       s.dScratch, s.hTotal, comp));
        STEP_TRY(hipMemcpyAsync(s.hTotal + 1, s.dNumHits, 4, hipMemcpyDeviceToHost, comp));
      } else {
      if (packed)
        STEP_RC(searchHitsPacked(g, (const uint64_t *)s.dIn, kmerLength, s.n,
                                 locate ? (struct AwFmSearchRange *)s.dRanges : nullptr, (uint32_t *)s.dCounts,
                                 (uint8_t *)s.dChars, comp, narrowCounts));
      else
        STEP_RC((narrowCounts ? awfmGpuSearchHitsSparse : awfmGpuSearchHits)(
            g, (const uint8_t *)s.dIn, nullptr, kmerLength, s.n, locate ? (struct AwFmSearchRange *)s.dRanges : nullptr,
            (uint32_t *)s.dCounts, comp));
      if (locate)
        STEP_RC(awfmGpuHitOffsetsAsync(g, narrowCounts ? (const uint32_t *)s.dCounts : nullptr,
                                       (const struct AwFmSearchRange *)s.dRanges, s.n, (uint64_t *)s.dHitOffsets, s.dScratch,
                                       s.hTotal, comp));
      }
      STEP_TRY(hipEventRecord(s.searched, comp));
    }
    if (t >= 1 && t - 1 < numChunks) { /* ---- B(t-1): locate; download ---- */
      StreamSlot &s = slots[(t - 1) % kStreamSlots];
      hipStream_t out = st.slotStream[(t - 1) % kStreamSlots], comp = out;
      if (sparse) {
        STEP_TRY(hipEventSynchronize(s.searched));
        u64 listed = *(const uint32_t *)(s.hTotal + 1);
        if (listed > s.capUsed) { /* more k-mers with hits than the list holds: this chunk again, densely */
          denseList = true;
          STEP_RC(ensureList(s, s.n, chunkKmers));
          STEP_RC(denseToList(g, s, packed, kmerLength, comp));
          s.capUsed = (uint32_t)s.n;
          STEP_RC(awfmGpuHitOffsetsAsync(g, nullptr, (const struct AwFmSearchRange *)s.dHitRanges, s.capUsed, (uint64_t *)s.dListOffsets,
                                         s.dScratch, s.hTotal, comp));
          STEP_TRY(hipMemcpyAsync(s.hTotal + 1, s.dNumHits, 4, hipMemcpyDeviceToHost, comp));
          STEP_TRY(hipStreamSynchronize(comp));
          listed = *(const uint32_t *)(s.hTotal + 1);
        }
        s.total = *s.hTotal;
        s.numHitKmers = listed;
        s.windowed = false;
        if (trace)
          fprintf(stderr, "[stream] t=%llu sparse chunk %llu: %llu k-mers with hits (list of %u), %llu hits, budget %llu\n", (unsigned long long)t,
                  (unsigned long long)(t - 1), (unsigned long long)listed, s.capUsed, (unsigned long long)s.total, (unsigned long long)hitBudget);
        if (locate && s.total > hitBudget) {
          setError("awfmGpuStream (sparse results): the hits of a chunk exceed the device's hit budget; use smaller chunks, or the "
                   "dense pipeline, which takes such a chunk in windows");
          drain();
          return AwFmAllocationFailure;
        }
        if (locate && s.total) {
          STEP_RC(ensurePositions(s, s.total));
          STEP_RC(awfmGpuLocateTo(g, (const struct AwFmSearchRange *)s.dHitRanges, (const uint64_t *)s.dListOffsets, s.capUsed, s.total,
                                  (uint64_t *)s.dPositions, (uint64_t *)s.dPositions, comp));
        }
        STEP_TRY(hipEventRecord(s.located, comp));
        STEP_TRY(hipStreamWaitEvent(out, s.located, 0));
        if (locate && s.total) STEP_TRY(hipMemcpyAsync(s.hPositions, s.dPositions, s.total * 8, hipMemcpyDeviceToHost, out));
        if (listed) STEP_TRY(hipMemcpyAsync(s.hHitKmers, s.dHitKmers, listed * 4, hipMemcpyDeviceToHost, out));
        STEP_TRY(hipMemcpyAsync(s.hListOffsets, s.dListOffsets, (listed + 1) * 8, hipMemcpyDeviceToHost, out));
        STEP_TRY(hipEventRecord(s.done, out));
      } else {
      if (locate) {
        STEP_TRY(hipEventSynchronize(s.searched));
        if (trace) fprintf(stderr, "[stream] t=%llu searched(%llu) seen %.2f ms\n", (unsigned long long)t, (unsigned long long)(t - 1), now());
        s.total = *s.hTotal;
        /* a chunk whose hit list exceeds what may be resident on the device is located window by window when it is its
         * turn to be handed over (stage C), so that chunks still arrive in order */
        s.windowed = s.total > hitBudget;
        if (s.total && !s.windowed) {
          STEP_RC(ensurePositions(s, s.total));
          /* (the finish kernel storing into the page-locked staging itself instead of a copy afterwards measured the same:
           * 10^8 planted 21-mers, 46-54 ms either way) */
          STEP_RC(awfmGpuLocateTo(g, (const struct AwFmSearchRange *)s.dRanges, (const uint64_t *)s.dHitOffsets, s.n, s.total,
                                  (uint64_t *)s.dPositions, (uint64_t *)s.dPositions, comp));
        }
        STEP_TRY(hipEventRecord(s.located, comp));
        STEP_TRY(hipStreamWaitEvent(out, s.located, 0));
        if (s.total && !s.windowed) STEP_TRY(hipMemcpyAsync(s.hPositions, s.dPositions, s.total * 8, hipMemcpyDeviceToHost, out));
      } else {
        STEP_TRY(hipStreamWaitEvent(out, s.searched, 0));
      }
      STEP_TRY(hipMemcpyAsync(s.hCounts, s.dCounts, s.n * 4, hipMemcpyDeviceToHost, out));
      STEP_TRY(hipEventRecord(s.done, out));
      }
    }
    if (t >= lag) { /* ---- C(t-lag): hand the chunk to the caller ---- */
      StreamSlot &s = slots[(t - lag) % kStreamSlots];
      STEP_TRY(hipEventSynchronize(s.done));
      if (trace) fprintf(stderr, "[stream] t=%llu done(%llu) seen %.2f ms\n", (unsigned long long)t, (unsigned long long)(t - lag), now());
      int stop = 0;
      if (sparse) {
        stop = sparseSink(user, s.first, s.n, s.numHitKmers, s.hHitKmers, (const uint64_t *)s.hListOffsets, locate ? (const uint64_t *)s.hPositions : nullptr,
                          locate ? s.total : 0);
      } else if (locate && s.windowed) {
        /* Hit-budgeted hand-over: the chunk's k-mers go to the sink in consecutive groups whose hit lists fit one window
         * (half the budget); a k-mer whose own list is longer goes alone, slice by slice (same firstKmer, numKmers = 1,
         * counts[0] its full count every time). */
        hipStream_t ws = st.slotStream[(t - lag) % kStreamSlots];
        if ((s.n + 1) * 8 > s.hCapOffsets) {
          if (s.hOffsets) (void)hipHostFree(s.hOffsets);
          s.hOffsets = nullptr;
          s.hCapOffsets = 0;
          STEP_TRY(hipHostMalloc((void **)&s.hOffsets, (s.capKmers + 1) * 8, hipHostMallocDefault));
          s.hCapOffsets = (s.capKmers + 1) * 8;
        }
        STEP_TRY(hipMemcpyAsync(s.hOffsets, s.dHitOffsets, (s.n + 1) * 8, hipMemcpyDeviceToHost, ws));
        STEP_TRY(hipStreamSynchronize(ws));
        const u64 window = hitBudget / 2 > 0 ? hitBudget / 2 : 1;
        STEP_RC(ensurePositions(s, window));
        const u64 *off = s.hOffsets;
        auto deliver = [&](u64 qb, u64 qe, u64 hb, u64 he) -> enum AwFmReturnCode {
          if (he > hb) {
            const enum AwFmReturnCode r = awfmGpuLocateWindow(g, (const struct AwFmSearchRange *)s.dRanges, (const uint64_t *)s.dHitOffsets,
                                                              qb, qe, hb, he, (uint64_t *)s.dPositions, (uint64_t *)s.dPositions, ws);
            if (r != AwFmSuccess) return r;
            if (hipMemcpyAsync(s.hPositions, s.dPositions, (he - hb) * 8, hipMemcpyDeviceToHost, ws) != hipSuccess ||
                hipStreamSynchronize(ws) != hipSuccess) {
              setError("awfmGpuStream: locating a window of hits failed", hipGetLastError());
              return AwFmGeneralFailure;
            }
          }
          stop = sink(user, s.first + qb, qe - qb, s.hCounts + qb, (const uint64_t *)s.hPositions, he - hb);
          return AwFmSuccess;
        };
        for (u64 q = 0; q < s.n && !stop;) {
          const u64 hb = off[q];
          u64 lo = q, hi = s.n; /* the last qe in (q, n] with off[qe] - hb <= window, if any */
          while (lo < hi) {
            const u64 mid = lo + (hi - lo + 1) / 2;
            if (off[mid] - hb <= window) lo = mid;
            else hi = mid - 1;
          }
          if (lo > q) {
            STEP_RC(deliver(q, lo, hb, off[lo]));
            q = lo;
          } else { /* one k-mer above a window */
            for (u64 h = hb; h < off[q + 1] && !stop; h += window)
              STEP_RC(deliver(q, q + 1, h, h + window < off[q + 1] ? h + window : off[q + 1]));
            q++;
          }
        }
      } else {
        stop = sink(user, s.first, s.n, s.hCounts, locate ? (const uint64_t *)s.hPositions : nullptr, s.total);
      }
      if (stop != 0) {
        setError("awfmGpuStream: the sink asked to stop");
        drain();
        return AwFmGeneralFailure;
      }
    }
  }
#undef STEP_TRY
#undef STEP_RC
  return rc;
}

enum AwFmReturnCode awfmGpuStreamPacked(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                        uint64_t numKmers, uint64_t chunkKmers, int locate, unsigned hostThreads,
                                        AwFmGpuChunkSink sink, void *user) {
  return streamBatch(g, packedKmers, 1, kmerLength, numKmers, chunkKmers, locate, hostThreads, sink, user);
}

enum AwFmReturnCode awfmGpuStreamChars(AwFmGpuIndex *g, const uint8_t *chars, uint32_t kmerLength, uint64_t numKmers,
                                       uint64_t chunkKmers, int locate, unsigned hostThreads, AwFmGpuChunkSink sink,
                                       void *user) {
  return streamBatch(g, chars, 0, kmerLength, numKmers, chunkKmers, locate, hostThreads, sink, user);
}

enum AwFmReturnCode awfmGpuStreamPackedSparse(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                              uint64_t numKmers, uint64_t chunkKmers, int locate, unsigned hostThreads,
                                              AwFmGpuSparseChunkSink sink, void *user) {
  if (!sink) {
    setError("awfmGpuStreamPackedSparse: null sink");
    return AwFmNullPtrError;
  }
  return streamBatch(g, packedKmers, 1, kmerLength, numKmers, chunkKmers, locate, hostThreads, nullptr, user, sink);
}

enum AwFmReturnCode awfmGpuStreamCharsSparse(AwFmGpuIndex *g, const uint8_t *chars, uint32_t kmerLength, uint64_t numKmers,
                                             uint64_t chunkKmers, int locate, unsigned hostThreads, AwFmGpuSparseChunkSink sink,
                                             void *user) {
  if (!sink) {
    setError("awfmGpuStreamCharsSparse: null sink");
    return AwFmNullPtrError;
  }
  return streamBatch(g, chars, 0, kmerLength, numKmers, chunkKmers, locate, hostThreads, nullptr, user, sink);
}

/* ---- whole-batch convenience on top of the pipeline: results gathered into caller arrays ---- */
struct GatherCtx {
  uint32_t *counts;
  uint64_t *positions;
  uint64_t capacity, used;
  unsigned threads;
  int failed;
};
static int gatherSink(void *user, uint64_t first, uint64_t n, const uint32_t *counts, const uint64_t *positions,
                      uint64_t numPositions) {
  GatherCtx *c = (GatherCtx *)user;
  {
    CopyCtx ctx = {(const uint8_t *)counts, (uint8_t *)(c->counts + first)};
    awfmParallelFor(c->threads, n * 4, copyRange, &ctx);
  }
  if (positions && numPositions) {
    if (c->used + numPositions > c->capacity) {
      const uint64_t want = (c->used + numPositions) * 2;
      uint64_t *grown = (uint64_t *)realloc(c->positions, want * 8);
      if (!grown) {
        c->failed = 1;
        return 1;
      }
      c->positions = grown;
      c->capacity = want;
    }
    CopyCtx ctx = {(const uint8_t *)positions, (uint8_t *)(c->positions + c->used)};
    awfmParallelFor(c->threads, numPositions * 8, copyRange, &ctx);
    c->used += numPositions;
  }
  return 0;
}

enum AwFmReturnCode awfmGpuCountPackedHost(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                           uint64_t numKmers, uint32_t *counts) {
  if (!counts && numKmers) {
    setError("awfmGpuCountPackedHost: null argument");
    return AwFmNullPtrError;
  }
  GatherCtx ctx = {counts, nullptr, 0, 0, 4, 0};
  return awfmGpuStreamPacked(g, packedKmers, kmerLength, numKmers, 0, 0, 4, gatherSink, &ctx);
}

enum AwFmReturnCode awfmGpuLocatePackedHost(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                            uint64_t numKmers, uint32_t *counts, uint64_t **positions,
                                            uint64_t *numPositions) {
  if ((!counts && numKmers) || !positions || !numPositions) {
    setError("awfmGpuLocatePackedHost: null argument");
    return AwFmNullPtrError;
  }
  *positions = nullptr;
  *numPositions = 0;
  GatherCtx ctx = {counts, (uint64_t *)malloc(8 * (numKmers + 16)), numKmers + 16, 0, 4, 0};
  if (!ctx.positions) {
    setError("awfmGpuLocatePackedHost: host allocation failed");
    return AwFmAllocationFailure;
  }
  const enum AwFmReturnCode rc = awfmGpuStreamPacked(g, packedKmers, kmerLength, numKmers, 0, 1, 4, gatherSink, &ctx);
  if (rc != AwFmSuccess || ctx.failed) {
    free(ctx.positions);
    if (ctx.failed) setError("awfmGpuLocatePackedHost: host allocation failed");
    return ctx.failed ? AwFmAllocationFailure : rc;
  }
  *positions = ctx.positions;
  *numPositions = ctx.used;
  return AwFmSuccess;
}

}  // extern "C"
