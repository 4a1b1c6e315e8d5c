/*
 * awfm_gpu_image.hip -- the device image of an index and its lifetime: upload + re-layout of the reference's arrays
 * (awfm_device.h), the registry behind awFmParallelSearch* (one image per index and device, lanes on it), the device-only
 * accelerators an image gets by its size (pair image, deeper seed table with next-step bits; the full suffix array:
 * awfm_gpu_dense_sa.hip), their setters, and what an image says of itself.  ref src/AwFmIndex.h:55-109, src/AwFmIndexStruct.c.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "awfm_device.h"
#include "awfm_search_kernel.h"
#include "awfm_locate_kernel.h"

thread_local hipStream_t awfmGpuSetupStream = nullptr; /* awfm_device.h */
static thread_local std::string tlsError;
void awfmGpuSetError(const char *what) { tlsError = what; }
void awfmGpuSetHipError(const char *what, hipError_t e) { tlsError = std::string(what) + ": " + hipGetErrorString(e); }

/* ------------------------------------------------------------------ host side */

namespace {

std::mutex tableMutex;
struct ImageEntry {
  const AwFmIndex *index;
  int device; /* HIP ordinal the image lives on */
  int lane;   /* 0 = the image itself; n = the n-th extra handle on it (a device named again in $AWFM_GPU_DEVICES) */
  AwFmGpuIndex *image;
};
std::vector<ImageEntry> imageTable;



}  // namespace
enum AwFmReturnCode awfmGpuEnsureWork(AwFmGpuIndex *g, size_t bytes) {
  if (bytes <= g->workBytes) return AwFmSuccess;
  if (g->dWork) (void)hipFree(g->dWork);
  g->dWork = nullptr;
  g->workBytes = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  AWFM_HIP_TRY(hipMalloc(&g->dWork, want), AwFmAllocationFailure);
  g->workBytes = want;
  return AwFmSuccess;
}


namespace {
void fillDevIndex(AwFmGpuIndex *g, const struct AwFmIndex *index, unsigned superShift, unsigned long long sentinelPos) {
  DevIndex &d = g->dev;
  d.blocks = (const uint4 *)g->dBlocks;
  d.super = (const unsigned long long *)g->dSuper;
  d.numSuper = (unsigned)awfmNumSuper(index->bwtLength, index->config.alphabetType == AwFmAlphabetAmino, superShift);
  d.nucSuperShift = superShift;
  d.seed = (const ulonglong2 *)g->dSeed;
  d.sa = (const unsigned long long *)g->dSa;
  d.bwtLength = index->bwtLength;
  d.sentinelPos = sentinelPos;
  d.seedLen = awfmKmerTableLength(index->config.alphabetType, index->config.kmerLengthInSeedTable);
  d.prefixSums = (const unsigned long long *)g->dPrefix;
  d.saRatio = index->config.suffixArrayCompressionRatio;
  d.saShift = 0xFFFFFFFFu;
  if ((d.saRatio & (d.saRatio - 1)) == 0) {
    d.saShift = 0;
    while ((1u << d.saShift) < d.saRatio) d.saShift++;
  }
  d.saWidth = index->suffixArray.valueBitWidth;
  d.seedK = index->config.kmerLengthInSeedTable;
  d.deepSeed = nullptr;
  d.deepK = 0;
  d.deepNarrow = 0;
  d.pairBlocks = nullptr;
  d.pairSuper = nullptr;
  d.pairSuper32 = nullptr;
  d.pairC = nullptr;
  d.numPairSuper = 0;
  d.pairSuperInLds = 0;
}
}  // namespace

/* lanes that cooperate on one query: image setting, else $AWFM_GPU_DIAG kernel=g4|g2|g1, else the default.  A device
 * block has 4 slices, so 4 lanes is the widest group (GROUP8 of the enum maps to it); amino slices are 32 B, 2 lanes
 * per query already hold 64 registers of block data */
int awfmGpuLanesPerQuery(const AwFmGpuIndex *g) {
  int lanes = 4;
  switch (g->kernel) {
    case AWFM_GPU_KERNEL_GROUP8:
    case AWFM_GPU_KERNEL_GROUP4: lanes = 4; break;
    case AWFM_GPU_KERNEL_GROUP2: lanes = 2; break;
    case AWFM_GPU_KERNEL_GROUP1: lanes = 1; break;
    default:
      /* measured on MI355X (scripts/ab_layout.sh): 10^8 random 21-mers against the GRCh38-sized index, general
       * kernel: g4 13.2-13.7 ms, g2 13.0-13.4, g1 13.9 (within the box-to-box spread: the kernel runs at the rate the
       * chip delivers random granules; g4 keeps 8 waves per SIMD without spilling); 5*10^7 amino 10-mers: g4 3.91 ms,
       * g2 3.72 */
      lanes = g->amino ? 2 : 4;
      if (const char *env = awfmGpuDiag("kernel")) { /* lanes per k-mer of the general kernel: g4 | g2 | g1 */
        if (!strcmp(env, "g8") || !strcmp(env, "g4")) lanes = 4;
        else if (!strcmp(env, "g2")) lanes = 2;
        else if (!strcmp(env, "g1")) lanes = 1;
      }
  }
  if (g->amino && lanes < 2) lanes = 2;
  return lanes;
}

constexpr unsigned kAutoDeepSeedMin = 14, kAutoDeepSeedMax = 16; /* depths of the device-only seed table large nucleotide images get by default */
extern "C" {
static enum AwFmReturnCode applyDeepSeedFromEnv(AwFmGpuIndex *g);
static enum AwFmReturnCode applyPairFromEnv(AwFmGpuIndex *g);
}

AwFmGpuIndex *awfmGpuIndexAdopt(const struct AwFmIndex *index, int device, void *dBlocks, void *dSuper, unsigned superShift,
                                void *dSeed, void *dSa, void *dPrefix, unsigned long long sentinelPos, uint64_t deviceBytes) {
  AwFmGpuIndex *g = new AwFmGpuIndex();
  g->device = device;
  g->amino = index->config.alphabetType == AwFmAlphabetAmino;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    g->numCUs = prop.multiProcessorCount;
  g->numBlocks = awfmDeviceBlocks(index->bwtLength);
  g->dBlocks = dBlocks;
  g->dSuper = dSuper;
  g->dSeed = dSeed;
  g->dSa = dSa;
  g->dPrefix = dPrefix;
  g->deviceBytes = deviceBytes;
  fillDevIndex(g, index, superShift, sentinelPos);
  if (const char *env = awfmKnob(AWFM_KNOB_FORCE_WIDE)) g->forceWide = atoi(env) != 0;
  (void)applyPairFromEnv(g);     /* first: the deeper table's next-step bits are computed through the pair image */
  (void)applyDeepSeedFromEnv(g); /* optional accelerator: on failure the image simply has no deeper table */
  (void)awfmGpuApplyDenseSaAuto(g);  /* the same: without it a locate walks */
  return g;
}

void awfmGpuIndexRegister(const struct AwFmIndex *index, AwFmGpuIndex *g) {
  std::lock_guard<std::mutex> lock(tableMutex);
  imageTable.push_back({index, g->device, 0, g});
}

bool awfmGpuRelayout(const void *dRefBlocks, uint64_t bwtLength, bool amino, unsigned superShift, void *dBlocks,
                     void *dSuper, unsigned long long *sentinelPosOut) {
  const uint64_t numRef = awfmNumBlocks(bwtLength);
  const unsigned numSuper = (unsigned)awfmNumSuper(bwtLength, amino, superShift);
  unsigned long long *dSentinel = nullptr;
  hipError_t e = hipMalloc((void **)&dSentinel, 8);
  if (e == hipSuccess) e = hipMemset(dSentinel, 0, 8);
  if (e == hipSuccess) {
    const unsigned words = numSuper * (amino ? kAminoSuperStride : 4u);
    hipLaunchKernelGGL(gatherSuperKernel, dim3((words + 255) / 256), dim3(256), 0, 0, (const unsigned long long *)dRefBlocks,
                       (unsigned long long)numRef, amino ? 1 : 0, superShift, numSuper, (unsigned long long *)dSuper);
    const uint64_t threads = numRef * 2 * kSlices;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (amino)
      hipLaunchKernelGGL(relayoutAminoKernel, dim3(grid), dim3(256), 0, 0, (const unsigned long long *)dRefBlocks,
                         (unsigned long long)numRef, (unsigned long long)bwtLength, (const unsigned long long *)dSuper,
                         (uint4 *)dBlocks, dSentinel);
    else
      hipLaunchKernelGGL(relayoutNucKernel, dim3(grid), dim3(256), 0, 0, (const unsigned long long *)dRefBlocks,
                         (unsigned long long)numRef, (unsigned long long)bwtLength, superShift,
                         (const unsigned long long *)dSuper, (uint4 *)dBlocks, dSentinel);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(sentinelPosOut, dSentinel, 8, hipMemcpyDeviceToHost);
  if (dSentinel) (void)hipFree(dSentinel);
  if (e != hipSuccess) {
    setError("awfmGpuRelayout", e);
    return false;
  }
  return true;
}

extern "C" {

int awfmGpuDeviceCount(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *awfmGpuLastError(void) { return tlsError.c_str(); }

static enum AwFmReturnCode createImage(const struct AwFmIndex *index, int device, AwFmGpuIndex **out, bool deferAccelerators);
enum AwFmReturnCode awfmGpuIndexCreate(const struct AwFmIndex *index, int device, AwFmGpuIndex **out) {
  return createImage(index, device, out, false);
}
static void accelBuilder(AwFmGpuIndex *g);
/* images whose builder thread has not been joined yet */
static std::mutex buildingMutex;
static std::vector<AwFmGpuIndex *> building;
static void forgetBuilder(AwFmGpuIndex *g) {
  std::lock_guard<std::mutex> lock(buildingMutex);
  for (size_t i = 0; i < building.size(); i++)
    if (building[i] == g) {
      building[i] = building.back();
      building.pop_back();
      break;
    }
}
static void settleBuildersAtExit() {
  std::vector<AwFmGpuIndex *> left;
  {
    std::lock_guard<std::mutex> lock(buildingMutex);
    left.swap(building);
  }
  for (AwFmGpuIndex *g : left)
    if (g->accelThread.joinable()) g->accelThread.join();
}
static unsigned chooseDeepSeedK(const AwFmGpuIndex *g, std::string &notes);
static enum AwFmReturnCode buildDeepSeed(AwFmGpuIndex *g, unsigned deepK, AwFmGpuIndex::PendingAccel *to);
static void installDeepSeed(AwFmGpuIndex *g, AwFmGpuIndex::PendingAccel *from, const std::vector<AwFmGpuIndex *> &laneList);
static enum AwFmReturnCode createImage(const struct AwFmIndex *index, int device, AwFmGpuIndex **out, bool deferAccelerators) {
  if (!index || !out) {
    setError("awfmGpuIndexCreate: null argument");
    return AwFmNullPtrError;
  }
  *out = nullptr;
  if (awfmGpuDeviceCount() <= 0) {
    setError("awfmGpuIndexCreate: no HIP device available (this library has no CPU search path)");
    return AwFmGeneralFailure;
  }
  if (device < 0) {
    const char *env = awfmKnob(AWFM_KNOB_DEVICE);
    if (env && *env) {
      device = atoi(env);
    } else if (hipGetDevice(&device) != hipSuccess) {
      device = 0;
    }
  }
  DeviceGuard guard(device);
  if (!guard.ok) {
    setError("awfmGpuIndexCreate: hipSetDevice failed");
    return AwFmGeneralFailure;
  }
  const bool amino = index->config.alphabetType == AwFmAlphabetAmino;
  const unsigned superShift = awfmSuperShift(amino, index->bwtLength);
  if (!amino && awfmNumSuper(index->bwtLength, false, superShift) > kMaxNucSuper) {
    setError("awfmGpuIndexCreate: nucleotide device images hold at most 64 superblocks (2^38 positions)");
    return AwFmUnsupportedVersionError;
  }
  if (index->config.suffixArrayCompressionRatio == 0) {
    setError("awfmGpuIndexCreate: suffixArrayCompressionRatio must be >= 1");
    return AwFmGeneralFailure;
  }

  AwFmGpuIndex *g = new AwFmGpuIndex();
  g->device = device;
  g->amino = amino;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    g->numCUs = prop.multiProcessorCount;
  g->numBlocks = awfmDeviceBlocks(index->bwtLength);
  const size_t refBytes = awfmNumBlocks(index->bwtLength) * awfmBlockBytes(index->config.alphabetType);
  const size_t devBlockBytes = g->numBlocks * awfmDeviceBlockBytes(amino);
  const size_t superBytes = awfmSuperBytes(index->bwtLength, amino, superShift);
  const uint64_t seedLen = awfmKmerTableLength(index->config.alphabetType, index->config.kmerLengthInSeedTable);
  const size_t seedBytes = seedLen * sizeof(struct AwFmSearchRange);
  const size_t saBytes = index->suffixArray.compressedByteLength;
  const size_t saAlloc = alignUp(saBytes, 16) + 256; /* the locate kernel reads a 128-byte window at a sample */

  auto fail = [&](enum AwFmReturnCode rc) {
    awfmGpuIndexDestroy(g);
    return rc;
  };
  void *dRef = nullptr;
#define TRY_OR_FAIL(call, rc)                 \
  do {                                        \
    hipError_t e__ = (call);                  \
    if (e__ != hipSuccess) {                  \
      setError(#call, e__);                   \
      if (dRef) (void)hipFree(dRef);          \
      return fail(rc);                        \
    }                                         \
  } while (0)

  TRY_OR_FAIL(hipMalloc(&g->dBlocks, devBlockBytes), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSuper, superBytes), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSeed, seedBytes ? seedBytes : 16), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSa, saAlloc), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&dRef, refBytes), AwFmAllocationFailure);
  g->deviceBytes = devBlockBytes + superBytes + seedBytes + saAlloc;

  TRY_OR_FAIL(hipMemcpy(dRef, index->bwtBlockList.asNucleotide, refBytes, hipMemcpyHostToDevice), AwFmGeneralFailure);
  unsigned long long sentinelPos = 0;
  if (!awfmGpuRelayout(dRef, index->bwtLength, amino, superShift, g->dBlocks, g->dSuper, &sentinelPos)) {
    (void)hipFree(dRef);
    return fail(AwFmGeneralFailure);
  }
  (void)hipFree(dRef);
  dRef = nullptr;

  TRY_OR_FAIL(hipMemcpy(g->dSeed, index->kmerSeedTable, seedBytes, hipMemcpyHostToDevice), AwFmGeneralFailure);
  {
    unsigned long long prefix[24] = {0};
    memcpy(prefix, index->prefixSums, awfmPrefixSumsLength(index->config.alphabetType) * sizeof(uint64_t));
    TRY_OR_FAIL(hipMalloc(&g->dPrefix, sizeof prefix), AwFmAllocationFailure);
    TRY_OR_FAIL(hipMemcpy(g->dPrefix, prefix, sizeof prefix, hipMemcpyHostToDevice), AwFmGeneralFailure);
  }

  /* sampled SA: from memory, or staged from the index file (keepSuffixArrayInMemory == false) */
  TRY_OR_FAIL(hipMemset(g->dSa, 0, saAlloc), AwFmGeneralFailure);
  if (index->suffixArray.values) {
    TRY_OR_FAIL(hipMemcpy(g->dSa, index->suffixArray.values, saBytes, hipMemcpyHostToDevice), AwFmGeneralFailure);
  } else {
    uint8_t *staged = awfmReadPackedSaFromFile(index);
    if (!staged) {
      setError("awfmGpuIndexCreate: index has no in-memory suffix array and it could not be read from its file");
      return fail(AwFmFileReadFail);
    }
    hipError_t e = hipMemcpy(g->dSa, staged, saBytes, hipMemcpyHostToDevice);
    free(staged);
    TRY_OR_FAIL(e, AwFmGeneralFailure);
  }
#undef TRY_OR_FAIL

  fillDevIndex(g, index, superShift, sentinelPos);
  if (const char *env = awfmKnob(AWFM_KNOB_FORCE_WIDE)) g->forceWide = atoi(env) != 0;
  (void)applyPairFromEnv(g); /* without it (no memory left) searches simply take one step per read */
  if (deferAccelerators) {
    /* the image is usable now (general kernel from the index's own table, pair steps, LF walk); the deeper table and the full
     * suffix array are built by a thread of their own, on a stream of their own, and installed between two calls */
    g->accelState.store(1);
    g->accelThread = std::thread(accelBuilder, g);
    { /* a program that exits while a builder is at work (no awFmDeallocIndex) must not tear the runtime down under it */
      std::lock_guard<std::mutex> lock(buildingMutex);
      static bool hooked = false;
      if (!hooked) hooked = atexit(settleBuildersAtExit) == 0;
      building.push_back(g);
    }
  } else {
    if (applyDeepSeedFromEnv(g) != AwFmSuccess) return fail(AwFmGeneralFailure);
    (void)awfmGpuApplyDenseSaAuto(g); /* optional accelerator: without it (no memory left) a locate walks */
  }
  *out = g;
  return AwFmSuccess;
}

/* the thread that builds an image's deeper table and full suffix array behind its first searches: the choices the
 * synchronous construction makes (by the image's size and the free memory), on a non-blocking stream of the thread's own */
static void accelBuilder(AwFmGpuIndex *g) {
  DeviceGuard guard(g->device);
  hipStream_t own = nullptr;
  if (hipStreamCreateWithFlags(&own, hipStreamNonBlocking) != hipSuccess) own = nullptr; /* (the null stream then: correct, not hidden) */
  awfmGpuSetupStream = own;
  AwFmGpuIndex::PendingAccel &to = g->pendingAccel;
  const unsigned deepK = chooseDeepSeedK(g, to.notes);
  if (deepK != 0 && buildDeepSeed(g, deepK, &to) != AwFmSuccess) to.notes += "deeper table: construction failed; ";
  (void)awfmGpuBuildDenseSaAuto(g, &to.dense, &to.denseWide, &to.denseBytes, &to.denseSeconds, &to.notes);
  (void)awfmGpuSetupSync();
  awfmGpuSetupStream = nullptr;
  if (own) (void)hipStreamDestroy(own);
  (void)hipGetLastError();
  g->accelState.store(2);
}

}  // extern "C"
void awfmGpuAdoptAccelerators(AwFmGpuIndex *g, bool wait, const std::vector<AwFmGpuIndex *> *lanesIn) {
  if (!g || g->shares) return;
  if (g->accelState.load() == 0) return;
  if (wait) {
    if (g->accelThread.joinable()) {
      g->accelThread.join();
      forgetBuilder(g);
    }
  } else if (g->accelState.load() != 2) {
    return;
  }
  if (g->accelState.load() != 2) return;
  std::vector<AwFmGpuIndex *> lanes = lanesIn ? *lanesIn : awfmGpuLanesOf(g);
  /* nobody is enqueuing a search through the image or one of its lanes while its view changes: all their locks, or (not
   * waiting) none and another time */
  std::vector<std::mutex *> want = {&g->aosMutex, &g->workMutex, &g->orderMutex};
  for (AwFmGpuIndex *lane : lanes) {
    want.push_back(&lane->aosMutex);
    want.push_back(&lane->workMutex);
    want.push_back(&lane->orderMutex);
  }
  size_t held = 0;
  for (; held < want.size(); held++) {
    if (wait) want[held]->lock();
    else if (!want[held]->try_lock()) break;
  }
  if (held == want.size() && g->accelState.load() == 2) {
    if (g->accelThread.joinable()) {
      g->accelThread.join();
      forgetBuilder(g);
    }
    AwFmGpuIndex::PendingAccel &from = g->pendingAccel;
    if (from.deepTable) installDeepSeed(g, &from, lanes);
    if (from.dense) {
      g->dDenseSa = from.dense;
      g->denseWide = from.denseWide;
      g->denseSaBytes = from.denseBytes;
      g->denseSaBuildSeconds = from.denseSeconds;
      from.dense = nullptr;
      for (AwFmGpuIndex *lane : lanes) {
        lane->dDenseSa = g->dDenseSa;
        lane->denseWide = g->denseWide;
      }
    }
    g->accelNotes += from.notes;
    from.notes.clear();
    g->accelState.store(0);
  }
  while (held > 0) want[--held]->unlock();
}
extern "C" {

void awfmGpuIndexDestroy(AwFmGpuIndex *g) {
  if (!g) return;
  if (g->accelThread.joinable()) g->accelThread.join();
  forgetBuilder(g);
  {
    DeviceGuard guard(g->device);
    if (g->pendingAccel.deepTable) (void)hipFree(g->pendingAccel.deepTable); /* (built, never installed) */
    if (g->pendingAccel.deepBig) (void)hipFree(g->pendingAccel.deepBig);
    if (g->pendingAccel.dense) (void)hipFree(g->pendingAccel.dense);
    awfmGpuStreamStateFree(g);
    if (!g->shares) { /* a lane owns only its staging */
      if (g->dBlocks) (void)hipFree(g->dBlocks);
      if (g->dSuper) (void)hipFree(g->dSuper);
      if (g->dSeed) (void)hipFree(g->dSeed);
      if (g->dSa) (void)hipFree(g->dSa);
      if (g->dPrefix) (void)hipFree(g->dPrefix);
      if (g->dDeepSeed) (void)hipFree(g->dDeepSeed);
      if (g->dDeepBig) (void)hipFree(g->dDeepBig);
      if (g->dDenseSa) (void)hipFree(g->dDenseSa);
      if (g->dLengthTable) (void)hipFree(g->dLengthTable);
      if (g->dLengthBig) (void)hipFree(g->dLengthBig);
      void *pairOwned[] = {g->dPairBlocks, g->dPairSuper, g->dPairSuper32, g->dPairC};
      for (void *p : pairOwned)
        if (p) (void)hipFree(p);
    }
    if (g->dWork) (void)hipFree(g->dWork);
    if (g->dHits) (void)hipFree(g->dHits);
    for (auto &slot : g->orderSlot) {
      if (slot.mem) (void)hipFree(slot.mem);
      if (slot.gate.done) (void)hipEventDestroy(slot.gate.done);
    }
    if (g->dSparse) (void)hipFree(g->dSparse);
    if (g->sparseGate.done) (void)hipEventDestroy(g->sparseGate.done);
    for (auto &entry : g->orderLog)
      for (int i = 0; i < 4; i++)
        if (entry.ev[i]) (void)hipEventDestroy(entry.ev[i]);
    for (int i = 0; i < 2; i++)
      if (g->windowEvent[i]) (void)hipEventDestroy(g->windowEvent[i]);
    for (int i = 0; i < 4; i++)
      if (g->pinned[i]) (void)hipHostFree(g->pinned[i]);
    if (g->predict.verdictHost) (void)hipHostFree(g->predict.verdictHost);
  }
  delete g;
}

/* device ordinals the AoS entry points shard over: $AWFM_GPU_DEVICES = "all" or a comma list (a device named
 * again gets a lane on its image); unset = the default device (-1) with three lanes, so that one chunk of a
 * list is packed / scattered on the host while others are on the PCIe bus or in the kernels (awfm_batch.c) */
static int aosDevices(int *devs, int maxOut) {
  int n = 0;
  const char *env = awfmKnob(AWFM_KNOB_DEVICES);
  if (env && !strcmp(env, "all")) {
    const int count = awfmGpuDeviceCount();
    for (int d = 0; d < count && n < maxOut; d++) devs[n++] = d;
  } else if (env && *env) {
    for (const char *c = env; *c && n < maxOut;) {
      devs[n++] = atoi(c);
      while (*c && *c != ',') c++;
      if (*c == ',') c++;
    }
  }
  if (n == 0) { /* three lanes on the default device: one packs or scatters while two are in their device stage */
    devs[n++] = -1;
    for (int lane = 1; lane < 3 && n < maxOut; lane++) devs[n++] = -1;
  }
  return n;
}

}  // extern "C"
/* the lanes of a primary image (call with tableMutex NOT held) */
std::vector<AwFmGpuIndex *> awfmGpuLanesOf(const AwFmGpuIndex *primary) {
  std::vector<AwFmGpuIndex *> lanes;
  std::lock_guard<std::mutex> lock(tableMutex);
  for (auto &e : imageTable)
    if (e.image->shares == primary) lanes.push_back(e.image);
  return lanes;
}

extern "C" {
static AwFmGpuIndex *makeLane(AwFmGpuIndex *primary) {
  AwFmGpuIndex *g = new AwFmGpuIndex();
  g->shares = primary;
  g->device = primary->device;
  g->amino = primary->amino;
  g->dev = primary->dev;
  g->dBlocks = primary->dBlocks;
  g->dSuper = primary->dSuper;
  g->dSeed = primary->dSeed;
  g->dSa = primary->dSa;
  g->dPrefix = primary->dPrefix;
  g->dDeepSeed = primary->dDeepSeed;
  g->dDenseSa = primary->dDenseSa;
  g->denseWide = primary->denseWide;
  g->numBlocks = primary->numBlocks;
  g->kernel = primary->kernel;
  g->forceWide = primary->forceWide;
  g->numCUs = primary->numCUs;
  return g;
}

int awfmGpuIndexAcquireAll(const struct AwFmIndex *index, AwFmGpuIndex **out, int maxOut) {
  int devs[64];
  const int numDevs = aosDevices(devs, 64);
  /* -1 = the default device: $AWFM_GPU_DEVICE, else the calling thread's current device.  Entries are keyed by
   * the resolved ordinal, so a list that changes between calls never hands out another device's image. */
  int fallback = 0;
  if (const char *env = awfmKnob(AWFM_KNOB_DEVICE); env && *env) fallback = atoi(env);
  else if (hipGetDevice(&fallback) != hipSuccess) fallback = 0;
  for (int i = 0; i < numDevs; i++)
    if (devs[i] < 0) devs[i] = fallback;
  std::lock_guard<std::mutex> lock(tableMutex);
  auto find = [&](int device, int lane) -> AwFmGpuIndex * {
    for (auto &e : imageTable)
      if (e.index == index && e.device == device && e.lane == lane) return e.image;
    return nullptr;
  };
  int n = 0;
  for (int slot = 0; slot < numDevs && n < maxOut; slot++) {
    int lane = 0; /* how often this device was named before */
    for (int earlier = 0; earlier < slot; earlier++) lane += devs[earlier] == devs[slot];
    AwFmGpuIndex *g = find(devs[slot], lane);
    if (!g) {
      if (lane > 0) { /* a device named again gets a lane on the image it already has */
        AwFmGpuIndex *primary = find(devs[slot], 0);
        if (!primary) return n;
        g = makeLane(primary);
      } else if (createImage(index, devs[slot], &g, true) != AwFmSuccess) {
        return n;
      }
      imageTable.push_back({index, devs[slot], lane, g});
    }
    out[n++] = g;
  }
  /* what the images' builder threads have finished since the last call is installed now, if nobody is inside a search */
  for (int i = 0; i < n; i++) {
    AwFmGpuIndex *primary = out[i]->shares ? out[i]->shares : out[i];
    if (primary->accelState.load() == 2) {
      std::vector<AwFmGpuIndex *> lanes;
      for (auto &e : imageTable)
        if (e.image->shares == primary) lanes.push_back(e.image);
      awfmGpuAdoptAccelerators(primary, false, &lanes);
    }
  }
  return n;
}

/* the explicit way to an index's image: complete -- whatever is built behind the first searches is waited for and installed */
AwFmGpuIndex *awfmGpuIndexAcquire(const struct AwFmIndex *index) {
  AwFmGpuIndex *g = nullptr;
  if (awfmGpuIndexAcquireAll(index, &g, 1) != 1) return nullptr;
  awfmGpuAdoptAccelerators(g->shares ? g->shares : g, true);
  return g;
}

void awfmGpuIndexRelease(const struct AwFmIndex *index) {
  std::vector<AwFmGpuIndex *> doomed;
  {
    std::lock_guard<std::mutex> lock(tableMutex);
    for (size_t i = 0; i < imageTable.size();) {
      if (imageTable[i].index == index) {
        doomed.push_back(imageTable[i].image);
        imageTable.erase(imageTable.begin() + (long)i);
      } else {
        i++;
      }
    }
  }
  for (AwFmGpuIndex *g : doomed)
    if (g->shares) awfmGpuIndexDestroy(g); /* lanes first: they point into their primary */
  for (AwFmGpuIndex *g : doomed)
    if (!g->shares) awfmGpuIndexDestroy(g);
}

void *awfmGpuPinnedBuffer(AwFmGpuIndex *g, int slot, uint64_t bytes) {
  if (!g || slot < 0 || slot > 3) return nullptr;
  if (bytes <= g->pinnedBytes[slot]) return g->pinned[slot];
  DeviceGuard guard(g->device);
  if (g->pinned[slot]) (void)hipHostFree(g->pinned[slot]);
  g->pinned[slot] = nullptr;
  g->pinnedBytes[slot] = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&g->pinned[slot], want, hipHostMallocDefault) != hipSuccess) {
    setError("awfmGpuPinnedBuffer: hipHostMalloc failed");
    g->pinned[slot] = nullptr;
    return nullptr;
  }
  g->pinnedBytes[slot] = want;
  return g->pinned[slot];
}
void awfmGpuAosLock(AwFmGpuIndex *g) {
  if (g) g->aosMutex.lock();
}
void awfmGpuAosUnlock(AwFmGpuIndex *g) {
  if (g) g->aosMutex.unlock();
}

uint64_t awfmGpuIndexDeviceBytes(const AwFmGpuIndex *g) {
  return g ? g->deviceBytes + g->deepSeedBytes + g->denseSaBytes + g->pairBytes + g->lengthTableBytes : 0;
}


/* replaces the deeper table of a primary image and of the given lanes; the caller holds whatever locks the image
 * needs (none for an image nobody else has a pointer to yet) */
static enum AwFmReturnCode applyDeepSeed(AwFmGpuIndex *g, unsigned deepK, const std::vector<AwFmGpuIndex *> &laneList);

enum AwFmReturnCode awfmGpuIndexSetDeepSeed(AwFmGpuIndex *g, unsigned deepK) {
  if (!g) {
    setError("awfmGpuIndexSetDeepSeed: null image");
    return AwFmNullPtrError;
  }
  if (g->shares) {
    setError("awfmGpuIndexSetDeepSeed: set it on the primary image, not on a lane");
    return AwFmIllegalPositionError;
  }
  awfmGpuAdoptAccelerators(g, true);
  DeviceGuard guard(g->device);
  AwFmGpuLaneLocks lanes(g); /* nobody searches through a lane while the table is replaced */
  std::lock_guard<std::mutex> lock(g->workMutex);
  return applyDeepSeed(g, deepK, lanes.lanes);
}

/* the deeper table of depth deepK with its next-step bits, built from the image as it is (nothing of the image is written):
 * into `to` */
static enum AwFmReturnCode buildDeepSeed(AwFmGpuIndex *g, unsigned deepK, AwFmGpuIndex::PendingAccel *to) {
  void *table = nullptr, *big = nullptr;
  uint64_t bytes = 0, peak = 0;
  unsigned format = 0, numBig = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (!awfmGpuBuildDeepSeedTable(g, deepK, &table, &bytes, &peak, &to->deepAllocSeconds, &format, &big)) return AwFmGeneralFailure;
  /* the next-step bits: images with pair blocks (format 1: the long lengths move to `big` with them) */
  const int next = awfmGpuDeepSeedAddNext(g, table, deepK, format, &big, &numBig);
  (void)awfmGpuSetupSync();
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (next < 0) {
    (void)hipFree(table);
    if (big) (void)hipFree(big);
    return AwFmGeneralFailure;
  }
  to->deepTable = table;
  to->deepBig = big;
  to->deepBytes = bytes;
  to->deepBigBytes = !big ? 0u
                     : format == 2u ? ((g->dev.bwtLength >> (g->amino ? kAminoWideBigShift : kDeepWideBigShift)) + 2u) * 8u
                                    : ((g->dev.bwtLength >> (g->amino ? kAminoDeepBigShift : kDeepBigShift)) + 5u) * 4u;
  to->deepTransient = peak > bytes ? peak - bytes : 0;
  to->deepK = deepK;
  to->deepFormat = format;
  to->deepNext = next > 0 ? 1u : 0u;
  to->numDeepBig = numBig;
  to->deepSeconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (awfmKnob(AWFM_KNOB_VERBOSE))
    fprintf(stderr, "[awfm deeper table] depth %u, entry format %u: %.2f GB in %.2f s; next-step bits %s; %u entries with long ranges\n", deepK, format,
            (double)bytes * 1e-9, to->deepSeconds, next > 0 ? "yes" : "no", numBig);
  return AwFmSuccess;
}
/* a built table becomes the image's (and its lanes'); the caller holds whatever locks the image needs */
static void installDeepSeed(AwFmGpuIndex *g, AwFmGpuIndex::PendingAccel *from, const std::vector<AwFmGpuIndex *> &laneList) {
  g->dDeepSeed = from->deepTable;
  g->dDeepBig = from->deepBig;
  g->deepSeedBytes = from->deepBytes + from->deepBigBytes;
  g->deepSeedBuildSeconds = from->deepSeconds;
  g->deepSeedAllocSeconds = from->deepAllocSeconds;
  g->deepSeedTransientBytes = from->deepTransient;
  g->dev.deepSeed = (const ulonglong2 *)from->deepTable;
  g->dev.deepNarrow = from->deepFormat;
  g->dev.deepNext = from->deepNext;
  g->dev.numDeepBig = from->numDeepBig;
  g->dev.deepBigBySp = (const unsigned *)from->deepBig;
  g->dev.deepK = from->deepTable ? from->deepK : 0u;
  from->deepTable = from->deepBig = nullptr;
  for (AwFmGpuIndex *lane : laneList) {
    lane->dDeepSeed = g->dDeepSeed;
    lane->dev.deepSeed = g->dev.deepSeed;
    lane->dev.deepNarrow = g->dev.deepNarrow;
    lane->dev.deepK = g->dev.deepK;
    lane->dev.deepNext = g->dev.deepNext;
    lane->dev.numDeepBig = g->dev.numDeepBig;
    lane->dev.deepBigBySp = g->dev.deepBigBySp;
  }
}

static enum AwFmReturnCode applyDeepSeed(AwFmGpuIndex *g, unsigned deepK, const std::vector<AwFmGpuIndex *> &laneList) {
  (void)hipDeviceSynchronize();
  if (g->dDeepSeed) (void)hipFree(g->dDeepSeed);
  if (g->dDeepBig) (void)hipFree(g->dDeepBig);
  { /* the tables of the shorter lengths go with the deeper table they complete; the next mixed-length batch builds them again */
    std::lock_guard<std::mutex> lock(g->lengthMutex);
    if (g->dLengthTable) (void)hipFree(g->dLengthTable);
    if (g->dLengthBig) (void)hipFree(g->dLengthBig);
    g->dLengthTable = nullptr;
    g->dLengthBig = nullptr;
    g->lengthDepths = 0;
    g->lengthTableBytes = 0;
    g->lengthTried = false;
  }
  AwFmGpuIndex::PendingAccel built; /* (nothing: the image without a deeper table) */
  enum AwFmReturnCode rc = AwFmSuccess;
  if (deepK != 0) {
    /* (the construction reads the image's view: without the table that is being replaced) */
    g->dev.deepSeed = nullptr;
    g->dev.deepK = 0;
    rc = buildDeepSeed(g, deepK, &built);
  }
  installDeepSeed(g, &built, laneList);
  return rc;
}

/* $AWFM_GPU_DEEP_SEED_K on an image that was just created or adopted: nobody else holds it and it has no lanes,
 * so no lock is taken -- awfmGpuIndexAcquireAll creates images while it holds the table lock, and the public
 * setter would ask for that lock again through lanesOf() */
static unsigned chooseDeepSeedK(const AwFmGpuIndex *g, std::string &notes);
static enum AwFmReturnCode applyDeepSeedFromEnv(AwFmGpuIndex *g) {
  const unsigned deepK = chooseDeepSeedK(g, g->accelNotes);
  if (deepK == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  return applyDeepSeed(g, deepK, {});
}
/* the depth of the device-only table an image gets by itself ($AWFM_GPU_DEEP_SEED_K / $AWFM_GPU_AMINO_DEEP_SEED_K, else by its size
 * and the memory that is free); 0: none */
static unsigned chooseDeepSeedK(const AwFmGpuIndex *g, std::string &notes) {
  int deepK = 0;
  if (const char *env = awfmKnob(g->amino ? AWFM_KNOB_AMINO_DEEP_SEED_K : AWFM_KNOB_DEEP_SEED_K)) {
    deepK = atoi(env); /* 0: none */
  } else if (g->amino) {
    /* Automatic, amino: an image of >= 2^26 positions whose own table is shallower gets the deepest table of up to 7
     * characters with at most 8 entries per text position, when three times its size is free on the device: 20^7 x 8 B =
     * 10.2 GB for a Swiss-Prot-sized text (2 * 10^8 residues), where 85 % of random 10-mers end at their entry (no such
     * 7-mer) and the rest start two steps further on.  Exact: an entry is what the stepping holds after those steps. */
    size_t freeBytes = 0, totalBytes = 0;
    DeviceGuard guard(g->device);
    if (g->dev.bwtLength >= (1ull << 26) && g->dev.bwtLength < (1ull << kDeepWideMaxBits) && g->dev.seedK >= 2 && hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess) {
      unsigned long long entries = 1;
      for (unsigned k = 1; k <= 7u; k++) {
        entries *= 20ull;
        if (k > g->dev.seedK && entries <= 8ull * g->dev.bwtLength && freeBytes / 3u >= entries * 8ull) deepK = (int)k;
        else if (k > g->dev.seedK && entries <= 8ull * g->dev.bwtLength && k > (unsigned)deepK) notes += "deeper table: depth " + std::to_string(k) + " not built (less than 3 x its size free); ";
      }
    } else {
      (void)hipGetLastError();
    }
  } else if (g->dev.bwtLength >= (1ull << 28) && g->dev.seedK >= 8 && g->dev.seedK < kAutoDeepSeedMin) {
    /* Automatic: an image far beyond the L2s gets the deepest table of 14..16 characters that has no more than two
     * entries per text position, when the device has room to spare (8 B -- 16 B from 2^32 positions -- x 4^K: 2.1 GB
     * at 14, 34 GB at 16; its construction holds the level below beside it; asked for: three times the table).  Every
     * level of the table replaces a dependent block read of EVERY k-mer by a wider spread of the one table read: 10^8
     * random 21-mers against a 3.1 Gbp image, seed-order search kernel 3.44 ms at 14, 3.21 at 15, 2.87 at 16 (the
     * index's own k = 12 table: 4.6); planted 21-mers 6.08 -> 5.13 ms.  Results are bit-identical (the table holds
     * what the stepping would compute, stop-at-first-invalid rule included). */
    size_t freeBytes = 0, totalBytes = 0;
    DeviceGuard guard(g->device);
    if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess) {
      const uint64_t entryBytes = g->dev.bwtLength < (1ull << kDeepWideMaxBits) ? 8u : 16u;
      for (unsigned k = kAutoDeepSeedMax; k >= kAutoDeepSeedMin && deepK == 0; k--)
        if ((1ull << (2u * k)) <= 2ull * g->dev.bwtLength && freeBytes / 3u >= (entryBytes << (2u * k))) deepK = (int)k;
      if (deepK == 0 && freeBytes / 4u >= (16ull << (2u * kAutoDeepSeedMin))) deepK = (int)kAutoDeepSeedMin;
      unsigned wanted = 0; /* the depth the image's size asks for */
      for (unsigned k = kAutoDeepSeedMax; k >= kAutoDeepSeedMin && wanted == 0; k--)
        if ((1ull << (2u * k)) <= 2ull * g->dev.bwtLength) wanted = k;
      if ((unsigned)deepK < wanted)
        notes += "deeper table: depth " + std::to_string(wanted) + " not built (less than 3 x its size free)" +
                         (deepK ? ", depth " + std::to_string(deepK) + " instead; " : "; ");
    } else {
      (void)hipGetLastError();
    }
  }
  return deepK <= 0 || (unsigned)deepK <= g->dev.seedK ? 0u : (unsigned)deepK; /* (nothing deeper than the index's own table) */
}
/* Pair image (awfm_pair.h) of a nucleotide image that was just created or adopted (nobody else holds it, no lanes, so
 * no lock): built unless $AWFM_GPU_PAIR=0.  It doubles the block bytes of the image (128 B per 128 positions beside
 * the 64 B of the one-letter blocks) and halves the dependent block reads of hits-only searches and of the LF walk. */
static enum AwFmReturnCode applyPairFromEnv(AwFmGpuIndex *g) {
  if (g->amino) return AwFmSuccess;
  if (const char *env = awfmKnob(AWFM_KNOB_PAIR))
    if (atoi(env) == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  const enum AwFmReturnCode rc = awfmGpuApplyPairImage(g, true);
  if (rc != AwFmSuccess) {
    (void)awfmGpuApplyPairImage(g, false);
    g->accelNotes += "pair image: not built (no device memory for 1 byte per position); ";
  }
  return rc;
}

enum AwFmReturnCode awfmGpuIndexSetPairImage(AwFmGpuIndex *g, int enable) {
  if (!g) {
    setError("awfmGpuIndexSetPairImage: null image");
    return AwFmNullPtrError;
  }
  if (g->shares) {
    setError("awfmGpuIndexSetPairImage: set it on the primary image, not on a lane");
    return AwFmIllegalPositionError;
  }
  awfmGpuAdoptAccelerators(g, true);
  DeviceGuard guard(g->device);
  AwFmGpuLaneLocks lanes(g); /* nobody searches through a lane while the image changes */
  std::lock_guard<std::mutex> lock(g->workMutex);
  const enum AwFmReturnCode rc = awfmGpuApplyPairImage(g, enable != 0);
  if (rc != AwFmSuccess) (void)awfmGpuApplyPairImage(g, false);
  for (AwFmGpuIndex *lane : lanes.lanes) {
    lane->dev.pairBlocks = g->dev.pairBlocks;
    lane->dev.pairSuper = g->dev.pairSuper;
    lane->dev.pairSuper32 = g->dev.pairSuper32;
    lane->dev.pairC = g->dev.pairC;
    lane->dev.numPairSuper = g->dev.numPairSuper;
  }
  return rc;
}
int awfmGpuIndexHasPairImage(const AwFmGpuIndex *g) { return g && g->dev.pairBlocks ? 1 : 0; }
unsigned awfmGpuIndexDeepSeedK(const AwFmGpuIndex *g) { return g ? g->dev.deepK : 0u; }
double awfmGpuIndexDeepSeedAllocSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->deepSeedAllocSeconds : 0.0; }
double awfmGpuIndexDeepSeedBuildSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->deepSeedBuildSeconds : 0.0; }
uint64_t awfmGpuIndexDeepSeedTransientBytes(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->deepSeedTransientBytes : 0; }

int awfmGpuIndexDevice(const AwFmGpuIndex *g) { return g ? g->device : -1; }
void awfmGpuIndexSetKernel(AwFmGpuIndex *g, enum AwFmGpuKernel kernel) {
  if (g) g->kernel = kernel;
}
int awfmGpuIndexIsWide(const AwFmGpuIndex *g) { return g && !awfmImageNarrow(g) ? 1 : 0; }
void awfmGpuIndexSetWide(AwFmGpuIndex *g, int wide) {
  if (!g) return;
  g->forceWide = wide != 0;
  if (!g->shares)
    for (AwFmGpuIndex *lane : awfmGpuLanesOf(g)) lane->forceWide = g->forceWide;
}


/* see include/awfm_gpu.h */
int awfmGpuIndexDescribe(const AwFmGpuIndex *g, char *out, int outBytes) {
  if (!g || !out || outBytes <= 0) return 0;
  const AwFmGpuIndex *p = g->shares ? g->shares : g;
  std::string text = std::string(p->amino ? "amino" : "nucleotide") + " image of " + std::to_string(p->dev.bwtLength) + " positions, " +
                     std::to_string(awfmGpuIndexDeviceBytes(p)) + " bytes on device " + std::to_string(p->device) + ": ";
  if (!p->amino) text += p->dev.pairBlocks ? "pair image yes; " : "pair image no; ";
  text += p->dev.deepK ? "deeper table depth " + std::to_string(p->dev.deepK) + (p->dev.deepNext ? " with next-step bits; " : "; ") : "deeper table no; ";
  text += p->dDenseSa ? "full suffix array yes; " : "full suffix array no; ";
  if (!p->amino) text += p->dLengthTable ? "tables per k-mer length 1.." + std::to_string(p->lengthDepths) + "; " : "tables per k-mer length not built (the first large mixed-length batch builds them); ";
  if (!p->accelNotes.empty()) text += "notes: " + p->accelNotes;
  while (!text.empty() && (text.back() == ' ' || text.back() == ';')) text.pop_back();
  const int n = (int)text.size() < outBytes - 1 ? (int)text.size() : outBytes - 1;
  memcpy(out, text.data(), (size_t)n);
  out[n] = 0;
  return (int)text.size();
}
/* the tables per k-mer length a mixed-length batch builds on first use (awfm_gpu_ordered.hip: ensureLengthTables) */
uint64_t awfmGpuIndexLengthTableBytes(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->lengthTableBytes : 0; }
double awfmGpuIndexLengthTableBuildSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->lengthTableBuildSeconds : 0.0; }

}  // extern "C"
