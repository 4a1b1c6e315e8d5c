/*
 * Batch search entry points: AwFmKmerSearchList lifetime and
 * awFmParallelSearchCount / awFmParallelSearchLocate.
 *
 * The reference runs an OpenMP loop over 8-query blocks on the CPU
 * (ref src/AwFmParallelSearch.c:95-220).  Here the host only packs the AoS
 * k-mers into one flat buffer, hands it to the HIP side (include/awfm_gpu.h)
 * and scatters the flat results back into the AoS.  There is no CPU search
 * path: if the device side fails, the failure is reported, never papered over.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "awfm_internal.h"

#define AWFM_DEFAULT_POSITION_LIST_CAPACITY 4 /* ref src/AwFmParallelSearch.c:13 */

/* ref src/AwFmParallelSearch.c:36-84 */
struct AwFmKmerSearchList *awFmCreateKmerSearchList(const size_t capacity) {
  struct AwFmKmerSearchList *list = malloc(sizeof *list);
  if (!list) return NULL;
  list->capacity = capacity;
  list->count = 0;
  list->kmerSearchData = malloc(capacity * sizeof(struct AwFmKmerSearchData));
  if (!list->kmerSearchData && capacity) {
    free(list);
    return NULL;
  }
  size_t made = 0;
  for (; made < capacity; made++) {
    struct AwFmKmerSearchData *d = &list->kmerSearchData[made];
    d->kmerString = NULL;
    d->kmerLength = 0;
    d->count = 0;
    d->capacity = AWFM_DEFAULT_POSITION_LIST_CAPACITY;
    d->positionList = malloc(AWFM_DEFAULT_POSITION_LIST_CAPACITY * sizeof(uint64_t));
    if (!d->positionList) break;
  }
  if (made < capacity) {
    for (size_t i = 0; i < made; i++) free(list->kmerSearchData[i].positionList);
    free(list->kmerSearchData);
    free(list);
    return NULL;
  }
  return list;
}

/* ref src/AwFmParallelSearch.c:86-93: k-mer strings are never freed */
void awFmDeallocKmerSearchList(struct AwFmKmerSearchList *_RESTRICT_ const searchList) {
  if (!searchList) return;
  for (size_t i = 0; i < searchList->capacity; i++) free(searchList->kmerSearchData[i].positionList);
  free(searchList->kmerSearchData);
  free(searchList);
}

#include <pthread.h>

#define AWFM_MAX_IMAGES 16
#define AWFM_MIN_SHARDED_LIST 65536u

/* ---- pack: AoS k-mers -> flat chars + CSR offsets ---- */
#define AWFM_MAX_PACK_THREADS 64 /* awfmParallelFor never uses more */

struct packCtx {
  const struct AwFmKmerSearchData *data; /* first query of the shard */
  uint64_t firstLength;
  /* pass 1 (per chunk of awfmParallelFor, indexed by tid): characters in the chunk, all lengths == firstLength */
  uint64_t chunkChars[AWFM_MAX_PACK_THREADS];
  bool chunkUniform[AWFM_MAX_PACK_THREADS];
  bool chunkUsed[AWFM_MAX_PACK_THREADS];
  /* pass 2 */
  uint64_t chunkStart[AWFM_MAX_PACK_THREADS]; /* character offset of the chunk's first query */
  uint64_t *offsets;                          /* NULL when every k-mer has firstLength characters */
  uint8_t *chars;
};

static void packMeasure(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  struct packCtx *c = p;
  uint64_t total = 0;
  bool uniform = true;
  for (uint64_t i = begin; i < end; i++) {
    const uint64_t len = c->data[i].kmerLength;
    total += len;
    uniform &= len == c->firstLength;
  }
  c->chunkChars[tid] = total;
  c->chunkUniform[tid] = uniform;
  c->chunkUsed[tid] = true;
}

static void packChars(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  struct packCtx *c = p;
  if (!c->offsets) {
    const uint64_t len = c->firstLength;
    for (uint64_t i = begin; i < end; i++) memcpy(c->chars + i * len, c->data[i].kmerString, len);
    return;
  }
  uint64_t at = c->chunkStart[tid];
  for (uint64_t i = begin; i < end; i++) {
    const uint64_t len = c->data[i].kmerLength;
    c->offsets[i] = at;
    memcpy(c->chars + at, c->data[i].kmerString, len);
    at += len;
  }
}

/* Packs n k-mers starting at data[0] into page-locked staging buffers of the image, with `threads` host threads
 * (lengths are summed per chunk first, so that every chunk knows where its characters go).  When every k-mer
 * has the same length the offsets array is dropped (*offsetsOut = NULL, *fixedOut = that length). */
static bool packQueries(AwFmGpuIndex *g, const struct AwFmKmerSearchData *data, uint64_t n, unsigned threads,
                        uint8_t **charsOut, uint64_t **offsetsOut, uint32_t *fixedOut) {
  struct packCtx ctx;
  memset(&ctx, 0, sizeof ctx);
  ctx.data = data;
  ctx.firstLength = data[0].kmerLength;
  awfmParallelFor(threads, n, packMeasure, &ctx);
  uint64_t total = 0;
  bool uniform = ctx.firstLength != 0 && ctx.firstLength <= 0xFFFFFFFFull;
  for (unsigned t = 0; t < AWFM_MAX_PACK_THREADS; t++) { /* chunks are in query order; a run may use fewer */
    ctx.chunkStart[t] = total;
    if (!ctx.chunkUsed[t]) continue;
    total += ctx.chunkChars[t];
    uniform &= ctx.chunkUniform[t];
  }
  if (!uniform) {
    ctx.offsets = awfmGpuPinnedBuffer(g, 1, (n + 1) * sizeof(uint64_t));
    if (!ctx.offsets) return false;
    ctx.offsets[n] = total;
  }
  ctx.chars = awfmGpuPinnedBuffer(g, 0, total ? total : 1);
  if (!ctx.chars) return false;
  awfmParallelFor(threads, n, packChars, &ctx);
  *charsOut = ctx.chars;
  *offsetsOut = ctx.offsets;
  *fixedOut = uniform ? (uint32_t)ctx.firstLength : 0;
  return true;
}

/* ---- scatter ---- */
struct countCtx {
  struct AwFmKmerSearchData *data;
  const uint32_t *counts;
};
static void scatterCounts(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct countCtx *c = p;
  for (uint64_t i = begin; i < end; i++) c->data[i].count = c->counts[i];
}

struct locateCtx {
  struct AwFmKmerSearchData *data;
  const uint64_t *hitOffsets;
  uint64_t n;
  unsigned threads;
  bool sized;
  int failed;
  /* the window being scattered */
  uint64_t queryBegin, hitBegin, hitEnd;
  const uint64_t *positions;
};

/* ref src/AwFmParallelSearch.c:327-328, :367-387 (setPositionListCount): count is set, the list grows by realloc to
 * exactly `count` only when capacity is too small */
static void sizeLists(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct locateCtx *c = p;
  for (uint64_t i = begin; i < end; i++) {
    struct AwFmKmerSearchData *d = &c->data[i];
    const uint32_t newCount = (uint32_t)(c->hitOffsets[i + 1] - c->hitOffsets[i]);
    if (d->capacity < newCount) {
      void *grown = realloc(d->positionList, (size_t)newCount * sizeof(uint64_t));
      if (!grown) {
        fprintf(stderr, "Critical memory failure: could not allocate memory for position list.\n");
        __atomic_store_n(&c->failed, 1, __ATOMIC_RELAXED);
        d->count = 0;
        continue;
      }
      d->positionList = grown;
      d->capacity = newCount;
    }
    d->count = newCount;
  }
}

/* ref src/AwFmParallelSearch.c:361: the part of every list that lies in the window [hitBegin, hitEnd) of the flat hit list
 * (a list whose hits exceed the device's hit budget arrives in several windows) */
static void scatterWindow(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct locateCtx *c = p;
  for (uint64_t j = begin; j < end; j++) {
    const uint64_t i = c->queryBegin + j;
    struct AwFmKmerSearchData *d = &c->data[i];
    const uint64_t from = c->hitOffsets[i], to = from + d->count; /* count is 0 for a list that could not be allocated */
    const uint64_t lo = from > c->hitBegin ? from : c->hitBegin, hi = to < c->hitEnd ? to : c->hitEnd;
    if (lo < hi) memcpy(d->positionList + (lo - from), c->positions + (lo - c->hitBegin), (size_t)(hi - lo) * sizeof(uint64_t));
  }
}

static int locateWindowSink(void *user, uint64_t queryBegin, uint64_t queryEnd, uint64_t hitBegin, uint64_t hitEnd,
                            const uint64_t *positions) {
  struct locateCtx *c = user;
  if (!c->sized) { /* the hit offsets are complete: every list gets its final size before the first position lands */
    awfmParallelFor(c->threads, c->n, sizeLists, c);
    c->sized = true;
  }
  c->queryBegin = queryBegin;
  c->hitBegin = hitBegin;
  c->hitEnd = hitEnd;
  c->positions = positions;
  awfmParallelFor(c->threads, queryEnd - queryBegin, scatterWindow, c);
  return 0;
}

/* A shard that failed reports no hits: there is no CPU search path to fall back to, awFmParallelSearchCount
 * returns void (ref src/AwFmIndex.h:400-403), and a caller must never read the counts of an earlier search as
 * the answer of this one.  The failure itself goes to stderr and to awfmGpuLastBatchStatus(). */
static void invalidateCounts(struct AwFmKmerSearchData *data, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) data[i].count = 0;
}

/* ---- one contiguous shard of the list on one device image ---- */
struct shardJob {
  AwFmGpuIndex *image;
  struct AwFmKmerSearchData *data; /* first query of the shard */
  uint64_t n;
  unsigned threads;
  bool locate;
  enum AwFmReturnCode rc;
  char error[256]; /* awfmGpuLastError() of the thread that ran the shard (the message is thread-local) */
};

static void *runShard(void *p) {
  struct shardJob *job = p;
  AwFmGpuIndex *g = job->image;
  job->rc = AwFmSuccess;
  if (job->n == 0) return NULL;
  awfmGpuAosLock(g);
  uint8_t *chars = NULL;
  uint64_t *offsets = NULL;
  uint32_t fixedLength = 0;
  void *out = awfmGpuPinnedBuffer(g, 2, (job->n + 1) * sizeof(uint64_t)); /* counts (u32) or hit offsets (u64) */
  if (!out || !packQueries(g, job->data, job->n, job->threads, &chars, &offsets, &fixedLength)) {
    job->rc = AwFmAllocationFailure;
  } else if (!job->locate) {
    job->rc = awfmGpuCountHost(g, chars, offsets, fixedLength, job->n, NULL, out);
    if (job->rc == AwFmSuccess) {
      struct countCtx ctx = {job->data, out};
      awfmParallelFor(job->threads, job->n, scatterCounts, &ctx);
    }
  } else {
    /* the hit list arrives in windows bounded by the device's hit budget (one window for all but hit-heavy batches),
     * in page-locked staging of the image that is valid while the sink runs */
    struct locateCtx ctx;
    memset(&ctx, 0, sizeof ctx);
    ctx.data = job->data;
    ctx.hitOffsets = out;
    ctx.n = job->n;
    ctx.threads = job->threads;
    job->rc = awfmGpuLocateHostWindows(g, chars, offsets, fixedLength, job->n, NULL, out, locateWindowSink, &ctx);
    if (job->rc == AwFmSuccess && !ctx.sized) awfmParallelFor(job->threads, job->n, sizeLists, &ctx); /* no hit at all: counts = 0 */
    if (job->rc == AwFmSuccess && ctx.failed) job->rc = AwFmAllocationFailure;
  }
  if (job->rc != AwFmSuccess) {
    snprintf(job->error, sizeof job->error, "%s", awfmGpuLastError());
    invalidateCounts(job->data, job->n);
  }
  awfmGpuAosUnlock(g);
  return NULL;
}

/* Shards the list contiguously over the images of $AWFM_GPU_DEVICES (one host thread per device; the
 * shards are independent, there is no exchange: ref src/AwFmParallelSearch.c:103-129 treats 8-query
 * blocks the same way).  Returns the first failure. */
static enum AwFmReturnCode runBatch(const struct AwFmIndex *index, struct AwFmKmerSearchList *list, uint32_t numThreads,
                                    bool locate, const char *who) {
  const uint64_t n = (uint32_t)list->count; /* the reference reads the count as uint32_t (:100, :164) */
  if (n == 0) return AwFmSuccess;
  AwFmGpuIndex *images[AWFM_MAX_IMAGES];
  /* without an explicit device list a small batch is one shard: splitting it over the two default lanes
   * would only add launches */
  const char *deviceList = getenv("AWFM_GPU_DEVICES");
  const bool oneShard = !(deviceList && *deviceList) && n < AWFM_MIN_SHARDED_LIST;
  const int numImages = awfmGpuIndexAcquireAll(index, images, oneShard ? 1 : AWFM_MAX_IMAGES);
  if (numImages <= 0) {
    fprintf(stderr, "%s: no device image: %s\n", who, awfmGpuLastError());
    invalidateCounts(list->kmerSearchData, n);
    return AwFmGeneralFailure;
  }
  struct shardJob jobs[AWFM_MAX_IMAGES];
  pthread_t threads[AWFM_MAX_IMAGES];
  bool spawned[AWFM_MAX_IMAGES] = {false};
  const uint64_t per = (n + (uint64_t)numImages - 1) / (uint64_t)numImages;
  const unsigned threadsPerShard = numThreads / (unsigned)numImages > 0 ? numThreads / (unsigned)numImages : 1;
  for (int i = 0; i < numImages; i++) {
    const uint64_t begin = per * (uint64_t)i < n ? per * (uint64_t)i : n;
    const uint64_t end = begin + per < n ? begin + per : n;
    jobs[i] = (struct shardJob){images[i], list->kmerSearchData + begin, end - begin, threadsPerShard, locate, AwFmSuccess, {0}};
  }
  for (int i = 1; i < numImages; i++) spawned[i] = pthread_create(&threads[i], NULL, runShard, &jobs[i]) == 0;
  runShard(&jobs[0]);
  int firstFailed = jobs[0].rc != AwFmSuccess ? 0 : -1;
  for (int i = 1; i < numImages; i++) {
    if (spawned[i])
      pthread_join(threads[i], NULL);
    else
      runShard(&jobs[i]);
    if (firstFailed < 0 && jobs[i].rc != AwFmSuccess) firstFailed = i;
  }
  if (firstFailed < 0) return AwFmSuccess;
  fprintf(stderr, "%s: GPU search failed (%d) on image %d: %s\n", who, (int)jobs[firstFailed].rc, firstFailed,
          jobs[firstFailed].error);
  return jobs[firstFailed].rc;
}

static _Thread_local enum AwFmReturnCode lastBatchStatus = AwFmSuccess;
/* return code of the calling thread's last awFmParallelSearchCount / awFmParallelSearchLocate */
enum AwFmReturnCode awfmGpuLastBatchStatus(void) { return lastBatchStatus; }

/* ref src/AwFmParallelSearch.c:159-220 */
void awFmParallelSearchCount(const struct AwFmIndex *_RESTRICT_ const index,
                             struct AwFmKmerSearchList *_RESTRICT_ const searchList, uint32_t numThreads) {
  lastBatchStatus = runBatch(index, searchList, numThreads, false, "awFmParallelSearchCount");
}

/* ref src/AwFmParallelSearch.c:95-157 */
enum AwFmReturnCode awFmParallelSearchLocate(const struct AwFmIndex *_RESTRICT_ const index,
                                             struct AwFmKmerSearchList *_RESTRICT_ const searchList,
                                             uint32_t numThreads) {
  return lastBatchStatus = runBatch(index, searchList, numThreads, true, "awFmParallelSearchLocate");
}
