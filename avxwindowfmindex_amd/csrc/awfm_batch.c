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
#include <time.h>

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

/* 8..32 characters (every seed-and-extend k-mer length): two overlapping fixed-size copies instead of a call with a
 * run-time length; both stay inside the string */
static inline void copyKmer(uint8_t *dst, const char *src, uint64_t len) {
  if (len >= 16 && len <= 32) {
    memcpy(dst, src, 16);
    memcpy(dst + len - 16, src + len - 16, 16);
  } else if (len >= 8 && len < 16) {
    memcpy(dst, src, 8);
    memcpy(dst + len - 8, src + len - 8, 8);
  } else {
    memcpy(dst, src, len);
  }
}

/* the usual list -- every k-mer of one length -- is packed in ONE pass over the AoS: each range copies as if the list
 * were uniform and says so if it met another length (the two passes below then redo the chunk) */
struct packUniformCtx {
  const struct AwFmKmerSearchData *data;
  uint64_t length;
  uint8_t *chars;
  int mixed;
};
static void packUniform(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct packUniformCtx *c = p;
  const uint64_t len = c->length;
  for (uint64_t i = begin; i < end; i++) {
    if (c->data[i].kmerLength != len) {
      __atomic_store_n(&c->mixed, 1, __ATOMIC_RELAXED);
      return;
    }
    copyKmer(c->chars + i * len, c->data[i].kmerString, len);
  }
}

/* Packs n k-mers starting at data[0] into page-locked staging buffers of the image, with `threads` host threads.
 * When every k-mer has the same length the offsets array is dropped (*offsetsOut = NULL, *fixedOut = that length);
 * otherwise lengths are summed per range first, so that every range knows where its characters go. */
static bool packQueries(AwFmGpuIndex *g, const struct AwFmKmerSearchData *data, uint64_t n, unsigned threads,
                        uint8_t **charsOut, uint64_t **offsetsOut, uint32_t *fixedOut) {
  const uint64_t firstLength = data[0].kmerLength;
  /* (the one-pass try sizes its buffer by the FIRST k-mer: bounded to the lengths seed-and-extend k-mers have, so that a
   * mixed list whose first string is long does not pin n x 4096 bytes it then throws away; if the buffer cannot be had
   * the two passes below, which need the summed lengths only, still run) */
  if (firstLength != 0 && firstLength <= 64) {
    struct packUniformCtx u = {data, firstLength, awfmGpuPinnedBuffer(g, 0, n * firstLength), 0};
    if (u.chars) awfmParallelFor(threads, n, packUniform, &u);
    if (u.chars && !u.mixed) {
      *charsOut = u.chars;
      *offsetsOut = NULL;
      *fixedOut = (uint32_t)firstLength;
      return true;
    }
  }
  struct packCtx ctx;
  memset(&ctx, 0, sizeof ctx);
  ctx.data = data;
  ctx.firstLength = firstLength;
  awfmParallelFor(threads, n, packMeasure, &ctx);
  uint64_t total = 0;
  bool uniform = ctx.firstLength != 0 && ctx.firstLength <= 0xFFFFFFFFull;
  for (unsigned t = 0; t < AWFM_MAX_PACK_THREADS; t++) { /* ranges are in query order; a run may use fewer */
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
  double waitMs, scatterMs; /* $AWFM_GPU_DIAG aos_trace */
};

/* ref src/AwFmParallelSearch.c:327-328, :367-387 (setPositionListCount): count is set, the list grows by realloc to
 * exactly `count` only when capacity is too small */
static inline void sizeList(struct locateCtx *c, uint64_t i) {
  struct AwFmKmerSearchData *d = &c->data[i];
  const uint32_t newCount = (uint32_t)(c->hitOffsets[i + 1] - c->hitOffsets[i]);
  if (d->capacity < newCount) {
    void *grown = realloc(d->positionList, (size_t)newCount * sizeof(uint64_t));
    if (!grown) {
      fprintf(stderr, "Critical memory failure: could not allocate memory for position list.\n");
      __atomic_store_n(&c->failed, 1, __ATOMIC_RELAXED);
      d->count = 0;
      return;
    }
    d->positionList = grown;
    d->capacity = newCount;
  }
  d->count = newCount;
}
static void sizeLists(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  for (uint64_t i = begin; i < end; i++) sizeList(p, i);
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

/* the whole hit list in one window (all but hit-heavy chunks): sizeLists and scatterWindow in one pass over the AoS */
static void sizeAndScatter(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct locateCtx *c = p;
  for (uint64_t i = begin; i < end; i++) {
    sizeList(c, i);
    struct AwFmKmerSearchData *d = &c->data[i];
    if (d->count) memcpy(d->positionList, c->positions + c->hitOffsets[i], (size_t)d->count * sizeof(uint64_t));
  }
}

static int locateWindowSink(void *user, uint64_t queryBegin, uint64_t queryEnd, uint64_t hitBegin, uint64_t hitEnd,
                            const uint64_t *positions) {
  struct locateCtx *c = user;
  if (!c->sized && hitBegin == 0 && hitEnd == c->hitOffsets[c->n]) {
    c->positions = positions;
    awfmParallelFor(c->threads, c->n, sizeAndScatter, c);
    c->sized = true;
    return 0;
  }
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

/* ---- the list in chunks, over the lanes of the device images ---- */
/* The host stages of a chunk (packing its k-mers, scattering its results) and its device stage (upload, kernels,
 * download) alternate, so a list is cut into chunks that the lanes -- one host thread per device image, three images
 * on the default device -- take in turn: while one lane waits for the device, another packs or scatters with ALL the
 * caller's threads.  The host stages take turns (hostStage); a lane never waits for the device while it holds the
 * turn.  $AWFM_GPU_AOS_CHUNK: k-mers per chunk; $AWFM_GPU_DIAG aos_trace=1: one line per chunk on stderr. */
#define AWFM_AOS_CHUNK_DEFAULT (1u << 20) /* 10^7 random 21-mers, locate, 32 threads: 8.5 ms at 2^20 or 2^21, 13.7 ms at 4*10^6 */
static pthread_mutex_t hostStage = PTHREAD_MUTEX_INITIALIZER;

static double nowMs(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}

/* what the last awFmParallelSearchCount / Locate of the process spent where, summed over its chunks (the lanes overlap, so the
 * sums exceed the wall time): reporting (awfmGpuAosLastStages; bench.py's end_to_end) */
static pthread_mutex_t stageLock = PTHREAD_MUTEX_INITIALIZER;
static struct {
  double wallMs, turnMs, packMs, deviceMs, scatterMs;
  uint64_t chunks, kmers, hits, packBytes, scatterBytes;
} lastStages;

struct laneJob {
  AwFmGpuIndex *image;
  struct AwFmKmerSearchData *data; /* the whole list */
  uint64_t n, chunk;
  unsigned lane, numLanes, threads;
  bool locate, trace;
  enum AwFmReturnCode rc;
  char error[256]; /* awfmGpuLastError() of the thread that ran the lane (the message is thread-local) */
};

static int locateWindowSinkTurn(void *user, uint64_t queryBegin, uint64_t queryEnd, uint64_t hitBegin, uint64_t hitEnd,
                                const uint64_t *positions) {
  struct locateCtx *c = user;
  const double t0 = nowMs();
  pthread_mutex_lock(&hostStage);
  const double t1 = nowMs();
  const int r = locateWindowSink(user, queryBegin, queryEnd, hitBegin, hitEnd, positions);
  pthread_mutex_unlock(&hostStage);
  c->waitMs += t1 - t0;
  c->scatterMs += nowMs() - t1;
  return r;
}

static enum AwFmReturnCode runChunk(struct laneJob *job, struct AwFmKmerSearchData *data, uint64_t n, uint64_t chunkNumber) {
  AwFmGpuIndex *g = job->image;
  enum AwFmReturnCode rc = AwFmSuccess;
  uint8_t *chars = NULL;
  uint64_t *offsets = NULL;
  uint32_t fixedLength = 0;
  double waitMs = 0, scatterMs = 0;
  const double t0 = nowMs();
  pthread_mutex_lock(&hostStage);
  const double t1 = nowMs();
  void *out = awfmGpuPinnedBuffer(g, 2, (n + 1) * sizeof(uint64_t)); /* counts (u32) or hit offsets (u64) */
  const bool packed = out && packQueries(g, data, n, job->threads, &chars, &offsets, &fixedLength);
  pthread_mutex_unlock(&hostStage);
  const double t2 = nowMs();
  if (!packed) {
    rc = AwFmAllocationFailure;
  } else if (!job->locate) {
    rc = awfmGpuCountHost(g, chars, offsets, fixedLength, n, NULL, out);
    if (rc == AwFmSuccess) {
      struct countCtx ctx = {data, out};
      const double t3 = nowMs();
      pthread_mutex_lock(&hostStage);
      const double t4 = nowMs();
      awfmParallelFor(job->threads, n, scatterCounts, &ctx);
      pthread_mutex_unlock(&hostStage);
      waitMs = t4 - t3;
      scatterMs = nowMs() - t4;
    }
  } else {
    /* the hit list arrives in windows bounded by the device's hit budget (one window for all but hit-heavy chunks),
     * in page-locked staging of the image that is valid while the sink runs */
    struct locateCtx ctx;
    memset(&ctx, 0, sizeof ctx);
    ctx.data = data;
    ctx.hitOffsets = out;
    ctx.n = n;
    ctx.threads = job->threads;
    rc = awfmGpuLocateHostWindows(g, chars, offsets, fixedLength, n, NULL, out, locateWindowSinkTurn, &ctx);
    if (rc == AwFmSuccess && !ctx.sized) { /* no hit at all: counts = 0 */
      pthread_mutex_lock(&hostStage);
      awfmParallelFor(job->threads, n, sizeLists, &ctx);
      pthread_mutex_unlock(&hostStage);
    }
    if (rc == AwFmSuccess && ctx.failed) rc = AwFmAllocationFailure;
    waitMs = ctx.waitMs;
    scatterMs = ctx.scatterMs;
  }
  const double t5 = nowMs();
  if (job->trace)
    fprintf(stderr, "[awfm aos] lane %u chunk %llu: %llu k-mers, turn %.2f ms, pack %.2f ms, device call %.2f ms (of which turn %.2f, scatter %.2f)\n",
            job->lane, (unsigned long long)chunkNumber, (unsigned long long)n, t1 - t0, t2 - t1, t5 - t2, waitMs, scatterMs);
  {
    /* bytes the host stages move for the chunk, at the granularity the memory system moves them: pack reads a 32-byte entry
     * and the k-mer's characters and writes the characters (+ 8 bytes of offset in a mixed-length chunk); scatter reads 8
     * bytes of hit offset (4 of count) per k-mer and 8 per position, rewrites every entry's line (count, possibly the list
     * pointer: 64 bytes read + written per 32-byte entry pair, i.e. 64 per k-mer), and for a k-mer with hits its position
     * list -- the reference's own 4-slot malloc per k-mer (ref src/AwFmParallelSearch.c:36-84): 48 bytes of heap, read for
     * ownership and written back; hits beyond the four slots at 16 bytes each */
    uint64_t chars = 0, hits = 0;
    if (packed) chars = offsets ? offsets[n] : (uint64_t)fixedLength * n;
    if (packed && job->locate && rc == AwFmSuccess) hits = ((const uint64_t *)out)[n];
    pthread_mutex_lock(&stageLock);
    lastStages.turnMs += (t1 - t0) + waitMs;
    lastStages.packMs += t2 - t1;
    lastStages.deviceMs += (t5 - t2) - waitMs - scatterMs;
    lastStages.scatterMs += scatterMs;
    lastStages.chunks++;
    lastStages.kmers += n;
    lastStages.hits += hits;
    lastStages.packBytes += 32u * n + 2u * chars + (offsets ? 8u * n : 0u);
    lastStages.scatterBytes += (job->locate ? 8u : 4u) * n + 64u * n + 96u * (hits < n ? hits : n) + 8u * hits + (hits > 4u * n ? 16u * (hits - 4u * n) : 0u);
    pthread_mutex_unlock(&stageLock);
  }
  return rc;
}

static void *runLane(void *p) {
  struct laneJob *job = p;
  AwFmGpuIndex *g = job->image;
  job->rc = AwFmSuccess;
  awfmGpuAosLock(g);
  for (uint64_t c = job->lane; c * job->chunk < job->n; c += job->numLanes) {
    const uint64_t begin = c * job->chunk, m = job->n - begin < job->chunk ? job->n - begin : job->chunk;
    job->rc = runChunk(job, job->data + begin, m, c);
    if (job->rc != AwFmSuccess) {
      snprintf(job->error, sizeof job->error, "%s", awfmGpuLastError());
      for (uint64_t d = c; d * job->chunk < job->n; d += job->numLanes) { /* this chunk and what the lane still had to do */
        const uint64_t b = d * job->chunk;
        invalidateCounts(job->data + b, job->n - b < job->chunk ? job->n - b : job->chunk);
      }
      break;
    }
  }
  awfmGpuAosUnlock(g);
  return NULL;
}

/* Chunks of the list go round-robin over the images of $AWFM_GPU_DEVICES (one host thread per image; the chunks are
 * independent, there is no exchange: ref src/AwFmParallelSearch.c:103-129 treats 8-query blocks the same way).
 * Returns the first failure. */
static enum AwFmReturnCode runBatch(const struct AwFmIndex *index, struct AwFmKmerSearchList *list, uint32_t numThreads,
                                    bool locate, const char *who) {
  const uint64_t n = (uint32_t)list->count; /* the reference reads the count as uint32_t (:100, :164) */
  if (n == 0) return AwFmSuccess;
  const double batchStart = nowMs();
  pthread_mutex_lock(&stageLock);
  memset(&lastStages, 0, sizeof lastStages);
  pthread_mutex_unlock(&stageLock);
  AwFmGpuIndex *images[AWFM_MAX_IMAGES];
  /* without an explicit device list a small batch is one chunk on one lane: splitting it would only add launches */
  const char *deviceList = awfmKnob(AWFM_KNOB_DEVICES);
  const bool oneShard = !(deviceList && *deviceList) && n < AWFM_MIN_SHARDED_LIST;
  const int numImages = awfmGpuIndexAcquireAll(index, images, oneShard ? 1 : AWFM_MAX_IMAGES);
  if (numImages <= 0) {
    fprintf(stderr, "%s: no device image: %s\n", who, awfmGpuLastError());
    invalidateCounts(list->kmerSearchData, n);
    return AwFmGeneralFailure;
  }
  uint64_t chunk = AWFM_AOS_CHUNK_DEFAULT;
  if (awfmKnob(AWFM_KNOB_AOS_CHUNK) && strtoull(awfmKnob(AWFM_KNOB_AOS_CHUNK), NULL, 10) > 0) chunk = strtoull(awfmKnob(AWFM_KNOB_AOS_CHUNK), NULL, 10);
  const uint64_t even = (n + (uint64_t)numImages - 1) / (uint64_t)numImages; /* a list of less than a chunk per image: even shares */
  if (chunk > even) chunk = even;
  struct laneJob jobs[AWFM_MAX_IMAGES];
  pthread_t threads[AWFM_MAX_IMAGES];
  bool spawned[AWFM_MAX_IMAGES] = {false};
  for (int i = 0; i < numImages; i++)
    jobs[i] = (struct laneJob){images[i], list->kmerSearchData, n, chunk, (unsigned)i, (unsigned)numImages, numThreads > 0 ? numThreads : 1,
                               locate, awfmKnob(AWFM_KNOB_DIAG) != NULL && strstr(awfmKnob(AWFM_KNOB_DIAG), "aos_trace") != NULL, AwFmSuccess, {0}};
  for (int i = 1; i < numImages; i++) spawned[i] = pthread_create(&threads[i], NULL, runLane, &jobs[i]) == 0;
  runLane(&jobs[0]);
  int firstFailed = jobs[0].rc != AwFmSuccess ? 0 : -1;
  for (int i = 1; i < numImages; i++) {
    if (spawned[i])
      pthread_join(threads[i], NULL);
    else
      runLane(&jobs[i]);
    if (firstFailed < 0 && jobs[i].rc != AwFmSuccess) firstFailed = i;
  }
  pthread_mutex_lock(&stageLock);
  lastStages.wallMs = nowMs() - batchStart;
  pthread_mutex_unlock(&stageLock);
  if (firstFailed < 0) return AwFmSuccess;
  fprintf(stderr, "%s: GPU search failed (%d) on image %d: %s\n", who, (int)jobs[firstFailed].rc, firstFailed,
          jobs[firstFailed].error);
  return jobs[firstFailed].rc;
}

/* see include/awfm_gpu.h */
void awfmGpuAosLastStages(double out[10]) {
  pthread_mutex_lock(&stageLock);
  out[0] = lastStages.wallMs;
  out[1] = lastStages.turnMs;
  out[2] = lastStages.packMs;
  out[3] = lastStages.deviceMs;
  out[4] = lastStages.scatterMs;
  out[5] = (double)lastStages.chunks;
  out[6] = (double)lastStages.kmers;
  out[7] = (double)lastStages.hits;
  out[8] = (double)lastStages.packBytes;
  out[9] = (double)lastStages.scatterBytes;
  pthread_mutex_unlock(&stageLock);
}

/* a copy of `bytes` bytes by `threads` threads of the pool the host stages run on: what this box's memory system gives them
 * (GB/s of bytes read + bytes written; the second of two passes -- the first touches the pages) */
struct copyCtx {
  uint8_t *dst;
  const uint8_t *src;
};
static void copyRange(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct copyCtx *c = p;
  memcpy(c->dst + begin, c->src + begin, (size_t)(end - begin));
}
double awfmHostCopyGBs(unsigned threads, uint64_t bytes) {
  if (bytes < 4096u) bytes = 4096u;
  struct copyCtx c = {malloc(bytes), malloc(bytes)};
  if (!c.dst || !c.src) {
    free(c.dst);
    free((void *)c.src);
    return 0.0;
  }
  memset((void *)c.src, 1, bytes);
  awfmParallelFor(threads ? threads : 1, bytes, copyRange, &c);
  double best = 0.0;
  for (int pass = 0; pass < 3; pass++) {
    const double t0 = nowMs();
    awfmParallelFor(threads ? threads : 1, bytes, copyRange, &c);
    const double gbs = 2.0 * (double)bytes / ((nowMs() - t0) * 1e-3) / 1e9;
    if (gbs > best) best = gbs;
  }
  free(c.dst);
  free((void *)c.src);
  return best;
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
