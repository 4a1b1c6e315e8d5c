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

/* ---- pack: AoS k-mers -> flat chars + CSR offsets ---- */
struct packCtx {
  const struct AwFmKmerSearchData *data;
  uint64_t *offsets;
  uint8_t *chars;
};

static void packChars(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct packCtx *c = p;
  for (uint64_t i = begin; i < end; i++)
    memcpy(c->chars + c->offsets[i], c->data[i].kmerString, c->data[i].kmerLength);
}

/* Packs the k-mers into page-locked staging buffers of the image.  When every k-mer has the same length
 * the offsets array is dropped (*offsetsOut = NULL, *fixedOut = that length). */
static bool packQueries(AwFmGpuIndex *g, const struct AwFmKmerSearchList *list, uint64_t n, unsigned threads,
                        uint8_t **charsOut, uint64_t **offsetsOut, uint32_t *fixedOut) {
  uint64_t *offsets = awfmGpuPinnedBuffer(g, 1, (n + 1) * sizeof(uint64_t));
  if (!offsets) return false;
  uint64_t total = 0;
  const uint64_t firstLength = list->kmerSearchData[0].kmerLength;
  bool uniform = firstLength != 0 && firstLength <= 0xFFFFFFFFull;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t len = list->kmerSearchData[i].kmerLength;
    offsets[i] = total;
    total += len;
    uniform &= len == firstLength;
  }
  offsets[n] = total;
  uint8_t *chars = awfmGpuPinnedBuffer(g, 0, total ? total : 1);
  if (!chars) return false;
  struct packCtx ctx = {list->kmerSearchData, offsets, chars};
  awfmParallelFor(threads, n, packChars, &ctx);
  *charsOut = chars;
  *offsetsOut = uniform ? NULL : offsets;
  *fixedOut = uniform ? (uint32_t)firstLength : 0;
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

/* ref src/AwFmParallelSearch.c:159-220 */
void awFmParallelSearchCount(const struct AwFmIndex *_RESTRICT_ const index,
                             struct AwFmKmerSearchList *_RESTRICT_ const searchList, uint32_t numThreads) {
  const uint64_t n = (uint32_t)searchList->count; /* the reference reads the count as uint32_t, :164 */
  if (n == 0) return;
  AwFmGpuIndex *g = awfmGpuIndexAcquire(index);
  if (!g) {
    fprintf(stderr, "awFmParallelSearchCount: no device image: %s\n", awfmGpuLastError());
    return;
  }
  awfmGpuAosLock(g);
  uint8_t *chars = NULL;
  uint64_t *offsets = NULL;
  uint32_t fixedLength = 0;
  uint32_t *counts = awfmGpuPinnedBuffer(g, 2, n * sizeof(uint32_t));
  if (!counts || !packQueries(g, searchList, n, numThreads, &chars, &offsets, &fixedLength)) {
    fprintf(stderr, "awFmParallelSearchCount: host staging allocation failed: %s\n", awfmGpuLastError());
    awfmGpuAosUnlock(g);
    return;
  }
  const enum AwFmReturnCode rc = awfmGpuCountHost(g, chars, offsets, fixedLength, n, NULL, counts);
  if (rc == AwFmSuccess) {
    struct countCtx ctx = {searchList->kmerSearchData, counts};
    awfmParallelFor(numThreads, n, scatterCounts, &ctx);
  } else {
    fprintf(stderr, "awFmParallelSearchCount: GPU search failed (%d): %s\n", (int)rc, awfmGpuLastError());
  }
  awfmGpuAosUnlock(g);
}

struct locateCtx {
  struct AwFmKmerSearchData *data;
  const uint64_t *hitOffsets;
  const uint64_t *positions;
  int failed;
};

/* ref src/AwFmParallelSearch.c:327-328, :361, :367-387: count is set, the list
 * grows by realloc to exactly `count` only when capacity is too small */
static void scatterPositions(void *p, uint64_t begin, uint64_t end, unsigned tid) {
  (void)tid;
  struct locateCtx *c = p;
  for (uint64_t i = begin; i < end; i++) {
    struct AwFmKmerSearchData *d = &c->data[i];
    const uint64_t hits = c->hitOffsets[i + 1] - c->hitOffsets[i];
    const uint32_t newCount = (uint32_t)hits;
    if (d->capacity < newCount) {
      void *grown = realloc(d->positionList, (size_t)newCount * sizeof(uint64_t));
      if (!grown) {
        fprintf(stderr, "Critical memory failure: could not allocate memory for position list.\n");
        __atomic_store_n(&c->failed, 1, __ATOMIC_RELAXED);
        continue;
      }
      d->positionList = grown;
      d->capacity = newCount;
    }
    d->count = newCount;
    memcpy(d->positionList, c->positions + c->hitOffsets[i], (size_t)newCount * sizeof(uint64_t));
  }
}

/* ref src/AwFmParallelSearch.c:95-157 */
enum AwFmReturnCode awFmParallelSearchLocate(const struct AwFmIndex *_RESTRICT_ const index,
                                             struct AwFmKmerSearchList *_RESTRICT_ const searchList,
                                             uint32_t numThreads) {
  const uint64_t n = (uint32_t)searchList->count; /* :100 */
  if (n == 0) return AwFmSuccess;
  AwFmGpuIndex *g = awfmGpuIndexAcquire(index);
  if (!g) {
    fprintf(stderr, "awFmParallelSearchLocate: no device image: %s\n", awfmGpuLastError());
    return AwFmGeneralFailure;
  }
  awfmGpuAosLock(g);
  uint8_t *chars = NULL;
  uint64_t *offsets = NULL;
  uint32_t fixedLength = 0;
  uint64_t *hitOffsets = awfmGpuPinnedBuffer(g, 2, (n + 1) * sizeof(uint64_t));
  uint64_t *positions = NULL;
  if (!hitOffsets || !packQueries(g, searchList, n, numThreads, &chars, &offsets, &fixedLength)) {
    awfmGpuAosUnlock(g);
    return AwFmAllocationFailure;
  }
  enum AwFmReturnCode rc = awfmGpuLocateHost(g, chars, offsets, fixedLength, n, NULL, hitOffsets, &positions);
  if (rc == AwFmSuccess) {
    struct locateCtx ctx = {searchList->kmerSearchData, hitOffsets, positions, 0};
    awfmParallelFor(numThreads, n, scatterPositions, &ctx);
    if (ctx.failed) rc = AwFmAllocationFailure;
  } else {
    fprintf(stderr, "awFmParallelSearchLocate: GPU search failed (%d): %s\n", (int)rc, awfmGpuLastError());
  }
  free(positions);
  awfmGpuAosUnlock(g);
  return rc;
}
