/* Minimal pthread parallel-for used for host-side pack/scatter (the reference
 * uses OpenMP for its whole search; here threads only move bytes). */
#include <pthread.h>
#include <stdlib.h>
#include "awfm_internal.h"

struct awfmTask {
  awfmRangeFn fn;
  void *ctx;
  uint64_t begin, end;
  unsigned tid;
};

static void *awfmTaskMain(void *p) {
  struct awfmTask *t = p;
  t->fn(t->ctx, t->begin, t->end, t->tid);
  return NULL;
}

void awfmParallelFor(unsigned numThreads, uint64_t n, awfmRangeFn fn, void *ctx) {
  if (numThreads > 64) numThreads = 64;
  if (numThreads <= 1 || n < 4096) {
    fn(ctx, 0, n, 0);
    return;
  }
  pthread_t threads[64];
  struct awfmTask tasks[64];
  bool spawned[64] = {false};
  const uint64_t chunk = (n + numThreads - 1) / numThreads;
  for (unsigned t = 1; t < numThreads; t++) {
    const uint64_t b = (uint64_t)t * chunk, e = b + chunk > n ? n : b + chunk;
    if (b >= n) break;
    tasks[t] = (struct awfmTask){fn, ctx, b, e, t};
    spawned[t] = pthread_create(&threads[t], NULL, awfmTaskMain, &tasks[t]) == 0;
    if (!spawned[t]) fn(ctx, b, e, t); /* could not start a thread: do the chunk here */
  }
  fn(ctx, 0, chunk > n ? n : chunk, 0);
  for (unsigned t = 1; t < numThreads; t++)
    if (spawned[t]) pthread_join(threads[t], NULL);
}
