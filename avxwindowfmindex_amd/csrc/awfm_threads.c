/* pthread parallel-for used for host-side pack/scatter (the reference uses OpenMP for its whole search; here threads
 * only move bytes).
 *
 * The workers are kept: a batch API call runs several short loops (1-3 ms each over a few million k-mers), and
 * creating and joining 31 threads per loop cost more than the loop (measured: 1.3 ms of a 2.2 ms pack of 2 M k-mers).
 * One loop runs on the pool at a time; a caller that finds it taken (two host threads driving two devices) runs its
 * loop on threads of its own, as every loop did before. */
#include <pthread.h>
#include <stdlib.h>
#include "awfm_internal.h"

#define AWFM_POOL_MAX 63 /* workers; the caller is the 64th thread of a loop */

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

static void spawnFor(unsigned numThreads, uint64_t n, awfmRangeFn fn, void *ctx) {
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

static struct {
  pthread_mutex_t owner; /* one loop at a time */
  pthread_mutex_t lock;  /* the fields below */
  pthread_cond_t start, done;
  unsigned workers;    /* threads created so far (they never exit) */
  uint64_t generation; /* bumped per loop */
  /* the loop being run */
  awfmRangeFn fn;
  void *ctx;
  uint64_t n, chunk;
  unsigned parts;   /* ranges 1 .. parts-1 belong to the workers (range 0 is the caller's) */
  unsigned pending; /* worker ranges not finished yet */
} pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, NULL, NULL, 0, 0, 0, 0};

static void *poolWorker(void *p) {
  const unsigned me = (unsigned)(uintptr_t)p; /* 1-based: the range this worker takes when a loop has that many parts */
  uint64_t seen = 0;
  pthread_mutex_lock(&pool.lock);
  for (;;) {
    while (pool.generation == seen) pthread_cond_wait(&pool.start, &pool.lock);
    seen = pool.generation;
    if (me >= pool.parts) continue; /* a narrower loop */
    const awfmRangeFn fn = pool.fn;
    void *ctx = pool.ctx;
    const uint64_t b = (uint64_t)me * pool.chunk, e = b + pool.chunk > pool.n ? pool.n : b + pool.chunk;
    pthread_mutex_unlock(&pool.lock);
    if (b < e) fn(ctx, b, e, me);
    pthread_mutex_lock(&pool.lock);
    if (--pool.pending == 0) pthread_cond_signal(&pool.done);
  }
  return NULL;
}

/* a forked child has none of the workers: it starts over with an empty pool */
static void poolAfterForkInChild(void) {
  pthread_mutex_init(&pool.owner, NULL);
  pthread_mutex_init(&pool.lock, NULL);
  pthread_cond_init(&pool.start, NULL);
  pthread_cond_init(&pool.done, NULL);
  pool.workers = 0;
  pool.pending = 0;
}
static void poolRegisterForkHandler(void) { pthread_atfork(NULL, NULL, poolAfterForkInChild); }
static pthread_once_t poolOnce = PTHREAD_ONCE_INIT;

void awfmParallelFor(unsigned numThreads, uint64_t n, awfmRangeFn fn, void *ctx) {
  if (numThreads > 64) numThreads = 64;
  if (numThreads <= 1 || n < 4096) {
    fn(ctx, 0, n, 0);
    return;
  }
  if (pthread_mutex_trylock(&pool.owner) != 0) {
    spawnFor(numThreads, n, fn, ctx);
    return;
  }
  pthread_once(&poolOnce, poolRegisterForkHandler);
  pthread_mutex_lock(&pool.lock);
  while (pool.workers + 1 < numThreads && pool.workers < AWFM_POOL_MAX) { /* grow to what this loop asks for */
    pthread_t t;
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_DETACHED);
    const int rc = pthread_create(&t, &attr, poolWorker, (void *)(uintptr_t)(pool.workers + 1));
    pthread_attr_destroy(&attr);
    if (rc != 0) break;
    pool.workers++;
  }
  if (pool.workers + 1 < numThreads) { /* the ranges of a loop depend on numThreads alone (callers pair loops by tid) */
    pthread_mutex_unlock(&pool.lock);
    pthread_mutex_unlock(&pool.owner);
    spawnFor(numThreads, n, fn, ctx);
    return;
  }
  const unsigned parts = numThreads;
  const uint64_t chunk = (n + parts - 1) / parts;
  pool.fn = fn;
  pool.ctx = ctx;
  pool.n = n;
  pool.chunk = chunk;
  pool.parts = parts;
  pool.pending = parts - 1;
  pool.generation++;
  if (parts > 1) pthread_cond_broadcast(&pool.start);
  pthread_mutex_unlock(&pool.lock);
  fn(ctx, 0, chunk > n ? n : chunk, 0);
  pthread_mutex_lock(&pool.lock);
  while (pool.pending != 0) pthread_cond_wait(&pool.done, &pool.lock);
  pthread_mutex_unlock(&pool.lock);
  pthread_mutex_unlock(&pool.owner);
}
