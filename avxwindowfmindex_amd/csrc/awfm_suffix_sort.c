/*
 * Host suffix-array construction for the index builder.
 *
 * The reference delegates to libdivsufsort (ref src/AwFmCreate.c:99-100); the
 * suffix array of a text is unique, so any correct construction yields the
 * same index bytes.  This one is a prefix-doubling sort seeded with a radix
 * sort on packed prefix keys:
 *   1. bytes are mapped to dense codes (b bits each) and every suffix gets a
 *      64-bit key holding its first 64/b characters;
 *   2. (key, position) pairs are LSD radix sorted;
 *   3. suffixes whose keys tie are refined Larsson-Sadakane style: within a
 *      tied group sort by the rank of the suffix h characters further on,
 *      doubling h until every group is a singleton.
 * Uniform random text is fully resolved by step 2 except for a handful of
 * groups; repetitive text costs O(log n) refinement rounds.
 */
#include <stdlib.h>
#include <string.h>
#include "awfm_internal.h"

typedef struct {
  uint64_t key;
  uint64_t pos;
} awfmPair;

static void radixSortPairs(awfmPair *a, awfmPair *tmp, uint64_t n) {
  for (unsigned shift = 0; shift < 64; shift += 8) {
    uint64_t count[256] = {0};
    for (uint64_t i = 0; i < n; i++) count[(a[i].key >> shift) & 255]++;
    bool trivial = false;
    for (unsigned b = 0; b < 256; b++)
      if (count[b] == n) trivial = true;
    if (trivial) continue;
    uint64_t sum = 0;
    for (unsigned b = 0; b < 256; b++) {
      const uint64_t c = count[b];
      count[b] = sum;
      sum += c;
    }
    for (uint64_t i = 0; i < n; i++) tmp[count[(a[i].key >> shift) & 255]++] = a[i];
    memcpy(a, tmp, n * sizeof(awfmPair));
  }
}

static int cmpPairs(const void *x, const void *y) {
  const awfmPair *a = x, *b = y;
  return a->key < b->key ? -1 : (a->key > b->key ? 1 : 0);
}

int awfmSuffixSort(const uint8_t *s, uint64_t n, uint64_t *sa) {
  if (n == 0) return 0;
  /* dense codes, 0 reserved for "past the end" */
  unsigned code[256] = {0}, distinct = 0;
  bool present[256] = {false};
  for (uint64_t i = 0; i < n; i++) present[s[i]] = true;
  for (unsigned c = 0; c < 256; c++)
    if (present[c]) code[c] = ++distinct;
  unsigned bits = 1;
  while ((1u << bits) <= distinct) bits++;
  const unsigned perKey = 64 / bits;

  awfmPair *pairs = malloc(n * sizeof(awfmPair));
  awfmPair *tmp = malloc(n * sizeof(awfmPair));
  uint64_t *rank = malloc(n * sizeof(uint64_t));
  if (!pairs || !tmp || !rank) {
    free(pairs);
    free(tmp);
    free(rank);
    return -1;
  }
  /* rolling key: key(i) = key(i+1) >> bits | code(s[i]) << topShift */
  const unsigned topShift = bits * (perKey - 1);
  uint64_t key = 0;
  for (uint64_t i = n; i-- > 0;) {
    key = (key >> bits) | ((uint64_t)code[s[i]] << topShift);
    pairs[i].key = key;
    pairs[i].pos = i;
  }
  radixSortPairs(pairs, tmp, n);

  /* rank = index of the group head; collect whether anything is tied */
  uint64_t unresolved = 0;
  for (uint64_t i = 0; i < n;) {
    uint64_t j = i + 1;
    while (j < n && pairs[j].key == pairs[i].key) j++;
    for (uint64_t t = i; t < j; t++) rank[pairs[t].pos] = i;
    if (j - i > 1) unresolved += j - i;
    i = j;
  }
  for (uint64_t i = 0; i < n; i++) sa[i] = pairs[i].pos;

  /* refinement: groups are maximal runs of equal rank[sa[.]] */
  for (uint64_t h = perKey; unresolved > 0; h *= 2) {
    unresolved = 0;
    /* pass 1: sort each tied group by the rank h characters ahead (old ranks) */
    for (uint64_t i = 0; i < n;) {
      uint64_t j = i + 1;
      const uint64_t r = rank[sa[i]];
      while (j < n && rank[sa[j]] == r) j++;
      if (j - i > 1) {
        for (uint64_t t = i; t < j; t++) {
          const uint64_t p = sa[t];
          pairs[t].pos = p;
          pairs[t].key = p + h < n ? rank[p + h] + 1 : 0;
        }
        qsort(pairs + i, j - i, sizeof(awfmPair), cmpPairs);
        for (uint64_t t = i; t < j; t++) sa[t] = pairs[t].pos;
      } else {
        pairs[i].key = ~0ULL; /* singleton marker, never compared */
      }
      i = j;
    }
    /* pass 2: split groups where the secondary key changes, assign new ranks */
    for (uint64_t i = 0; i < n;) {
      uint64_t j = i + 1;
      const uint64_t r = rank[sa[i]];
      while (j < n && rank[sa[j]] == r) j++;
      if (j - i > 1) {
        uint64_t head = i;
        tmp[i].key = head;
        for (uint64_t t = i + 1; t < j; t++) {
          if (pairs[t].key != pairs[t - 1].key) head = t;
          tmp[t].key = head;
        }
        /* count what is still tied */
        for (uint64_t t = i; t < j;) {
          uint64_t u = t + 1;
          while (u < j && tmp[u].key == tmp[t].key) u++;
          if (u - t > 1) unresolved += u - t;
          t = u;
        }
      } else {
        tmp[i].key = i;
      }
      i = j;
    }
    for (uint64_t i = 0; i < n; i++) rank[sa[i]] = tmp[i].key;
  }
  free(pairs);
  free(tmp);
  free(rank);
  return 0;
}
