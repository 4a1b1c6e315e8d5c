/*
 * kmer_locate.c -- a program written against the reference's public API (AwFmIndex.h) only: build an index of a
 * seeded synthetic DNA text, count and locate a batch of k-mers, print an order-sensitive digest of the results.
 * The same source compiles against TravisWheelerLab/AvxWindowFmIndex; here it links libawfmindex_amd.so, where
 * awFmParallelSearchCount / awFmParallelSearchLocate run on the GPU.
 *
 *   cc -std=gnu11 -O2 examples/kmer_locate.c -Iinclude -Lavxwindowfmindex_amd -lawfmindex_amd \
 *      -Wl,-rpath,$PWD/avxwindowfmindex_amd -o kmer_locate && ./kmer_locate 200000 20000 14
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "AwFmIndex.h"

static uint64_t splitmix64(uint64_t *state) {
  uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static uint64_t fnv1a(uint64_t h, const void *data, size_t bytes) {
  const uint8_t *p = data;
  for (size_t i = 0; i < bytes; i++) h = (h ^ p[i]) * 0x100000001B3ull;
  return h;
}

int main(int argc, char **argv) {
  const size_t textLength = argc > 1 ? strtoull(argv[1], NULL, 10) : 200000;
  const size_t numKmers = argc > 2 ? strtoull(argv[2], NULL, 10) : 20000;
  const size_t kmerLength = argc > 3 ? strtoull(argv[3], NULL, 10) : 14;
  static const char letters[4] = {'a', 'c', 'g', 't'};
  uint64_t rng = 12345;
  uint8_t *text = malloc(textLength);
  for (size_t i = 0; i < textLength; i++) text[i] = (uint8_t)letters[splitmix64(&rng) & 3];

  struct AwFmIndexConfiguration config = {.suffixArrayCompressionRatio = 8,
                                          .kmerLengthInSeedTable = 8,
                                          .alphabetType = AwFmAlphabetDna,
                                          .keepSuffixArrayInMemory = true,
                                          .storeOriginalSequence = false};
  struct AwFmIndex *index = NULL;
  enum AwFmReturnCode rc = awFmCreateIndex(&index, &config, text, textLength, "kmer_locate.awfmi");
  if (awFmReturnCodeIsFailure(rc)) {
    fprintf(stderr, "awFmCreateIndex failed: %d\n", rc);
    return 2;
  }

  /* every other k-mer is drawn from the text, the rest are random */
  struct AwFmKmerSearchList *list = awFmCreateKmerSearchList(numKmers);
  char *kmers = malloc(numKmers * kmerLength);
  for (size_t i = 0; i < numKmers; i++) {
    char *kmer = kmers + i * kmerLength;
    if (i & 1) {
      memcpy(kmer, text + splitmix64(&rng) % (textLength - kmerLength), kmerLength);
    } else {
      for (size_t j = 0; j < kmerLength; j++) kmer[j] = letters[splitmix64(&rng) & 3];
    }
    list->kmerSearchData[i].kmerString = kmer;
    list->kmerSearchData[i].kmerLength = kmerLength;
  }
  list->count = numKmers;

  awFmParallelSearchCount(index, list, 4);
  uint64_t digest = 0xCBF29CE484222325ull, totalCount = 0;
  for (size_t i = 0; i < numKmers; i++) {
    digest = fnv1a(digest, &list->kmerSearchData[i].count, sizeof(uint32_t));
    totalCount += list->kmerSearchData[i].count;
  }
  rc = awFmParallelSearchLocate(index, list, 4);
  if (awFmReturnCodeIsFailure(rc)) {
    fprintf(stderr, "awFmParallelSearchLocate failed: %d\n", rc);
    return 3;
  }
  uint64_t totalHits = 0;
  for (size_t i = 0; i < numKmers; i++) {
    const struct AwFmKmerSearchData *d = &list->kmerSearchData[i];
    digest = fnv1a(digest, d->positionList, (size_t)d->count * sizeof(uint64_t));
    totalHits += d->count;
    for (uint32_t h = 0; h < d->count; h++) /* every reported position really holds the k-mer */
      if (memcmp(text + d->positionList[h], d->kmerString, kmerLength) != 0) {
        fprintf(stderr, "k-mer %zu: position %" PRIu64 " does not match\n", i, d->positionList[h]);
        return 4;
      }
  }
  printf("kmers %zu counted %" PRIu64 " located %" PRIu64 " digest %016" PRIx64 "\n", numKmers, totalCount, totalHits,
         digest);
  awFmDeallocKmerSearchList(list);
  awFmDeallocIndex(index);
  free(kmers);
  free(text);
  remove("kmer_locate.awfmi");
  return 0;
}
