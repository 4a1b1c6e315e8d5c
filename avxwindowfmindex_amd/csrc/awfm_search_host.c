/*
 * Host-side scalar primitives on the reference block layout.
 *
 * These serve (a) the index builder (seed-table fill) and (b) the single-query
 * entry points of AwFmIndex.h.  They are NOT used by awFmParallelSearchCount /
 * awFmParallelSearchLocate, which run on the GPU only (awfm_batch.c).
 */
#include <stdlib.h>
#include <string.h>
#include "awfm_internal.h"

static inline const uint8_t *blockAt(const struct AwFmIndex *ix, uint64_t block) {
  return (const uint8_t *)ix->bwtBlockList.asNucleotide + block * awfmBlockBytes(ix->config.alphabetType);
}

/* popcount of occVec(letter) over bits 0..p of one block
 * (ref src/AwFmOccurrence.c:8-135 + src/AwFmSimdConfig.c:89-114) */
static inline unsigned blockRank(const uint8_t *blk, unsigned planes, uint8_t ones, uint8_t zeros, unsigned p) {
  const uint64_t *w = (const uint64_t *)blk; /* plane j, word q at w[4*j + q] */
  unsigned total = 0;
  const unsigned last = p >> 6;
  for (unsigned q = 0; q <= last; q++) {
    uint64_t v = ~0ULL;
    for (unsigned j = 0; j < planes; j++) {
      const uint64_t plane = w[4 * j + q];
      const uint64_t wantOne = (uint64_t)0 - ((ones >> j) & 1u);
      const uint64_t wantZero = (uint64_t)0 - ((zeros >> j) & 1u);
      v &= (plane | ~wantOne) & (~plane | ~wantZero);
    }
    if (q == last) v &= ~0ULL >> (63 - (p & 63));
    total += (unsigned)__builtin_popcountll(v);
  }
  return total;
}

/* Occ(letter, q) (ref src/AwFmSearch.c:48-63) */
uint64_t awfmHostOcc(const struct AwFmIndex *ix, uint8_t letter, uint64_t q) {
  const bool amino = awfmIsAmino(ix);
  const unsigned planes = amino ? 5 : 3;
  const uint8_t *blk = blockAt(ix, q / AW_FM_POSITIONS_PER_FM_BLOCK);
  const uint64_t *base = (const uint64_t *)(blk + 32 * planes);
  const uint8_t ones = amino ? awfmAminoOnes[letter] : awfmNucOnes[letter];
  const uint8_t zeros = amino ? awfmAminoZeros[letter] : awfmNucZeros[letter];
  return base[letter] + blockRank(blk, planes, ones, zeros, (unsigned)(q % AW_FM_POSITIONS_PER_FM_BLOCK));
}

static void hostStep(const struct AwFmIndex *ix, struct AwFmSearchRange *range, uint8_t letter) {
  const uint64_t c = ix->prefixSums[letter];
  const uint64_t sp = c + awfmHostOcc(ix, letter, range->startPtr - 1);
  const uint64_t ep = c + awfmHostOcc(ix, letter, range->endPtr) - 1;
  range->startPtr = sp;
  range->endPtr = ep;
}

/* ref src/AwFmSearch.c:42-103 */
void awFmNucleotideIterativeStepBackwardSearch(const struct AwFmIndex *_RESTRICT_ const index,
                                               struct AwFmSearchRange *_RESTRICT_ const range,
                                               const uint8_t letterIndex) {
  hostStep(index, range, letterIndex);
}

/* ref src/AwFmSearch.c:105-159 */
void awFmAminoIterativeStepBackwardSearch(const struct AwFmIndex *_RESTRICT_ const index,
                                          struct AwFmSearchRange *_RESTRICT_ const range, const uint8_t letterIndex) {
  hostStep(index, range, letterIndex);
}

static inline uint8_t letterIndexOf(const struct AwFmIndex *ix, char c) {
  return awfmIsAmino(ix) ? awfmAminoAsciiToIndex((uint8_t)c) : awfmNucAsciiToIndex((uint8_t)c);
}

/* ref src/AwFmSearch.c:27-40 */
struct AwFmSearchRange awFmCreateInitialQueryRangeFromChar(const struct AwFmIndex *_RESTRICT_ const index,
                                                           const char letter) {
  const uint8_t a = letterIndexOf(index, letter);
  return (struct AwFmSearchRange){index->prefixSums[a], index->prefixSums[a + 1] - 1};
}

/* ref src/AwFmSearch.c:6-25 */
struct AwFmSearchRange awFmCreateInitialQueryRange(const struct AwFmIndex *_RESTRICT_ const index,
                                                   const char *_RESTRICT_ const query, const uint64_t queryLength) {
  return awFmCreateInitialQueryRangeFromChar(index, query[queryLength - 1]);
}

/* ref src/AwFmSearch.c:317-358 */
struct AwFmSearchRange awFmFindSearchRangeForString(const struct AwFmIndex *_RESTRICT_ const index,
                                                    const char *_RESTRICT_ const kmer, const size_t kmerLength) {
  size_t pos = kmerLength - 1;
  struct AwFmSearchRange range = awFmCreateInitialQueryRangeFromChar(index, kmer[pos]);
  while (range.startPtr <= range.endPtr && pos-- != 0) hostStep(index, &range, letterIndexOf(index, kmer[pos]));
  return range;
}

/* letter stored at a BWT position + LF step
 * (ref src/AwFmOccurrence.c:170-217, src/AwFmSearch.c:369-427); the sentinel maps to 0 */
uint64_t awfmHostLf(const struct AwFmIndex *ix, uint64_t p, uint8_t *letterOut) {
  const bool amino = awfmIsAmino(ix);
  const unsigned planes = amino ? 5 : 3;
  const uint8_t *blk = blockAt(ix, p / AW_FM_POSITIONS_PER_FM_BLOCK);
  const unsigned local = (unsigned)(p % AW_FM_POSITIONS_PER_FM_BLOCK);
  unsigned code = 0;
  for (unsigned j = 0; j < planes; j++) code |= ((blk[32 * j + local / 8] >> (local % 8)) & 1u) << j;
  const uint8_t letter = amino ? awfmAminoCodeToIndex((uint8_t)code) : awfmNucCodeToIndex((uint8_t)code);
  if (letterOut) *letterOut = letter;
  if (letter == (amino ? 21 : 5)) return 0;
  return ix->prefixSums[letter] + awfmHostOcc(ix, letter, p) - 1;
}

/* ref src/AwFmSearch.c:429-455; on the sentinel the position is left unchanged
 * and 0 is returned, as the reference does */
uint8_t awFmNucleotideBacktraceReturnPreviousLetterIndex(const struct AwFmIndex *_RESTRICT_ const index,
                                                         uint64_t *bwtPosition) {
  uint8_t letter;
  const uint64_t next = awfmHostLf(index, *bwtPosition, &letter);
  if (letter == 5) return 0;
  *bwtPosition = next;
  return letter;
}

/* ref src/AwFmSearch.c:457-483 */
uint8_t awFmAminoBacktraceReturnPreviousLetterIndex(const struct AwFmIndex *_RESTRICT_ const index,
                                                    uint64_t *bwtPosition) {
  uint8_t letter;
  const uint64_t next = awfmHostLf(index, *bwtPosition, &letter);
  if (letter == 21) return 0;
  *bwtPosition = next;
  return letter;
}

/* sampled SA value i, from memory or from the index file
 * (ref src/AwFmSuffixArray.c:149-177) */
static enum AwFmReturnCode sampledSaValue(const struct AwFmIndex *ix, uint64_t i, uint64_t *out) {
  if (ix->config.keepSuffixArrayInMemory && ix->suffixArray.values) {
    *out = awfmSaGet(ix->suffixArray.values, ix->suffixArray.valueBitWidth, i);
    return AwFmSuccess;
  }
  size_t v = 0;
  const enum AwFmReturnCode rc = awfmSaValueFromFile(ix, i, &v);
  *out = v;
  return rc;
}

/* ref src/AwFmSearch.c:248-282 */
uint64_t awFmFindDatabaseHitPositionSingle(const struct AwFmIndex *_RESTRICT_ const index,
                                           const uint64_t bwtPosition,
                                           enum AwFmReturnCode *_RESTRICT_ fileAccessResult) {
  const uint64_t ratio = index->config.suffixArrayCompressionRatio;
  uint64_t p = bwtPosition, offset = 0, value = 0;
  while (p % ratio != 0) {
    p = awfmHostLf(index, p, NULL);
    offset++;
  }
  if (sampledSaValue(index, p / ratio, &value) != AwFmSuccess) {
    *fileAccessResult = AwFmFileReadFail;
    return 0;
  }
  *fileAccessResult = AwFmFileReadOkay;
  return (value + offset) % index->bwtLength;
}

/* ref src/AwFmSearch.c:161-246: NULL + AwFmGeneralFailure for an empty range */
uint64_t *awFmFindDatabaseHitPositions(const struct AwFmIndex *_RESTRICT_ const index,
                                       const struct AwFmSearchRange *_RESTRICT_ const searchRange,
                                       enum AwFmReturnCode *_RESTRICT_ fileAccessResult) {
  const uint64_t hits = awFmSearchRangeLength(searchRange);
  if (hits == 0) {
    *fileAccessResult = AwFmGeneralFailure;
    return NULL;
  }
  uint64_t *positions = malloc(hits * sizeof(uint64_t));
  if (!positions) {
    *fileAccessResult = AwFmAllocationFailure;
    return NULL;
  }
  for (uint64_t i = 0; i < hits; i++) {
    enum AwFmReturnCode rc;
    positions[i] = awFmFindDatabaseHitPositionSingle(index, searchRange->startPtr + i, &rc);
    if (rc != AwFmFileReadOkay) {
      *fileAccessResult = AwFmFileReadFail;
      return positions;
    }
  }
  *fileAccessResult = AwFmFileReadOkay;
  return positions;
}
