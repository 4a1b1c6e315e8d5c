/*
 * Host index builder: text -> suffix array -> BWT bit-plane blocks, prefix
 * sums, k-mer seed table, bit-packed sampled SA -> .awfmi file.
 * Behavioural contract: ref src/AwFmCreate.c:31-137, :281-450.  The arrays it
 * produces are byte-identical to the reference's (they feed the GPU image and
 * the oracle alike).
 */
#include <stdlib.h>
#include <string.h>
#include "awfm_internal.h"

#define AWFM_GPU_BUILD_MIN_LENGTH (1ull << 20)

/* Fill the blocks from the suffix array: BWT[i] = text[SA[i]-1], '$' when
 * SA[i]==0; each block starts with a copy of the running letter counts
 * (ref src/AwFmCreate.c:291-336, :350-395).  Also derives the prefix sums
 * (ref src/AwFmCreate.c:338-344, :397-403). */
static void fillBlocksAndPrefixSums(struct AwFmIndex *ix, const uint8_t *text, const uint64_t *sa) {
  const bool amino = awfmIsAmino(ix);
  const unsigned planes = amino ? 5 : 3;
  const unsigned counters = amino ? 24 : 8;
  const size_t blockBytes = awfmBlockBytes(ix->config.alphabetType);
  uint8_t *blocks = (uint8_t *)ix->bwtBlockList.asNucleotide;
  uint64_t running[24] = {0};
  const uint64_t numBlocks = awfmNumBlocks(ix->bwtLength);

  for (uint64_t b = 0; b < numBlocks; b++) {
    uint64_t planeWords[5][4];
    memset(planeWords, 0, sizeof planeWords);
    uint64_t *countsOut = (uint64_t *)(blocks + b * blockBytes + 32 * planes);
    memcpy(countsOut, running, counters * sizeof(uint64_t));
    const uint64_t first = b * AW_FM_POSITIONS_PER_FM_BLOCK;
    const uint64_t last = first + AW_FM_POSITIONS_PER_FM_BLOCK < ix->bwtLength ? first + AW_FM_POSITIONS_PER_FM_BLOCK
                                                                              : ix->bwtLength;
    for (uint64_t i = first; i < last; i++) {
      const uint64_t textPos = sa[i];
      uint8_t letter;
      if (textPos == 0)
        letter = amino ? 21 : 5;
      else
        letter = amino ? awfmAminoAsciiToIndex(text[textPos - 1]) : awfmNucAsciiToIndex(text[textPos - 1]);
      running[letter]++;
      const unsigned code = amino ? awfmAminoIndexToCode(letter) : awfmNucIndexToCode(letter);
      const unsigned local = (unsigned)(i - first);
      for (unsigned j = 0; j < planes; j++) planeWords[j][local >> 6] |= (uint64_t)((code >> j) & 1u) << (local & 63);
    }
    memcpy(blocks + b * blockBytes, planeWords, 32 * planes);
  }
  const unsigned n = awfmPrefixSumsLength(ix->config.alphabetType);
  ix->prefixSums[0] = 1; /* the sentinel sorts first */
  for (unsigned i = 1; i < n; i++) ix->prefixSums[i] = ix->prefixSums[i - 1] + running[i - 1];
}

/* Seed table: entry of k-mer w = range reached by blind backward stepping,
 * i.e. with NO validity check, so absent k-mers hold the invalid ranges the
 * stepping produces (ref src/AwFmCreate.c:407-450).  Index of w: last letter
 * least significant (ref src/AwFmKmerTable.c:21-35).  Iterative DFS. */
static void fillSeedTable(struct AwFmIndex *ix) {
  const unsigned card = awfmCardinality(ix->config.alphabetType);
  const unsigned k = ix->config.kmerLengthInSeedTable;
  if (k == 0) { /* |A|^0 = 1 entry; the reference never writes it, leave it zeroed */
    ix->kmerSeedTable[0] = (struct AwFmSearchRange){0, 0};
    return;
  }
  struct AwFmSearchRange rangeAt[256];
  uint64_t indexAt[256], multAt[256];
  unsigned nextLetter[256];
  for (unsigned a0 = 0; a0 < card; a0++) {
    unsigned depth = 1;
    rangeAt[1] = (struct AwFmSearchRange){ix->prefixSums[a0], ix->prefixSums[a0 + 1] - 1};
    indexAt[1] = a0;
    multAt[1] = card;
    nextLetter[1] = 0;
    while (depth >= 1) {
      if (depth == k) {
        ix->kmerSeedTable[indexAt[depth]] = rangeAt[depth];
        depth--;
        continue;
      }
      const unsigned a = nextLetter[depth];
      if (a == card) {
        depth--;
        continue;
      }
      nextLetter[depth] = a + 1;
      struct AwFmSearchRange r = rangeAt[depth];
      if (awfmIsAmino(ix))
        awFmAminoIterativeStepBackwardSearch(ix, &r, (uint8_t)a);
      else
        awFmNucleotideIterativeStepBackwardSearch(ix, &r, (uint8_t)a);
      rangeAt[depth + 1] = r;
      indexAt[depth + 1] = indexAt[depth] + a * multAt[depth];
      multAt[depth + 1] = multAt[depth] * card;
      nextLetter[depth + 1] = 0;
      depth++;
    }
  }
}

/* ref src/AwFmCreate.c:31-137 */
enum AwFmReturnCode awFmCreateIndex(struct AwFmIndex *_RESTRICT_ *index,
                                    struct AwFmIndexConfiguration *_RESTRICT_ const config,
                                    const uint8_t *_RESTRICT_ const sequence, const size_t sequenceLength,
                                    const char *_RESTRICT_ const fileSrc) {
  return awfmCreateIndexWithFasta((struct AwFmIndex **)index, config, sequence, sequenceLength, fileSrc, NULL);
}

/* the build shared by awFmCreateIndex and awFmCreateIndexFromFasta (ref src/AwFmCreate.c:31-137, :140-279);
 * takes ownership of fastaVector on success */
enum AwFmReturnCode awfmCreateIndexWithFasta(struct AwFmIndex **index, const struct AwFmIndexConfiguration *config,
                                             const uint8_t *sequence, size_t sequenceLength, const char *fileSrc,
                                             struct FastaVector *fastaVector) {
  if (!config || !sequence || !fileSrc) return AwFmNullPtrError;
  *index = NULL;
  const bool amino = config->alphabetType == AwFmAlphabetAmino;
  const uint64_t saLength = (uint64_t)sequenceLength + 1;

  /* Texts of a megabase and more are built on the GPU when there is one (suffix sort, BWT planes, seed table and
   * SA packing in seconds for a 3.1 Gbp genome; the arrays and the file are byte-identical to this builder's,
   * tests/test_gpu_build.py).  $AWFM_HOST_BUILD=1 keeps the build on the host; so does any failure of the GPU
   * build (no device, not enough device memory: about 25 bytes per position, 37 from 2^32-1 positions on, where
   * suffix positions and ranks are 64-bit). */
  if (sequenceLength >= AWFM_GPU_BUILD_MIN_LENGTH && config->suffixArrayCompressionRatio != 0) {
    const char *hostOnly = awfmKnob(AWFM_KNOB_HOST_BUILD);
    if (!(hostOnly && *hostOnly && *hostOnly != '0') && awfmGpuDeviceCount() > 0) {
      const enum AwFmReturnCode rc =
          awfmGpuCreateIndexWithFasta(index, config, sequence, sequenceLength, 0, fileSrc, -1, fastaVector);
      if (*index) return rc; /* built (rc is the file write status, as below) */
    }
  }

  /* sanitized copy + '$' (ref src/AwFmCreate.c:53-66, :452-466) */
  uint8_t *text = malloc(saLength);
  if (!text) return AwFmAllocationFailure;
  for (size_t i = 0; i < sequenceLength; i++) text[i] = amino ? awfmAminoSanitize(sequence[i]) : awfmNucSanitize(sequence[i]);
  text[sequenceLength] = '$';

  struct AwFmIndex *ix = awfmIndexAlloc(config, saLength);
  uint64_t *sa = malloc(saLength * sizeof(uint64_t));
  if (!ix || !sa) {
    free(text);
    free(sa);
    awFmDeallocIndex(ix);
    return AwFmAllocationFailure;
  }
  ix->versionNumber = AWFM_VERSION_NUMBER;
  ix->featureFlags = fastaVector ? (1u << AWFM_FEATURE_BIT_FASTA_VECTOR) : 0u;

  if (awfmSuffixSort(text, saLength, sa) != 0) {
    free(text);
    free(sa);
    awFmDeallocIndex(ix);
    return AwFmSuffixArrayCreationFailure;
  }
  fillBlocksAndPrefixSums(ix, text, sa);
  free(text);
  fillSeedTable(ix);

  /* sampled, bit-packed SA (ref src/AwFmCreate.c:117-121) */
  ix->suffixArray.valueBitWidth = awfmSaWidth(saLength);
  ix->suffixArray.compressedByteLength = awfmSaPackedBytes(saLength, config->suffixArrayCompressionRatio);
  ix->suffixArray.values = malloc(ix->suffixArray.compressedByteLength);
  if (!ix->suffixArray.values) {
    free(sa);
    awFmDeallocIndex(ix);
    return AwFmAllocationFailure;
  }
  awfmSaPack(sa, saLength, config->suffixArrayCompressionRatio, ix->suffixArray.values);
  free(sa);
  ix->suffixArrayFileOffset = awfmSuffixArrayFileOffset(ix);
  ix->sequenceFileOffset = awfmSequenceFileOffset(ix);

  ix->fastaVector = fastaVector; /* the trailer of the file is written from it */
  const enum AwFmReturnCode rc = awFmWriteIndexToFile(ix, sequence, sequenceLength, fileSrc);
  if (awFmReturnCodeIsFailure(rc)) ix->fastaVector = NULL; /* the caller keeps ownership on failure */

  if (!config->keepSuffixArrayInMemory) { /* ref src/AwFmCreate.c:128-131 */
    free(ix->suffixArray.values);
    ix->suffixArray.values = NULL;
  }
  *index = ix;
  return rc;
}
