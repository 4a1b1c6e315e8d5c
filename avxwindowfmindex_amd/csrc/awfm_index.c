/* AwFmIndex allocation and the small public helpers (ref src/AwFmIndexStruct.c). */
#include <stdlib.h>
#include <string.h>
#include "awfm_internal.h"

/* ref src/AwFmIndexStruct.c:77-86: |A|^k entries */
uint64_t awfmKmerTableLength(enum AwFmAlphabetType a, unsigned k) {
  uint64_t len = 1;
  for (unsigned i = 0; i < k; i++) len *= awfmCardinality(a);
  return len;
}

/* ref src/AwFmIndexStruct.c:9-55: zeroed struct, prefix sums, 32-byte aligned
 * block list, seed table */
struct AwFmIndex *awfmIndexAlloc(const struct AwFmIndexConfiguration *config, uint64_t bwtLength) {
  struct AwFmIndex *ix = calloc(1, sizeof *ix);
  if (!ix) return NULL;
  ix->config = *config;
  ix->bwtLength = bwtLength;
  ix->fileDescriptor = -1;
  const size_t blockBytes = awfmNumBlocks(bwtLength) * awfmBlockBytes(config->alphabetType);
  ix->prefixSums = malloc(awfmPrefixSumsLength(config->alphabetType) * sizeof(uint64_t));
  ix->bwtBlockList.asNucleotide = aligned_alloc(32, blockBytes); /* block sizes are multiples of 32 */
  ix->kmerSeedTable =
      malloc(awfmKmerTableLength(config->alphabetType, config->kmerLengthInSeedTable) * sizeof(struct AwFmSearchRange));
  if (!ix->prefixSums || !ix->bwtBlockList.asNucleotide || !ix->kmerSeedTable) {
    awFmDeallocIndex(ix);
    return NULL;
  }
  return ix;
}

/* ref src/AwFmIndexStruct.c:57-70 */
void awFmDeallocIndex(struct AwFmIndex *index) {
  if (!index) return;
  awfmGpuIndexRelease(index);
  if (index->fileHandle) fclose(index->fileHandle);
  free(index->bwtBlockList.asNucleotide);
  free(index->prefixSums);
  free(index->kmerSeedTable);
  free(index->suffixArray.values);
  awfmFastaVectorFree(index->fastaVector);
  free(index);
}

/* ref src/AwFmIndexStruct.c:126-130 */
size_t awFmSearchRangeLength(const struct AwFmSearchRange *_RESTRICT_ const range) {
  return range->startPtr <= range->endPtr ? range->endPtr - range->startPtr + 1 : 0;
}

/* ref src/AwFmIndexStruct.c:141-147 */
bool awFmReturnCodeIsFailure(const enum AwFmReturnCode rc) { return rc < 0; }
bool awFmReturnCodeIsSuccess(const enum AwFmReturnCode rc) { return rc >= 0; }
