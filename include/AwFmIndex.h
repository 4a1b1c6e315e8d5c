/*
 * AwFmIndex.h -- drop-in public API of the MI355X-native FM-index library
 * (libawfmindex_amd.so).
 *
 * Binary-compatible with the reference header /root/reference/src/AwFmIndex.h:
 * same struct layouts (LP64), same enum values, same function names, argument
 * meaning and return conventions.  Each declaration cites the reference
 * declaration it replaces.  Differences, all ABI-neutral:
 *   - FastaVector.h / <immintrin.h> are not included; `struct FastaVector` is
 *     opaque and the 256-bit plane type is a 32-byte aligned uint64_t[4].
 *   - `struct FastaVector` is this library's own record table (the reference's comes from a
 *     submodule that is not part of its tree); only its pointer appears in the ABI.
 *   - awFmParallelSearchCount / awFmParallelSearchLocate run on the GPU
 *     (HIP, gfx950).  There is no CPU fallback: without a usable device Locate
 *     returns AwFmGeneralFailure and Count leaves the list untouched and
 *     reports through awfmGpuLastError() (include/awfm_gpu.h).
 */
#ifndef AW_FM_INDEX_STRUCTS_H
#define AW_FM_INDEX_STRUCTS_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
#define _RESTRICT_ __restrict__
extern "C" {
#else
#define _RESTRICT_ restrict
#endif

/* ref src/AwFmIndex.h:16-26 */
#ifndef AW_FM_NUM_CONCURRENT_QUERIES
#define AW_FM_NUM_CONCURRENT_QUERIES 8
#endif
#define AW_FM_POSITIONS_PER_FM_BLOCK 256
#define AW_FM_CACHE_LINE_SIZE_IN_BYTES 64
#define AW_FM_NUCLEOTIDE_VECTORS_PER_WINDOW 3
#define AW_FM_NUCLEOTIDE_CARDINALITY 4
#define AW_FM_AMINO_VECTORS_PER_WINDOW 5
#define AW_FM_AMINO_CARDINALITY 20

/* ref src/AwFmIndex.h:29-33 */
enum AwFmAlphabetType { AwFmAlphabetAmino = 1, AwFmAlphabetDna = 2, AwFmAlphabetRna = 3 };
/* ref src/AwFmIndex.h:36 */
enum AwFmBwtType { AwFmBwtTypeBackwardOnly = 1, AwFmBwtTypeBiDirectional = 2 };

/* ref src/AwFmIndex.h:38-52: one 256-position bit plane, 32 bytes, 32-byte aligned */
typedef struct AwFmSimdVec256 {
  uint64_t qword[4];
} __attribute__((aligned(32))) AwFmSimdVec256;

/* ref src/AwFmIndex.h:55-59: 5 planes + 24 counters = 352 bytes */
struct AwFmAminoBlock {
  AwFmSimdVec256 letterBitVectors[AW_FM_AMINO_VECTORS_PER_WINDOW];
  uint64_t baseOccurrences[AW_FM_AMINO_CARDINALITY + 4];
};

/* ref src/AwFmIndex.h:61-65: 3 planes + 8 counters = 160 bytes */
struct AwFmNucleotideBlock {
  AwFmSimdVec256 letterBitVectors[AW_FM_NUCLEOTIDE_VECTORS_PER_WINDOW];
  uint64_t baseOccurrences[AW_FM_NUCLEOTIDE_CARDINALITY + 4];
};

/* ref src/AwFmIndex.h:67-70 */
union AwFmBwtBlockList {
  struct AwFmNucleotideBlock *asNucleotide;
  struct AwFmAminoBlock *asAmino;
};

/* ref src/AwFmIndex.h:74-80 (12 bytes) */
struct AwFmIndexConfiguration {
  uint8_t suffixArrayCompressionRatio;
  uint8_t kmerLengthInSeedTable;
  enum AwFmAlphabetType alphabetType;
  bool keepSuffixArrayInMemory;
  bool storeOriginalSequence;
};

/* ref src/AwFmIndex.h:82-86 */
struct AwFmCompressedSuffixArray {
  uint8_t valueBitWidth;
  uint8_t *values;
  uint64_t compressedByteLength;
};

/* ref src/AwFmIndex.h:88-91: inclusive on both ends */
struct AwFmSearchRange {
  uint64_t startPtr;
  uint64_t endPtr;
};

struct FastaVector; /* opaque here; ref src/AwFmIndex.h:8,107 */

/* ref src/AwFmIndex.h:94-109 (112 bytes); host arrays stay in reference layout,
 * the device image lives in a side table keyed by the index address */
struct AwFmIndex {
  uint32_t versionNumber;
  uint32_t featureFlags;
  uint64_t bwtLength;
  union AwFmBwtBlockList bwtBlockList;
  uint64_t *prefixSums;
  struct AwFmSearchRange *kmerSeedTable;
  FILE *fileHandle;
  struct AwFmIndexConfiguration config;
  int fileDescriptor;
  size_t suffixArrayFileOffset;
  size_t sequenceFileOffset;
  struct FastaVector *fastaVector;
  struct AwFmCompressedSuffixArray suffixArray;
};

/* ref src/AwFmIndex.h:111-117 (32 bytes); kmerString is not owned and not NUL
 * terminated, positionList is malloc-family memory owned by the list */
struct AwFmKmerSearchData {
  char *kmerString;
  uint64_t kmerLength;
  uint64_t *positionList;
  uint32_t count;
  uint32_t capacity;
};

/* ref src/AwFmIndex.h:119-123 */
struct AwFmKmerSearchList {
  size_t capacity;
  size_t count;
  struct AwFmKmerSearchData *kmerSearchData;
};

/* ref src/AwFmIndex.h:126-129 */
struct AwFmBacktrace {
  uint64_t position;
  uint64_t offset;
};

/* ref src/AwFmIndex.h:132-138: positive = success flavours, negative = failure */
enum AwFmReturnCode {
  AwFmSuccess = 1,
  AwFmFileReadOkay = 2,
  AwFmFileWriteOkay = 3,
  AwFmGeneralFailure = -1,
  AwFmUnsupportedVersionError = -2,
  AwFmAllocationFailure = -3,
  AwFmNullPtrError = -4,
  AwFmSuffixArrayCreationFailure = -5,
  AwFmIllegalPositionError = -6,
  AwFmNoFileSrcGiven = -7,
  AwFmNoDatabaseSequenceGiven = -8,
  AwFmFileFormatError = -9,
  AwFmFileOpenFail = -10,
  AwFmFileReadFail = -11,
  AwFmFileWriteFail = -12,
  AwFmErrorDbSequenceNull = -13,
  AwFmErrorSuffixArrayNull = -14,
  AwFmFileAlreadyExists = -15
};

/* ---- index lifetime ---------------------------------------------------- */

/* ref src/AwFmIndex.h:164-169 / src/AwFmCreate.c:31-137.  Returns
 * AwFmFileWriteOkay on success; *index is NULL on failure. */
enum AwFmReturnCode awFmCreateIndex(struct AwFmIndex *_RESTRICT_ *index,
                                    struct AwFmIndexConfiguration *_RESTRICT_ const config,
                                    const uint8_t *_RESTRICT_ const sequence, const size_t sequenceLength,
                                    const char *_RESTRICT_ const fileSrc);

/* ref src/AwFmIndex.h:196-200 / src/AwFmCreate.c:140-279.  The records of a multi-FASTA
 * file are indexed as one text with a terminator after every record (it becomes the
 * ambiguity letter, so no k-mer matches across records); headers and record boundaries
 * are kept for the two lookups below and stored in the .awfmi trailer. */
enum AwFmReturnCode awFmCreateIndexFromFasta(struct AwFmIndex *_RESTRICT_ *index,
                                             struct AwFmIndexConfiguration *_RESTRICT_ const config,
                                             const char *fastaSrc, const char *_RESTRICT_ const indexFileSrc);

/* ref src/AwFmIndex.h:212 / src/AwFmIndexStruct.c:57-70; also drops the device image */
void awFmDeallocIndex(struct AwFmIndex *index);

/* ref src/AwFmIndex.h:234-238 / src/AwFmFile.c:20-193 (.awfmi version 8) */
enum AwFmReturnCode awFmWriteIndexToFile(struct AwFmIndex *_RESTRICT_ const index,
                                         const uint8_t *_RESTRICT_ const sequence, const uint64_t sequenceLength,
                                         const char *_RESTRICT_ const fileSrc);

/* ref src/AwFmIndex.h:260-262 / src/AwFmFile.c:195-449 */
enum AwFmReturnCode awFmReadIndexFromFile(struct AwFmIndex *_RESTRICT_ *_RESTRICT_ index, const char *fileSrc,
                                          const bool keepSuffixArrayInMemory);

/* ---- batch search: the GPU hot path ------------------------------------- */

/* ref src/AwFmIndex.h:308 / src/AwFmParallelSearch.c:36-84 */
struct AwFmKmerSearchList *awFmCreateKmerSearchList(const size_t capacity);

/* ref src/AwFmIndex.h:326-327 / src/AwFmParallelSearch.c:86-93 */
void awFmDeallocKmerSearchList(struct AwFmKmerSearchList *_RESTRICT_ const searchList);

/* ref src/AwFmIndex.h:364-367 / src/AwFmParallelSearch.c:95-157.  numThreads is
 * used for the host-side pack/scatter threads. */
enum AwFmReturnCode awFmParallelSearchLocate(const struct AwFmIndex *_RESTRICT_ const index,
                                             struct AwFmKmerSearchList *_RESTRICT_ const searchList,
                                             uint32_t numThreads);

/* ref src/AwFmIndex.h:400-403 / src/AwFmParallelSearch.c:159-220 */
void awFmParallelSearchCount(const struct AwFmIndex *_RESTRICT_ const index,
                             struct AwFmKmerSearchList *_RESTRICT_ const searchList, uint32_t numThreads);

/* ---- single-query helpers (host-side scalar code, not the hot path) ------ */

/* ref src/AwFmIndex.h:286-289 / src/AwFmSearch.c:317-358 */
struct AwFmSearchRange awFmFindSearchRangeForString(const struct AwFmIndex *_RESTRICT_ const index,
                                                    const char *_RESTRICT_ const kmer, const size_t kmerLength);

/* ref src/AwFmIndex.h:429-433 / src/AwFmFile.c:451-482 */
enum AwFmReturnCode awFmReadSequenceFromFile(const struct AwFmIndex *_RESTRICT_ const index,
                                             const size_t sequenceStartPosition, const size_t sequenceSegmentLength,
                                             char *const sequenceBuffer);

/* ref src/AwFmIndex.h:455-458 / src/AwFmSearch.c:6-25 */
struct AwFmSearchRange awFmCreateInitialQueryRange(const struct AwFmIndex *_RESTRICT_ const index,
                                                   const char *_RESTRICT_ const query, const uint64_t queryLength);

/* ref src/AwFmIndex.h:477-478 / src/AwFmSearch.c:27-40 */
struct AwFmSearchRange awFmCreateInitialQueryRangeFromChar(const struct AwFmIndex *_RESTRICT_ const index,
                                                           const char letter);

/* ref src/AwFmIndex.h:494-496 / src/AwFmSearch.c:42-103 */
void awFmNucleotideIterativeStepBackwardSearch(const struct AwFmIndex *_RESTRICT_ const index,
                                               struct AwFmSearchRange *_RESTRICT_ const range,
                                               const uint8_t letterIndex);

/* ref src/AwFmIndex.h:512-514 / src/AwFmSearch.c:105-159 */
void awFmAminoIterativeStepBackwardSearch(const struct AwFmIndex *_RESTRICT_ const index,
                                          struct AwFmSearchRange *_RESTRICT_ const range, const uint8_t letterIndex);

/* ref src/AwFmIndex.h:547-550 / src/AwFmSearch.c:161-246; caller frees the result */
uint64_t *awFmFindDatabaseHitPositions(const struct AwFmIndex *_RESTRICT_ const index,
                                       const struct AwFmSearchRange *_RESTRICT_ const searchRange,
                                       enum AwFmReturnCode *_RESTRICT_ fileAccessResult);

/* ref src/AwFmIndex.h:574-576 / src/AwFmSearch.c:248-282 */
uint64_t awFmFindDatabaseHitPositionSingle(const struct AwFmIndex *_RESTRICT_ const index,
                                           const uint64_t bwtPosition,
                                           enum AwFmReturnCode *_RESTRICT_ fileAccessResult);

/* ref src/AwFmIndex.h:602-604 / src/AwFmSearch.c:284-301: AwFmUnsupportedVersionError for an
 * index that was not built from FASTA, AwFmIllegalPositionError outside every record */
enum AwFmReturnCode awFmGetLocalSequencePositionFromIndexPosition(const struct AwFmIndex *_RESTRICT_ const index,
                                                                  size_t globalPosition, size_t *sequenceNumber,
                                                                  size_t *localSequencePosition);

/* ref src/AwFmIndex.h:621-622 / src/AwFmSearch.c:429-455 */
uint8_t awFmNucleotideBacktraceReturnPreviousLetterIndex(const struct AwFmIndex *_RESTRICT_ const index,
                                                         uint64_t *bwtPosition);

/* ref src/AwFmIndex.h:639-640 / src/AwFmSearch.c:457-483 */
uint8_t awFmAminoBacktraceReturnPreviousLetterIndex(const struct AwFmIndex *_RESTRICT_ const index,
                                                    uint64_t *bwtPosition);

/* ref src/AwFmIndex.h:662-664 / src/AwFmSearch.c:303-315: *headerBuffer points into the index
 * (not NUL terminated) */
enum AwFmReturnCode awFmGetHeaderStringFromSequenceNumber(const struct AwFmIndex *_RESTRICT_ const index,
                                                          size_t sequenceNumber, char **headerBuffer,
                                                          size_t *headerLength);

/* ref src/AwFmIndex.h:680-681 / src/AwFmIndexStruct.c:126-130 */
size_t awFmSearchRangeLength(const struct AwFmSearchRange *_RESTRICT_ const range);

/* ref src/AwFmIndex.h:694,706 / src/AwFmIndexStruct.c:141-147 */
bool awFmReturnCodeIsFailure(const enum AwFmReturnCode rc);
bool awFmReturnCodeIsSuccess(const enum AwFmReturnCode rc);

/* ref src/AwFmIndex.h:720 / src/AwFmIndexStruct.c:149-155: 1 for an index not built from FASTA */
uint32_t awFmGetNumSequences(const struct AwFmIndex *_RESTRICT_ const index);

#ifdef __cplusplus
}
#endif
#endif /* AW_FM_INDEX_STRUCTS_H */
