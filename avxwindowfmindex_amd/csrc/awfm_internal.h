/*
 * awfm_internal.h -- declarations shared by the host-side C sources of
 * libawfmindex_amd.so.  Host code is plain C11; the device side lives in
 * awfm_gpu.hip behind the C ABI of include/awfm_gpu.h.
 */
#ifndef AWFM_INTERNAL_H
#define AWFM_INTERNAL_H

#include "AwFmIndex.h"
#include "awfm_gpu.h"
#include "awfm_knobs.h"

#define AWFM_VERSION_NUMBER 8u          /* ref src/AwFmIndexStruct.h:9 */
#define AWFM_FEATURE_BIT_FASTA_VECTOR 0 /* ref src/AwFmIndexStruct.h:10 */
#define AWFM_SA_PAD_BYTES 8             /* ref src/AwFmSuffixArray.c:9 */

/* record table of a multi-FASTA index (this library's own definition of the type the reference header
 * only names; ref src/AwFmIndex.h:107) */
struct AwfmFastaRecord {
  size_t headerEndPosition;   /* end of this record's header in the concatenated header string */
  size_t sequenceEndPosition; /* end of this record's residues in the concatenated text (terminator excluded) */
};
struct FastaVector {
  char *headers;
  size_t headerLength;
  struct AwfmFastaRecord *records;
  size_t numRecords;
};

#ifdef __cplusplus
extern "C" {
#endif

/* ---- awfm_fasta.c / awfm_build.c ---- */
void awfmFastaVectorFree(struct FastaVector *fv);
enum AwFmReturnCode awfmCreateIndexWithFasta(struct AwFmIndex **index, const struct AwFmIndexConfiguration *config,
                                             const uint8_t *sequence, size_t sequenceLength, const char *fileSrc,
                                             struct FastaVector *fastaVector);
/* the same build on the GPU (awfm_gpu_build.hip): byte-identical arrays; sequence is a host pointer unless
 * sequenceOnDevice; fails (negative code, awfmGpuLastError) without a device or beyond 2^32-2 positions */
enum AwFmReturnCode awfmGpuCreateIndexWithFasta(struct AwFmIndex **index, const struct AwFmIndexConfiguration *config,
                                                const uint8_t *sequence, uint64_t sequenceLength, int sequenceOnDevice,
                                                const char *fileSrc, int device, struct FastaVector *fastaVector);

/* ---- awfm_letters.c (ref src/AwFmLetter.c) ---- */
uint8_t awfmNucAsciiToIndex(uint8_t c);
uint8_t awfmAminoAsciiToIndex(uint8_t c);
uint8_t awfmNucSanitize(uint8_t c);
uint8_t awfmAminoSanitize(uint8_t c);
uint8_t awfmNucIndexToCode(uint8_t letterIndex);
uint8_t awfmAminoIndexToCode(uint8_t letterIndex);
uint8_t awfmNucCodeToIndex(uint8_t code);
uint8_t awfmAminoCodeToIndex(uint8_t code);
bool awfmLetterIsAmbiguous(uint8_t c, enum AwFmAlphabetType alphabet);
/* per letter: planes that must be 1 / must be 0 in the occurrence vector
 * (ref src/AwFmOccurrence.c:8-36, :52-135) */
extern const uint8_t awfmNucOnes[5], awfmNucZeros[5];
extern const uint8_t awfmAminoOnes[21], awfmAminoZeros[21];

/* ---- awfm_index.c (ref src/AwFmIndexStruct.c) ---- */
static inline bool awfmIsAmino(const struct AwFmIndex *ix) { return ix->config.alphabetType == AwFmAlphabetAmino; }
static inline unsigned awfmCardinality(enum AwFmAlphabetType a) {
  return a == AwFmAlphabetAmino ? AW_FM_AMINO_CARDINALITY : AW_FM_NUCLEOTIDE_CARDINALITY;
}
static inline size_t awfmBlockBytes(enum AwFmAlphabetType a) {
  return a == AwFmAlphabetAmino ? sizeof(struct AwFmAminoBlock) : sizeof(struct AwFmNucleotideBlock);
}
static inline uint64_t awfmNumBlocks(uint64_t bwtLength) { return 1 + (bwtLength - 1) / AW_FM_POSITIONS_PER_FM_BLOCK; }
static inline unsigned awfmPrefixSumsLength(enum AwFmAlphabetType a) { return awfmCardinality(a) + 2; }
uint64_t awfmKmerTableLength(enum AwFmAlphabetType a, unsigned k);
struct AwFmIndex *awfmIndexAlloc(const struct AwFmIndexConfiguration *config, uint64_t bwtLength);

/* ---- awfm_sa.c (ref src/AwFmSuffixArray.c) ---- */
uint8_t awfmSaWidth(uint64_t saLength);
uint64_t awfmSaSampleCount(uint64_t bwtLength, uint64_t ratio);
uint64_t awfmSaPackedBytes(uint64_t saLength, uint8_t ratio);
void awfmSaPack(const uint64_t *fullSa, uint64_t saLength, uint8_t ratio, uint8_t *out);
uint64_t awfmSaGet(const uint8_t *values, uint8_t width, uint64_t i);

/* ---- awfm_suffix_sort.c ---- */
/* suffix array of s[0..n) where s[n-1] is a unique smallest byte; 0 on success */
int awfmSuffixSort(const uint8_t *s, uint64_t n, uint64_t *sa);

/* ---- awfm_file.c (ref src/AwFmFile.c) ---- */
size_t awfmSequenceFileOffset(const struct AwFmIndex *ix);
size_t awfmSuffixArrayFileOffset(const struct AwFmIndex *ix);
enum AwFmReturnCode awfmSaValueFromFile(const struct AwFmIndex *ix, size_t i, size_t *valueOut);
/* reads the whole packed sampled SA of a file-backed index into a malloc'ed buffer */
uint8_t *awfmReadPackedSaFromFile(const struct AwFmIndex *ix);

/* ---- awfm_search_host.c (host scalar, single-query API + builder) ---- */
uint64_t awfmHostOcc(const struct AwFmIndex *ix, uint8_t letter, uint64_t q);
uint64_t awfmHostLf(const struct AwFmIndex *ix, uint64_t p, uint8_t *letterOut);

/* ---- awfm_threads.c ---- */
typedef void (*awfmRangeFn)(void *ctx, uint64_t begin, uint64_t end, unsigned tid);
void awfmParallelFor(unsigned numThreads, uint64_t n, awfmRangeFn fn, void *ctx);

#ifdef __cplusplus
}
#endif
#endif
