/*
 * .awfmi version-8 file writer/reader (format contract: ref src/AwFmFile.c).
 * Layout: "AwFmIndex\n" | u32 version | u32 featureFlags | u8 saRatio |
 * u8 seedK | u8 alphabet | u8 storeSequence | u64 bwtLength | blocks |
 * prefix sums | seed table | [sequence] | packed sampled SA.
 */
#define _XOPEN_SOURCE 700
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "awfm_internal.h"

static const char kMagic[10] = {'A', 'w', 'F', 'm', 'I', 'n', 'd', 'e', 'x', '\n'};

/* ref src/AwFmFile.c:524-541 */
size_t awfmSequenceFileOffset(const struct AwFmIndex *ix) {
  return sizeof kMagic + 12 + sizeof(uint64_t) + awfmNumBlocks(ix->bwtLength) * awfmBlockBytes(ix->config.alphabetType) +
         awfmPrefixSumsLength(ix->config.alphabetType) * sizeof(uint64_t) +
         awfmKmerTableLength(ix->config.alphabetType, ix->config.kmerLengthInSeedTable) * sizeof(struct AwFmSearchRange);
}

/* ref src/AwFmFile.c:543-551 */
size_t awfmSuffixArrayFileOffset(const struct AwFmIndex *ix) {
  return awfmSequenceFileOffset(ix) + (ix->config.storeOriginalSequence ? ix->bwtLength - 1 : 0);
}

static bool put(FILE *f, const void *p, size_t bytes) { return bytes == 0 || fwrite(p, 1, bytes, f) == bytes; }
static bool get(FILE *f, void *p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }

/* FASTA trailer (ref src/AwFmFile.c:157-187): header length, record count, header characters, records */
static bool putFastaTrailer(FILE *f, const struct AwFmIndex *index) {
  const struct FastaVector *fv = index->fastaVector;
  if (!fv || (index->featureFlags & (1u << AWFM_FEATURE_BIT_FASTA_VECTOR)) == 0) return true;
  return put(f, &fv->headerLength, sizeof(size_t)) && put(f, &fv->numRecords, sizeof(size_t)) &&
         put(f, fv->headers, fv->headerLength) && put(f, fv->records, fv->numRecords * sizeof(struct AwfmFastaRecord));
}

/* ref src/AwFmFile.c:360-440 */
static bool getFastaTrailer(FILE *f, struct AwFmIndex *ix) {
  if ((ix->featureFlags & (1u << AWFM_FEATURE_BIT_FASTA_VECTOR)) == 0) return true;
  if (fseek(f, (long)(ix->suffixArrayFileOffset + ix->suffixArray.compressedByteLength), SEEK_SET) != 0) return false;
  struct FastaVector *fv = calloc(1, sizeof *fv);
  if (!fv) return false;
  bool ok = get(f, &fv->headerLength, sizeof(size_t)) && get(f, &fv->numRecords, sizeof(size_t));
  if (ok) {
    fv->headers = malloc(fv->headerLength ? fv->headerLength : 1);
    fv->records = malloc((fv->numRecords ? fv->numRecords : 1) * sizeof(struct AwfmFastaRecord));
    ok = fv->headers && fv->records && get(f, fv->headers, fv->headerLength) &&
         get(f, fv->records, fv->numRecords * sizeof(struct AwfmFastaRecord));
  }
  if (!ok) {
    awfmFastaVectorFree(fv);
    return false;
  }
  ix->fastaVector = fv;
  return true;
}

/* ref src/AwFmFile.c:20-193 */
enum AwFmReturnCode awFmWriteIndexToFile(struct AwFmIndex *_RESTRICT_ const index,
                                         const uint8_t *_RESTRICT_ const sequence, const uint64_t sequenceLength,
                                         const char *_RESTRICT_ const fileSrc) {
  if (!fileSrc) return AwFmNoFileSrcGiven;
  if (!index || !sequence) return AwFmNullPtrError;
  if (index->fileHandle) fclose(index->fileHandle);
  index->fileHandle = fopen(fileSrc, "w+b");
  if (!index->fileHandle) return AwFmFileAlreadyExists; /* sic, ref src/AwFmFile.c:44-46 */
  FILE *f = index->fileHandle;
  const uint8_t header[4] = {index->config.suffixArrayCompressionRatio, index->config.kmerLengthInSeedTable,
                             (uint8_t)index->config.alphabetType, (uint8_t)index->config.storeOriginalSequence};
  const bool ok =
      put(f, kMagic, sizeof kMagic) && put(f, &index->versionNumber, 4) && put(f, &index->featureFlags, 4) &&
      put(f, header, 4) && put(f, &index->bwtLength, 8) &&
      put(f, index->bwtBlockList.asNucleotide,
          awfmNumBlocks(index->bwtLength) * awfmBlockBytes(index->config.alphabetType)) &&
      put(f, index->prefixSums, awfmPrefixSumsLength(index->config.alphabetType) * sizeof(uint64_t)) &&
      put(f, index->kmerSeedTable,
          awfmKmerTableLength(index->config.alphabetType, index->config.kmerLengthInSeedTable) *
              sizeof(struct AwFmSearchRange)) &&
      (!index->config.storeOriginalSequence || put(f, sequence, sequenceLength)) &&
      put(f, index->suffixArray.values, index->suffixArray.compressedByteLength) && putFastaTrailer(f, index);
  if (!ok) {
    fclose(f);
    index->fileHandle = NULL;
    return AwFmFileWriteFail;
  }
  fflush(f);
  index->fileDescriptor = fileno(f);
  return AwFmFileWriteOkay;
}

/* ref src/AwFmFile.c:195-449 (with the FASTA trailer when the feature flag is set) */
enum AwFmReturnCode awFmReadIndexFromFile(struct AwFmIndex *_RESTRICT_ *_RESTRICT_ index, const char *fileSrc,
                                          const bool keepSuffixArrayInMemory) {
  if (!fileSrc) return AwFmNoFileSrcGiven;
  FILE *f = fopen(fileSrc, "rb");
  if (!f) return AwFmFileOpenFail;
  char magic[sizeof kMagic];
  uint32_t version, flags;
  uint8_t header[4];
  uint64_t bwtLength;
  if (!get(f, magic, sizeof magic)) {
    fclose(f);
    return AwFmFileReadFail;
  }
  if (memcmp(magic, kMagic, sizeof kMagic) != 0) {
    fclose(f);
    return AwFmFileFormatError;
  }
  if (!get(f, &version, 4)) {
    fclose(f);
    return AwFmFileReadFail;
  }
  if (version != AWFM_VERSION_NUMBER) {
    fclose(f);
    return AwFmUnsupportedVersionError;
  }
  if (!get(f, &flags, 4) || !get(f, header, 4) || !get(f, &bwtLength, 8)) {
    fclose(f);
    return AwFmFileReadFail;
  }
  struct AwFmIndexConfiguration config = {.suffixArrayCompressionRatio = header[0],
                                          .kmerLengthInSeedTable = header[1],
                                          .alphabetType = (enum AwFmAlphabetType)header[2],
                                          .keepSuffixArrayInMemory = keepSuffixArrayInMemory,
                                          .storeOriginalSequence = header[3] != 0};
  struct AwFmIndex *ix = awfmIndexAlloc(&config, bwtLength);
  if (!ix) {
    fclose(f);
    return AwFmAllocationFailure;
  }
  ix->versionNumber = version;
  ix->featureFlags = flags;
  ix->fileHandle = f; /* from here on awFmDeallocIndex closes the file */
  ix->fileDescriptor = fileno(f);
  if (!get(f, ix->bwtBlockList.asNucleotide, awfmNumBlocks(bwtLength) * awfmBlockBytes(config.alphabetType)) ||
      !get(f, ix->prefixSums, awfmPrefixSumsLength(config.alphabetType) * sizeof(uint64_t)) ||
      !get(f, ix->kmerSeedTable,
           awfmKmerTableLength(config.alphabetType, config.kmerLengthInSeedTable) * sizeof(struct AwFmSearchRange))) {
    awFmDeallocIndex(ix);
    return AwFmFileReadFail;
  }
  ix->suffixArray.valueBitWidth = awfmSaWidth(bwtLength);
  ix->suffixArray.compressedByteLength = awfmSaPackedBytes(bwtLength, config.suffixArrayCompressionRatio);
  ix->suffixArrayFileOffset = awfmSuffixArrayFileOffset(ix);
  ix->sequenceFileOffset = awfmSequenceFileOffset(ix);
  if (keepSuffixArrayInMemory) {
    ix->suffixArray.values = awfmReadPackedSaFromFile(ix);
    if (!ix->suffixArray.values) {
      awFmDeallocIndex(ix);
      return AwFmFileReadFail;
    }
  }
  if (!getFastaTrailer(f, ix)) {
    awFmDeallocIndex(ix);
    return AwFmFileReadFail;
  }
  *index = ix;
  return AwFmFileReadOkay;
}

uint8_t *awfmReadPackedSaFromFile(const struct AwFmIndex *ix) {
  if (ix->fileDescriptor < 0) return NULL;
  const size_t bytes = ix->suffixArray.compressedByteLength;
  uint8_t *buf = malloc(bytes);
  if (!buf) return NULL;
  size_t done = 0;
  while (done < bytes) {
    const ssize_t r = pread(ix->fileDescriptor, buf + done, bytes - done, (off_t)(ix->suffixArrayFileOffset + done));
    if (r <= 0) {
      free(buf);
      return NULL;
    }
    done += (size_t)r;
  }
  return buf;
}

/* ref src/AwFmFile.c:451-482 */
enum AwFmReturnCode awFmReadSequenceFromFile(const struct AwFmIndex *_RESTRICT_ const index,
                                             const size_t sequenceStartPosition, const size_t sequenceSegmentLength,
                                             char *const sequenceBuffer) {
  if (!index->config.storeOriginalSequence) return AwFmUnsupportedVersionError;
  if (sequenceStartPosition + sequenceSegmentLength > index->bwtLength) return AwFmIllegalPositionError;
  const ssize_t r = pread(index->fileDescriptor, sequenceBuffer, sequenceSegmentLength,
                          (off_t)(index->sequenceFileOffset + sequenceStartPosition));
  if (r < 0 || (size_t)r != sequenceSegmentLength) return AwFmFileReadFail;
  sequenceBuffer[sequenceSegmentLength] = 0;
  return AwFmFileReadOkay;
}

/* ref src/AwFmFile.c:484-522: one sampled-SA value straight from the file */
enum AwFmReturnCode awfmSaValueFromFile(const struct AwFmIndex *ix, size_t i, size_t *valueOut) {
  const unsigned width = ix->suffixArray.valueBitWidth;
  const uint64_t tailBits = (i % 8) * width;
  const uint64_t byteOffset = (i / 8) * width + tailBits / 8;
  const unsigned bitOffset = (unsigned)(tailBits % 8);
  uint8_t window[16] = {0};
  const size_t want = (bitOffset + width + 7) / 8;
  size_t done = 0;
  while (done < want) {
    const ssize_t r =
        pread(ix->fileDescriptor, window + done, want - done, (off_t)(ix->suffixArrayFileOffset + byteOffset + done));
    if (r <= 0) return AwFmFileReadFail;
    done += (size_t)r;
  }
  unsigned __int128 bitsValue = 0;
  for (size_t b = 0; b < want; b++) bitsValue |= (unsigned __int128)window[b] << (8 * b);
  bitsValue >>= bitOffset;
  *valueOut = width >= 64 ? (size_t)bitsValue : (size_t)(bitsValue & (((unsigned __int128)1 << width) - 1));
  return AwFmSuccess;
}
