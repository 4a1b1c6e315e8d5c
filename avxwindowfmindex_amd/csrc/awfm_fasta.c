/*
 * Multi-FASTA support: awFmCreateIndexFromFasta and the sequence-number / header lookups.
 *
 * The reference delegates this to the FastaVector library (git submodule, empty in the mount), so the
 * record bookkeeping here is written from the call sites and tests only
 * (ref src/AwFmCreate.c:140-279, src/AwFmSearch.c:284-315, src/AwFmFile.c:157-187, :360-440,
 *  test/multiSequenceIndexTest/AwFmMultiSequenceTest.c:627-753):
 *   - the indexed text is the records' residues concatenated with one NUL terminator after every record
 *     (the sanitisers turn it into the ambiguity letter, so no k-mer can match across two records);
 *   - per record: headerEndPosition = end of its header in the concatenated header string,
 *     sequenceEndPosition = end of its residues (terminator excluded) in the concatenated text;
 *   - a global position maps to the record whose [start, sequenceEndPosition) contains it, where
 *     start = previous sequenceEndPosition + 1.
 * The .awfmi trailer keeps the reference's layout: header length, record count (size_t each), header
 * characters, then {headerEndPosition, sequenceEndPosition} pairs.
 */
#include <stdlib.h>
#include <string.h>
#include "awfm_internal.h"

static void fastaFree(struct FastaVector *fv) {
  if (!fv) return;
  free(fv->headers);
  free(fv->records);
  free(fv);
}

void awfmFastaVectorFree(struct FastaVector *fv) { fastaFree(fv); }

/* reads a whole file; caller frees */
static char *slurp(const char *path, size_t *lengthOut) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  size_t cap = 1 << 16, len = 0;
  char *buf = malloc(cap);
  while (buf) {
    const size_t got = fread(buf + len, 1, cap - len, f);
    len += got;
    if (got == 0) break;
    if (len == cap) {
      cap *= 2;
      char *grown = realloc(buf, cap);
      if (!grown) {
        free(buf);
        buf = NULL;
      } else {
        buf = grown;
      }
    }
  }
  fclose(f);
  *lengthOut = len;
  return buf;
}

/* parses FASTA text into the concatenated residue buffer (NUL after every record) and the record table */
static struct FastaVector *parseFasta(const char *data, size_t length, uint8_t **textOut, size_t *textLengthOut) {
  struct FastaVector *fv = calloc(1, sizeof *fv);
  uint8_t *text = malloc(length + 2);
  char *headers = malloc(length + 1);
  size_t recordCap = 16;
  struct AwfmFastaRecord *records = malloc(recordCap * sizeof *records);
  if (!fv || !text || !headers || !records) {
    free(fv);
    free(text);
    free(headers);
    free(records);
    return NULL;
  }
  size_t textLen = 0, headerLen = 0, numRecords = 0, i = 0;
  bool open = false;
  while (i < length) {
    size_t lineEnd = i;
    while (lineEnd < length && data[lineEnd] != '\n') lineEnd++;
    size_t contentEnd = lineEnd;
    while (contentEnd > i && (data[contentEnd - 1] == '\r' || data[contentEnd - 1] == ' ' || data[contentEnd - 1] == '\t'))
      contentEnd--;
    if (contentEnd > i && data[i] == '>') {
      if (open) { /* close the previous record */
        records[numRecords - 1].sequenceEndPosition = textLen;
        text[textLen++] = 0;
      }
      if (numRecords == recordCap) {
        recordCap *= 2;
        struct AwfmFastaRecord *grown = realloc(records, recordCap * sizeof *records);
        if (!grown) break;
        records = grown;
      }
      memcpy(headers + headerLen, data + i + 1, contentEnd - i - 1);
      headerLen += contentEnd - i - 1;
      records[numRecords].headerEndPosition = headerLen;
      records[numRecords].sequenceEndPosition = textLen;
      numRecords++;
      open = true;
    } else if (open) {
      for (size_t c = i; c < contentEnd; c++)
        if (data[c] != ' ' && data[c] != '\t') text[textLen++] = (uint8_t)data[c];
    }
    i = lineEnd + 1;
  }
  if (open) {
    records[numRecords - 1].sequenceEndPosition = textLen;
    text[textLen++] = 0;
  }
  fv->headers = headers;
  fv->headerLength = headerLen;
  fv->records = records;
  fv->numRecords = numRecords;
  *textOut = text;
  *textLengthOut = textLen;
  return fv;
}

/* ref src/AwFmCreate.c:140-279 */
enum AwFmReturnCode awFmCreateIndexFromFasta(struct AwFmIndex *_RESTRICT_ *index,
                                             struct AwFmIndexConfiguration *_RESTRICT_ const config,
                                             const char *fastaSrc, const char *_RESTRICT_ const indexFileSrc) {
  if (!config || !fastaSrc || !indexFileSrc) return AwFmNullPtrError;
  *index = NULL;
  size_t fileLength = 0;
  char *data = slurp(fastaSrc, &fileLength);
  if (!data) return AwFmFileOpenFail;
  uint8_t *text = NULL;
  size_t textLength = 0;
  struct FastaVector *fv = parseFasta(data, fileLength, &text, &textLength);
  free(data);
  if (!fv) return AwFmAllocationFailure;
  /* build exactly like awFmCreateIndex, then attach the record table and rewrite the file with its trailer */
  struct AwFmIndex *ix = NULL;
  enum AwFmReturnCode rc = awfmCreateIndexWithFasta(&ix, config, text, textLength, indexFileSrc, fv);
  free(text);
  if (awFmReturnCodeIsFailure(rc) || !ix) {
    if (ix) awFmDeallocIndex(ix); /* its fastaVector pointer was cleared on failure */
    fastaFree(fv);
    return rc;
  }
  *index = ix;
  return rc;
}

/* ref src/AwFmSearch.c:284-301 */
enum AwFmReturnCode awFmGetLocalSequencePositionFromIndexPosition(const struct AwFmIndex *_RESTRICT_ const index,
                                                                  size_t globalPosition, size_t *sequenceNumber,
                                                                  size_t *localSequencePosition) {
  const struct FastaVector *fv = index->fastaVector;
  if (!fv) return AwFmUnsupportedVersionError;
  size_t lo = 0, hi = fv->numRecords; /* first record whose end is beyond the position */
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (fv->records[mid].sequenceEndPosition > globalPosition)
      hi = mid;
    else
      lo = mid + 1;
  }
  if (lo == fv->numRecords) return AwFmIllegalPositionError;
  const size_t start = lo == 0 ? 0 : fv->records[lo - 1].sequenceEndPosition + 1;
  if (globalPosition < start) return AwFmIllegalPositionError; /* a terminator between two records */
  *sequenceNumber = lo;
  *localSequencePosition = globalPosition - start;
  return AwFmSuccess;
}

/* ref src/AwFmSearch.c:303-315: *headerBuffer points into the index (not NUL terminated) */
enum AwFmReturnCode awFmGetHeaderStringFromSequenceNumber(const struct AwFmIndex *_RESTRICT_ const index,
                                                          size_t sequenceNumber, char **headerBuffer,
                                                          size_t *headerLength) {
  const struct FastaVector *fv = index->fastaVector;
  if (!fv || (index->featureFlags & (1u << AWFM_FEATURE_BIT_FASTA_VECTOR)) == 0) return AwFmUnsupportedVersionError;
  if (sequenceNumber >= fv->numRecords) return AwFmIllegalPositionError;
  const size_t start = sequenceNumber == 0 ? 0 : fv->records[sequenceNumber - 1].headerEndPosition;
  *headerBuffer = fv->headers + start;
  *headerLength = fv->records[sequenceNumber].headerEndPosition - start;
  return AwFmSuccess;
}

/* ref src/AwFmIndexStruct.c:149-155 */
uint32_t awFmGetNumSequences(const struct AwFmIndex *_RESTRICT_ const index) {
  return index->fastaVector ? (uint32_t)index->fastaVector->numRecords : 1u;
}
