/*
 * awfm_pack.c -- host-side bit-packing of fixed-length k-mers for the flat batch API (include/awfm_gpu.h):
 * one 64-bit word per k-mer, first character most significant; nucleotide 2 bits per character, amino 5 bits =
 * the letter index of ref src/AwFmLetter.c:55-67.
 */
#include "awfm_internal.h"

enum AwFmReturnCode awfmPackKmers(enum AwFmAlphabetType alphabet, const uint8_t *chars, uint32_t kmerLength,
                                  uint64_t numKmers, uint64_t *packedOut, uint64_t *firstUnpackable) {
  if ((!chars || !packedOut) && numKmers) return AwFmNullPtrError;
  const bool amino = alphabet == AwFmAlphabetAmino;
  if (kmerLength == 0 || kmerLength > (amino ? 12u : 32u)) return AwFmIllegalPositionError;
  for (uint64_t j = 0; j < numKmers; j++) {
    const uint8_t *k = chars + j * kmerLength;
    uint64_t w = 0;
    for (uint32_t c = 0; c < kmerLength; c++) {
      const uint8_t a = amino ? awfmAminoAsciiToIndex(k[c]) : awfmNucAsciiToIndex(k[c]);
      if (a >= (amino ? 20u : 4u)) { /* ambiguity character, sentinel, not a letter */
        if (firstUnpackable) *firstUnpackable = j;
        return AwFmIllegalPositionError;
      }
      w = (w << (amino ? 5 : 2)) | a;
    }
    packedOut[j] = w;
  }
  return AwFmSuccess;
}
