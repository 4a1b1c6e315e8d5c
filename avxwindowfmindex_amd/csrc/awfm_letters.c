/* Letter encodings.  The table VALUES are format constants of the .awfmi
 * index and of the BWT bit planes (ref src/AwFmLetter.c); the code is new. */
#include "awfm_internal.h"

/* ref src/AwFmLetter.c:4-22: a0 c1 g2 t/u3, '$'5, everything else 4 */
uint8_t awfmNucAsciiToIndex(uint8_t c) {
  static const uint8_t lowerMap[26] = {/*a*/ 0, 4, /*c*/ 1, 4, 4, 4, /*g*/ 2, 4, 4, 4, 4, 4, 4,
                                       4,       4, 4,       4, 4, 4, /*t*/ 3, /*u*/ 3, 4, 4, 4, 4, 4};
  const uint8_t l = c | 0x20;
  if (l >= 'a' && l <= 'z') return lowerMap[l - 'a'];
  return l == '$' ? 5 : 4;
}

/* ref src/AwFmLetter.c:24-42 */
uint8_t awfmNucSanitize(uint8_t c) {
  const uint8_t l = c | 0x20;
  const uint8_t idx = awfmNucAsciiToIndex(c);
  return idx == 4 ? 'x' : l;
}

/* ref src/AwFmLetter.c:44-47 */
uint8_t awfmNucIndexToCode(uint8_t letterIndex) {
  static const uint8_t code[6] = {6, 5, 3, 1, 2, 4};
  return code[letterIndex];
}

/* ref src/AwFmLetter.c:49-53 */
uint8_t awfmNucCodeToIndex(uint8_t code) {
  static const uint8_t idx[8] = {5, 3, 4, 2, 5, 1, 0, 0};
  return idx[code & 7];
}

/* ref src/AwFmLetter.c:55-67 */
uint8_t awfmAminoAsciiToIndex(uint8_t c) {
  static const uint8_t lut[32] = {20, 0,  20, 1,  2,  3,  4,  5,  6,  7,  20, 8,  9,  10, 11, 20,
                                  12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 20, 20, 20, 20, 20, 20};
  return c == '$' ? 21 : lut[c & 0x1F];
}

/* ref src/AwFmLetter.c:69-79 */
uint8_t awfmAminoSanitize(uint8_t c) {
  const uint8_t l = c | 0x20;
  return (l == 'b' || l == 'x' || c == 0) ? 'z' : c;
}

/* ref src/AwFmLetter.c:81-87 */
uint8_t awfmAminoIndexToCode(uint8_t letterIndex) {
  static const uint8_t code[22] = {0x0C, 0x17, 0x03, 0x06, 0x1E, 0x1A, 0x1B, 0x19, 0x15, 0x1C, 0x1D,
                                   0x08, 0x09, 0x04, 0x13, 0x0A, 0x05, 0x16, 0x01, 0x02, 0x1F, 0x00};
  return code[letterIndex];
}

/* ref src/AwFmLetter.c:89-96 */
uint8_t awfmAminoCodeToIndex(uint8_t code) {
  static const uint8_t idx[32] = {21, 18, 19, 2,  13, 16, 3,  20, 11, 12, 15, 20, 0, 20, 20, 20,
                                  20, 20, 20, 14, 20, 8,  17, 1,  20, 7,  5,  6,  9, 10, 4,  20};
  return idx[code & 31];
}

/* ref src/AwFmLetter.c:98-125 */
bool awfmLetterIsAmbiguous(uint8_t c, enum AwFmAlphabetType alphabet) {
  const uint8_t l = (c >= 'A' && c <= 'Z') ? (uint8_t)(c + ('a' - 'A')) : c;
  if (alphabet == AwFmAlphabetAmino) return l == 'z' || l == 'x' || l == 'b';
  return !(l == 'a' || l == 'c' || l == 'g' || l == 't' || l == 'u');
}

/* Occurrence-vector literals, bit j = plane j (ref src/AwFmOccurrence.c:18-31
 * for nucleotides, :66-128 for amino acids).  Planes in neither mask are
 * don't-care, as in the reference's shortened Boolean forms. */
const uint8_t awfmNucOnes[5] = {0x6, 0x5, 0x3, 0x1, 0x2};
const uint8_t awfmNucZeros[5] = {0x0, 0x0, 0x0, 0x6, 0x5};
const uint8_t awfmAminoOnes[21] = {0x0C, 0x07, 0x03, 0x06, 0x0E, 0x10, 0x0B, 0x10, 0x10, 0x10, 0x0D,
                                   0x08, 0x09, 0x04, 0x10, 0x0A, 0x05, 0x10, 0x01, 0x02, 0x0F};
const uint8_t awfmAminoZeros[21] = {0x10, 0x08, 0x10, 0x10, 0x01, 0x05, 0x04, 0x06, 0x0A, 0x03, 0x02,
                                    0x07, 0x10, 0x0B, 0x0C, 0x10, 0x10, 0x09, 0x0E, 0x0D, 0x00};
