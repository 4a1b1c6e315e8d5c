/* Bit-packed sampled suffix array codec (ref src/AwFmSuffixArray.c). */
#include <string.h>
#include "awfm_internal.h"

/* ref src/AwFmSuffixArray.c:12-18; the reference evaluates clz(0) for a one-entry suffix array (empty text),
 * which is undefined: one bit is used there */
uint8_t awfmSaWidth(uint64_t saLength) { return saLength <= 1 ? 1 : (uint8_t)(64 - __builtin_clzll(saLength - 1)); }

/* ref src/AwFmSuffixArray.c:144-147 */
uint64_t awfmSaSampleCount(uint64_t bwtLength, uint64_t ratio) { return (bwtLength + ratio - 1) / ratio; }

/* ref src/AwFmSuffixArray.c:41-53: bytes of the bit stream, rounded up, + 8 pad */
uint64_t awfmSaPackedBytes(uint64_t saLength, uint8_t ratio) {
  const unsigned __int128 bits = (unsigned __int128)awfmSaSampleCount(saLength, ratio) * awfmSaWidth(saLength);
  return (uint64_t)((bits + 7) / 8) + AWFM_SA_PAD_BYTES;
}

/* ref src/AwFmSuffixArray.c:58-112: sample i = SA[i*ratio] at bit i*width of a
 * little-endian bit stream.  (The reference packs in place inside the SA
 * buffer, so its 8 pad bytes hold leftovers of the unpacked array; here they
 * are zero.  No reader looks at them beyond masking.) */
void awfmSaPack(const uint64_t *fullSa, uint64_t saLength, uint8_t ratio, uint8_t *out) {
  const unsigned width = awfmSaWidth(saLength);
  const uint64_t samples = awfmSaSampleCount(saLength, ratio);
  memset(out, 0, awfmSaPackedBytes(saLength, ratio));
  unsigned __int128 acc = 0; /* bits not yet flushed, LSB first */
  unsigned accBits = 0;
  uint64_t byte = 0;
  for (uint64_t i = 0; i < samples; i++) {
    acc |= (unsigned __int128)fullSa[i * ratio] << accBits;
    accBits += width;
    while (accBits >= 8) {
      out[byte++] = (uint8_t)acc;
      acc >>= 8;
      accBits -= 8;
    }
  }
  if (accBits) out[byte] = (uint8_t)acc;
}

/* ref src/AwFmSuffixArray.c:114-142 with :22-39 folded in */
uint64_t awfmSaGet(const uint8_t *values, uint8_t width, uint64_t i) {
  const uint64_t tailBits = (i % 8) * width;
  const uint64_t byteOffset = (i / 8) * width + tailBits / 8;
  const unsigned bitOffset = (unsigned)(tailBits % 8);
  uint64_t window;
  memcpy(&window, values + byteOffset, 8);
  window >>= bitOffset;
  if (width > 57 && bitOffset) window |= (uint64_t)values[byteOffset + 8] << (64 - bitOffset);
  return width >= 64 ? window : window & ((1ULL << width) - 1);
}
