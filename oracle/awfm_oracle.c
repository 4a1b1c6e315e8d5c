/*
 * awfm_oracle.c -- CPU restatement of the AvxWindowFmIndex search path.
 * TEST INFRASTRUCTURE ONLY (see awfm_oracle.h for the pinning statement).
 * Plain scalar C; citations are file:line under /root/reference.
 */
#include "awfm_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ letters */

/* src/AwFmLetter.c:4-22 */
uint8_t orc_nuc_ascii_to_index(uint8_t c) {
  switch (c | 0x20) {
  case 'a': return 0;
  case 'c': return 1;
  case 'g': return 2;
  case 't': return 3;
  case 'u': return 3;
  case '$': return 5; /* '$' == 0x24 already has bit 0x20 set */
  default: return 4;
  }
}

/* src/AwFmLetter.c:24-42 */
uint8_t orc_nuc_sanitize(uint8_t c) {
  const uint8_t l = c | 0x20;
  if (l == 'a' || l == 'c' || l == 'g' || l == 't' || l == 'u' || l == '$') return l;
  return 'x';
}

/* src/AwFmLetter.c:44-47 */
uint8_t orc_nuc_index_to_code(uint8_t letterIndex) {
  static const uint8_t code[6] = {6, 5, 3, 1, 2, 4};
  return code[letterIndex];
}

/* src/AwFmLetter.c:49-53 */
uint8_t orc_nuc_code_to_index(uint8_t code) {
  static const uint8_t idx[8] = {5, 3, 4, 2, 5, 1, 0, 0 /* code 7 never stored */};
  return idx[code & 7];
}

/* src/AwFmLetter.c:55-67 */
uint8_t orc_amino_ascii_to_index(uint8_t c) {
  static const uint8_t lut[32] = {20, 0,  20, 1,  2,  3,  4,  5,  6,  7,  20, 8,  9,  10, 11, 20,
                                  12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 20, 20, 20, 20, 20, 20};
  if (c == '$') return 21;
  return lut[c & 0x1F];
}

/* src/AwFmLetter.c:69-79 */
uint8_t orc_amino_sanitize(uint8_t c) {
  const uint8_t l = c | 0x20;
  if (l == 'b' || l == 'x' || c == 0) return 'z';
  return c;
}

/* src/AwFmLetter.c:81-87 */
uint8_t orc_amino_index_to_code(uint8_t letterIndex) {
  static const uint8_t code[22] = {0x0C, 0x17, 0x03, 0x06, 0x1E, 0x1A, 0x1B, 0x19, 0x15, 0x1C, 0x1D,
                                   0x08, 0x09, 0x04, 0x13, 0x0A, 0x05, 0x16, 0x01, 0x02, 0x1F, 0x00};
  return code[letterIndex];
}

/* src/AwFmLetter.c:89-96 */
uint8_t orc_amino_code_to_index(uint8_t code) {
  static const uint8_t idx[32] = {21, 18, 19, 2,  13, 16, 3,  20, 11, 12, 15, 20, 0, 20, 20, 20,
                                  20, 20, 20, 14, 20, 8,  17, 1,  20, 7,  5,  6,  9, 10, 4,  20};
  return idx[code & 31];
}

/* src/AwFmLetter.c:98-125 (tolower on ASCII letters == |0x20 for 'A'..'Z') */
int orc_letter_is_ambiguous(uint8_t c, uint8_t alphabet) {
  const uint8_t l = (c >= 'A' && c <= 'Z') ? (uint8_t)(c | 0x20) : c;
  if (alphabet == ORC_ALPHABET_AMINO) return l == 'z' || l == 'x' || l == 'b';
  return !(l == 'a' || l == 'c' || l == 'g' || l == 't' || l == 'u');
}

static inline uint8_t ascii_to_index(const OrcIndex *ix, uint8_t c) {
  return ix->alphabet == ORC_ALPHABET_AMINO ? orc_amino_ascii_to_index(c) : orc_nuc_ascii_to_index(c);
}
static inline int num_planes(const OrcIndex *ix) { return ix->alphabet == ORC_ALPHABET_AMINO ? 5 : 3; }
static inline int cardinality(uint8_t alphabet) { return alphabet == ORC_ALPHABET_AMINO ? 20 : 4; }
static inline uint8_t sentinel_index(const OrcIndex *ix) { return ix->alphabet == ORC_ALPHABET_AMINO ? 21 : 5; }

/* --------------------------------------------------------- masked popcount */

/* src/AwFmSimdConfig.c:89-114: number of set bits among bit positions 0..p
 * (inclusive) of a 256-bit vector stored LSB-first. */
uint32_t orc_masked_popcount(const uint8_t vec[32], uint8_t p) {
  uint32_t total = 0;
  for (unsigned bit = 0; bit <= p; bit++) total += (vec[bit >> 3] >> (bit & 7)) & 1u;
  return total;
}

/* ------------------------------------------------------ occurrence vectors */

/* Every case of src/AwFmOccurrence.c:8-36 (nucleotide) and :52-135 (amino) is a
 * conjunction of plane literals; the tables hold, per letter, the planes that
 * must be 1 and the planes that must be 0 (planes not listed are don't-care,
 * exactly as in the reference's shortened Boolean forms). */
static const uint8_t NUC_ONES[5] = {0x6, 0x5, 0x3, 0x1, 0x2};
static const uint8_t NUC_ZEROS[5] = {0x0, 0x0, 0x0, 0x6, 0x5};
static const uint8_t AMINO_ONES[21] = {0x0C, 0x07, 0x03, 0x06, 0x0E, 0x10, 0x0B, 0x10, 0x10, 0x10, 0x0D,
                                       0x08, 0x09, 0x04, 0x10, 0x0A, 0x05, 0x10, 0x01, 0x02, 0x0F};
static const uint8_t AMINO_ZEROS[21] = {0x10, 0x08, 0x10, 0x10, 0x01, 0x05, 0x04, 0x06, 0x0A, 0x03, 0x02,
                                        0x07, 0x10, 0x0B, 0x0C, 0x10, 0x10, 0x09, 0x0E, 0x0D, 0x00};

static inline const uint8_t *block_ptr(const OrcIndex *ix, uint64_t block) {
  return ix->blocks + block * (uint64_t)ix->blockBytes;
}
static inline uint64_t block_base(const OrcIndex *ix, uint64_t block, uint8_t letter) {
  uint64_t v;
  memcpy(&v, block_ptr(ix, block) + 32 * num_planes(ix) + 8 * (size_t)letter, 8);
  return v;
}

/* word-wise form of orc_masked_popcount(occVec(letter), p); the byte-wise
 * definition above is what tests pin against the known answers, and the unit
 * tests check the two against each other. */
static inline uint32_t block_rank(const OrcIndex *ix, const uint8_t *blk, uint8_t letter, uint8_t p) {
  const int np = num_planes(ix);
  const uint8_t ones = np == 5 ? AMINO_ONES[letter] : NUC_ONES[letter];
  const uint8_t zeros = np == 5 ? AMINO_ZEROS[letter] : NUC_ZEROS[letter];
  uint32_t total = 0;
  const unsigned lastWord = p >> 6;
  for (unsigned w = 0; w <= lastWord; w++) {
    uint64_t acc = ~0ULL;
    for (int j = 0; j < np; j++) {
      uint64_t plane;
      memcpy(&plane, blk + 32 * j + 8 * w, 8);
      if ((ones >> j) & 1) acc &= plane;
      if ((zeros >> j) & 1) acc &= ~plane;
    }
    if (w == lastWord) acc &= ~0ULL >> (63 - (p & 63));
    total += (uint32_t)__builtin_popcountll(acc);
  }
  return total;
}

/* exported for unit tests: the 32-byte occurrence vector itself */
void orc_occ_vector(const OrcIndex *ix, uint64_t block, uint8_t letter, uint8_t out[32]) {
  const int np = num_planes(ix);
  const uint8_t ones = np == 5 ? AMINO_ONES[letter] : NUC_ONES[letter];
  const uint8_t zeros = np == 5 ? AMINO_ZEROS[letter] : NUC_ZEROS[letter];
  const uint8_t *blk = block_ptr(ix, block);
  for (int b = 0; b < 32; b++) {
    uint8_t acc = 0xFF;
    for (int j = 0; j < np; j++) {
      if ((ones >> j) & 1) acc &= blk[32 * j + b];
      if ((zeros >> j) & 1) acc &= (uint8_t)~blk[32 * j + b];
    }
    out[b] = acc;
  }
}

/* Occ(a,q) = base[q/256][a] + popcount(occVec(a) & bits[0..q%256])
 * src/AwFmSearch.c:48-63, src/AwFmIndexStruct.c:117-124 */
uint64_t orc_occ(const OrcIndex *ix, uint8_t letter, uint64_t q) {
  const uint64_t block = q / ORC_POSITIONS_PER_BLOCK;
  const uint8_t p = (uint8_t)(q % ORC_POSITIONS_PER_BLOCK);
  return block_base(ix, block, letter) + block_rank(ix, block_ptr(ix, block), letter, p);
}

/* src/AwFmSearch.c:42-103 (nucleotide), :105-159 (amino): no validity check */
void orc_step(const OrcIndex *ix, uint64_t *sp, uint64_t *ep, uint8_t letter) {
  const uint64_t c = ix->prefixSums[letter];
  const uint64_t newSp = c + orc_occ(ix, letter, *sp - 1);
  const uint64_t newEp = c + orc_occ(ix, letter, *ep) - 1;
  *sp = newSp;
  *ep = newEp;
}

/* src/AwFmOccurrence.c:170-184, :202-217 */
uint8_t orc_letter_at(const OrcIndex *ix, uint64_t p) {
  const uint8_t *blk = block_ptr(ix, p / ORC_POSITIONS_PER_BLOCK);
  const unsigned local = (unsigned)(p % ORC_POSITIONS_PER_BLOCK);
  const unsigned byte = local / 8, bit = local % 8;
  unsigned code = 0;
  for (int j = 0; j < num_planes(ix); j++) code |= ((blk[32 * j + byte] >> bit) & 1u) << j;
  return ix->alphabet == ORC_ALPHABET_AMINO ? orc_amino_code_to_index((uint8_t)code)
                                            : orc_nuc_code_to_index((uint8_t)code);
}

/* src/AwFmSearch.c:369-397, :399-427 */
uint64_t orc_lf(const OrcIndex *ix, uint64_t p) {
  const uint8_t letter = orc_letter_at(ix, p);
  if (letter == sentinel_index(ix)) return 0;
  return ix->prefixSums[letter] + orc_occ(ix, letter, p) - 1;
}

/* src/AwFmIndexStruct.c:126-130 */
uint64_t orc_range_length(uint64_t sp, uint64_t ep) { return sp <= ep ? ep - sp + 1 : 0; }

/* src/AwFmSearch.c:317-358 */
void orc_range_for_string(const OrcIndex *ix, const char *kmer, uint64_t len, uint64_t *sp, uint64_t *ep) {
  uint64_t pos = len - 1;
  uint8_t letter = ascii_to_index(ix, (uint8_t)kmer[pos]);
  *sp = ix->prefixSums[letter];
  *ep = ix->prefixSums[letter + 1] - 1;
  while (*sp <= *ep && pos-- != 0) {
    letter = ascii_to_index(ix, (uint8_t)kmer[pos]);
    orc_step(ix, sp, ep, letter);
  }
}

/* ------------------------------------------------------------- sampled SA */

/* src/AwFmSuffixArray.c:12-18: width = 64 - clz(saLength-1) */
uint8_t orc_sa_width(uint64_t saLength) {
  return saLength <= 1 ? 1 : (uint8_t)(64 - __builtin_clzll(saLength - 1)); /* clz(0) is undefined: one bit */
}

/* src/AwFmSuffixArray.c:144-147 */
uint64_t orc_sa_num_samples(uint64_t bwtLength, uint64_t ratio) { return (bwtLength + ratio - 1) / ratio; }

/* src/AwFmSuffixArray.c:22-39 */
static inline void sa_offset(uint8_t width, uint64_t i, uint64_t *byteOff, unsigned *bitOff) {
  const uint64_t endingBits = (i % 8) * width;
  *byteOff = (i / 8) * width + endingBits / 8;
  *bitOff = (unsigned)(endingBits % 8);
}

/* src/AwFmSuffixArray.c:41-53 (+8 pad, :9) */
uint64_t orc_sa_packed_bytes(uint64_t saLength, uint8_t ratio) {
  uint64_t byteOff;
  unsigned bitOff;
  sa_offset(orc_sa_width(saLength), orc_sa_num_samples(saLength, ratio), &byteOff, &bitOff);
  return byteOff + (bitOff ? 1 : 0) + 8;
}

/* src/AwFmSuffixArray.c:58-112: little-endian bit stream of samples SA[i*ratio] */
void orc_sa_pack(const uint64_t *fullSa, uint64_t saLength, uint8_t ratio, uint8_t *out) {
  const uint8_t width = orc_sa_width(saLength);
  const uint64_t samples = orc_sa_num_samples(saLength, ratio);
  memset(out, 0, orc_sa_packed_bytes(saLength, ratio));
  for (uint64_t i = 0; i < samples; i++) {
    const uint64_t v = fullSa[i * ratio];
    const uint64_t firstBit = i * (uint64_t)width;
    for (unsigned b = 0; b < width; b++)
      if ((v >> b) & 1) out[(firstBit + b) >> 3] |= (uint8_t)(1u << ((firstBit + b) & 7));
  }
}

/* src/AwFmSuffixArray.c:114-142 */
uint64_t orc_sa_get(const uint8_t *values, uint8_t width, uint64_t i) {
  uint64_t byteOff, buffer;
  unsigned bitOff;
  sa_offset(width, i, &byteOff, &bitOff);
  memcpy(&buffer, values + byteOff, 8);
  buffer >>= bitOff;
  if (width > 57) {
    uint64_t last = values[byteOff + 8];
    last <<= 1;
    last <<= (63 - bitOff);
    buffer |= last;
  }
  const uint64_t mask = width >= 64 ? ~0ULL : ((1ULL << width) - 1);
  return buffer & mask;
}

/* src/AwFmParallelSearch.c:338-361 + src/AwFmSuffixArray.c:179-191 */
uint64_t orc_locate_one(const OrcIndex *ix, uint64_t p, uint64_t *lfStepsOut) {
  uint64_t offset = 0;
  while (p % ix->saRatio != 0) { /* src/AwFmIndexStruct.c:88-91 */
    p = orc_lf(ix, p);
    offset++;
  }
  if (lfStepsOut) *lfStepsOut += offset;
  return (orc_sa_get(ix->sa, ix->saWidth, p / ix->saRatio) + offset) % ix->bwtLength;
}

/* ------------------------------------------------------------ index build */

/* Suffix array by prefix doubling (the reference calls divsufsort64,
 * src/AwFmCreate.c:99-100; the SA of a text is unique so any correct
 * construction is a restatement).  Not re-entrant (qsort comparator state). */
static const uint64_t *g_rank;
static uint64_t g_h, g_n;
static int cmp_doubling(const void *a, const void *b) {
  const uint64_t i = *(const uint64_t *)a, j = *(const uint64_t *)b;
  if (g_rank[i] != g_rank[j]) return g_rank[i] < g_rank[j] ? -1 : 1;
  const uint64_t ri = i + g_h < g_n ? g_rank[i + g_h] + 1 : 0;
  const uint64_t rj = j + g_h < g_n ? g_rank[j + g_h] + 1 : 0;
  return ri < rj ? -1 : (ri > rj ? 1 : 0);
}

void orc_suffix_array(const uint8_t *text, uint64_t n, uint64_t *sa) {
  uint64_t *rank = malloc(n * sizeof(uint64_t)), *tmp = malloc(n * sizeof(uint64_t));
  for (uint64_t i = 0; i < n; i++) {
    sa[i] = i;
    rank[i] = text[i];
  }
  for (uint64_t h = 1;; h *= 2) {
    g_rank = rank;
    g_h = h;
    g_n = n;
    /* first round sorts by (text[i], text[i+1]) */
    qsort(sa, n, sizeof(uint64_t), cmp_doubling);
    tmp[sa[0]] = 0;
    for (uint64_t i = 1; i < n; i++) tmp[sa[i]] = tmp[sa[i - 1]] + (cmp_doubling(&sa[i - 1], &sa[i]) != 0);
    memcpy(rank, tmp, n * sizeof(uint64_t));
    if (rank[sa[n - 1]] == n - 1) break;
  }
  free(rank);
  free(tmp);
}

/* src/AwFmCreate.c:407-450: DFS that prepends letters with NO validity check */
static void seed_dfs(OrcIndex *ix, uint64_t sp, uint64_t ep, unsigned curLen, uint64_t curIndex, uint64_t mult) {
  if (curLen == ix->seedK) {
    ix->seedTable[2 * curIndex] = sp;
    ix->seedTable[2 * curIndex + 1] = ep;
    return;
  }
  const int card = cardinality(ix->alphabet);
  for (int a = 0; a < card; a++) {
    uint64_t nsp = sp, nep = ep;
    orc_step(ix, &nsp, &nep, (uint8_t)a);
    seed_dfs(ix, nsp, nep, curLen + 1, curIndex + (uint64_t)a * mult, mult * (uint64_t)card);
  }
}

static uint64_t ipow(uint64_t b, unsigned e) {
  uint64_t r = 1;
  while (e--) r *= b;
  return r;
}

/* src/AwFmCreate.c:31-137, :281-405 */
OrcIndex *orc_build(const uint8_t *text, uint64_t n, uint8_t alphabet, uint8_t saRatio, uint8_t seedK) {
  OrcIndex *ix = calloc(1, sizeof(OrcIndex));
  const int amino = alphabet == ORC_ALPHABET_AMINO;
  ix->alphabet = alphabet;
  ix->saRatio = saRatio;
  ix->seedK = seedK;
  ix->ownsArrays = 1;
  ix->bwtLength = n + 1;
  ix->blockBytes = amino ? ORC_AMINO_BLOCK_BYTES : ORC_NUC_BLOCK_BYTES;
  ix->numBlocks = 1 + (ix->bwtLength - 1) / ORC_POSITIONS_PER_BLOCK; /* src/AwFmIndexStruct.c:104-106 */

  /* sanitize + '$' (src/AwFmCreate.c:62-66, :452-466) */
  uint8_t *s = malloc(n + 1);
  for (uint64_t i = 0; i < n; i++) s[i] = amino ? orc_amino_sanitize(text[i]) : orc_nuc_sanitize(text[i]);
  s[n] = '$';

  ix->fullSa = malloc((n + 1) * sizeof(uint64_t));
  orc_suffix_array(s, n + 1, ix->fullSa);

  /* BWT bit planes + running counts copied at the head of each block
   * (src/AwFmCreate.c:291-336, :350-395) */
  ix->blocks = calloc(ix->numBlocks, ix->blockBytes);
  const int np = amino ? 5 : 3;
  const int counters = amino ? 24 : 8;
  uint64_t running[24] = {0};
  for (uint64_t i = 0; i < ix->bwtLength; i++) {
    uint8_t *blk = ix->blocks + (i / 256) * (uint64_t)ix->blockBytes;
    const unsigned local = (unsigned)(i % 256);
    if (local == 0) memcpy(blk + 32 * np, running, counters * sizeof(uint64_t));
    const uint64_t textPos = ix->fullSa[i];
    uint8_t letter, code;
    if (textPos == 0) {
      letter = amino ? 21 : 5;
    } else {
      letter = amino ? orc_amino_ascii_to_index(s[textPos - 1]) : orc_nuc_ascii_to_index(s[textPos - 1]);
    }
    code = amino ? orc_amino_index_to_code(letter) : orc_nuc_index_to_code(letter);
    running[letter]++;
    for (int j = 0; j < np; j++) blk[32 * j + local / 8] |= (uint8_t)(((code >> j) & 1u) << (local % 8));
  }
  /* prefix sums (src/AwFmCreate.c:338-344, :397-403): [0]=1, [i]=1+sum_{j<i} */
  const int card = cardinality(alphabet);
  ix->prefixSums[0] = 1;
  for (int i = 1; i < card + 2; i++) ix->prefixSums[i] = ix->prefixSums[i - 1] + running[i - 1];
  free(s);

  /* seed table (src/AwFmCreate.c:407-417) */
  ix->seedLen = ipow((uint64_t)card, seedK);
  ix->seedTable = malloc(ix->seedLen * 16);
  for (int a = 0; a < card; a++)
    seed_dfs(ix, ix->prefixSums[a], ix->prefixSums[a + 1] - 1, 1, (uint64_t)a, (uint64_t)card);

  /* sampled SA (src/AwFmSuffixArray.c:58-112) */
  ix->saWidth = orc_sa_width(ix->bwtLength);
  ix->saBytes = orc_sa_packed_bytes(ix->bwtLength, saRatio);
  ix->sa = malloc(ix->saBytes);
  orc_sa_pack(ix->fullSa, ix->bwtLength, saRatio, ix->sa);
  return ix;
}

OrcIndex *orc_wrap(uint8_t alphabet, uint8_t saRatio, uint8_t seedK, uint64_t bwtLength, uint8_t *blocks,
                   const uint64_t *prefixSums, uint64_t *seedTable, uint8_t *sa) {
  OrcIndex *ix = calloc(1, sizeof(OrcIndex));
  const int amino = alphabet == ORC_ALPHABET_AMINO;
  ix->alphabet = alphabet;
  ix->saRatio = saRatio;
  ix->seedK = seedK;
  ix->bwtLength = bwtLength;
  ix->blockBytes = amino ? ORC_AMINO_BLOCK_BYTES : ORC_NUC_BLOCK_BYTES;
  ix->numBlocks = 1 + (bwtLength - 1) / ORC_POSITIONS_PER_BLOCK;
  ix->blocks = blocks;
  memcpy(ix->prefixSums, prefixSums, (size_t)(cardinality(alphabet) + 2) * 8);
  ix->seedLen = ipow((uint64_t)cardinality(alphabet), seedK);
  ix->seedTable = seedTable;
  ix->saWidth = orc_sa_width(bwtLength);
  ix->saBytes = orc_sa_packed_bytes(bwtLength, saRatio);
  ix->sa = sa;
  return ix;
}

void orc_free(OrcIndex *ix) {
  if (!ix) return;
  if (ix->ownsArrays) {
    free(ix->blocks);
    free(ix->seedTable);
    free(ix->sa);
    free(ix->fullSa);
  }
  free(ix);
}

/* ------------------------------------------------------------ batch search */

#define ORC_CONCURRENT 8 /* AW_FM_NUM_CONCURRENT_QUERIES, src/AwFmIndex.h:16-18 */

static inline void prefetch_block(const OrcIndex *ix, uint64_t q) {
  const uint8_t *p = block_ptr(ix, q / ORC_POSITIONS_PER_BLOCK);
  for (uint32_t off = 0; off < ix->blockBytes; off += 64) __builtin_prefetch(p + off, 0, 0);
}

static inline void tally_step(OrcTally *t, uint64_t sp, uint64_t ep) {
  t->steps++;
  t->blocks += ((sp - 1) / ORC_POSITIONS_PER_BLOCK == ep / ORC_POSITIONS_PER_BLOCK) ? 1 : 2;
}

/* src/AwFmParallelSearch.c:222-271 for one query */
static void seed_one(const OrcIndex *ix, const char *kmer, uint64_t len, uint64_t *sp, uint64_t *ep, OrcTally *t) {
  const uint64_t k = ix->seedK;
  int useTable = len >= k; /* src/AwFmKmerTable.c:4-19 */
  if (useTable)
    for (uint64_t i = len - k; i < len; i++)
      if (orc_letter_is_ambiguous((uint8_t)kmer[i], ix->alphabet)) {
        useTable = 0;
        break;
      }
  if (useTable) { /* src/AwFmKmerTable.c:21-51 */
    const uint64_t card = (uint64_t)cardinality(ix->alphabet);
    uint64_t index = 0;
    for (uint64_t i = len - k; i < len; i++) index = index * card + ascii_to_index(ix, (uint8_t)kmer[i]);
    *sp = ix->seedTable[2 * index];
    *ep = ix->seedTable[2 * index + 1];
    t->seeded++;
    return;
  }
  /* src/AwFmSearch.c:485-520 on the last min(len,k) characters */
  const uint64_t start = len < k ? 0 : len - k;
  const uint64_t sub = len < k ? len : k;
  const char *tail = kmer + start;
  uint64_t pos = sub - 1;
  uint8_t letter = ascii_to_index(ix, (uint8_t)tail[pos]);
  *sp = ix->prefixSums[letter];
  *ep = ix->prefixSums[letter + 1] - 1;
  while (pos-- != 0 && *sp <= *ep) {
    tally_step(t, *sp, *ep);
    orc_step(ix, sp, ep, ascii_to_index(ix, (uint8_t)tail[pos]));
  }
}

/* one block of up to 8 queries: seeds, then lock-step extension
 * (src/AwFmParallelSearch.c:273-313) */
static void search_block(const OrcIndex *ix, const char *chars, const uint64_t *offsets, uint64_t i0, uint64_t i1,
                         uint64_t *sp, uint64_t *ep, OrcTally *t) {
  for (uint64_t i = i0; i < i1; i++) {
    const uint64_t len = offsets[i + 1] - offsets[i];
    t->queries++;
    t->chars += len;
    if (len == 0) { /* documented UB in the reference (src/AwFmIndex.h:348-352); defined here as an empty range */
      sp[i] = 1;
      ep[i] = 0;
      continue;
    }
    seed_one(ix, chars + offsets[i], len, &sp[i], &ep[i], t);
    if (sp[i] <= ep[i]) {
      prefetch_block(ix, sp[i] - 1);
      prefetch_block(ix, ep[i]);
    }
  }
  uint64_t cur = ix->seedK;
  int active = 1;
  while (active) {
    cur++;
    active = 0;
    for (uint64_t i = i0; i < i1; i++) {
      const uint64_t len = offsets[i + 1] - offsets[i];
      if (len >= cur && sp[i] <= ep[i]) {
        active = 1;
        const uint8_t letter = ascii_to_index(ix, (uint8_t)chars[offsets[i] + (len - cur)]);
        tally_step(t, sp[i], ep[i]);
        orc_step(ix, &sp[i], &ep[i], letter);
        prefetch_block(ix, sp[i] - 1);
        prefetch_block(ix, ep[i]);
      }
    }
  }
}

static void tally_add(OrcTally *dst, const OrcTally *src) {
  dst->queries += src->queries;
  dst->seeded += src->seeded;
  dst->steps += src->steps;
  dst->blocks += src->blocks;
  dst->hits += src->hits;
  dst->lfSteps += src->lfSteps;
  dst->chars += src->chars;
}

/* src/AwFmParallelSearch.c:159-220 (and the search half of :95-157) */
void orc_batch_search(const OrcIndex *ix, const char *chars, const uint64_t *offsets, uint64_t n, uint64_t *sp,
                      uint64_t *ep, uint32_t *count, OrcTally *tally, int threads) {
  OrcTally total;
  memset(&total, 0, sizeof total);
  const int64_t numBlocks = (int64_t)((n + ORC_CONCURRENT - 1) / ORC_CONCURRENT);
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 1 ? threads : 1)
#endif
  {
    OrcTally local;
    memset(&local, 0, sizeof local);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t b = 0; b < numBlocks; b++) {
      const uint64_t i0 = (uint64_t)b * ORC_CONCURRENT;
      const uint64_t i1 = i0 + ORC_CONCURRENT > n ? n : i0 + ORC_CONCURRENT;
      search_block(ix, chars, offsets, i0, i1, sp, ep, &local);
      if (count)
        for (uint64_t i = i0; i < i1; i++) count[i] = (uint32_t)orc_range_length(sp[i], ep[i]);
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    tally_add(&total, &local);
  }
  if (tally) tally_add(tally, &total);
}

/* src/AwFmParallelSearch.c:315-365: hits in BWT order sp, sp+1, ..., ep */
void orc_batch_locate(const OrcIndex *ix, const uint64_t *sp, const uint64_t *ep, uint64_t n,
                      const uint64_t *hitOffsets, uint64_t *positions, OrcTally *tally, int threads) {
  OrcTally total;
  memset(&total, 0, sizeof total);
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 1 ? threads : 1)
#endif
  {
    OrcTally local;
    memset(&local, 0, sizeof local);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (int64_t i = 0; i < (int64_t)n; i++) {
      const uint64_t hits = orc_range_length(sp[i], ep[i]);
      for (uint64_t h = 0; h < hits; h++) {
        positions[hitOffsets[i] + h] = orc_locate_one(ix, sp[i] + h, &local.lfSteps);
        local.hits++;
      }
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    tally_add(&total, &local);
  }
  if (tally) tally_add(tally, &total);
}

/* FNV-1a 64 */
uint64_t orc_fnv1a(const void *data, uint64_t bytes, uint64_t seed) {
  const uint8_t *p = data;
  uint64_t h = seed ? seed : 0xcbf29ce484222325ULL;
  for (uint64_t i = 0; i < bytes; i++) {
    h ^= p[i];
    h *= 0x100000001b3ULL;
  }
  return h;
}
