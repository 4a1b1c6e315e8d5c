/*
 * awfm_ordered_kernel.h -- the ordered, hits-only search of large fixed-length nucleotide batches
 * (awfmGpuSearchHits).
 *
 * The backward search of a batch of unrelated k-mers reads BWT blocks at random: every step of every query is
 * one 128-B line from somewhere in the image, and the chip delivers about 5*10^10 such lines per second
 * whatever the kernel does.  The order of the queries inside a batch is not part of the
 * result, so this path picks the order that makes those reads local:
 *
 *   1. encodeQueriesKernel: one thread per query packs the k-mer into 2-bit codes (a 16-byte record) and derives
 *      a 16-bit key = the leading bits of its seed-table index.  Queries with ambiguity characters get key
 *      0x8000 and are searched by the general kernel afterwards (searchKernel<INDIRECT>).
 *   2. rocPRIM radix sort of (key, record): two 8-bit passes.
 *   3. orderedSearchKernel: the same seed lookup and the same backward steps as searchKernel (nucFastStep) over
 *      the records in key order.  Neighbours in that order start in neighbouring seed entries and, step after
 *      step, land in neighbouring blocks (the range of cP lies in the c-section of the BWT in the order of P),
 *      so most block reads hit the L2.  Each XCD has its own L2: workgroup b runs on XCD b % 8, and every XCD
 *      walks one contiguous eighth of the order.
 *
 * Putting results back under the original query numbers is a scatter of one partial line per query, which costs
 * as much as the ordering saves -- unless it is sparse.  This path therefore reports HITS: the
 * output arrays are first filled with "no hit" ({1,0} / 0) by a streaming kernel, and only queries whose final
 * range is non-empty store it.  For a query without hits the reference's batch API reports count 0 and nothing
 * else, so nothing is lost; the exact empty range the stepping ended in is what awfmGpuSearch returns.
 *
 * Results of queries with hits are those of the reference algorithm (same table entry, same steps):
 * ref src/AwFmParallelSearch.c:222-313, src/AwFmKmerTable.c:4-51, src/AwFmSearch.c:42-159.
 */
#ifndef AWFM_ORDERED_KERNEL_H
#define AWFM_ORDERED_KERNEL_H

#include "awfm_search_kernel.h"

namespace {

constexpr unsigned kOrderKeyBits = 16;        /* sort key width */
constexpr unsigned kOrderGeneralKey = 0x8000; /* key of the queries left to the general kernel */

/* 4 characters -> 4 two-bit codes (first character in bits 7..6) and 4 "not a,c,g,t,u" flags (bit i = character i);
 * same SWAR decode as the window decode of searchKernel */
__device__ __forceinline__ void decodeWord(unsigned word, unsigned &packed, unsigned &badBits) {
  unsigned t = (word >> 1) & 0x03030303u;
  t ^= (t >> 1) & 0x01010101u;
  const unsigned b0 = t & 0x01010101u, b1 = (t >> 1) & 0x01010101u, b01 = b0 & b1;
  const unsigned expect = 0x61616161u + (b0 << 1) + b1 * 6u + b01 * 11u; /* 'a','c','g','t' */
  unsigned diff = ((word | 0x20202020u) ^ expect) & ~b01;
  diff |= diff >> 4;
  diff |= diff >> 2;
  diff |= diff >> 1;
  diff &= 0x01010101u;
  badBits = (diff & 1u) | ((diff >> 7) & 2u) | ((diff >> 14) & 4u) | ((diff >> 21) & 8u);
  packed = ((t & 3u) << 6) | ((t >> 4) & 0x30u) | ((t >> 14) & 0x0Cu) | (t >> 24);
}

/* "no hit" everywhere: the ordered search only stores the queries that have hits */
__global__ void __launch_bounds__(256)
    fillNoHitKernel(ulonglong2 *__restrict__ ranges, unsigned *__restrict__ counts, const unsigned long long n) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n; i += stride) {
    if (ranges) ranges[i] = make_ulonglong2(1ull, 0ull);
    if (counts) counts[i] = 0u;
  }
}

/* fixed-length batch, depth <= len <= 32: record + key per query; depth = the table the search starts from */
__global__ void __launch_bounds__(256)
    encodeQueriesKernel(const unsigned char *__restrict__ chars, const unsigned len, const unsigned depth,
                        const unsigned long long numQueries, unsigned short *__restrict__ keys,
                        QueryRec *__restrict__ recs, unsigned *__restrict__ generalCount) {
  const unsigned long long t = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  const bool live = t < numQueries;
  unsigned long long codes = 0;
  unsigned bad = 0;
  if (live) {
    /* aligned dwords that hold the query's bytes; a dword is only read when it contains one of them */
    const unsigned long long at = (unsigned long long)chars + t * len;
    const unsigned *first = (const unsigned *)(at & ~3ull);
    const unsigned shift = (unsigned)at & 3u;
    const unsigned numDwords = (shift + len + 3u) >> 2;
    unsigned dw[9];
#pragma unroll
    for (int j = 0; j < 9; j++) dw[j] = (unsigned)j < numDwords ? first[j] : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      unsigned packed = 0, badBits = 0;
      if (4u * j < len) decodeWord(__builtin_amdgcn_alignbyte(dw[j + 1], dw[j], shift), packed, badBits);
      codes = (codes << 8) | packed;
      bad |= badBits << (4 * j);
    }
    codes >>= 2u * (32u - len); /* character 0 was in bits 63..62: now the last character is in bits 1..0 */
    bad &= len >= 32u ? ~0u : ((1u << len) - 1u);
  }
  const bool fast = live && bad == 0u;
  if (live) {
    unsigned key = kOrderGeneralKey;
    if (fast) {
      const unsigned long long index = codes & ((1ull << (2u * depth)) - 1ull);
      key = 2u * depth >= 15u ? (unsigned)(index >> (2u * depth - 15u)) : (unsigned)(index << (15u - 2u * depth));
    }
    keys[t] = (unsigned short)key;
    QueryRec r;
    r.codes = codes;
    r.index = (unsigned)t;
    r.length = fast ? len : 0xFFFFFFFFu;
    recs[t] = r;
  }
  const unsigned long long general = __ballot(live && !fast);
  if ((threadIdx.x & 63u) == 0u && general != 0ull) atomicAdd(generalCount, (unsigned)__popcll(general));
}

/*
 * Backward search over the ordered records.  Group/lane layout and the step are those of searchKernel; a query
 * comes from a 16-byte record (read one iteration ahead) and only a non-empty final range is stored, under the
 * original query number.
 */
template <int G, bool NARROW>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(G >= 4 ? 8 : 2, 8)))
    orderedSearchKernel(const DevIndex ix, const QueryRec *__restrict__ recs, const unsigned long long numRecs,
                        const unsigned *__restrict__ generalCount, const unsigned len, const unsigned depth,
                        const ulonglong2 *__restrict__ table, ulonglong2 *__restrict__ ranges,
                        unsigned *__restrict__ counts, unsigned *__restrict__ tickets, const int xcdMap = 0) {
  constexpr int S = 8 / G;
  constexpr int kGroups = kThreads / G;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sMask[256 * 8];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  for (unsigned e = threadIdx.x; e < 256u * 8u; e += kThreads) sMask[e] = sliceMask(e >> 3, e & 7u);
  __syncthreads();

  const unsigned gl = threadIdx.x % G;
  const unsigned firstPiece = gl * S;
  /* the records the fast path covers come first in the order; each XCD takes a contiguous eighth of them */
  const unsigned long long covered = numRecs - (unsigned long long)*generalCount;
  const unsigned xcds = (gridDim.x & 7u) == 0u && xcdMap != 2 ? 8u : 1u;
  const unsigned perXcd = gridDim.x / xcds;
  const unsigned xcd = xcds == 8u ? (xcdMap == 1 ? blockIdx.x / perXcd : (blockIdx.x & 7u)) : 0u;
  const unsigned blockInXcd = xcds == 8u ? (xcdMap == 1 ? blockIdx.x % perXcd : (blockIdx.x >> 3)) : blockIdx.x;
  const unsigned long long share = (covered + xcds - 1ull) / xcds;
  const unsigned long long begin = share * xcd;
  const unsigned long long end = begin + share < covered ? begin + share : covered;
  (void)blockInXcd;
  /* The workgroups of an XCD take chunks of kGroups consecutive records from a per-XCD ticket counter instead
   * of a fixed stride: free-running workgroups drift apart, and with a fixed stride the records in flight on an
   * XCD would then span many more buckets than its L2 holds the blocks of.  The ticket for the next chunk is
   * drawn one iteration ahead, so its latency and the record read hide behind the current chunk. */
  __shared__ unsigned sTicket[2];
  unsigned *ticket = tickets + xcd * 64u; /* one counter per XCD, 256 bytes apart */
  unsigned parity = 0;
  if (threadIdx.x == 0) {
    sTicket[0] = atomicAdd(ticket, 1u);
    sTicket[1] = atomicAdd(ticket, 1u);
  }
  __syncthreads();
  unsigned long long q = begin + (unsigned long long)sTicket[0] * kGroups + threadIdx.x / G;
  unsigned long long qNext = begin + (unsigned long long)sTicket[1] * kGroups + threadIdx.x / G;

  const unsigned long long tableMask = (1ull << (2u * depth)) - 1ull;
  ulonglong2 raw = make_ulonglong2(0ull, 0ull); /* the prefetched record as two 64-bit words */
  if (q < end) raw = *(const ulonglong2 *)(recs + q);
  /* every thread of the workgroup leaves the loop in the same iteration: chunks are whole, q - threadIdx.x / G is uniform */
  while (q - threadIdx.x / G < end) {
    const bool live = q < end;
    const unsigned long long codes = raw.x;
    const unsigned index = (unsigned)raw.y;
    /* draw the ticket after next, fetch the next chunk's record */
    __syncthreads(); /* sTicket[parity] (the current chunk) has been read by everyone */
    if (threadIdx.x == 0) sTicket[parity] = atomicAdd(ticket, 1u);
    if (qNext < end) raw = *(const ulonglong2 *)(recs + qNext);
    /* ---- seed (ref src/AwFmKmerTable.c:4-51): the index table, or the deeper device-only one ---- */
    pos_t sp = 1, ep = 0;
    int pos = -1;
    if (live) {
      const ulonglong2 r = table[codes & tableMask];
      sp = (pos_t)r.x;
      ep = (pos_t)r.y;
      pos = (int)(len - depth) - 1;
    }
    unsigned long long rem = codes >> (2u * depth); /* code of character `pos` in bits 1..0 */

    /* ---- extension (ref src/AwFmParallelSearch.c:273-313) ---- */
    while (pos >= 0 && sp <= ep) {
      nucFastStep<G, NARROW>(ix, sC, sMask, firstPiece, (unsigned)rem & 3u, sp, ep);
      pos--;
      rem >>= 2;
    }
    if (live && gl == 0 && sp <= ep) {
      if (ranges) ranges[index] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
      if (counts) counts[index] = (unsigned)(ep - sp + (pos_t)1);
    }
    __syncthreads(); /* the ticket drawn at the top of this iteration is visible */
    q = qNext;
    qNext = begin + (unsigned long long)sTicket[parity] * kGroups + threadIdx.x / G;
    parity ^= 1u;
  }
}

}  // namespace

#endif
