/*
 * awfm_gpu.h -- C ABI of the HIP (gfx950) side of libawfmindex_amd.so.
 *
 * This is the thin shim the host C code (and any FFI: ctypes, cgo, JNI ...)
 * calls.  Plain pointers and sizes only; `stream` arguments are hipStream_t
 * passed as void* (NULL = the null stream).  Pointers prefixed `d` are device
 * addresses (hipMalloc or any allocator that shares the HIP context, e.g. a
 * torch tensor's data_ptr()); all others are host addresses.
 *
 * What each entry point replaces in the reference:
 *   awfmGpuSearch          seed + extend phases of awFmParallelSearchCount /
 *                          awFmParallelSearchLocate
 *                          (ref src/AwFmParallelSearch.c:159-220, :222-313) on top of
 *                          the rank primitives (ref src/AwFmOccurrence.c:8-135,
 *                          src/AwFmSimdConfig.c:89-114, src/AwFmSearch.c:42-159,
 *                          :485-520, src/AwFmKmerTable.c:4-51)
 *   awfmGpuHitOffsets      the per-query sizing of setPositionListCount
 *                          (ref src/AwFmParallelSearch.c:327-328, :367-387) as
 *                          one exclusive scan
 *   awfmGpuLocate          parallelSearchTracebackPositionLists
 *                          (ref src/AwFmParallelSearch.c:315-365): LF walk
 *                          (ref src/AwFmSearch.c:369-427, src/AwFmOccurrence.c:170-217)
 *                          + sampled-SA read (ref src/AwFmSuffixArray.c:114-142, :179-191)
 *   awfmGpuCountHost /     the whole of awFmParallelSearchCount / ...Locate
 *   awfmGpuLocateHost      for host-resident flat query buffers
 *   awfmGpuIndexCreate     the reference has no analogue: it builds the
 *                          device image (re-laid-out BWT blocks, seed table,
 *                          sampled SA) of a host AwFmIndex.
 *
 * Results are bit-identical to the reference semantics:
 * a query stops at the first invalid range and keeps it; hits are listed in
 * BWT order sp, sp+1, ..., ep.
 */
#ifndef AWFM_GPU_H
#define AWFM_GPU_H

#include "AwFmIndex.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct AwFmGpuIndex AwFmGpuIndex; /* opaque device image of one index on one GPU */

/* search kernel variants: how many lanes cooperate on one query (AUTO picks the default) */
enum AwFmGpuKernel {
  AWFM_GPU_KERNEL_AUTO = 0,
  AWFM_GPU_KERNEL_GROUP8 = 1, /* 8 lanes x 16 B: one load instruction per 128-B block, 8 queries per wave */
  AWFM_GPU_KERNEL_GROUP4 = 2, /* 4 lanes x 32 B, 16 queries per wave */
  AWFM_GPU_KERNEL_GROUP2 = 3, /* 2 lanes x 64 B, 32 queries per wave (nucleotide only) */
  AWFM_GPU_KERNEL_GROUP1 = 4  /* 1 lane x 128 B, 64 queries per wave (nucleotide only) */
};

/* ---- runtime ---- */
int awfmGpuDeviceCount(void);           /* 0 when no usable HIP device */
const char *awfmGpuLastError(void);     /* thread-local text of the last failure, "" if none */
/* Return code of the calling thread's last awFmParallelSearchCount / awFmParallelSearchLocate.  Count returns void in
 * the reference API (ref src/AwFmIndex.h:400-403) and there is no CPU search path here, so a failed call (no device,
 * allocation or kernel failure) sets the `count` of every k-mer of the failed shard(s) to 0, prints the reason to
 * stderr, and leaves its code here. */
enum AwFmReturnCode awfmGpuLastBatchStatus(void);

/* ---- environment ----
 * The library reads 20 variables, all through csrc/awfm_knobs.h (INTEGRATION.md section 7 explains each); none changes a
 * result.  $AWFM_GPU_DIAG = "key=value,key=value,..." holds the test and diagnostics hooks, none of which selects a faster path:
 *   walk_give_up=N      LF steps after which an ordinary locate's walk is parked for finishKernel to walk on
 *   park_list=N         capacity of the full-suffix-array builder's list of parked walks (0: an entry per position)
 *   build_wide=1        the GPU builder's 64-bit suffix sort on any text
 *   kernel=g1|g2|g4     lanes per k-mer of the general kernel (awfmGpuIndexSetKernel does the same per image)
 *   tally_with_deep=1   awfmGpuSearchTally starts from the deeper table (default: the index's own, the reference's bytes)
 *   nuc_super_shift=13..31|auto   nucleotide superblocks of 2^shift positions: the arithmetic of images of 2^32 positions
 *                       and more on small ones
 *   stream_trace=1, aos_trace=1   host timelines of the chunked pipelines / the AoS lanes on stderr */

/* ---- device image ---- */
/* Builds the device image of `index` on GPU `device` (-1: current device or
 * $AWFM_GPU_DEVICE).  When the index has no in-memory sampled SA
 * (keepSuffixArrayInMemory == false) it is staged from index->fileDescriptor.  The image is complete when the call returns:
 * its device-only accelerators (deeper seed table, full suffix array: below) are built before it does. */
enum AwFmReturnCode awfmGpuIndexCreate(const struct AwFmIndex *index, int device, AwFmGpuIndex **out);
void awfmGpuIndexDestroy(AwFmGpuIndex *g);
/* Side table used by awFmParallelSearch*: image for a host index, created on first use.  Round 6: the image the drop-in entry
 * points make (awfmGpuIndexAcquireAll) is usable as soon as its blocks, pair image and copied tables are on the device; its
 * deeper table and full suffix array are built by a thread of their own, on a stream of their own, and installed between two
 * calls of the entry points -- the first awFmParallelSearchLocate on a GRCh38-sized index returns after 0.2 s instead of 0.9-7 s,
 * and searches issued meanwhile run on what is there (same results).  awfmGpuIndexAcquire hands over the COMPLETE image: it
 * waits for that thread. */
AwFmGpuIndex *awfmGpuIndexAcquire(const struct AwFmIndex *index);
/* Handles on the device images of a host index for every entry of $AWFM_GPU_DEVICES ("all" or a comma list of
 * ordinals), created on first use; awFmParallelSearch* deal the chunks of a list to them, one host thread each.  A device
 * named again gets a lane: a handle with its own staging buffers and locks on the image that device already
 * has (no second copy of the index), so that its chunks overlap the others' transfers and kernels.  Unset:
 * the default device with three lanes.  Returns how many handles were written to out[0..maxOut). */
int awfmGpuIndexAcquireAll(const struct AwFmIndex *index, AwFmGpuIndex **out, int maxOut);
/* Drops the side-table entries of the index (called by awFmDeallocIndex). */
void awfmGpuIndexRelease(const struct AwFmIndex *index);
uint64_t awfmGpuIndexDeviceBytes(const AwFmGpuIndex *g);
int awfmGpuIndexDevice(const AwFmGpuIndex *g);
/* Optional, nucleotide images: builds a device-only seed table of depth deepK (seedK < deepK <= 16,
 * 4^deepK x 8 bytes of HBM on images below 2^32 positions -- {sp, length}: 2.1 GB at 14, 34 GB at 16 -- and 16 bytes
 * {sp, ep} beyond) whose entries equal what the reference algorithm
 * reaches after the seed lookup plus deepK-seedK extension steps (stopping at the first invalid range), so
 * results stay bit-identical while those steps' block reads disappear.  deepK = 0 drops it.  The host
 * index, its seed table and the .awfmi file are untouched.  When an image is created: $AWFM_GPU_DEEP_SEED_K (0: none)
 * if set; otherwise images of 2^28 positions and more whose own table is shallower get depth 14 when four times the
 * table's 4.3 GB are free on the device. */
enum AwFmReturnCode awfmGpuIndexSetDeepSeed(AwFmGpuIndex *g, unsigned deepK);
unsigned awfmGpuIndexDeepSeedK(const AwFmGpuIndex *g); /* depth of the deeper table the image has, 0: none */
/* reporting: wall seconds the construction of that table took (levels + next-step bits; whoever built it: the automatic
 * choice at awfmGpuIndexCreate / Acquire, or awfmGpuIndexSetDeepSeed), and the device memory the construction held
 * beyond the table itself at its peak (the level below the deepest, 16 B x 4^(k-1)) */
double awfmGpuIndexDeepSeedBuildSeconds(const AwFmGpuIndex *g);
/* ... of which spent inside hipMalloc (a process's first allocation of tens of GB can take seconds on some boxes: device
 * memory handed back from, or scrubbed after, the process before -- not the construction) */
double awfmGpuIndexDeepSeedAllocSeconds(const AwFmGpuIndex *g);
uint64_t awfmGpuIndexDeepSeedTransientBytes(const AwFmGpuIndex *g);
/* The full suffix array on the device (32-bit entries, 4 x bwtLength bytes: 12.4 GB for a GRCh38-sized index),
 * reconstructed once from the sampled SA with the LF-walk kernel (walks capped at 32 x ratio steps; the ones that have
 * not met a sample by then -- positions inside long runs of one letter -- are completed from each other by pointer jumping),
 * so that locating a hit is one read instead of a chain of about ratio-1 dependent block reads.  Positions are bit-identical (the walk wrote them).  Built by default for
 * images of 2^26 .. 2^32 positions with a sampled array when four times its size is free on the device
 * ($AWFM_GPU_DENSE_SA=0|1: never / always); enable = 0 drops it, 1 builds it.  Needs bwtLength < 2^32. */
enum AwFmReturnCode awfmGpuIndexSetDenseSa(AwFmGpuIndex *g, int enable);
int awfmGpuIndexHasDenseSa(const AwFmGpuIndex *g);
/* One line of text about what the image holds: its size, which of the optional accelerators it has (pair image, deeper table
 * and its depth, full suffix array, tables per k-mer length) and which ones its size asked for and it did NOT get, with the
 * reason (every one of them is dropped silently when device memory is short: the searches run without it, results are the
 * same).  Returns the length of the whole text; at most outBytes - 1 characters and a 0 are written. */
int awfmGpuIndexDescribe(const AwFmGpuIndex *g, char *out, int outBytes);
double awfmGpuIndexDenseSaBuildSeconds(const AwFmGpuIndex *g); /* reporting: wall seconds of the automatic construction */
/* Device-only tables per k-mer length (nucleotide images below 2^32 positions with the narrow deeper table of depth D): for
 * every length d = 1 .. D-1 the 8-byte entry {first position, length} of the range of EVERY d-letter string -- what the
 * reference reaches for a k-mer of exactly d characters (ref src/AwFmSearch.c:485-520 below the seed table's length,
 * src/AwFmKmerTable.c:4-51 at it, src/AwFmParallelSearch.c:273-313 above it) -- (4^D - 4) / 3 entries, 11.5 GB for D = 16.
 * Built on the device by the first mixed-length batch (CSR offsets) that takes the lookup-first kernel
 * ($AWFM_GPU_MIXED_LOOKUP=0: never), kept with the image; bytes / wall seconds of that construction (0: none yet).
 * THAT ONE CALL IS NOT ASYNCHRONOUS: the tables are built on the null stream and the call waits for the device (0.02 s for
 * D = 16) before it launches its search on the caller's stream; they are built only when three times their size is free on
 * the device, and a construction that found no room is tried again 64 mixed-length searches later. */
uint64_t awfmGpuIndexLengthTableBytes(const AwFmGpuIndex *g);
double awfmGpuIndexLengthTableBuildSeconds(const AwFmGpuIndex *g);
/* Nucleotide images carry, beside the one-letter blocks, a pair image: for every BWT position the pair of its two
 * preceding text characters, in 128-byte blocks of 128 positions with 16 base counts, so that two backward steps (two
 * LF steps) are one rank over a 16-letter sequence and one block read (csrc/awfm_pair.h).  The searches and the LF
 * walk of awfmGpuLocate use it; results are those of the letter-by-letter steps (ref src/AwFmSearch.c:42-103,
 * :369-427), bit for bit -- awfmGpuSearch's final range of a k-mer without hits included (a k-mer that dies inside a
 * pair step gets the range of the single step that emptied it).  Built with every nucleotide image unless
 * $AWFM_GPU_PAIR=0; enable = 0 drops it (the image then takes one step per read), enable != 0 rebuilds it.  It adds
 * 1 byte per BWT position of device memory; the host index and the .awfmi file are untouched. */
enum AwFmReturnCode awfmGpuIndexSetPairImage(AwFmGpuIndex *g, int enable);
int awfmGpuIndexHasPairImage(const AwFmGpuIndex *g);
/* Selects the search kernel variant for this image (default AUTO). */
void awfmGpuIndexSetKernel(AwFmGpuIndex *g, enum AwFmGpuKernel kernel);
/* The kernels keep BWT positions in 32 bits whenever bwtLength < 2^32 and in 64 bits otherwise (the reference is
 * 64-bit throughout: ref src/AwFmIndex.h:88-91, src/AwFmSuffixArray.c:114-142).  wide != 0 selects the 64-bit
 * instantiations on any image -- same results, used by the parity tests to cover the code indices of 2^32 or more
 * positions run; also read from $AWFM_GPU_FORCE_WIDE when an image is created. */
void awfmGpuIndexSetWide(AwFmGpuIndex *g, int wide);
int awfmGpuIndexIsWide(const AwFmGpuIndex *g); /* 1 when searches on this image run the 64-bit instantiations */

/* ---- index construction on the GPU ---- */
/* Same contract and byte-identical arrays as awFmCreateIndex (ref src/AwFmCreate.c:31-137), built on
 * the device: suffix sort (radix + prefix doubling), BWT bit planes, base counts, seed table, sampled
 * SA.  `sequence` is a host pointer, or a device pointer when sequenceOnDevice != 0.  fileSrc may be
 * NULL (no .awfmi file is written; an extension over the reference).  The device image stays
 * resident and is the one awFmParallelSearch* will use.  Suffix positions and ranks are 32-bit on the device while
 * bwtLength <= 2^32-2 (about 25 bytes of HBM per position at the peak) and 64-bit beyond (about 37 bytes per
 * position: a 4.4 Gbp text builds in 12 s, a two-strand human genome of 6.2 Gbp fits one MI355X);
 * $AWFM_GPU_DIAG build_wide=1 selects the 64-bit suffix sort on any text (tests). */
enum AwFmReturnCode awfmGpuCreateIndex(struct AwFmIndex **index, const struct AwFmIndexConfiguration *config,
                                       const uint8_t *sequence, uint64_t sequenceLength, int sequenceOnDevice,
                                       const char *fileSrc, int device);

/* ---- flat batch API on device buffers ---- */
/* Queries: dChars = concatenated ASCII k-mers; either dOffsets (numQueries+1
 * CSR offsets into dChars) or, with dOffsets == NULL, fixedLength characters
 * per query.  Outputs (each may be NULL): dRanges[numQueries] = final {sp,ep},
 * dCounts[numQueries] = range length truncated to u32
 * (ref src/AwFmIndexStruct.c:126-130).  Asynchronous on `stream`; allocates no table: large batches read the image's
 * device-only tables where it has them (the deeper table; the tables per k-mer length once a hits-only mixed-length batch
 * has built them), and scratch of 8 bytes per k-mer in one of the image's two slots. */
enum AwFmReturnCode awfmGpuSearch(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                  uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                  uint32_t *dCounts, void *stream);

/* Hits-only variant of awfmGpuSearch, for callers that go on to count or locate (what
 * awFmParallelSearchCount/Locate report: ref src/AwFmParallelSearch.c:159-220, :315-365): queries with hits get
 * exactly the range and count awfmGpuSearch gives them; a query WITHOUT hits gets count 0 and some empty range
 * (sp > ep), not necessarily the one the stepping ended in.  That freedom lets large nucleotide batches (>= 2^23
 * k-mers against >= 2^28 positions, fixed length or CSR; $AWFM_GPU_ORDERED=0|1 or awfmGpuIndexSetOrdered override)
 * be searched in seed order: the k-mers are packed into 8- or 16-byte records, partitioned by the leading bits of the
 * string their search starts from, searched in that order so that neighbouring queries read neighbouring blocks out of
 * the L2, and only the non-empty results are stored under their query numbers over a "no hit" fill.
 * Other batches run awfmGpuSearch.  Scratch: 16-36 bytes per query, owned by the image and re-used; searches on one
 * image are ordered across streams.  Every launch is asynchronous on `stream`: a fixed-length ASCII batch of >= 2^20
 * k-mers that starts from the deeper table is sampled first (awfmGpuLastOrderedKernelIsLookup below), and the sample's
 * verdict stays on the device -- the kernels of both front ends are launched and the one it does not choose returns at
 * once ($AWFM_GPU_LOOKUP_FIRST=0|1: no sample). */
enum AwFmReturnCode awfmGpuSearchHits(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                      uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                      uint32_t *dCounts, void *stream);
/* The same when the caller reads the ranges through the counts, as awfmGpuHitOffsetsFromCounts + awfmGpuLocate do:
 * dCounts[i] (required) is written for every k-mer, dRanges[i] for the k-mers with hits -- the range of a k-mer with
 * dCounts[i] == 0 may be left as the caller passed it (the seed-order path then streams 4 instead of 20 bytes of
 * "no hit" per k-mer over the outputs).  Hit offsets must then come from the counts (awfmGpuHitOffsets reads every
 * range), i.e. from images below 2^32 positions. */
enum AwFmReturnCode awfmGpuSearchHitsSparse(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                            uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                            uint32_t *dCounts, void *stream);
/* Sparse results: the k-mers with hits as a list instead of a range / count under every query number -- for batches in
 * which few k-mers occur (10^8 random 21-mers against a human-sized text: 7 * 10^4), where writing, scanning and moving
 * 10^8 "no hit" records is most of what happens after the search.  Only batches that take the seed-order path
 * (awfmGpuSearchHitsIsOrdered; AwFmUnsupportedVersionError otherwise -- run awfmGpuSearchHits + awfmGpuCompactHits then).
 * dHitKmers[capacity] / dHitRanges[capacity] are first filled with {0xFFFFFFFF, empty range}; every k-mer with hits then
 * appends {its number in the batch, its range} (order: as the waves come); *dNumHits (device) = how many there are, which
 * may exceed capacity -- the list is then incomplete and the caller repeats densely (a dense batch should not be searched
 * this way in the first place: its appends contend for one counter; capacity = numQueries / 64 keeps a mistaken attempt
 * cheap).  `packed`: dChars is one 64-bit word per k-mer (awfmGpuSearchHitsPacked).  awfmGpuSortHits orders the list by
 * k-mer number (entries beyond the hits sort last); awfmGpuHitOffsets / awfmGpuLocate then take the list as if it were
 * the batch (numQueries = capacity or the number of hits). */
enum AwFmReturnCode awfmGpuSearchHitsCompact(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                             uint32_t fixedLength, uint64_t numQueries, int packed, uint32_t *dHitKmers,
                                             struct AwFmSearchRange *dHitRanges, uint32_t capacity, uint32_t *dNumHits,
                                             void *stream);
/* Results IN SEARCH ORDER, for batches in which most k-mers have hits: entry q = {number of the k-mer the seed-order search
 * took q-th, its range (exact when it has hits, empty otherwise)}, every k-mer of the batch exactly once, written as whole
 * lines -- instead of 10^8 partial-line stores under the original k-mer numbers.  awfmGpuHitOffsets / awfmGpuLocate take
 * dOrderRanges as if it were the batch: hit offsets and positions then follow the search order too (neighbouring entries
 * are neighbours in the BWT, which the walk's first steps share), and dOrderKmers says whose they are -- what a consumer
 * that scatters into per-k-mer lists anyway (awFmParallelSearchLocate does: ref src/AwFmParallelSearch.c:327-361) needs.
 * Only batches that take the seed-order path (AwFmUnsupportedVersionError otherwise). */
enum AwFmReturnCode awfmGpuSearchHitsInOrder(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                             uint32_t fixedLength, uint64_t numQueries, int packed, uint32_t *dOrderKmers,
                                             struct AwFmSearchRange *dOrderRanges, void *stream);
/* the same with the 32-bit counts in that order as well (dOrderCounts, may be NULL; ref src/AwFmIndexStruct.c:126-130: the
 * list's count is a u32): awfmGpuHitOffsetsOnDevice then scans 4 instead of 16 bytes per k-mer (round 6: 10^8 planted
 * 21-mers, the scan 0.78 -> 0.3 ms) */
enum AwFmReturnCode awfmGpuSearchHitsInOrderCounts(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                                   uint32_t fixedLength, uint64_t numQueries, int packed, uint32_t *dOrderKmers,
                                                   struct AwFmSearchRange *dOrderRanges, uint32_t *dOrderCounts, void *stream);
/* the same list from dense results (dCounts / dRanges of awfmGpuSearchHits or awfmGpuSearch), already in k-mer order:
 * dFlagOffsets[numQueries + 1] and dScratch (awfmGpuScanScratchBytes) are work space */
enum AwFmReturnCode awfmGpuCompactHits(AwFmGpuIndex *g, const uint32_t *dCounts, const struct AwFmSearchRange *dRanges,
                                       uint64_t numQueries, uint64_t *dFlagOffsets, void *dScratch, uint32_t *dHitKmers,
                                       struct AwFmSearchRange *dHitRanges, uint32_t capacity, uint32_t *dNumHits, void *stream);
enum AwFmReturnCode awfmGpuSortHits(AwFmGpuIndex *g, uint32_t *dHitKmers, struct AwFmSearchRange *dHitRanges,
                                    uint32_t numEntries, void *stream);
/* The same order without the host knowing the list's length: the first min(*dNumHits, capacity) entries of the list
 * awfmGpuSearchHitsCompact appended (k-mer numbers distinct and below numQueries, the batch's size) are put in k-mer order
 * by ranking them in a bitmap of the batch; *dNumHits is read on the device, every launch is asynchronous on `stream`, and
 * the entries beyond the hits stay {0xFFFFFFFF, empty range}.  With awfmGpuHitOffsetsOnDevice / awfmGpuLocateOnDevice a
 * batch is searched, listed and located without one host wait. */
enum AwFmReturnCode awfmGpuSortHitsOnDevice(AwFmGpuIndex *g, uint32_t *dHitKmers, struct AwFmSearchRange *dHitRanges,
                                            uint32_t capacity, const uint32_t *dNumHits, uint64_t numQueries, void *stream);

/* The whole tail of a step whose results are the list, in one launch (lists of up to 2^18 entries; longer ones through the
 * three calls above): the first min(*dNumHits, capacity) entries of the list awfmGpuSearchHitsCompact appended -- left as
 * they are -- come out in k-mer order in dSortedKmers / dSortedRanges (entries beyond the hits: {0xFFFFFFFF, empty range}),
 * dHitOffsets[0 .. capacity] are the hit offsets over the sorted list (every entry from the list's length on holds the
 * total), and dPositions[0 .. capacityHits) the first capacityHits positions in that order (may be NULL: offsets only) --
 * what awfmGpuSortHitsOnDevice + awfmGpuHitOffsetsOnDevice + awfmGpuLocateOnDevice leave, from seven dependent launches less
 * (ref src/AwFmParallelSearch.c:315-365: the reference sizes and fills one list per k-mer on the host).  Nothing waits for
 * the host; the sorted arrays must not be the unsorted ones. */
enum AwFmReturnCode awfmGpuListLocateOnDevice(AwFmGpuIndex *g, const uint32_t *dHitKmers, const struct AwFmSearchRange *dHitRanges,
                                              uint32_t capacity, const uint32_t *dNumHits, uint64_t numQueries, uint32_t *dSortedKmers,
                                              struct AwFmSearchRange *dSortedRanges, uint64_t *dHitOffsets, uint64_t capacityHits,
                                              uint64_t *dPositions, void *stream);

/* -1 = automatic (default), 0 = never, 1 = whenever the ordered path applies */
void awfmGpuIndexSetOrdered(AwFmGpuIndex *g, int mode);
/* 1 when awfmGpuSearchHits would search such a batch in seed order on this image (reporting, bench.py) */
int awfmGpuSearchHitsIsOrdered(const AwFmGpuIndex *g, int hasOffsets, uint32_t fixedLength, uint64_t numQueries);
/* STREAMS AND THE IMAGE'S SCRATCH.  The seed-order searches and the list's ordering share scratch memory that belongs to the
 * image, and the image orders its use across streams with events.  For a stream the caller created, the event of a use is
 * not recorded when the use is enqueued (a recorded event leaves the queue idle for ~5 us; a search followed by another on the
 * SAME stream needs none) but when a search on ANOTHER stream needs the scratch -- on the first stream, behind whatever it has
 * been given since.  The image therefore keeps the handle of the last stream that used each piece of scratch.  Rule: a stream
 * that has searched on an image must either outlive the image, or be RETIRED before it is destroyed:
 * awfmGpuStreamRetire(g, stream) records what is still owed on it and forgets the handle (cheap; no wait unless an event
 * cannot be recorded).  The null stream and hipStreamPerThread need nothing: their uses are recorded at once. */
void awfmGpuStreamRetire(AwFmGpuIndex *g, void *stream);

/* measurement hook: with $AWFM_GPU_TIME_ORDERED set, awfmGpuSearchHits brackets its dominant kernel (orderedSearchKernel,
 * or encodeLookupKernel when the batch was one for "lookup first") with HIP events on the launch stream; this returns the
 * last bracket in ms (<0: none).  Reporting calls: this one and the two below may wait for the device. */
double awfmGpuLastOrderedKernelMs(AwFmGpuIndex *g);
/* 1 when the last seed-order search on the image looked the table entries up while encoding ("lookup first": batches of
 * ASCII k-mers of which, by a sample, fewer than a quarter are still alive after the deeper table; only those are then
 * ordered and searched; $AWFM_GPU_LOOKUP_FIRST=0 / 1: never / whenever it applies) -- the timed kernel is then
 * encodeLookupKernel */
int awfmGpuLastOrderedKernelIsLookup(AwFmGpuIndex *g);
/* Which front end(s) the last SAMPLED search on the image launched (reporting): 0 both -- the sample's verdict stays on the
 * device and the kernels of the front end it does not choose return at once --, 1 the lookup kernel only, 2 the ordering
 * passes and the ordered kernel only; -1: no sampled search yet.  1 and 2 happen when the verdict of an earlier search of
 * the same k-mer length has reached the host (it is published in page-locked memory by the kernel that takes the sample;
 * nothing waits for it): a stream of like batches then pays for one front end's launches, not two ($AWFM_GPU_LOOKUP_PREDICT=0:
 * always both).  Either front end alone searches any batch correctly; a verdict that contradicts the mode its own search
 * ran in switches the prediction off for the next 8, 16, ... searches. */
int awfmGpuLastLookupFront(AwFmGpuIndex *g);
/* Whether the last awfmGpuSearch on the image (also the one behind a hits-only search that the seed-order path did not take)
 * went through exactLookupSearchKernel -- the exact ranges from one table entry per k-mer -- (1) or the general kernel (0);
 * reporting only */
int awfmGpuLastSearchWasExactLookup(AwFmGpuIndex *g);
/* with $AWFM_GPU_TIME_ORDERED: orderedSearchKernel's own bracket of the last search, whichever kernel was the dominant one
 * (after encodeLookupKernel it searched only the k-mers that kernel kept); < 0: none */
double awfmGpuLastOrderedSearchKernelMs(AwFmGpuIndex *g);
/* the brackets of EVERY timed search since the last call (at most the last 1024), oldest first: frontMs[i] =
 * encodeLookupKernel's (< 0: that search had none), kernelMs[i] = orderedSearchKernel's; returns how many and empties the
 * log.  A loop of searches is timed kernel by kernel without a host wait inside the loop (bench.py). */
int awfmGpuOrderedKernelLog(AwFmGpuIndex *g, double *frontMs, double *kernelMs, int max);
/* k-mers the last seed-order search with 8-byte records ordered and searched: the batch, or what the lookup-first pass kept
 * of it (reporting; waits for the device) */
uint64_t awfmGpuLastOrderedKept(AwFmGpuIndex *g);

/* Instrumented run of the same kernel for the roofline accounting: tallyOut =
 * {queries that used the seed table, backward steps executed, distinct blocks over those steps,
 * query characters}.  Synchronous, not for timing. */
enum AwFmReturnCode awfmGpuSearchTally(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                       uint32_t fixedLength, uint64_t numQueries, uint64_t tallyOut[4]);

/* What the lookup-first kernel of mixed-length batches (awfm_mixed_lookup_kernel.h: one table entry per k-mer, from the table
 * of its own length or from the deeper table, then the steps of the k-mers still alive) has to read for this batch: an
 * instrumented pass over the same k-mers with every 128-B line marked in a bitmap.  tallyOut = {lines of the length tables
 * touched, lines of the deeper table touched, distinct (search level, 128-B line) pairs of the pair image, the same of the
 * one-letter image, k-mers still alive after their table entry, k-mers with hits, k-mers left to the general kernel,
 * block lines the steps of the k-mers still alive read as executed (no line shared between two k-mers)}.  Builds the length tables when the image can have them and has none yet;
 * AwFmUnsupportedVersionError when it cannot (amino, 2^32 positions or more, no narrow deeper table).  Synchronous, not for
 * timing. */
enum AwFmReturnCode awfmGpuMixedLookupLineTally(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                                uint64_t numQueries, uint64_t tallyOut[8]);

/* Instrumented run of the seed-order path of awfmGpuSearchHits on the same batch (encode + sort + search with every
 * line the search kernel reads marked in per-level bitmaps): the COMPULSORY memory traffic of that kernel, i.e. what an
 * ideal cache would still have to fetch.  tallyOut = {128-B lines of the seed table touched, lines of the deeper
 * device-only table touched, distinct (search level, 128-B line) pairs of the pair image, the same of the one-letter
 * image, k-mers the seed-order kernel searched, bytes of sorted record + key it reads per k-mer, k-mers with hits,
 * k-mers left to the general kernel}.  AwFmUnsupportedVersionError when the batch would not take the seed-order path
 * (awfmGpuSearchHitsIsOrdered).  Synchronous, not for timing. */
enum AwFmReturnCode awfmGpuSearchHitsLineTally(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                               uint32_t fixedLength, uint64_t numQueries, uint64_t tallyOut[8]);

/* dHitOffsets[numQueries+1] = exclusive scan of the range lengths; the total is
 * also copied to *totalHits (host) -- this call synchronises `stream`.
 * dScratch must hold awfmGpuScanScratchBytes(numQueries) bytes. */
uint64_t awfmGpuScanScratchBytes(uint64_t numQueries);
enum AwFmReturnCode awfmGpuHitOffsets(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges, uint64_t numQueries,
                                      uint64_t *dHitOffsets, void *dScratch, uint64_t *totalHits, void *stream);
/* The same from the 32-bit counts awfmGpuSearch / awfmGpuSearchHits wrote (a quarter of the bytes of the
 * ranges); exact, hence allowed, only for images below 2^32 positions. */
enum AwFmReturnCode awfmGpuHitOffsetsFromCounts(AwFmGpuIndex *g, const uint32_t *dCounts, uint64_t numQueries,
                                                uint64_t *dHitOffsets, void *dScratch, uint64_t *totalHits, void *stream);

/* The scan without the read-back: dHitOffsets[numQueries] (device) holds the total, nothing waits for the host.  From
 * the 32-bit counts (images below 2^32 positions) when dCounts is given, from the ranges otherwise. */
enum AwFmReturnCode awfmGpuHitOffsetsOnDevice(AwFmGpuIndex *g, const uint32_t *dCounts, const struct AwFmSearchRange *dRanges,
                                              uint64_t numQueries, uint64_t *dHitOffsets, void *dScratch, void *stream);
/* awfmGpuLocate with the number of hits read ON THE DEVICE (dHitOffsets[numQueries]): dPositions holds capacityHits
 * entries, the first min(total, capacityHits) hits are located; the caller learns the total whenever it next reads
 * dHitOffsets[numQueries] and repeats with a larger buffer if it was too small.  Asynchronous on `stream`. */
enum AwFmReturnCode awfmGpuLocateOnDevice(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges, const uint64_t *dHitOffsets,
                                          uint64_t numQueries, uint64_t capacityHits, uint64_t *dPositions, void *stream);

/* dPositions[dHitOffsets[i] + h] = text position of hit h (BWT order) of query i.
 * Asynchronous on `stream`. */
enum AwFmReturnCode awfmGpuLocate(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges,
                                  const uint64_t *dHitOffsets, uint64_t numQueries, uint64_t totalHits,
                                  uint64_t *dPositions, void *stream);

/* The same with the final positions written to outPositions instead of over dPositions (which stays the work array of
 * the walk): outPositions may be any memory the device can store to, e.g. page-locked host memory (awfmGpuHostAlloc),
 * in which case the kernel that produces the positions also delivers them and no device-to-host copy is needed. */
enum AwFmReturnCode awfmGpuLocateTo(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges,
                                    const uint64_t *dHitOffsets, uint64_t numQueries, uint64_t totalHits,
                                    uint64_t *dPositions, uint64_t *outPositions, void *stream);

/* A window of the batch's hit list: the hits numbered hitBegin .. hitEnd-1 (in the numbering of dHitOffsets, i.e. hit h of
 * query i has number dHitOffsets[i] + h), which belong to the queries queryBegin .. queryEnd-1 (any superset of the
 * queries whose lists meet the window will do); dPositions / outPositions hold hitEnd - hitBegin entries, entry 0 is hit
 * hitBegin.  A window may begin and end inside the list of one k-mer: this is how a locate whose hit list exceeds a
 * device-memory budget is taken in pieces (the reference grows every positionList on its own, ref
 * src/AwFmParallelSearch.c:315-387; here the list is flat and the budget bounds what is resident). */
enum AwFmReturnCode awfmGpuLocateWindow(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges, const uint64_t *dHitOffsets,
                                        uint64_t queryBegin, uint64_t queryEnd, uint64_t hitBegin, uint64_t hitEnd,
                                        uint64_t *dPositions, uint64_t *outPositions, void *stream);

/* Reporting for the drop-in AoS entry points: what the process's last awFmParallelSearchCount / Locate spent where, summed over
 * its chunks -- out = {wall ms, ms waiting for the host stages' turn, ms packing, ms in the device calls, ms scattering, chunks,
 * k-mers, hits, bytes the pack stage read + wrote, bytes the scatter stage read + wrote} (the lanes overlap: the sums of the
 * stages exceed the wall time) -- and the copy rate this box gives `threads` of the same thread pool (GB/s, bytes read + bytes
 * written, over `bytes`): the pack and scatter stages move bytes and nothing else, so that rate bounds them. */
void awfmGpuAosLastStages(double out[10]);
double awfmHostCopyGBs(unsigned threads, uint64_t bytes);

/* ---- pinned staging for the drop-in AoS entry points ---- */
/* A grow-only page-locked host buffer cached in the image (slot 0..3); valid until the next call for the
 * same slot.  awfmGpuAosLock/Unlock serialise the AoS entry points that share these buffers. */
void *awfmGpuPinnedBuffer(AwFmGpuIndex *g, int slot, uint64_t bytes);
void awfmGpuAosLock(AwFmGpuIndex *g);
void awfmGpuAosUnlock(AwFmGpuIndex *g);

/* ---- flat batch API on host buffers (upload, kernels, download) ---- */
/* Synchronous for the caller; the work is issued on the calling thread's own stream (hipStreamPerThread), so
 * several host threads (or the lanes of the AoS entry points) overlap on the device.
 * ranges / counts may be NULL. */
enum AwFmReturnCode awfmGpuCountHost(AwFmGpuIndex *g, const uint8_t *chars, const uint64_t *offsets,
                                     uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *ranges,
                                     uint32_t *counts);
/* hitOffsets[numQueries+1] is filled; *positions is a malloc'ed array of
 * hitOffsets[numQueries] entries that the caller frees. */
enum AwFmReturnCode awfmGpuLocateHost(AwFmGpuIndex *g, const uint8_t *chars, const uint64_t *offsets,
                                      uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *ranges,
                                      uint64_t *hitOffsets, uint64_t **positions);

/* The same with the hit list delivered in windows, for batches whose hits exceed what may be resident on the device
 * ($AWFM_GPU_HIT_BUDGET_BYTES; default a quarter of the free device memory, at most 2^31 hits): hitOffsets[0..numQueries]
 * is complete before the first call of the sink; the sink then receives consecutive windows [hitBegin, hitEnd) of the
 * flat hit list (hit h of query i has number hitOffsets[i] + h), `positions` holding hitEnd - hitBegin entries in
 * page-locked staging that stays valid until the sink returns, and the queries queryBegin .. queryEnd-1 whose lists meet
 * the window (the first and the last may be cut).  While the sink runs, the next window is walked and downloaded.  A
 * non-zero return stops the batch.  A batch that fits the budget is one window.  This is what the drop-in AoS entry
 * points run on (hold awfmGpuAosLock: the staging is slot 3 of the image's pinned buffers). */
typedef int (*AwFmGpuHitWindowSink)(void *user, uint64_t queryBegin, uint64_t queryEnd, uint64_t hitBegin, uint64_t hitEnd,
                                    const uint64_t *positions);
enum AwFmReturnCode awfmGpuLocateHostWindows(AwFmGpuIndex *g, const uint8_t *chars, const uint64_t *offsets,
                                             uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *ranges,
                                             uint64_t *hitOffsets, AwFmGpuHitWindowSink sink, void *user);

/* ---- packed k-mers and the chunked host-buffer pipeline (awfm_gpu_stream.hip) ----
 * The reference's batch is an array of structs: a kmerString pointer in and a malloc'ed positionList out per
 * k-mer (ref src/AwFmParallelSearch.c:36-93, :367-387).  These entry points take the batch flat and bit-packed
 * and return it flat, cut into chunks whose upload, kernels and download overlap.
 *
 * Packed k-mer: one 64-bit word per k-mer of kmerLength characters, first character most significant, last
 * character in the lowest bits; nucleotide 2 bits per character (a 0, c 1, g 2, t/u 3; kmerLength <= 32), amino
 * 5 bits per character = the letter index of ref src/AwFmLetter.c:55-67 (a 0, c 1, d 2, ... y 19; 20..31 search
 * as the ambiguity letter x; kmerLength <= 12).  Nucleotide ambiguity characters cannot be expressed: batches
 * that contain them go through the ASCII entry points. */
/* host-side packing of n fixed-length ASCII k-mers; returns AwFmIllegalPositionError and the number of the first
 * k-mer that cannot be expressed in *firstUnpackable (may be NULL) */
enum AwFmReturnCode awfmPackKmers(enum AwFmAlphabetType alphabet, const uint8_t *chars, uint32_t kmerLength,
                                  uint64_t numKmers, uint64_t *packedOut, uint64_t *firstUnpackable);
/* the same on device buffers, and back.  K-mers that cannot be expressed are counted in *numUnpackable and become the
 * all-ones word -- itself a k-mer ('t' x 32), so the call then returns AwFmIllegalPositionError like awfmPackKmers: the
 * output of such a batch must not be searched (search it as ASCII instead) */
enum AwFmReturnCode awfmGpuPackKmers(AwFmGpuIndex *g, const uint8_t *dChars, uint32_t kmerLength, uint64_t numKmers,
                                     uint64_t *dPacked, uint64_t *numUnpackable, void *stream);
enum AwFmReturnCode awfmGpuUnpackKmers(AwFmGpuIndex *g, const uint64_t *dPacked, uint32_t kmerLength, uint64_t numKmers,
                                       uint8_t *dChars, void *stream);
/* awfmGpuSearchHits for bit-packed k-mers resident on the device.  Nucleotide batches that take the seed-order path are
 * searched straight from the packed words; other batches are unpacked into dCharsScratch (kmerLength bytes per k-mer;
 * may be NULL when the caller knows awfmGpuSearchHitsIsOrdered) and searched as ASCII.  Same outputs and contract. */
enum AwFmReturnCode awfmGpuSearchHitsPacked(AwFmGpuIndex *g, const uint64_t *dPacked, uint32_t kmerLength, uint64_t numKmers,
                                            struct AwFmSearchRange *dRanges, uint32_t *dCounts, uint8_t *dCharsScratch,
                                            void *stream);
/* page-locked host memory: batches handed over in it are read by the DMA engine directly, anything else is first
 * copied into the pipeline's own staging by hostThreads threads */
void *awfmGpuHostAlloc(uint64_t bytes);
void awfmGpuHostFree(void *p);
/* Receives the results of k-mers firstKmer .. firstKmer+numKmers-1: counts[i] hits of k-mer firstKmer+i (the
 * reference's uint32 count, ref src/AwFmIndex.h:112-118), and -- locate -- numPositions text positions, the
 * hits of k-mer firstKmer+i starting where those of firstKmer+i-1 end, each list in BWT order (what
 * awFmParallelSearchLocate puts into positionList).  The arrays are page-locked staging of the pipeline, valid until
 * the sink returns; chunks arrive in order; a non-zero return stops the batch.  The sink runs on the calling thread
 * while the image's pipeline is locked: it must not start another batch on the same image.
 * A chunk whose hits exceed the device's hit budget ($AWFM_GPU_HIT_BUDGET_BYTES; default a quarter of the free device
 * memory) arrives in several calls: consecutive groups of whole k-mers, and a k-mer whose own list exceeds a window
 * alone, in consecutive calls with the same firstKmer and numKmers == 1, each with the next slice of its list
 * (counts[0] is its full count every time).  Concatenating the positions of all calls gives the batch's flat list. */
typedef int (*AwFmGpuChunkSink)(void *user, uint64_t firstKmer, uint64_t numKmers, const uint32_t *counts,
                                const uint64_t *positions, uint64_t numPositions);
/* Counts (locate == 0) or locates numKmers packed host-resident k-mers in chunks of chunkKmers (0: 2^24) through
 * three pipeline slots: while the sink consumes chunk t-2 on the calling thread, chunk t-1 is in the kernels and
 * chunk t on its way to the device.  One batch at a time per image. */
enum AwFmReturnCode awfmGpuStreamPacked(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                        uint64_t numKmers, uint64_t chunkKmers, int locate, unsigned hostThreads,
                                        AwFmGpuChunkSink sink, void *user);
/* the same pipeline for fixed-length ASCII k-mers (any character the ASCII API accepts) */
enum AwFmReturnCode awfmGpuStreamChars(AwFmGpuIndex *g, const uint8_t *chars, uint32_t kmerLength, uint64_t numKmers,
                                       uint64_t chunkKmers, int locate, unsigned hostThreads, AwFmGpuChunkSink sink,
                                       void *user);
/* The same pipelines with SPARSE results: per chunk the k-mers with hits as a list -- hitKmers[j] = number of the j-th such
 * k-mer relative to firstKmer (ascending), its hits positions[hitOffsets[j] .. hitOffsets[j + 1]) in BWT order
 * (hitOffsets has numHitKmers + 1 entries; locate == 0: positions is NULL and the offsets only say how many hits each
 * has) -- instead of a count for every k-mer of the chunk: for a batch in which few k-mers occur the download shrinks
 * from 4 bytes per k-mer to 12 bytes per k-mer WITH hits, and nothing of the chunk's size is written after the search.
 * A chunk with more than numKmers / 64 k-mers with hits is searched again densely and its list made from the counts (and
 * so are the chunks after it): correct for any batch, fast for sparse ones.  A chunk whose hits exceed the device's hit
 * budget fails with AwFmAllocationFailure (the dense pipeline takes such chunks in windows). */
typedef int (*AwFmGpuSparseChunkSink)(void *user, uint64_t firstKmer, uint64_t numKmers, uint64_t numHitKmers,
                                      const uint32_t *hitKmers, const uint64_t *hitOffsets, const uint64_t *positions,
                                      uint64_t numPositions);
enum AwFmReturnCode awfmGpuStreamPackedSparse(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                              uint64_t numKmers, uint64_t chunkKmers, int locate, unsigned hostThreads,
                                              AwFmGpuSparseChunkSink sink, void *user);
enum AwFmReturnCode awfmGpuStreamCharsSparse(AwFmGpuIndex *g, const uint8_t *chars, uint32_t kmerLength, uint64_t numKmers,
                                             uint64_t chunkKmers, int locate, unsigned hostThreads, AwFmGpuSparseChunkSink sink,
                                             void *user);

/* whole batch into caller arrays: counts[numKmers]; *positions is malloc'ed (caller frees), *numPositions entries */
enum AwFmReturnCode awfmGpuCountPackedHost(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                           uint64_t numKmers, uint32_t *counts);
enum AwFmReturnCode awfmGpuLocatePackedHost(AwFmGpuIndex *g, const uint64_t *packedKmers, uint32_t kmerLength,
                                            uint64_t numKmers, uint32_t *counts, uint64_t **positions,
                                            uint64_t *numPositions);

/* ---- seeded synthetic inputs on the device (bench and full-size tests) ---- */
/* text characters start..start+count-1 of the stream `seed`; amino != 0 selects the 20-letter alphabet */
enum AwFmReturnCode awfmGpuSynthText(uint8_t *dOut, uint64_t start, uint64_t count, uint64_t seed, int amino,
                                     void *stream);
/* `count` uniform random k-mers (query ids first..first+count-1) of `length` characters, row-major */
enum AwFmReturnCode awfmGpuSynthRandomQueries(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length,
                                              uint64_t seedQ, int amino, void *stream);
/* `count` k-mers copied from the (unsanitised) device text at seeded uniform offsets */
enum AwFmReturnCode awfmGpuSynthPlantedQueries(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length,
                                               uint64_t seedQ, const uint8_t *dText, uint64_t textLength, void *stream);

/* A genome-shaped nucleotide text of `length` characters (csrc/awfm_synth.hip, synth.py genome_text): interspersed repeat
 * families (a 300-character unit in about length/3000 copies at 10 % divergence, a 6000-character one in cut copies at
 * 5 %), tandem repeats, 24 runs of 'n' of up to 10^7 characters, unique sequence in between. */
enum AwFmReturnCode awfmGpuSynthGenomeText(uint8_t *dOut, uint64_t length, uint64_t seed, void *stream);
/* awfmGpuSynthPlantedQueries with every character that is not a,c,g,t replaced by a seeded random letter */
enum AwFmReturnCode awfmGpuSynthPlantedQueriesClean(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length,
                                                    uint64_t seedQ, const uint8_t *dText, uint64_t textLength, void *stream);

/* `count` k-mers copied from the UNIQUE sequence of the genome-shaped text of awfmGpuSynthGenomeText(textSeed): the first of up
 * to 64 seeded offsets whose window lies in blocks that are no repeat's and holds only a,c,g,t; dOffsetsOut (may be NULL)
 * gets the offset each k-mer was taken from */
enum AwFmReturnCode awfmGpuSynthPlantedQueriesUnique(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length, uint64_t seedQ,
                                                     const uint8_t *dText, uint64_t textLength, uint64_t textSeed,
                                                     uint64_t *dOffsetsOut, void *stream);

/* mixed-length set: lengths lo..hi, even ids random, odd ids copied from the text.
 * Lengths first; the caller turns them into count+1 exclusive-scan offsets; then the characters. */
enum AwFmReturnCode awfmGpuSynthMixedLengths(uint64_t *dLengths, uint64_t first, uint64_t count, uint32_t lo,
                                             uint32_t hi, uint64_t seedQ, void *stream);
enum AwFmReturnCode awfmGpuSynthMixedQueries(uint8_t *dOut, const uint64_t *dOffsets, uint64_t first, uint64_t count,
                                             uint64_t seedQ, const uint8_t *dText, uint64_t textLength, int amino,
                                             void *stream);

/* ---- seed-bucket sharding of dense-hit batches over the GPUs of a node (round 6) ----
 * The reference treats the k-mers of a batch as independent (ref src/AwFmParallelSearch.c:103-129), so ANY split of a batch
 * over index replicas gives the same results.  A contiguous split of the batch hands every rank a THIN slice of the seed order
 * (one k-mer per two block lines where the whole batch has four per line): its search re-reads nothing from the L2 and an 8-way
 * split of 10^8 k-mers drawn from the text scales to 0.67.  These calls let N ranks split the ORDER instead: every rank
 * orders its own contiguous shard (awfmGpuOrderKmers: the counting and the partition pass, records {rest of the code string,
 * number in the WHOLE batch} in bucket order), the ranks exchange the records bucket range by bucket range (rank j gets the
 * buckets [j B / N, (j + 1) B / N) of everybody: contiguous slices, whose per-bucket runs the receiver puts together bucket by
 * bucket -- the caller's all-to-all; avxwindowfmindex_amd/dist.py does it over torch.distributed), and every rank searches the
 * dense N-th of the order it then holds (awfmGpuSearchOrderedRecords: results in that order, {number in the whole batch,
 * range}, hit offsets and positions by the calls that follow awfmGpuSearchHitsInOrder).  K-mers the seed-order kernel does not
 * take (ambiguity characters) stay with the rank that holds their characters (awfmGpuSearchGeneralRecords).
 * dBucketStart: awfmGpuOrderBuckets() + 3 words -- [b] = records before bucket b, [buckets] = records the seed-order kernel
 * takes, [buckets + 1] = numQueries, [buckets + 2] = the k-mers left to the general kernel (the records' tail). */
uint32_t awfmGpuOrderBuckets(const AwFmGpuIndex *g, uint32_t fixedLength, uint64_t totalQueries); /* 0: not a batch for it */
enum AwFmReturnCode awfmGpuOrderKmers(AwFmGpuIndex *g, const uint8_t *dChars, uint32_t fixedLength, uint64_t numQueries,
                                      uint64_t firstNumber, uint64_t totalQueries, uint64_t *dRecords, uint32_t *dBucketStart,
                                      void *stream);
/* What a rank holds after the exchange, put in bucket order, in one launch: `dReceived` = the numSlices slices the ranks sent
 * (slice j begins at record dSliceAt[j]; each holds the sender's records of the buckets [firstBucket, endBucket), bucket by
 * bucket), dSliceStarts[j * (endBucket - firstBucket + 1) + b] = records of slice j before its bucket firstBucket + b (the
 * last one: the slice's length).  dRecords gets the runs of a bucket from all slices next to each other, slice by slice (any
 * order inside a bucket will do), dBucketStart the buckets + 3 words awfmGpuSearchOrderedRecords wants for an array that holds
 * those buckets only. */
enum AwFmReturnCode awfmGpuMergeBucketRuns(AwFmGpuIndex *g, const uint64_t *dReceived, const uint64_t *dSliceAt, const uint32_t *dSliceStarts,
                                           uint32_t numSlices, uint32_t firstBucket, uint32_t endBucket, uint32_t buckets, uint64_t *dRecords,
                                           uint32_t *dBucketStart, void *stream);
/* the buckets [firstBucket, endBucket) of `dRecords` (bucket order, dBucketStart as above); entry e of the outputs belongs to
 * record dBucketStart[firstBucket] + e */
enum AwFmReturnCode awfmGpuSearchOrderedRecords(AwFmGpuIndex *g, const uint64_t *dRecords, const uint32_t *dBucketStart,
                                                uint32_t firstBucket, uint32_t endBucket, uint32_t fixedLength, uint64_t totalQueries,
                                                uint32_t *dOrderKmers, struct AwFmSearchRange *dOrderRanges, void *stream);
/* ... with the 32-bit counts in that order as well (dOrderCounts may be NULL), as awfmGpuSearchHitsInOrderCounts */
enum AwFmReturnCode awfmGpuSearchOrderedRecordsCounts(AwFmGpuIndex *g, const uint64_t *dRecords, const uint32_t *dBucketStart,
                                                      uint32_t firstBucket, uint32_t endBucket, uint32_t fixedLength, uint64_t totalQueries,
                                                      uint32_t *dOrderKmers, struct AwFmSearchRange *dOrderRanges, uint32_t *dOrderCounts,
                                                      void *stream);
/* the tail of a shard's own records through the general kernel: entries [dBucketStart[buckets], numQueries) of the outputs */
enum AwFmReturnCode awfmGpuSearchGeneralRecords(AwFmGpuIndex *g, const uint8_t *dChars, uint32_t fixedLength, uint64_t numQueries,
                                                uint64_t firstNumber, uint64_t totalQueries, const uint64_t *dRecords,
                                                const uint32_t *dBucketStart, uint32_t *dOrderKmers,
                                                struct AwFmSearchRange *dOrderRanges, void *stream);

#ifdef __cplusplus
}
#endif
#endif
