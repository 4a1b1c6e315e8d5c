/* Every environment variable the library reads, in ONE table, read through ONE function (round 5's verdict, item 6: "the
 * release build reads <= 20 documented knobs").  C and HIP sources both include this; nothing else in csrc/ calls getenv.
 * A knob selects a device, bounds memory, switches an accelerator or a front end on / off for comparison runs, or asks for
 * reporting -- results never depend on one (tests/test_gpu_parity.py runs the parity cases under each).  INTEGRATION.md
 * ("Environment") carries the same table for users; tests/test_knobs.py holds the two against each other. */
#ifndef AWFM_KNOBS_H
#define AWFM_KNOBS_H
#include <stdlib.h>

enum AwFmKnob {
  AWFM_KNOB_DEVICE = 0,
  AWFM_KNOB_DEVICES,
  AWFM_KNOB_HIT_BUDGET_BYTES,
  AWFM_KNOB_AOS_CHUNK,
  AWFM_KNOB_ORDERED,
  AWFM_KNOB_LOOKUP_FIRST,
  AWFM_KNOB_LOOKUP_PREDICT,
  AWFM_KNOB_MIXED_LOOKUP,
  AWFM_KNOB_AMINO_LOOKUP,
  AWFM_KNOB_EXACT_LOOKUP,
  AWFM_KNOB_PAIR,
  AWFM_KNOB_DEEP_SEED_K,
  AWFM_KNOB_AMINO_DEEP_SEED_K,
  AWFM_KNOB_DEEP_NEXT,
  AWFM_KNOB_DENSE_SA,
  AWFM_KNOB_FORCE_WIDE,
  AWFM_KNOB_TIME_ORDERED,
  AWFM_KNOB_VERBOSE,
  AWFM_KNOB_HOST_BUILD,
  AWFM_KNOB_DIAG,
  AWFM_KNOB_COUNT
};

struct AwFmKnobEntry {
  const char *name, *what;
};

static const struct AwFmKnobEntry awfmKnobTable[AWFM_KNOB_COUNT] = {
    {"AWFM_GPU_DEVICE", "ordinal of the device an image is placed on when the caller names none (default 0)"},
    {"AWFM_GPU_DEVICES", "comma-separated ordinals the drop-in entry points spread a host k-mer list over (`all` or a list; default: three lanes on AWFM_GPU_DEVICE)"},
    {"AWFM_GPU_HIT_BUDGET_BYTES", "bytes of located positions resident on the device at once per caller (default: a quarter of the free memory shared by three callers)"},
    {"AWFM_GPU_AOS_CHUNK", "k-mers per chunk of the host-list pipeline behind awFmParallelSearchLocate / Count (default 2^20)"},
    {"AWFM_GPU_ORDERED", "0: nucleotide batches never take the seed-order search; 1: whenever it applies; unset: >= 2^23 k-mers against >= 2^28 positions"},
    {"AWFM_GPU_LOOKUP_FIRST", "0: no lookup kernel in front of the seed-order search; 1: always; unset: by a sample of the batch"},
    {"AWFM_GPU_LOOKUP_PREDICT", "0: every search launches both front ends and lets the sample choose on the device"},
    {"AWFM_GPU_MIXED_LOOKUP", "0: mixed-length batches go to the general kernel; 1: the per-length tables whenever they apply"},
    {"AWFM_GPU_AMINO_LOOKUP", "0: amino batches go to the general kernel; 1: the amino lookup kernel whenever it applies"},
    {"AWFM_GPU_EXACT_LOOKUP", "0: awfmGpuSearch never goes through the device-only tables; 1: whenever they apply"},
    {"AWFM_GPU_PAIR", "0: no pair image (one step per block read everywhere)"},
    {"AWFM_GPU_DEEP_SEED_K", "depth of the device-only deeper seed table of a nucleotide image; 0: none; unset: 14..16 by size and free memory"},
    {"AWFM_GPU_AMINO_DEEP_SEED_K", "the same for amino images (unset: up to 7)"},
    {"AWFM_GPU_DEEP_NEXT", "0: deeper-table entries without the next-step bits"},
    {"AWFM_GPU_DENSE_SA", "0: no full suffix array on the device (locate walks LF to a sampled position); 1 / auto: built when it fits"},
    {"AWFM_GPU_FORCE_WIDE", "1: an image below 2^32 positions runs the 64-bit instantiations (tests)"},
    {"AWFM_GPU_TIME_ORDERED", "HIP events around the dominant kernel of a seed-order search (awfmGpuLastOrderedKernelMs reads them)"},
    {"AWFM_VERBOSE", "build and image-construction timings on stderr"},
    {"AWFM_HOST_BUILD", "1: awFmCreateIndex builds on the host even when a device is present"},
    {"AWFM_GPU_DIAG", "\"key=value,...\": test and diagnostics hooks, none selects a faster path (include/awfm_gpu.h lists the keys)"},
};

/* the knob's value in the environment, or NULL */
static inline const char *awfmKnob(enum AwFmKnob which) { return getenv(awfmKnobTable[which].name); }

#endif
