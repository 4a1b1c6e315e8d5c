"""Compile-time guard for the two kernels the headline depends on: the default search kernel and the default
walk kernel must keep 8 waves per SIMD (<= 64 VGPRs), spill nothing, and keep their LDS footprint (the code is
cross-compiled for gfx950 with hipcc; no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "avxwindowfmindex_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernel_metadata(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    text = ""
    for source in ("awfm_gpu.hip", "awfm_gpu_locate.hip", "awfm_gpu_ordered.hip", "awfm_gpu_mixed.hip"):
        out = tmp_path_factory.mktemp("isa") / (source + ".s")
        subprocess.check_call([HIPCC, "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-Wno-unused-function", "-S", "--cuda-device-only", "-o", str(out),
                               os.path.join(CSRC, source)], stderr=subprocess.DEVNULL)
        text += out.read_text()
    meta = {}
    for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.name:\s+(\S+)\n(?:.*\n)*?"
                         r"\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n"
                         r"\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(2)] = {"lds": int(m.group(1)), "scratch": int(m.group(3)), "vgpr": int(m.group(4)),
                            "spill": int(m.group(5))}
    assert meta, "no kernel metadata found in the assembly"
    return meta


def _one(meta, pattern):
    names = [n for n in meta if re.search(pattern, n)]
    assert len(names) == 1, (pattern, names)
    return meta[names[0]]


def test_default_search_kernel_keeps_full_occupancy(kernel_metadata):
    # searchKernel<AMINO=false, G=4, CSR=false, TALLY=false, NARROW=true, INDIRECT=false, PAIR=false>
    k = _one(kernel_metadata, r"[0-9]searchKernelILb0ELi4ELb0ELb0ELb1ELb0ELb0EE")
    assert k["vgpr"] <= 64 and k["spill"] == 0 and k["scratch"] == 0
    assert k["lds"] <= 16 * 1024  # 8 workgroups of 256 threads per CU must fit the 160 KB of LDS
    # the hits-only variant with pair steps holds two blocks of 32 B per lane: 6 waves per SIMD
    k = _one(kernel_metadata, r"[0-9]searchKernelILb0ELi4ELb0ELb0ELb1ELb0ELb1EE")
    assert k["vgpr"] <= 80 and k["spill"] == 0 and k["scratch"] == 0


def test_default_walk_kernel_keeps_full_occupancy(kernel_metadata):
    # walkKernel<AMINO=false, G=4, POW2=true, NARROW=true, PAIR>: one LF step per read, and two (the default)
    # (..., PERLANE = 4: batches of 16 hits; the pair variant also exists with batches of 4 for short hit lists)
    for pair in ("Lb0ELj4EE", "Lb1ELj4EE", "Lb1ELj1EE"):
        k = _one(kernel_metadata, r"walkKernelILb0ELi4ELb1ELb1E" + pair)
        assert k["vgpr"] <= 64 and k["spill"] == 0 and k["scratch"] == 0, pair
        assert k["lds"] <= 1024  # small arrays that are indexed dynamically get moved to LDS by hipcc: must not happen


def test_ordered_search_kernels_keep_full_occupancy(kernel_metadata):
    # orderedSearchKernel<G=4, NARROW=true, VARLEN, PAIR, TOUCH=false, BUCKET, LIST=false>: bucketed 8-byte records (the
    # default of fixed-length batches), the 16-byte-record and the CSR variant, with one step per block read and with two
    # (pair image, the default)
    for varlen, bucket in (("Lb0E", "Lb1E"), ("Lb0E", "Lb0E"), ("Lb1E", "Lb0E")):
        for pair in ("Lb0E", "Lb1E"):
            k = _one(kernel_metadata, r"orderedSearchKernelILi4ELb1E" + varlen + pair + "Lb0E" + bucket + "Lb0EE")
            # mixed-length, pair and bucketed variants: 7 waves per SIMD (72 registers); the others 8.  The bucketed pair
            # variant holds the next chunk's codes and table entry as well and may spill two registers (measured faster
            # than the spill-free build at 80 registers and 6 waves: awfm_ordered_kernel.h)
            limit = 72 if (bucket == "Lb1E" or varlen == "Lb1E" or pair == "Lb1E") else 64
            spills = 2 if (bucket == "Lb1E" and pair == "Lb1E") else 0
            assert k["vgpr"] <= limit and k["spill"] <= spills and k["scratch"] <= 8 * spills, varlen + pair + bucket
            assert k["lds"] <= 16 * 1024  # static; the pair variant adds 64 B per 2^23 positions of dynamic LDS


def test_bucketed_list_variant_resources(kernel_metadata):
    # orderedSearchKernel<4, true, false, PAIR, false, true, LIST=true>: the bucketed variant that collects its hits for
    # the list per wave (2.5 KB of LDS more); the pair one spills a few registers more than its twin without the list
    for pair, spills in (("Lb0E", 0), ("Lb1E", 6)):
        k = _one(kernel_metadata, r"orderedSearchKernelILi4ELb1ELb0E" + pair + "Lb0ELb1ELb1EE")
        assert k["vgpr"] <= 72 and k["spill"] <= spills and k["scratch"] <= 8 * spills and k["lds"] <= 16 * 1024, pair


def test_no_search_or_walk_variant_uses_scratch(kernel_metadata):
    bucketed_pair = r"orderedSearchKernelILi4ELb[01]ELb0ELb1ELb[01]ELb1ELb[01]EE"  # a few spilled registers by choice (see above; 32- and 64-bit positions, and their instrumented twins)
    bad = {n: v for n, v in kernel_metadata.items()
           if ("searchKernel" in n or "walkKernel" in n or "orderedSearchKernel" in n) and v["scratch"]
           and not re.search(bucketed_pair, n)}
    assert not bad, bad


def test_lookup_search_kernels_resources(kernel_metadata):
    """the kernels that look the deeper table up and search what is still alive (round 4): lookupSearchKernel<21> -- the
    dominant kernel of the headline batch -- must keep 7 waves per SIMD (72 registers; it spills 14 in its decode phase by
    choice: 80 registers and 6 waves left 25 spilled and fewer lookups in flight), aminoLookupSearchKernel<10> spills nothing;
    the first keeps its LDS small enough for 7 workgroups per CU beside the pair image's superblock bases"""
    k = _one(kernel_metadata, r"lookupSearchKernelILj21ELb1EEE")
    assert k["vgpr"] <= 72 and k["spill"] <= 16 and k["scratch"] <= 64 and k["lds"] <= 12 * 1024
    # its 64-bit instantiation (round 6: images of 2^32 positions and more): 6 waves per SIMD, next to nothing spilled
    k = _one(kernel_metadata, r"lookupSearchKernelILj21ELb0EEE")
    assert k["vgpr"] <= 80 and k["spill"] <= 4 and k["scratch"] <= 16 and k["lds"] <= 16 * 1024
    # (round 5: a slot for every k-mer of a round -- 256 per wave, 25 KB of LDS per workgroup, and groups that refill as they
    # finish: 6 workgroups per CU, 6 waves per SIMD)
    k = _one(kernel_metadata, r"[0-9]aminoLookupSearchKernelILj10ELb1EEE")
    assert k["vgpr"] <= 80 and k["spill"] == 0 and k["scratch"] == 0 and k["lds"] <= 26 * 1024
    # its 64-bit instantiation (round 6): 4 waves per SIMD, nothing spilled, 64-bit ranges in the slots
    k = _one(kernel_metadata, r"[0-9]aminoLookupSearchKernelILj10ELb0EEE")
    print("aminoLookupSearchKernel<10, false>", k)
    assert k["vgpr"] <= 128 and k["spill"] == 0 and k["scratch"] == 0 and k["lds"] <= 34 * 1024
    # mixedLookupSearchKernel (mixed-length batches): four decoded k-mers and their entries per lane, 256 survivor slots per
    # wave (every k-mer of a round may survive): 5 waves per SIMD (<= 96 registers; builds held to 80 measured slower), no
    # spills, 5 workgroups' LDS (30 KB each) per CU
    k = _one(kernel_metadata, r"[0-9]mixedLookupSearchKernelILb1EE")
    assert k["vgpr"] <= 96 and k["spill"] == 0 and k["scratch"] == 0 and k["lds"] <= 32 * 1024
