#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs -> index-array digests, {sp,ep,count} and hit positions.

The vectors come from the CPU oracle after it has been pinned (tests/test_oracle_pin.py): the reference
itself cannot be built in the authoring container (empty FastaVector / libdivsufsort submodules).
Re-run after an intentional change of the generators only; the fixtures are data, the script is the
provenance.  Usage: python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avxwindowfmindex_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [
    # name, alphabet, text seed, n, sa ratio, seed k, query seed, queries, min len, max len
    ("dna_toy", "dna", 31, 29, 1, 3, 131, 200, 1, 12),
    ("dna_4k", "dna", 32, 4096, 3, 4, 132, 1500, 1, 40),
    ("dna_64k", "dna", 33, 65536, 8, 8, 133, 2500, 5, 40),
    ("dna_300k_r255", "dna", 34, 300000, 255, 6, 134, 1500, 6, 30),
    ("amino_toy", "amino", 41, 31, 1, 1, 141, 200, 1, 8),
    ("amino_64k", "amino", 42, 65536, 8, 3, 142, 2500, 2, 25),
    ("amino_200k_r16", "amino", 43, 200000, 16, 4, 143, 1500, 2, 20),
]


def build_case(case):
    name, alpha, tseed, n, ratio, k, qseed, nq, lo, hi = case
    letters = synth.AMINO_ALPHABET if alpha == "amino" else synth.DNA_ALPHABET
    txt = synth.text(tseed, n, letters).copy()
    if n > 100:
        txt[10:14] = ord("x")
        txt[n // 2] = ord("N") if alpha == "dna" else ord("b")
    chars, offsets = synth.mixed_queries(qseed, nq, txt, letters, lo, min(hi, n))
    chars = chars.copy()
    rng = np.random.default_rng(qseed)
    amb = rng.random(chars.size) < 0.01
    chars[amb] = ord("x") if alpha == "dna" else ord("z")
    up = rng.random(chars.size) < 0.2
    chars[up] = chars[up] & 0xDF
    return txt, chars, offsets


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for case in CASES:
        name, alpha, tseed, n, ratio, k, qseed, nq, lo, hi = case
        txt, chars, offsets = build_case(case)
        oi = O.Index.from_text(txt.tobytes(), O.AMINO if alpha == "amino" else O.DNA, ratio, k)
        sp, ep, cnt, tally = oi.batch_search(chars, offsets)
        hit_off, pos, t2 = oi.batch_locate(sp, ep)
        digests = np.array([O.fnv1a(oi.blocks()), O.fnv1a(oi.prefix_sums()), O.fnv1a(oi.seed_table()),
                            O.fnv1a(oi.packed_sa())], dtype=np.uint64)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), sp=sp, ep=ep, count=cnt, hit_offsets=hit_off,
                            positions=pos, digests=digests, bwt_length=np.uint64(oi.bwt_length),
                            prefix_sums=oi.prefix_sums(),
                            tally=np.array([tally["steps"], tally["blocks"], tally["seeded"], t2["hits"], t2["lfSteps"]],
                                           dtype=np.uint64))
        print(name, "queries", nq, "hits", int(hit_off[-1]), "bytes", os.path.getsize(os.path.join(out_dir, name + ".npz")))


if __name__ == "__main__":
    main()
