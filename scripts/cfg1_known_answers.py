#!/usr/bin/env python3
"""BASELINE configs[0] (1 M random 12-mers against a 1 Mbp synthetic index) through the CPU oracle: prints the
constants tests/test_gpu_configs.py::CFG1 holds.  CPU only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avxwindowfmindex_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

txt = synth.text(1, 1_000_000)
chars, offsets = synth.fixed_csr(synth.random_queries(101, 1_000_000, 12))
for k in (12, 8):
    oi = O.Index.from_text(txt.tobytes(), O.DNA, 8, k)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=os.cpu_count() or 1)
    ho, pos, _ = oi.batch_locate(sp, ep, threads=os.cpu_count() or 1)
    print(f"seed_k {k}: present {int((cnt > 0).sum())} hits {int(cnt.sum())} counts_fnv {O.fnv1a(cnt):#x} "
          f"ranges_fnv {O.fnv1a(np.stack([sp, ep], 1)):#x} positions_fnv {O.fnv1a(pos):#x}")
