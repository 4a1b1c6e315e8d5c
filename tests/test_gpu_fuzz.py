"""A few seconds of scripts/fuzz_ordered.py: the ordered hits-only search against the general kernel over random
index sizes, seed depths, deeper tables, fixed and mixed k-mer lengths, ambiguity characters and alignments."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,wide", [(11, False), (12, True)])
def test_ordered_search_differential_fuzz(require_gpu, seed, wide):
    """wide: the ordered search and the locate under test run with 64-bit positions, the general kernel they are
    compared with keeps 32-bit ones"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_ordered.py"), "6", str(seed)],
                         env=dict(os.environ, FUZZ_WIDE="1" if wide else "0", **({"AWFM_GPU_DIAG": "nuc_super_shift=auto"} if wide else {})),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
