"""examples/kmer_locate.c is a C program written against the reference's public header only; it is compiled with
gcc against include/AwFmIndex.h + libawfmindex_amd.so.  Without a GPU it must fail loudly; on the GPU its digest of
all counts and position lists must equal the one computed here from the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK = (1 << 64) - 1


def _compile(tmp_path):
    exe = str(tmp_path / "kmer_locate")
    lib_dir = os.path.join(ROOT, "avxwindowfmindex_amd")
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "examples", "kmer_locate.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-lawfmindex_amd",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def _splitmix(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return state, z ^ (z >> 31)


def _fnv(h, data):
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & MASK
    return h


def test_example_fails_loudly_without_a_gpu(awfm, tmp_path):
    from avxwindowfmindex_amd import _lib
    if _lib.lib().awfmGpuDeviceCount() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([_compile(tmp_path), "5000", "50", "10"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and "no CPU search path" in out.stderr


@pytest.mark.gpu
def test_example_digest_equals_the_oracle(oracle, awfm, require_gpu, tmp_path):
    n, count, k = 60000, 4000, 12
    out = subprocess.run([_compile(tmp_path), str(n), str(count), str(k)], cwd=tmp_path, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    # the program's generator, restated
    letters = b"acgt"
    state = 12345
    text = bytearray(n)
    for i in range(n):
        state, r = _splitmix(state)
        text[i] = letters[r & 3]
    kmers = []
    for i in range(count):
        if i & 1:
            state, r = _splitmix(state)
            at = r % (n - k)
            kmers.append(bytes(text[at:at + k]))
        else:
            q = bytearray(k)
            for j in range(k):
                state, r = _splitmix(state)
                q[j] = letters[r & 3]
            kmers.append(bytes(q))
    oi = oracle.Index.from_text(bytes(text), oracle.DNA, 8, 8)
    sp, ep, cnt, _ = oi.search_list(kmers)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    digest = 0xCBF29CE484222325
    for c in cnt:
        digest = _fnv(digest, int(c).to_bytes(4, "little"))
    digest = _fnv(digest, np.ascontiguousarray(pos, dtype="<u8").tobytes())
    expect = f"kmers {count} counted {int(cnt.sum())} located {len(pos)} digest {digest:016x}"
    assert out.stdout.strip() == expect
