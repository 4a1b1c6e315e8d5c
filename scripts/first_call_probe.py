#!/usr/bin/env python3
"""The drop-in user's first call in a FRESH process: awFmReadIndexFromFile of a GRCh38-sized .awfmi, then the first
awFmParallelSearchLocate (device image upload, pair image, deeper table, full suffix array, the search), then a second call.
Two child processes -- one builds and writes the index, the other reads and searches it; the parent never touches the GPU.
usage: scripts/first_call_probe.py [text length] [file]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 3_100_000_000
path = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "/tmp/first_call_probe.awfmi"

if "--write" in sys.argv or "--read" in sys.argv:
    sys.path.insert(0, ROOT)
    import numpy as np
    from avxwindowfmindex_amd import _lib, api, synth
    if "--write" in sys.argv:
        import torch
        L = _lib.lib()
        d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
        L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None)
        t0 = time.time()
        ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n, file_src=path)
        print(json.dumps({"build_and_write_s": round(time.time() - t0, 2), "file_bytes": os.path.getsize(path)}), flush=True)
    else:
        import ctypes as C
        t0 = time.time()
        ix = api.read_index_from_file(path)
        t1 = time.time()
        m, K = 1_000_000, 21
        q = np.ascontiguousarray(synth.random_queries(102, m, K))
        lst = api.KmerSearchList(m)
        arr = np.ctypeslib.as_array(C.cast(lst.ptr.contents.kmerSearchData, C.POINTER(C.c_uint64)), shape=(m, 4))
        arr[:, 0] = q.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(K)
        arr[:, 1] = K
        lst.ptr.contents.count = m
        t2 = time.time()
        rc = api.parallel_search_locate(ix, lst, 32)
        t3 = time.time()
        api.parallel_search_locate(ix, lst, 32)
        t4 = time.time()
        g = api.GpuIndex(ix, acquire=True)
        print(json.dumps({"read_index_from_file_s": round(t1 - t0, 2), "first_locate_s": round(t3 - t2, 3), "second_locate_s": round(t4 - t3, 4),
                          "rc": rc, "kmers": m, "image_bytes": g.device_bytes, "deep_seed_k": g.deep_seed_k, "deep_seed_build": g.deep_seed_build,
                          "dense_sa": g.has_dense_sa, "dense_sa_build_s": round(g.dense_sa_build_s, 3)}), flush=True)
        g.handle = None
    sys.exit(0)

for mode in ("--write", "--read"):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), path, mode], capture_output=True, text=True)
    print(mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:], flush=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-2000:])
os.remove(path)
