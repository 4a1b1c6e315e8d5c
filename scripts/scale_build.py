import sys, time, os
sys.path.insert(0, '/root/repo')
os.environ['AWFM_VERBOSE']='1'
import torch, numpy as np
from avxwindowfmindex_amd import api, _lib, synth
L=_lib.lib()
for n in [int(x) for x in sys.argv[1:]]:
    d = torch.empty(n, dtype=torch.uint8, device='cuda')
    L.awfmGpuSynthText(d.data_ptr(), 0, n, 2, 0, None); torch.cuda.synchronize()
    t0=time.time()
    ix = api.gpu_create_index(d.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n)
    t1=time.time()
    print(f"n={n} build {t1-t0:.2f}s bwt={ix.bwt_length} prefix={ix.prefix_sums()}", flush=True)
    ix.dealloc(); del d; torch.cuda.empty_cache()
