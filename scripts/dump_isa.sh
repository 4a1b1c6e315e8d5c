#!/bin/bash
# device ISA of awfm_gpu.hip -> gpurun_out/isa/awfm_gpu.s, and one kernel (by mangled-name substring) -> gpurun_out/isa/k.s
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out/isa
cd $ROOT/avxwindowfmindex_amd/csrc
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -S --cuda-device-only -o $ROOT/gpurun_out/isa/awfm_gpu.s awfm_gpu.hip 2>/dev/null
K=${1:-searchKernelILb0ELi4ELb0ELb0ELb1E}
awk -v k="$K" '$0 ~ "^_Z.*"k".*:" {p=1} p {print} p && /s_endpgm/ {exit}' $ROOT/gpurun_out/isa/awfm_gpu.s > $ROOT/gpurun_out/isa/k.s
awk -v k="$K" '$0 ~ "^_Z.*"k".*:" {p=1} p && /NumVgprs|TotalNumSgprs|Occupancy|LDSByteSize|ScratchSize/ {print} p && /Occupancy/ {exit}' $ROOT/gpurun_out/isa/awfm_gpu.s
wc -l $ROOT/gpurun_out/isa/k.s
