#!/bin/bash
# kernel + memory-copy timeline of one packed-pipeline batch (10^8 planted 21-mers, locate) -> gpurun_out/stream_timeline.txt
# shows what overlaps: uploads run on the DMA engines (MEMORY_COPY_HOST_TO_DEVICE), downloads as __amd_rocclr_copyBuffer kernels
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf /tmp/ps; mkdir -p /tmp/ps "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ps -- python3 "$ROOT/scripts/stream_probe.py" 3.1e9 1e8 ${1:-planted} 2>&1 | grep "^run" > "$ROOT/gpurun_out/stream_timeline.txt"
python3 - "$ROOT/gpurun_out/stream_timeline.txt" <<'PY'
import csv, glob, sys
k = list(csv.DictReader(open(glob.glob('/tmp/ps/*/*kernel_trace.csv')[0])))
m = list(csv.DictReader(open(glob.glob('/tmp/ps/*/*memory_copy_trace.csv')[0])))
end = max(int(r['End_Timestamp']) for r in k)
# the last batch: from the last-but-... find the start of the last run = first unpack/encode after a gap > 5 ms
ev = []
for r in k:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48]))
for r in m:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'DMA ' + r['Direction']))
ev.sort()
runs = [float(l.split()[2]) for l in open(sys.argv[1]) if l.startswith('run')]
last_start = max(e for s, e, n in ev) - int((runs[-1] + 0.5) * 1e6)  # the last batch: its wall time back from the last event
out = open(sys.argv[1], 'a')
out.write("\ntimeline of the last batch (ms; events of 0.1 ms and longer; DMA = copy engine, __amd_rocclr_copyBuffer = shader copy)\n")
for s, e, n in ev:
    if s >= last_start and (e - s) >= 100_000:
        out.write("%8.2f -> %8.2f  (%5.2f)  %s\n" % ((s - last_start) / 1e6, (e - last_start) / 1e6, (e - s) / 1e6, n))
PY
tail -5 "$ROOT/gpurun_out/stream_timeline.txt"
