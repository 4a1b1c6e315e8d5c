#!/bin/bash
cd "$(dirname "$0")/.."
for c in 8388608 16777216 33554432 50000000; do for w in planted random; do echo "== chunk $c $w"; python scripts/stream_probe.py 3.1e9 1e8 $w $c 2>&1 | grep "^run [123]"; done; done
