#!/bin/bash
# CPU test suite with the host C sources built under AddressSanitizer + UBSan (no GPU needed; the GPU pool
# refuses sanitizer runs).  Prints every sanitizer report; exit status 1 if there is one.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/avxwindowfmindex_amd/csrc" -s asan || exit 2
make -C "$ROOT/oracle" -s liboracle.so || exit 2
LOG=$(mktemp)
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
AWFM_LIB_PATH="$ROOT/avxwindowfmindex_amd/csrc/build/asan/libawfmindex_amd_asan.so" \
  python3 -m pytest "$ROOT/tests" -q -s -m "not gpu" -p no:cacheprovider > "$LOG" 2>&1
tail -2 "$LOG"
REPORTS=$(grep -E "runtime error|AddressSanitizer" "$LOG" | sort | uniq -c)
if [ -n "$REPORTS" ]; then echo "$REPORTS"; exit 1; fi
echo "no sanitizer reports"
