#!/bin/bash
# tools/ab_perf_sheet.sh <grep pattern> lib.so...: the perf sheet lines matching the pattern for every library build, same box
PAT=$1; shift
for L in "$@"; do echo "== $L"; RDYN_LIB_PATH=$PWD/$L python3 tools/perf_sheet.py 2>/dev/null | grep -E "$PAT"; done
