#!/bin/bash
set -u
cd "$(dirname "$0")/.."
for v in dbg1 dbg4 dbg5; do echo "== $v"; SKYRIM_GRAPHCAST_LIB=skyrim_amd/lib/variants/libgc_$v.so bash tools/r4_gc_call10.sh 2>&1 | grep "^node"; done
