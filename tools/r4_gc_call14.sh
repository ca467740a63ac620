#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/r4_gc_call13.sh 2>&1 | head -8
bash tools/r4_gc_call10.sh 2>&1 | grep "^node"
