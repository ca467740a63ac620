#!/bin/bash
# Where does the strided GEMM's time go?  Builds libskyrim_sfno variants with one phase of gemm.h's main loop removed
# (SKP_PROBE_NO_LOAD: operands fetched once; SKP_PROBE_NO_MFMA: no fragment reads / MFMAs; SKP_PROBE_NO_EPILOGUE: no stores) or with
# another tile (SKP_STRIDED_TILE / SKP_STRIDED_BK), then times `bench.py --model sfno` per stage with each.  Results are NOT valid
# forecasts -- timing only.   On the GPU box:  bash tools/sfno_probe.sh
set -e
cd "$(dirname "$0")/.."
C=skyrim_amd/csrc; L=$PWD/skyrim_amd/lib; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w"
build() { hipcc $F "$2" -c $C/sfno_ops.hip -o /tmp/probe_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libskyrim_sfno_$1.so /tmp/probe_$1.o $L/obj/sfno_chain.o $L/obj/aux.o; }
source tools/sfno_variants.sh
run default
for v in NO_LOAD NO_MFMA NO_EPILOGUE; do build $v -DSKP_PROBE_$v; run $v SKYRIM_SFNO_LIB=/tmp/libskyrim_sfno_$v.so; done
build bk64 -DSKP_STRIDED_BK=64; run bk64 SKYRIM_SFNO_LIB=/tmp/libskyrim_sfno_bk64.so
build t128x128 "-DSKP_STRIDED_TILE=128,128,32,2,2"; run t128x128 SKYRIM_SFNO_LIB=/tmp/libskyrim_sfno_t128x128.so
# the fused chains with ONE hidden chunk instead of all of them: what their operand loads / stores alone cost
hipcc $F -DSKP_PROBE_NO_COMPUTE -c $C/sfno_chain.hip -o /tmp/probe_chain.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libskyrim_sfno_nc.so $L/obj/sfno_ops.o /tmp/probe_chain.o $L/obj/aux.o
run chain_1chunk SKYRIM_SFNO_LIB=/tmp/libskyrim_sfno_nc.so
run chain_4waves SKSFNO_CHAIN_WAVES=4
