#!/bin/bash
# GPU box: the block kernel with four MFMA accumulator chains per loop (SKP_BLK2_CHAINS4, csrc/fused_block2.hip) against the in-tree build.
# Builds the variant if it is not there (skyrim_amd/lib/variants/ travels with gpurun), then parity + step time of the variant, step time of the default.
cd "$(dirname "$0")/.."
V=skyrim_amd/lib/variants/libskyrim_pangu_ch4.so
if [ ! -f $V ]; then
  C=skyrim_amd/csrc; O=skyrim_amd/lib/obj; mkdir -p skyrim_amd/lib/variants
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSKP_BLK2_CHAINS4 -c $C/fused_block2.hip -o /tmp/fb2_ch4.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC -o $V $O/api.o $O/attention.o $O/aux.o $O/ops_embed_recover.o $O/ops_attn.o $O/ops_mlp.o $O/ops_updown.o $O/fused_mlp.o $O/rowtile.o $O/fused_block.o /tmp/fb2_ch4.o || exit 1
fi
SKYRIM_PANGU_LIB=$PWD/$V python tools/blk2_chains.py 2>&1 | grep -v amdgpu.ids
python tools/blk2_chains.py --no-parity 2>&1 | grep -v amdgpu.ids
