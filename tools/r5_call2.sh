#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_pangu_numerics_gpu.py -q -x -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -40
timeout 300 python -m pytest tests/test_pangu_gpu.py -q -x -m gpu -k "one_term_block or f16x2m" 2>&1 | tail -3
SKYRIM_PANGU_CALIBRATION=off SKYRIM_PANGU_ROUNDING=nearest timeout 300 python tools/mode_times.py f16x2m f16x1m 2>&1 | grep -v amdgpu.ids
