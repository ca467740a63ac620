#!/bin/bash
# GPU box: rocprofv3 counter passes over a short bench.py run, one directory per pass under gpurun_out/ (summarise with
# tools/pmc_summary.py).  Counters go in their own runs with --kernel-trace only (MI355X_MICROARCH.md, "rocprofv3 PMC slots":
# SQ has 8 slots per pass, FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   bash tools/pmc_collect.sh <tag> [bench.py arguments...]
set -u
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export SKYRIM_PANGU_CALIBRATION=off   # the one-time calibration launches (PanguEngine.load_params) stay out of the per-step totals; timings do not depend on it
export SKYRIM_PANGU_GUARD=off         # ... and so do the load-time guard's (it would also replace the uncalibrated plan by three terms)
run() {   # <pass name> <counters...> --
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  local out=gpurun_out/pmc_${tag}_${name}
  rm -rf "$out"
  timeout 600 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$out" -o p -- python bench.py "$@" > "$out.log" 2>&1
  echo "pass $name rc=$? $(ls $out 2>/dev/null | tr '\n' ' ')"
}
ARGS=("$@" --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-alt-modes --no-models)
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- "${ARGS[@]}"
run sq2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -- "${ARGS[@]}"
run fetch FETCH_SIZE -- "${ARGS[@]}"
run write WRITE_SIZE -- "${ARGS[@]}"
