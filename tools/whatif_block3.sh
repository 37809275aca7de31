#!/bin/bash
# Timing experiments on the layer3 kernel (results are WRONG in these builds): which resource bounds the K walk?
# Rebuilds block_fused.hip (debug-hooks library) with MAGAT_WHATIF_* switches on the GPU box and prints the per-wave phase
# cycles (tools/chain_phase_probe.py) - CYCLES, not wall time: fake operands change the power draw and with it the clock.
#   WHATIF_LIST="NONE NO_LDS NO_W NO_LDS+NO_W" bash tools/whatif_block3.sh
cd $(dirname $0)/..
for V in ${WHATIF_LIST:-NONE NO_LDS NO_W}; do
  touch magat_pathplanning_amd/csrc/block_fused.hip
  F="-DMAGAT_EXPERIMENT_BUILD"; for X in ${V//+/ }; do [ $X = NONE ] || F="$F -DMAGAT_WHATIF_$X"; done
  MAGAT_EXTRA_FLAGS="$F" python -m magat_pathplanning_amd.build_native --debug > /dev/null 2>&1 || { echo "build failed $V"; continue; }
  echo "== $V"
  MAGAT_ALLOW_EXPERIMENT_BUILD=1 MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_debug.so python tools/chain_phase_probe.py 2>&1 | tail -20
done
touch magat_pathplanning_amd/csrc/block_fused.hip
python -m magat_pathplanning_amd.build_native --debug > /dev/null 2>&1
