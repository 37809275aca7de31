#!/bin/bash
# Timing experiments on gat_mfma_kernel (results are WRONG in these builds): rebuilds gat_mfma.hip (debug-hooks library) with
# GM_WHATIF_* switches on the GPU box and prints wave 0's phase cycles (tools/gat_mfma_probe.py).
#   WHATIF_LIST="NONE NOQW NOAW NOYST" bash tools/whatif_gat_mfma.sh
cd $(dirname $0)/..
for V in ${WHATIF_LIST:-NONE NOQW NOAW NOYST}; do
  touch magat_pathplanning_amd/csrc/gat_mfma.hip
  F="-DMAGAT_EXPERIMENT_BUILD"; for X in ${V//+/ }; do [ $X = NONE ] || F="$F -DGM_WHATIF_$X"; done
  MAGAT_EXTRA_FLAGS="$F" python -m magat_pathplanning_amd.build_native --debug > /dev/null 2>&1 || { echo "build failed $V"; continue; }
  echo "== $V"
  MAGAT_ALLOW_EXPERIMENT_BUILD=1 MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_debug.so python tools/gat_mfma_probe.py $PROBE_ARGS 2>&1 | grep -A12 "layer:\|wave 0"
done
touch magat_pathplanning_amd/csrc/gat_mfma.hip
python -m magat_pathplanning_amd.build_native --debug > /dev/null 2>&1
