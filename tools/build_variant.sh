#!/bin/bash
# Builds lib/libmagat_hip_<name>.so: the release objects with ONE source recompiled under extra -D flags (same-box A/B runs of
# kernel experiments: MAGAT_LIB_PATH=.../libmagat_hip_<name>.so).   tools/build_variant.sh <name> <source.hip> "<flags>"
set -e
NAME=$1; SRC=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/magat_pathplanning_amd/lib
mkdir -p $L/obj_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-inline-asm $( [ "$SRC" == block_fused.hip ] && echo "-mllvm -amdgpu-mfma-vgpr-form" ) $FLAGS \
  -c $R/magat_pathplanning_amd/csrc/$SRC -o $L/obj_$NAME/${SRC%.hip}.o
OBJS=""
for o in $L/obj/*.o; do
  b=$(basename $o)
  if [ "$b" == "${SRC%.hip}.o" ]; then OBJS="$OBJS $L/obj_$NAME/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $L/libmagat_hip_$NAME.so
echo $L/libmagat_hip_$NAME.so
