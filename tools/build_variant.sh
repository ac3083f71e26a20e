#!/bin/bash
# Experiment build of ONE kernel file with extra defines: tools/build_variant.sh NAME file.hip "-DX=1 ..." -> tvqaplus_amd/libstage_hip_NAME.so
# (use with LIB=tvqaplus_amd/libstage_hip_NAME.so python tools/k1_bwd_times.py)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; DEFS=$3
base=$(basename $SRC .hip)
make -j8 >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $DEFS -c tvqaplus_amd/csrc/$base.hip -o build/var_${NAME}.o
objs=$(ls build/*.o | grep -v "build/var_" | grep -v "build/t2_" | grep -v "build/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tvqaplus_amd/libstage_hip_${NAME}.so $objs build/var_${NAME}.o
echo built tvqaplus_amd/libstage_hip_${NAME}.so
