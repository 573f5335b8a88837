#!/bin/bash
# (experiment library: compiled with -DPGRF_ABLATION_BUILD, reports a negative pgrf_abi_version(); load it with
#  PANOGRF_HIP_LIB=... PGRF_ALLOW_ABLATION_LIB=1)
# build a variant of the library with extra -D flags for ONE source file (cost_volume | conv3d | conv3d_wino | aux_kernels | capi):
#   tools/build_variant_src.sh NAME SRC "-DX=1 ..."   -> panogrf_amd/libpanogrf_hip_NAME.so  (select with PANOGRF_HIP_LIB)
set -e
cd "$(dirname "$0")/../panogrf_amd/csrc"
NAME=$1; SRC=$2; FLAGS=$3
[ "$SRC" = conv3d_wino ] && FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form=1"
mkdir -p build/var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I../../include -DPGRF_ABLATION_BUILD $FLAGS -c $SRC.hip -o build/var_$NAME/$SRC.o
OBJS=$(ls build/*.o | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build/var_$NAME/$SRC.o -o ../libpanogrf_hip_$NAME.so
echo built libpanogrf_hip_$NAME.so
