#!/bin/bash
# (experiment library: compiled with -DPGRF_ABLATION_BUILD, reports a negative pgrf_abi_version(); load it with
#  PANOGRF_HIP_LIB=... PGRF_ALLOW_ABLATION_LIB=1)
# build a variant of the library with extra -D flags for one render-pass instance (default: the KT=2,V=2 production
# kernel, f32):   tools/build_variant.sh NAME "-DX=1 ..." [KT] [V] [DBG] [PREC]
set -e
cd "$(dirname "$0")/../panogrf_amd/csrc"
NAME=$1; FLAGS=$2; KT=${3:-2}; V=${4:-2}; DBG=${5:-0}; PREC=${6:-0}; INST=${KT}_${V}_${DBG}_${PREC}
mkdir -p build/var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I../../include -DPGRF_ABLATION_BUILD $FLAGS -DPGRF_KT=$KT -DPGRF_V=$V -DPGRF_DBG=$DBG -DPGRF_PREC=$PREC -Rpass-analysis=kernel-resource-usage -c render_pass_inst.hip -o build/var_$NAME/rp_$INST.o 2>&1 | grep -E "VGPRs:|VGPRs Spill|Occupancy|error" | tr '\n' ' '
echo
OBJS=$(ls build/*.o | grep -v rp_$INST.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build/var_$NAME/rp_$INST.o -o ../libpanogrf_hip_$NAME.so
