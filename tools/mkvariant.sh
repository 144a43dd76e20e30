#!/bin/bash
# A tuning build of libcales_hip.so with extra compile flags for ONE translation unit:  tools/mkvariant.sh NAME "FLAGS" UNIT
#   -> tools/variants/libcales_NAME.so (git-ignored; travels to the GPU box); select it with CALES_LIB=tools/variants/libcales_NAME.so
set -e
NAME=$1; FLAGS=$2; UNIT=${3:-k_sgs}
cd "$(dirname "$0")/../cales_amd/csrc"
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value $FLAGS -c $UNIT.hip -o /tmp/var_${NAME}_$UNIT.o
OBJ=""; for o in api host_setup comm_rccl k_stencil k_momrk k_bound k_sgs k_solver; do if [ $o = $UNIT ]; then OBJ="$OBJ /tmp/var_${NAME}_$UNIT.o"; else OBJ="$OBJ $o.o"; fi; done
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../../tools/variants/libcales_$NAME.so $OBJ -ldl
echo built tools/variants/libcales_$NAME.so
