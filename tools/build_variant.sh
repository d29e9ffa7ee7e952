#!/bin/bash
# Builds another libkpilqr.so with extra compiler flags for kernel A/B runs:
#   tools/build_variant.sh NAME -DKP_RECT=0 ...   ->  trajoptkp_amd/lib/variants/NAME/libkpilqr.so   (use with KPILQR_LIB=...)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/trajoptkp_amd/lib/variants/$NAME
mkdir -p $D/obj
make -C $ROOT/trajoptkp_amd/csrc OUT=$D OBJ=$D/obj COMMON="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function $*" >/dev/null
echo $D/libkpilqr.so
