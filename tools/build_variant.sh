#!/bin/bash
# Measurement build: the library with extra -D switches for ONE source file, as tmp_variants/lib_<tag>.so (git-ignored; travels to
# the GPU box).  Load it with VU_LIB_PATH=$PWD/tmp_variants/lib_<tag>.so.   usage: tools/build_variant.sh <tag> <file.hip> "<-D...>"
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; DEFS=$3
mkdir -p tmp_variants
CS=vit-unet_amd/csrc
(cd $CS && make -s >/dev/null)
EXTRA=""; [ "$SRC" = vu_flash.hip ] && EXTRA="-fno-honor-nans"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable $EXTRA $DEFS -c $CS/$SRC -o tmp_variants/${SRC%.hip}_$TAG.o
OBJS=$(ls $CS/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tmp_variants/lib_$TAG.so $OBJS tmp_variants/${SRC%.hip}_$TAG.o
echo tmp_variants/lib_$TAG.so
