#!/bin/bash
# Build a variant of the HIP library with extra compiler flags: bash tools/build_variant.sh <tag> <flags...>  -> vqacl_amd/libvlt5_<tag>.so
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/build/var_$TAG
mkdir -p $W
make -C $ROOT/vqacl_amd/csrc -j8 OBJDIR=$W TARGET=$ROOT/vqacl_amd/libvlt5_$TAG.so CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wno-unused-result $*" 2>&1 | grep -E "error|Error" 
ls -la $ROOT/vqacl_amd/libvlt5_$TAG.so
