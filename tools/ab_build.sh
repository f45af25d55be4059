#!/bin/bash
# A/B builds side by side: tools/ab_build.sh <name> "<extra compiler flags>"  ->  abtest/lib_<name>.so (picked up with MCA_HIP_LIB=...)
# The sources are copied to a scratch directory, so the shipped objects and library are left alone.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/mca_ab_XXXX)
mkdir -p $tmp/mcarray_amd/csrc $tmp/include $tmp/tools $root/abtest
cp $root/tools/check_isa.py $tmp/tools/
cp $root/mcarray_amd/csrc/*.hip $root/mcarray_amd/csrc/*.h $root/mcarray_amd/csrc/Makefile $tmp/mcarray_amd/csrc/
cp -r $root/include/* $tmp/include/
make -C $tmp/mcarray_amd/csrc -j8 -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*"
cp $tmp/mcarray_amd/libmcarray_hip.so $root/abtest/lib_$name.so
rm -rf $tmp
echo "built abtest/lib_$name.so with: $*"
