#!/bin/bash
# usage: tools/build_variant.sh <name> "<EXTRA flags>" : builds variants/lib_<name>.so from a scratch copy of csrc
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
w=/tmp/vbuild_$name
rm -rf $w && mkdir -p $w/volsurfs_amd && cp -r $root/volsurfs_amd/csrc $w/volsurfs_amd/csrc && cp -r $root/include $w/include
rm -rf $w/volsurfs_amd/csrc/build
mkdir -p $root/variants
make -C $w/volsurfs_amd/csrc -j8 EXTRA="$extra" TARGET=$root/variants/lib_$name.so 2>&1 | grep -E "error" | head
ls -la $root/variants/lib_$name.so
