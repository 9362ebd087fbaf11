#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFOO -DBAR"   -> scratch/variants/libNAME.so (conv.hip rebuilt with the defines)
set -e
cd "$(dirname "$0")/.."
name=$1; defs=$2
mkdir -p scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $defs -c dafne_amd/csrc/${SRC:-conv}.hip -o scratch/variants/conv_$name.o
objs=$(ls dafne_amd/build/*.o | grep -v "/${SRC:-conv}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/lib$name.so scratch/variants/conv_$name.o $objs
rm scratch/variants/conv_$name.o
echo scratch/variants/lib$name.so
