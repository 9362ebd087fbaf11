#!/bin/bash
# usage: scratch/stagger_variant.sh STEM SLEEP  -> scratch/variants/libstag_STEM_SLEEP.so
# Timing experiment (round 4, end): are the ingest-bound kernels slowed by every CU of an XCD streaming the SAME weight bytes at the
# SAME moment (one L2 channel at a time serving 32 CUs)?  A PATCHED COPY of csrc/STEM.hip (the product source is not touched) delays
# workgroup b by ((b >> 3) & 7) * SLEEP s_sleep(8) periods (~0.25 us each) at kernel entry, so that the CUs of one XCD run up to
# 7 * SLEEP periods apart.  Results are unchanged (a delay only); the launch pays the largest delay once.
set -e
cd "$(dirname "$0")/.."
stem=$1; sl=$2
mkdir -p scratch/variants
src=scratch/variants/${stem}_stag${sl}.hip
python3 - "$stem" "$sl" <<'E'
import re, sys
stem, sl = sys.argv[1], int(sys.argv[2])
s = open("dafne_amd/csrc/%s.hip" % stem).read()
delay = ("\n    { const int stag_ph = ((int)blockIdx.x >> 3) & 7; for (int stag_i = 0; stag_i < stag_ph * %d; stag_i++) "
         "__builtin_amdgcn_s_sleep(8); }\n" % sl)
# first statement of every __global__ kernel body that declares dynamic LDS
n = 0
def rep(m):
    global n
    n += 1
    return m.group(0) + delay
s = re.sub(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char lds\[\];", rep, s)
assert n > 0, "no kernel entry found"
s = s.replace('#include "common.h"', '#include "../../dafne_amd/csrc/common.h"')
open("scratch/variants/%s_stag%d.hip" % (stem, sl), "w").write(s)
print("patched", n, "kernel(s)")
E
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Idafne_amd/csrc -Iinclude -c $src -o scratch/variants/${stem}_stag${sl}.o
objs=$(ls dafne_amd/build/*.o | grep -v "/${stem}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/libstag_${stem}_${sl}.so scratch/variants/${stem}_stag${sl}.o $objs
rm scratch/variants/${stem}_stag${sl}.o
echo scratch/variants/libstag_${stem}_${sl}.so
