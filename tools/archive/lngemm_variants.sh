#!/bin/bash
# Diagnosis builds of the panel kernel (LNG_DBG bit set: 1 no W loads, 2 no A-fragment reads, 4 no stores) linked
# against the objects of the production library; run tools/lngemm_bench.py with PANDORA_LIB=build/lngemm_dbgN.so.
set -e
cd "$(dirname "$0")/.."
C=open-pandora_amd/csrc; O=build/obj; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
for f in gemm attn norm misc; do
  if [ ! -f $O/$f.o ] || [ $C/$f.hip -nt $O/$f.o ] || [ $C/common.hpp -nt $O/$f.o ]; then /opt/rocm/bin/hipcc $FL -c $C/$f.hip -o $O/$f.o & fi
done
wait
for d in "$@"; do
  /opt/rocm/bin/hipcc $FL -DLNG_DBG=$d -c $C/lngemm.hip -o $O/lngemm_dbg$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/gemm.o $O/attn.o $O/norm.o $O/misc.o $O/lngemm_dbg$d.o -o build/lngemm_dbg$d.so
done
ls -la build/*.so
