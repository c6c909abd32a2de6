#!/bin/bash
# Build a variant of libscone_hip.so for A/B runs (tools/ab_multi.sh, tools/variants.sh):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DSCONE_PREF_MASK=12]
# Only the per-format gather translation units are recompiled with the flags; the other objects are the in-tree
# ones (run `make -C scone_amd/csrc` first).  Output: gpurun_ab/lib<name>.so (git-ignored, travels with gpurun).
set -eu
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/scone_amd/csrc
B=$R/build/variant_$NAME
mkdir -p $B $R/gpurun_ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-pass-failed -Wno-unused-variable"
pids=()
for f in scone_gather_f32 scone_gather_f16 scone_gather_i8 scone_gather_i4 scone_gather; do
  ( cd $C && hipcc $FLAGS "$@" -c $f.hip -o $B/$f.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc -shared --offload-arch=gfx950 -o $R/gpurun_ab/lib$NAME.so $B/*.o $C/scone_api.o $C/scone_index.o $C/scone_table.o \
  $C/scone_fit.o $C/scone_stage.o $C/scone_shard.o $C/scone_ipc.o
echo "built gpurun_ab/lib$NAME.so ($*)"
