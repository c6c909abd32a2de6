#!/bin/bash
# Round 6: the interleaved walk (-DSCONE_WALK_INTERLEAVED: chunk c takes sequences c, c + n_chunks, ...: one compact write window
# instead of n_chunks windows 34 MB apart) against the shipped contiguous walk: whole-batch parity, the placement matrix, bench.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06p}
mkdir -p $O
cd $R
SCONE_HIP_LIB=$R/gpurun_ab/libwalk_il.so timeout -k 10 400 python -m pytest tests/test_gpu_bench_shape.py -x -q -k "whole_bench_batch" 2>&1 | tail -2
for v in tree walk_il tree walk_il; do
  if [ $v = tree ]; then unset SCONE_HIP_LIB; else export SCONE_HIP_LIB=$R/gpurun_ab/lib$v.so; fi
  echo "== $v"; timeout -k 10 200 python3 tools/placement_sensitivity.py 2>/dev/null | grep "table\[0\]:\|table\[1\]:"
done | tee $O/matrix.txt
unset SCONE_HIP_LIB
tools/ab_multi.sh ${1:-r06p}/plain 3 scone_amd/csrc/libscone_hip.so gpurun_ab/libwalk_il.so -- --out-candidates 1 | tee $O/bench_plain.txt
tools/ab_multi.sh ${1:-r06p}/tuned 2 scone_amd/csrc/libscone_hip.so gpurun_ab/libwalk_il.so -- --out-candidates 6 | tee $O/bench_tuned.txt
