#!/bin/bash
# Round 6: which of the kernel's streams feels the output buffer's placement?  tools/placement_sensitivity.py (one table x five
# output buffers is the line read here) through the shipped library, the no-store probe and the stores-only probe.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06q}
mkdir -p $O
cd $R
for v in tree no_store no_reads tree; do
  if [ $v = tree ]; then unset SCONE_HIP_LIB; else export SCONE_HIP_LIB=$R/gpurun_ab/lib$v.so; fi
  echo "== $v"; timeout -k 10 200 python3 tools/placement_sensitivity.py 2>/dev/null | grep "table\[1\]:\|table\[2\]:"
done | tee $O/matrix.txt
