#!/bin/bash
# the chunk pipeline two chunks ahead (default) against three (variant build: SCONE_STAGE_NBUF 7, SCONE_STAGE_AHEAD 3), copy grid 128 / 96 / 64
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04x4}
mkdir -p $O
cd $R
for cfg in "scone_amd/csrc/libscone_hip.so 128" "gpurun_ab/libahead3.so 128" "gpurun_ab/libahead3.so 96" "gpurun_ab/libahead3.so 64" "gpurun_ab/libahead3.so 160" "scone_amd/csrc/libscone_hip.so 128" "gpurun_ab/libahead3.so 128"; do
  set -- $cfg
  n=$(basename $1 .so)_b$2
  SCONE_HIP_LIB=$R/$1 SCONE_STAGE_COPY_BLOCKS=$2 timeout -k 10 300 python tools/c4_zipf_probe.py --mode cached --cache-rows 16000000 \
     --stage-tokens 262144 --prefetch-next --steps 40 --warmup 400 > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -3 $O/$n.err; exit 1; }
  python -c "
import json
j=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: %.3f ms' % j['ms_per_step'], 'copied', round(j['per_step']['rows_copied']), 'status', j['status'], j['checksum_last'])"
done
