#!/bin/bash
# lookup grid of 3 (default) / 6 / 12 / 24 residency rounds (SCONE_WAVE_BLOCKS_FIXED variants): alone, and beside the cache's copy
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04r2}
mkdir -p $O
cd $R
for lib in scone_amd/csrc/libscone_hip.so gpurun_ab/librounds6.so gpurun_ab/librounds12.so gpurun_ab/librounds24.so scone_amd/csrc/libscone_hip.so; do
  n=$(basename $lib .so)
  SCONE_HIP_LIB=$R/$lib timeout -k 10 300 python tools/c4_zipf_probe.py --mode cached --cache-rows 16000000 \
     --stage-tokens 262144 --prefetch-next --steps 40 --warmup 400 > $O/cached_$n.json 2> $O/cached_$n.err || { echo "$n failed"; tail -3 $O/cached_$n.err; exit 1; }
  SCONE_HIP_LIB=$R/$lib timeout -k 10 300 python tools/c4_zipf_probe.py --mode zero --hot 100000000 --steps 40 --warmup 20 > $O/hbm_$n.json 2> $O/hbm_$n.err || { echo "$n failed"; tail -3 $O/hbm_$n.err; exit 1; }
  python -c "
import json
a=json.loads(open('$O/cached_$n.json').read().strip().splitlines()[-1]); b=json.loads(open('$O/hbm_$n.json').read().strip().splitlines()[-1])
print('$n: cached %.3f ms  all-in-HBM %.3f ms' % (a['ms_per_step'], b['ms_per_step']), a['checksum_last'], b['checksum_last'])"
done
