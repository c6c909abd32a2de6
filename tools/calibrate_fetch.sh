#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on k_embed_wave's own access pattern (tools/calibrate_fetch.py); run via gpurun.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-calib}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/calibrate_fetch.py > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/calibrate_fetch.py > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/tools/calibrate_fetch.py --report $O/fetch $O/write | tee $O/calibration.json
