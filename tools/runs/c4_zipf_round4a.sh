#!/bin/bash
# Round-4 baseline for the pinned-host path on the Zipf stream (one gpurun call): zero-copy and the per-chunk staged
# prefetch with a different batch every step and with one repeated batch, + a kernel trace of the staged form.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04a}
mkdir -p $O
cd $R
for m in zero staged; do
  timeout -k 10 400 python tools/c4_zipf_probe.py --mode $m --steps 20 > $O/zipf_$m.json 2> $O/zipf_$m.err
  echo "$m fresh: $(tail -c 400 $O/zipf_$m.json | head -c 300)"
  timeout -k 10 400 python tools/c4_zipf_probe.py --mode $m --steps 20 --same-batch > $O/zipf_${m}_same.json 2> $O/zipf_${m}_same.err
  echo "$m same: $(tail -c 400 $O/zipf_${m}_same.json | head -c 300)"
done
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_staged -- python3 $R/tools/c4_zipf_probe.py --mode staged --steps 10 > $O/trace_staged.log 2>&1 )
cp $(ls $O/trace_staged/*/*kernel_stats.csv | head -1) $O/staged_kernel_stats.csv
cp $(ls $O/trace_staged/*/*kernel_trace.csv | head -1) $O/staged_kernel_trace.csv 2>/dev/null
rm -rf $O/trace_staged
head -20 $O/staged_kernel_stats.csv
