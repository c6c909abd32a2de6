#!/bin/bash
# the seven secondary HBM workloads: quick bench + kernel trace + FETCH_SIZE / WRITE_SIZE passes (tools/profile_round4.sh without the headline)
set -u
TAG=${1:-r04z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
sed -n '/^quick() {/,/^}/p' tools/profile_round4.sh > /tmp/quick_fn.sh
. /tmp/quick_fn.sh
quick ${TAG}_hbm_variant --rows 10000000 --keygen structured
quick ${TAG}_c4_int4_100m --rows 100000000 --format int4 --dim 1024 --keygen structured
quick ${TAG}_c2_fp16 --format fp16
quick ${TAG}_c3_int8_10m_d1024 --rows 10000000 --dim 1024
quick ${TAG}_int4_1m_d1024 --format int4 --dim 1024
quick ${TAG}_int8_1m_d1280 --dim 1280
quick ${TAG}_zipf --stream zipf
du -sh gpurun_out/${TAG}_* | tail -8
