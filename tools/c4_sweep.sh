cd $GRAFT_REPO_ROOT
for st in 0 32768 65536 131072 262144 524288; do
  for hot in 1000000; do
    python bench.py --quick --steps 8 --warmup 2 --rows 100000000 --format int4 --dim 1024 --keygen structured --placement pinned_host --hot-rows $hot --stage-tokens $st > gpurun_out/c4_$st.json 2>/dev/null && python -c "
import json; r=json.load(open('gpurun_out/c4_$st.json')); print('stage_tokens', $st, 'hot', $hot, 'Mtok/s %.1f' % (r['value']/1e6), 'step ms %.3f' % r['ms_per_step'], flush=True)"
  done
done
