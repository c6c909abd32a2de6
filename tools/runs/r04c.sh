#!/bin/bash
# Round 4, third GPU call: CU-mask layout probe, the hop's own cost, and rank 0's step at C5 scale with the transfers in flight
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04c}
mkdir -p $O
cd $R
timeout -k 10 120 python tools/cu_mask_probe.py > $O/cu_mask_probe.json 2> $O/cu_mask_probe.err; cat $O/cu_mask_probe.json | head -c 3000; echo
timeout -k 10 300 python tools/cu_reserve_curve.py --configs headline --reserves 0,8,16 --full-mask > $O/cu_reserve_full_mask.json 2> $O/cu_reserve_full_mask.err; tail -c 800 $O/cu_reserve_full_mask.json; echo
timeout -k 10 900 python tools/c5_rank0_step.py --variants 2,3 --transport-standin --cu-reserve 0,16 --rounds 5 > $O/c5_standin_ch2.json 2> $O/c5_standin_ch2.err
python - $O/c5_standin_ch2.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line)
        print(d.get("transport_standin"))
        for k,v in d["variants"].items(): print(f'{v["rank0_step_one_stream_ms"]:.4f} {v["rank0_step_split_phase_loop_ms"]:.4f}  {k[-90:]}')
        print("same output:", d["all_variants_same_output"])
PY
tail -3 $O/c5_standin_ch2.err
