#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04g}
mkdir -p $O
cd $R
show() {
python - $1 <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line)
        print(d.get("transport_standin"), d.get("reserve_mode"))
        for k,v in d["variants"].items(): print(f'{v["rank0_step_one_stream_ms"]:.4f} {v["rank0_step_split_phase_loop_ms"]:.4f}  {k[-100:]}')
        print("same output:", d["all_variants_same_output"])
PY
}
timeout -k 10 900 python tools/c5_rank0_step.py --variants 2,3 --transport-standin sdma --cu-reserve 0 --rounds 7 > $O/c5_standin_sdma.json 2> $O/c5_standin_sdma.err
show $O/c5_standin_sdma.json; tail -2 $O/c5_standin_sdma.err
timeout -k 10 900 python tools/c5_rank0_step.py --variants 2,3 --transport-standin kernel --channels 8 --threads 512 --cu-reserve 0,32 --rounds 5 > $O/c5_standin_ch8.json 2> $O/c5_standin_ch8.err
show $O/c5_standin_ch8.json; tail -2 $O/c5_standin_ch8.err
bash tools/r04f.sh $1
