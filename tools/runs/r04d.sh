#!/bin/bash
# Round 4: rank 0's step at C5 scale with the transfers in flight (RCCL-shaped stand-in kernels), with and without a CU reserve
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04d}
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
timeout -k 10 900 python tools/c5_rank0_step.py --variants 2,3 --transport-standin --channels 2 --cu-reserve 0,16,32 --rounds 5 > $O/c5_standin_ch2.json 2> $O/c5_standin_ch2.err
show $O/c5_standin_ch2.json; tail -2 $O/c5_standin_ch2.err
timeout -k 10 900 python tools/c5_rank0_step.py --variants 2 --transport-standin --channels 4 --threads 512 --cu-reserve 0,16 --rounds 5 > $O/c5_standin_ch4.json 2> $O/c5_standin_ch4.err
show $O/c5_standin_ch4.json; tail -2 $O/c5_standin_ch4.err
timeout -k 10 900 python tools/c5_rank0_step.py --variants 2 --transport-standin --channels 2 --cu-reserve 16 --reserve-mode hop --rounds 5 > $O/c5_standin_ch2_hop.json 2> $O/c5_standin_ch2_hop.err
show $O/c5_standin_ch2_hop.json; tail -2 $O/c5_standin_ch2_hop.err
