#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04k}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sync_free or row_exchange or slice_exchange" > $O/pytest_sf.log 2>&1
echo "pytest rc=$? $(tail -2 $O/pytest_sf.log | head -c 400)"
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
timeout -k 10 900 python tools/c5_rank0_step.py --variants 3 --transport-standin kernel --standin-directions out-read --channels 8 --threads 512 --cu-reserve 0,16,32 --rounds 5 > $O/c5_standin_kernel_outread_ch8.json 2> $O/c5_standin_kernel_outread_ch8.err
show $O/c5_standin_kernel_outread_ch8.json; tail -2 $O/c5_standin_kernel_outread_ch8.err
timeout -k 10 900 python tools/c5_rank0_step.py --variants 3 --transport-standin kernel --standin-directions out-read --channels 4 --threads 512 --cu-reserve 0,16 --rounds 5 > $O/c5_standin_kernel_outread_ch4.json 2> $O/c5_standin_kernel_outread_ch4.err
show $O/c5_standin_kernel_outread_ch4.json; tail -2 $O/c5_standin_kernel_outread_ch4.err
