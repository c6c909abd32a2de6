#!/bin/bash
# Round 6: the one-launch kernel's probe -- bitmap word and home bucket requested together (the tree) against the bitmap first
# (gpurun_ab/libfused_serial.so = -DSCONE_FUSED_SERIAL_PROBE).  Alternating processes on one box, tools/api_latency-style loop.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06c}
mkdir -p $O
cd $R
for round in 1 2 3; do
  for v in tree serial; do
    if [ $v = serial ]; then export SCONE_HIP_LIB=$R/gpurun_ab/libfused_serial.so; else unset SCONE_HIP_LIB; fi
    timeout -k 10 200 python3 - $v $round >> $O/latency.txt 2>> $O/err.log <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench
from scone_amd import EmbeddingCache
from scone_amd import synthetic as S
d = 768
vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
lat = bench.latency_block(cache, wte, wpe, d, extra_shapes=((16, 1024), (32, 1024)), calls=1000)
print(sys.argv[1], sys.argv[2], " ".join(f"{k}: call {v['call_us']:.2f} kern {v['kernel_us']:.2f} graph {v.get('graph_us', 0):.2f} |" for k, v in lat.items()), flush=True)
PY
  done
done
cat $O/latency.txt
