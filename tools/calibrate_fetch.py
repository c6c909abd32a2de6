#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS kernel's access pattern (MI355X_MICROARCH.md, HBM:
"other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").

Two launches of k_embed_wave whose unique bytes are known exactly and far exceed every cache:
  rows : N = 4M unigram f-grams over tokens 0..N-1, stream = a permutation of the tokens (every token hits
         exactly ONE row, every row is read exactly once), no wte / wpe
         -> reads N x 768 B of INT8 rows (8 + 4 B per lane) + N x 32 B of id records + N x 2 B of scales
  wte  : same stream against a vocabulary it never hits (K = 0), wte = [N, 768] fp16 read once per token
         -> reads N x 1536 B of fp16 rows (16 + 8 B per lane) + N x 32 B of records + N x 4 B of tokens
  both write N x 1536 B (fp16 output, non-temporal 16 + 8 B per lane).
Run once per counter:   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/calibrate_fetch.py
then:                    python tools/calibrate_fetch.py --report <dir_fetch> <dir_write>
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

N = 4 * 1024 * 1024
D = 768
T = 512


def run():
    import numpy as np
    import torch
    from scone_amd import EmbeddingCache, NGramExtractor
    rng = np.random.default_rng(3)
    perm = rng.permutation(N).astype(np.int32).reshape(N // T, T)
    tok = torch.from_numpy(perm).cuda()
    out = torch.empty(N // T, T, D, dtype=torch.float16, device="cuda")
    # rows: every token is a unigram f-gram with its own row
    keys = np.zeros((N, 3), dtype=np.uint32)
    keys[:, 0] = np.arange(N, dtype=np.uint32)
    ex = NGramExtractor.from_arrays(keys, np.ones(N, dtype=np.uint8), max_n=3)
    cache = EmbeddingCache.from_synthetic(ex, D, table_format="int8")
    for _ in range(3):
        cache.embed_tokens(tok, out=out, out_dtype=torch.float16)
    torch.cuda.synchronize()
    del cache
    # wte: nothing matches, the base rows are read once each
    keys2 = np.zeros((16, 3), dtype=np.uint32)
    keys2[:, 0] = np.arange(16, dtype=np.uint32) + 0x7F000000
    ex2 = NGramExtractor.from_arrays(keys2, np.ones(16, dtype=np.uint8), max_n=3)
    cache2 = EmbeddingCache.from_synthetic(ex2, D, table_format="int8")
    wte = torch.empty(N, D, dtype=torch.float16, device="cuda").normal_()
    for _ in range(3):
        cache2.embed_tokens(tok, wte=wte, out=out)
    torch.cuda.synchronize()
    print("calibration launches done", flush=True)


def counters(d, name):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = []
    for r in csv.DictReader(open(f)):
        if "k_embed_wave" in r["Kernel_Name"] and r["Counter_Name"] == name:
            vals.append(float(r["Counter_Value"]))
    return vals


def report(d_fetch, d_write):
    f, w = counters(d_fetch, "FETCH_SIZE"), counters(d_write, "WRITE_SIZE")
    assert len(f) == 6 and len(w) == 6, (len(f), len(w))
    known = {
        "rows": {"read": N * (D + 32 + 2), "write": N * D * 2},
        "wte": {"read": N * (D * 2 + 32 + 4), "write": N * D * 2},
    }
    res = {}
    for i, k in enumerate(("rows", "wte")):
        fk = sum(f[3 * i:3 * i + 3]) / 3 * 1024
        wk = sum(w[3 * i:3 * i + 3]) / 3 * 1024
        res[k] = {"known_read_bytes": known[k]["read"], "FETCH_SIZE_bytes": fk, "read_factor": known[k]["read"] / fk,
                  "known_write_bytes": known[k]["write"], "WRITE_SIZE_bytes": wk, "write_factor": known[k]["write"] / wk}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--report":
        report(sys.argv[2], sys.argv[3])
    else:
        run()
