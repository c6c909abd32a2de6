#!/bin/bash
# Lists every k_embed_wave / k_embed_fused / k_match_ell instantiation that uses scratch memory (register spills):
# a spill inside the per-token loop turns into HBM traffic (WRITE_SIZE +2 % and -5 % speed on the INT4 kernel before
# the occupancy estimate of scone_embed_wave.h accounted for its scale registers).  CPU only (hipcc -S).
cd "$(dirname "$0")/../scone_amd/csrc" || exit 1
rc=0
for src in scone_gather_i8 scone_gather_i4 scone_gather_f16 scone_gather_f32 scone_index; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S $src.hip -o /tmp/$src.s 2>/dev/null || { echo "compile failed: $src"; exit 1; }
  python3 - /tmp/$src.s $src <<'PY' || rc=1
import re, sys
name, vg, bad = None, None, 0
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\S+):", line)
    if m:
        name = m.group(1)
    elif "; NumVgprs:" in line:
        vg = int(line.split()[-1])
    elif "; ScratchSize:" in line:
        sc = int(line.split()[-1])
        if sc and name and any(k in name for k in ("k_embed_wave", "k_embed_fused", "k_embed_csr_wave", "k_match_ell")):
            print(f"{sys.argv[2]}: scratch {sc} B/lane, {vg} VGPRs: {name[:110]}")
            bad += 1
sys.exit(1 if bad else 0)
PY
done
[ $rc = 0 ] && echo "no wave kernel uses scratch"
exit $rc
