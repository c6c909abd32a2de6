#!/bin/bash
# SQ / TCC counters of the gather kernel for several bench configurations on one box:
#   tools/pmc_compare.sh <tag> "<name1>|<bench args>" "<name2>|<bench args>" ...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%|*}; args=${spec#*|}
  i=0
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${name}_$i -- python3 $R/bench.py --steps 5 --warmup 2 --quick $args > $O/${name}_$i.log 2>&1 || echo "pass $i of $name failed: $(tail -1 $O/${name}_$i.log)"
  done
done
python3 - $O <<'PY'
import collections, csv, glob, os, sys
O = sys.argv[1]
tab = collections.defaultdict(dict)
for f in glob.glob(os.path.join(O, "*", "*", "*counter_collection.csv")):
    name = os.path.basename(os.path.dirname(os.path.dirname(f))).rsplit("_", 1)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_embed_wave" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        tab[c][name] = sum(v) / len(v)
names = sorted({n for c in tab.values() for n in c})
print("%-30s" % "counter" + "".join("%18s" % n for n in names))
for c in sorted(tab):
    print("%-30s" % c + "".join("%18.4g" % tab[c].get(n, float("nan")) for n in names))
PY
