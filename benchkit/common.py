"""Constants shared by bench.py's modules (measurement / record / launcher)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PCIE_PEAK_GBPS = 64.0       # PCIe Gen5 x16, one direction
# xGMI: 7 links per GPU, ~153.6 GB/s each counting both directions (task statement / DESIGN section 6) = 76.8 GB/s into a
# GPU per link; a rank receives over min(W - 1, 7) links at once on the fully connected node
XGMI_LINK_GBPS_PER_DIRECTION = 76.8
T0_ENV = "SCONE_BENCH_T0"   # wall-clock start of the job's FIRST process: children of `self_launch` inherit the deadline
