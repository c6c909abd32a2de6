#!/usr/bin/env python3
"""Plain device-memory rates of this box for comparison with the lookup kernel's timing probes: a fill (writes only), a
read-only reduction, a copy (reads + writes) over buffers far larger than the caches.  HIP-event timed, best of 20."""
import json
import torch

def best(fn, n=20):
    t = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    return min(t), sorted(t)[len(t) // 2]

n = 1 << 30                      # 2 GiB of fp16
x = torch.empty(n, dtype=torch.float16, device="cuda")
y = torch.empty(n, dtype=torch.float16, device="cuda")
x.zero_(); y.zero_(); torch.cuda.synchronize()
gb = n * 2 / 1e9
res = {}
m, med = best(lambda: x.fill_(1.0));                 res["fill 2.1 GB (write only)"] = {"ms_min": m, "ms_median": med, "TBps": gb / m}
m, med = best(lambda: x.view(torch.int32).sum());    res["sum 2.1 GB (read only)"] = {"ms_min": m, "ms_median": med, "TBps": gb / m}
m, med = best(lambda: y.copy_(x));                   res["copy 2.1 GB (read + write 4.3 GB)"] = {"ms_min": m, "ms_median": med, "TBps": 2 * gb / m}
print(json.dumps(res, indent=1))
