import torch, time
n = 1 << 30
x = torch.empty(n, dtype=torch.float16, device="cuda")
y = torch.empty(n, dtype=torch.float16, device="cuda")
x.zero_(); y.zero_(); torch.cuda.synchronize()
def loop(fn, k):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    ev[0].record()
    for i in range(k):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(k)]
for name, fn in (("fill 2.1 GB", lambda: x.fill_(1.0)), ("copy 2.1 GB", lambda: y.copy_(x))):
    time.sleep(1.0)
    s = loop(fn, 120)
    print(name, "steps 1-24:", " ".join("%.3f" % v for v in s[:24]), "| 100-120 avg %.3f" % (sum(s[100:]) / 20), flush=True)
