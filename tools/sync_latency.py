import time, torch
x = torch.zeros(1024, device="cuda:0")
torch.cuda.synchronize()
for name, fn in (("stream.synchronize", lambda: torch.cuda.current_stream().synchronize()),
                 ("event.synchronize", None)):
    ts = []
    for _ in range(2000):
        t0 = time.perf_counter()
        x.add_(1.0)
        if fn: fn()
        else:
            e = torch.cuda.Event(); e.record(); e.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: launch + 3 us kernel + wake: median {ts[len(ts)//2]*1e6:.1f} us, p10 {ts[len(ts)//10]*1e6:.1f}, p90 {ts[int(len(ts)*0.9)]*1e6:.1f}")
ts = []
for _ in range(2000):
    t0 = time.perf_counter(); x.add_(1.0); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
ts.sort(); print(f"launch only: median {ts[len(ts)//2]*1e6:.1f} us")
