import torch
dev = torch.device("cuda:0")
def t(fn, reps=10):
    for _ in range(3): fn()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    return sorted(ms)[len(ms)//2]
for nbytes in (1 << 28, 1 << 30):
    x = torch.rand(nbytes // 4, device=dev)
    y = torch.empty_like(x)
    ms = t(lambda: x.sum())
    print(f"sum (read only) {nbytes/2**20:.0f} MiB: {ms:.3f} ms  {nbytes/ms/1e6:.0f} GB/s")
    ms = t(lambda: y.fill_(1.0))
    print(f"fill (write only) {nbytes/2**20:.0f} MiB: {ms:.3f} ms  {nbytes/ms/1e6:.0f} GB/s")
    ms = t(lambda: y.copy_(x))
    print(f"copy {nbytes/2**20:.0f} MiB: {ms:.3f} ms  {2*nbytes/ms/1e6:.0f} GB/s")
    ms = t(lambda: torch.add(x, 1.0, out=y))
    print(f"add scalar (r+w) {nbytes/2**20:.0f} MiB: {ms:.3f} ms  {2*nbytes/ms/1e6:.0f} GB/s")
