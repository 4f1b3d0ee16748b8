"""kernel time of the streaming kernels at the training size (N ~ 2^18), HIP events, median of 50"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfacc_amd import cuda as C
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
R = 13120
cnts = (torch.rand(R, device=dev, generator=g) < 0.45).long() * torch.randint(10, 80, (R,), device=dev, generator=g)
ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
N = ri.shape[0]
ts = torch.rand(N, device=dev, generator=g) * 4; te = ts + 5e-3
sig = torch.rand(N, device=dev, generator=g) * 30; rgb = torch.rand(N, 3, device=dev, generator=g)
bk = torch.ones(3, device=dev)
def t(fn, reps=50):
    for _ in range(5): fn()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    return sorted(ms)[len(ms)//2] * 1e3
col, opa, dep, w, T, a = C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
gc = torch.rand_like(col)
print("N", N, "tile", os.environ.get("NFA_TILE", "auto"),
      "rendering_fwd %.1f us" % t(lambda: C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)),
      "rendering_bwd %.1f us" % t(lambda: C.rendering_bwd(ri, ts, te, sig, rgb, w, T, a, opa, dep, R, bk, True, gc, None, None, None, None, None)),
      "weight_fwd %.1f us" % t(lambda: C.render_weight_from_density_fwd(ri, ts, te, sig, None)),
      "visibility %.1f us" % t(lambda: C.visibility_compact(ri, ts, te, sig * 0.02, False, 1e-4, 0.0)))
