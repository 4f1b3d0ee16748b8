"""host-side (Python + ctypes + allocation) cost of each op: wall time of issuing the call
without waiting for the GPU, averaged over many calls (queue drained before each batch)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerfacc_amd as nerfacc
from nerfacc_amd import cuda as C
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
R = 13120
cnts = (torch.rand(R, device=dev, generator=g) < 0.45).long() * torch.randint(10, 80, (R,), device=dev, generator=g)
ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts); N = ri.shape[0]
ts = torch.rand(N, device=dev, generator=g) * 4; te = ts + 5e-3
sig = (torch.rand(N, device=dev, generator=g) * 30).requires_grad_(True); rgb = torch.rand(N, 3, device=dev, generator=g).requires_grad_(True)
bk = torch.ones(3, device=dev)
def host(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / reps * 1e6
col, opa, dep, w, T, a = C.rendering_fwd(ri, ts, te, sig.detach(), rgb.detach(), R, bk, True)
print("N", N)
print("C.rendering_fwd            %.1f us" % host(lambda: C.rendering_fwd(ri, ts, te, sig.detach(), rgb.detach(), R, bk, True)))
print("nerfacc.rendering (fwd)    %.1f us" % host(lambda: nerfacc.rendering(ts, te, ri, R, rgb_sigma_fn=lambda *_: (rgb, sig), render_bkgd=bk)))
def fb():
    c, o, d, _ = nerfacc.rendering(ts, te, ri, R, rgb_sigma_fn=lambda *_: (rgb, sig), render_bkgd=bk)
    c.sum().backward()
print("rendering fwd+bwd          %.1f us" % host(fb, 100))
print("torch.empty                %.1f us" % host(lambda: torch.empty(N, device=dev)))
print("sig * 2 (one torch op)     %.1f us" % host(lambda: sig.detach() * 2))
print("current_stream().cuda_stream %.1f us" % host(lambda: torch.cuda.current_stream(dev).cuda_stream))
print("C.render_weight_fwd        %.1f us" % host(lambda: C.render_weight_from_density_fwd(ri, ts, te, sig.detach(), None)))
