"""host time of nerfacc.rendering itself (closure returns cached tensors; asynchronous, no sync inside the loop)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerfacc_amd as nerfacc
dev = torch.device("cuda:0")
R = 6500
cnts = torch.randint(20, 60, (R,), device=dev)
ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
N = ri.shape[0]
ts = torch.rand(N, device=dev) * 4; te = ts + 5e-3
sig = (torch.rand(N, device=dev) * 30).requires_grad_(True)
rgb = torch.rand(N, 3, device=dev).requires_grad_(True)
bk = torch.ones(3, device=dev)
fn = lambda a, b, c: (rgb, sig)
for _ in range(20):
    nerfacc.rendering(ts, te, ri, n_rays=R, rgb_sigma_fn=fn, render_bkgd=bk)
torch.cuda.synchronize()
K = 2000
t0 = time.perf_counter()
for _ in range(K):
    out = nerfacc.rendering(ts, te, ri, n_rays=R, rgb_sigma_fn=fn, render_bkgd=bk)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("nerfacc.rendering host time per call: %.1f us (N = %d)" % ((t1 - t0) / K * 1e6, N))
from nerfacc_amd.cuda import _backend
C = _backend._C
t0 = time.perf_counter()
for _ in range(K):
    out = C.rendering(ri, ts, te, sig, rgb, R, bk, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("  of which the extension call (autograd node included): %.1f us" % ((t1 - t0) / K * 1e6))
t0 = time.perf_counter()
for _ in range(K):
    out = C.rendering_fwd(ri, ts, te, sig.detach(), rgb.detach(), R, bk, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("  of which rendering_fwd without autograd: %.1f us" % ((t1 - t0) / K * 1e6))
