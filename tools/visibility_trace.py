"""per-kernel times of visibility_compact at N ~ 2^logn (rocprofv3 --kernel-trace --stats around this script)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from nerfacc_amd import cuda as C
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(42)
logn = int(sys.argv[1])
keep_scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
R = (1 << logn) // 96
cnts = torch.randint(0, 193, (R,), device=dev, generator=g)
ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
N = ri.shape[0]
ts = torch.rand(N, device=dev, generator=g) * 4
te = ts + 5e-3
sig = torch.rand(N, device=dev, generator=g) * 30 * keep_scale
for _ in range(10):
    out = C.visibility_compact(ri, ts, te, sig, False, 1e-4, 0.0)
torch.cuda.synchronize()
print("N", N, "kept", out[0].shape[0])
