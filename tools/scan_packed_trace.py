import os, sys, torch
sys.path.insert(0, os.getcwd())
from nerfacc_amd import cuda as C
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(42)
logn = int(sys.argv[1])
R = (1 << logn) // 96
cnts = torch.randint(0, 193, (R,), device=dev, generator=g)
N = int(cnts.sum())
st = torch.cumsum(cnts, 0) - cnts
x = torch.rand(N, device=dev, generator=g)
for _ in range(10):
    C.exclusive_sum(st, cnts, x, False, False)
torch.cuda.synchronize()
