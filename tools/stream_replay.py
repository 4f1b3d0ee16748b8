"""Every streaming kernel of the path a few times at N = 2^logn, plus a tuned copy of known size as the calibration point of
the memory-side counters (tools/pmc_streaming.sh runs this under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes).
usage: python tools/stream_replay.py [log2_N] [reps]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "ubench"))
from nerfacc_amd import cuda as C
import ceiling

dev = torch.device("cuda:0")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device=dev).manual_seed(42)
R = (1 << logn) // 96
cnts = torch.randint(0, 193, (R,), device=dev, generator=g)
ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
N = ri.shape[0]
pk = torch.stack([torch.cumsum(cnts, 0) - cnts, cnts], -1)
ts = torch.rand(N, device=dev, generator=g) * 4
te = ts + 5e-3
sig = torch.rand(N, device=dev, generator=g) * 30
rgb = torch.rand(N, 3, device=dev, generator=g)
bk = torch.ones(3, device=dev)
col, opa, dep, w, T, a = C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
gc, go, gd, gw = torch.rand_like(col), torch.rand_like(opa), torch.rand_like(dep), torch.rand_like(w)
x = torch.rand(N, device=dev, generator=g)
L = ceiling._lib()
src = torch.rand(N * 4, device=dev); dst = torch.empty_like(src)        # 16 N bytes each way
P = lambda t: ctypes.c_void_p(t.data_ptr())
print(f"N = {N} R = {R} copy_bytes = {16 * N}")
for _ in range(reps):
    L.run(0, 1, 0, P(src), P(dst), ctypes.c_int64(N), 16384, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    C.render_weight_from_density_fwd(ri, ts, te, sig, None)
    C.render_weight_from_density_bwd(ri, ts, te, sig, T, a, gw, None, None)
    C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
    C.rendering_bwd(ri, ts, te, sig, rgb, w, T, a, opa, dep, R, bk, True, gc, go, gd, None, None, None)
    C.accumulate_along_rays(ri, w, rgb, R)
    C.accumulate_along_rays(ri, w, None, R)
    C.exclusive_sum_cub(ri, x, False)
    C.exclusive_sum(pk[:, 0].contiguous(), pk[:, 1].contiguous(), x, False, False)
    C.visibility_compact(ri, ts, te, sig * 0.01, False, 1e-4, 0.0)
torch.cuda.synchronize()
