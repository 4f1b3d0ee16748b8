"""host-side cost per call of the two faces of the C ABI (torch extension vs ctypes): wall time of issuing the call
without waiting for the GPU (async calls), and end-to-end latency of the calls that read a count back"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from nerfacc_amd.cuda import _backend as B
from gpu_utils import lego_like, t
dev = torch.device("cuda:0")
o, d, aabb, occ = lego_like(5, 6500, res=128)
O, D, Bn, A = t(o), t(d), t(occ), t(aabb)
R = O.shape[0]
near, far = torch.zeros(R, device=dev), torch.full((R,), 1e10, device=dev)
faces = {"ext": B._C, "ctypes": B._CtypesC}
ri, ts, te, pk = faces["ext"].sample_occgrid(O, D, Bn, A, near, far, 5e-3, 0.0)
N = ri.shape[0]
sig = torch.rand(N, device=dev) * 10; rgb = torch.rand(N, 3, device=dev); bk = torch.ones(3, device=dev)
print("rays", R, "samples", N)
def host(fn, reps=300, sync_each=False):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / reps * 1e6
for name, C in faces.items():
    fa = C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
    g = [torch.rand_like(x) for x in fa]
    print(f"--- {name}")
    print("  sample_occgrid (sync inside)   %.1f us" % host(lambda: C.sample_occgrid(O, D, Bn, A, near, far, 5e-3, 0.0)))
    print("  visibility_compact (sync)      %.1f us" % host(lambda: C.visibility_compact(ri, ts, te, sig, False, 1e-4, 0.0)))
    print("  rendering_fwd                  %.1f us" % host(lambda: C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)))
    print("  rendering_bwd                  %.1f us" % host(lambda: C.rendering_bwd(ri, ts, te, sig, rgb, fa[3], fa[4], fa[5], fa[1], fa[2], R, bk, True, *g)))
    print("  sample_positions               %.1f us" % host(lambda: C.sample_positions(O, D, ri, ts, te)))
    print("  render_weight_fwd              %.1f us" % host(lambda: C.render_weight_from_density_fwd(ri, ts, te, sig, None)))
print("torch.empty                    %.1f us" % host(lambda: torch.empty(N, device=dev)))
print("sig * 2                        %.1f us" % host(lambda: sig * 2))
