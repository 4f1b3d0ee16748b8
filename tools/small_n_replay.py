"""Why do the path's streaming kernels take 1.3-1.9x longer inside bench.py than back to back?  The four of
them — visibility filter (mask + compaction), rendering forward, rendering backward — at the bench's size (6 564 rays, ~2.5e5
candidate samples, ~2.2e5 rendered) in three regimes:
    same       every repetition reads the same input tensors again (what profiles/r02_bench_kernels.md's back-to-back rows measured)
    produced   every input is written by a torch kernel right before the call (as the field / the sampling call do in the bench:
               the producer ran on whatever XCDs it ran on, the consumer's XCD has to fetch the lines from the memory side)
    evicted    as `produced`, and a 768 MB fill runs between the producer and the call (L2s and the 256 MB Infinity Cache hold
               nothing of the inputs: the bench's field moves ~1 GB per step between two of these kernels)
Run under rocprofv3 (tools/experiments/r06_small_n.sh): --kernel-trace --stats for the durations, --pmc TCC_HIT_sum TCC_MISS_sum and
--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY in separate passes.      python tools/small_n_replay.py <regime> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import nerfacc_amd
from nerfacc_amd import cuda as C

regime = sys.argv[1] if len(sys.argv) > 1 else "same"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
st = np.load(os.path.join(ROOT, "profiles", "r02_sampling_state.npz"))
res = tuple(int(x) for x in st["res"])
binaries = torch.from_numpy(np.unpackbits(st["binaries_bits"])[: int(np.prod(res))].astype(bool).reshape(res)).to(dev)
aabbs = torch.from_numpy(st["aabbs"]).to(dev)
O, D = torch.from_numpy(st["rays_o"]).to(dev), torch.from_numpy(st["rays_d"]).to(dev)
step = float(st["render_step"])
R = O.shape[0]
nerfacc_amd.set_option("fused_vis", 0)          # the three-kernel filter: mask pass and compaction show up as their own rows
ri, ts, te, _ = C.sample_occgrid(O, D, binaries, aabbs, None, None, step, 0.0, near_plane=0.0, far_plane=1e10)
N = ri.shape[0]
g = torch.Generator(device=dev).manual_seed(1)
sig = torch.rand(N, device=dev, generator=g) * 20.0
bk = torch.ones(3, device=dev)
junk = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
fri, fts, fte, _ = C.visibility_compact(ri, ts, te, sig, False, 1e-4, 0.0, False)
K = fri.shape[0]
rgb = torch.rand(K, 3, device=dev, generator=g)
sg2 = torch.rand(K, device=dev, generator=g) * 20.0
col, opa, dep, w, T, a = C.rendering_fwd(fri, fts, fte, sg2, rgb, R, bk, True)
gc = torch.rand_like(col)
print(f"regime {regime}: rays {R} candidates {N} rendered {K}")

def fresh(*tensors):
    """the same values, written by a kernel just now (into new storage in the `produced` / `evicted` regimes)"""
    if regime == "same":
        return tensors
    out = tuple(t.clone() for t in tensors)
    if regime == "evicted":
        junk.fill_(1)
    return out

for _ in range(reps):
    i_ri, i_ts, i_te, i_sig = fresh(ri, ts, te, sig)
    f = C.visibility_compact(i_ri, i_ts, i_te, i_sig, False, 1e-4, 0.0, False)
    j_ri, j_ts, j_te, j_sg, j_rgb = fresh(fri, fts, fte, sg2, rgb)
    o = C.rendering_fwd(j_ri, j_ts, j_te, j_sg, j_rgb, R, bk, True)
    k_ri, k_ts, k_te, k_rgb, k_w, k_T, k_a, k_gc = fresh(fri, fts, fte, rgb, o[3], o[4], o[5], gc)
    C.rendering_bwd(k_ri, k_ts, k_te, j_sg, k_rgb, k_w, k_T, k_a, o[1], o[2], R, bk, True, k_gc, None, None, None, None, None)
torch.cuda.synchronize()
