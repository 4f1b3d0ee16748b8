"""Achieved HBM GB/s of every streaming kernel of the path at roofline-scale N.

For each kernel: NeRF-like ragged rays (<= 1000 samples per ray), N ~ 2^24 samples (0.5-1 GB of
traffic, past the 256 MiB Infinity Cache), HIP events on the launch stream around the C-ABI
call, median of 10 after 3 warm-ups; GB/s = algorithmic bytes / time.
usage: python tools/roofline_sweep.py [log2_N] [out.md]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfacc_amd import cuda as C

PEAK = 8000.0
dev = torch.device("cuda:0")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = torch.Generator(device=dev).manual_seed(42)
R = (1 << logn) // 96
cnts = torch.randint(0, 193, (R,), device=dev, generator=g)
if os.environ.get("NFA_SWEEP_CNT"):                    # constant ray length (alignment experiments)
    R = (1 << logn) // int(os.environ["NFA_SWEEP_CNT"])
    cnts = torch.full((R,), int(os.environ["NFA_SWEEP_CNT"]), device=dev)
ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
N = ri.shape[0]
pk = torch.stack([torch.cumsum(cnts, 0) - cnts, cnts], -1)
ts = torch.rand(N, device=dev, generator=g) * 4
te = ts + 5e-3
sig = torch.rand(N, device=dev, generator=g) * 30
rgb = torch.rand(N, 3, device=dev, generator=g)
bk = torch.ones(3, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]


rows = []


def bench(name, nbytes, fn):
    ms = timeit(fn)
    gbs = nbytes / ms / 1e6
    rows.append((name, nbytes / 1e6, ms, gbs, gbs / PEAK))
    print(f"{name:46s} {nbytes/1e6:9.1f} MB {ms:8.3f} ms {gbs:8.1f} GB/s {100*gbs/PEAK:5.1f} % of 8 TB/s", flush=True)


col, opa, dep, w, T, a = C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
gc, go, gd = torch.rand_like(col), torch.rand_like(opa), torch.rand_like(dep)
gw = torch.rand_like(w)
x = torch.rand(N, device=dev, generator=g)
# copy baseline: what this chip + torch give for a plain float4 stream
buf = torch.empty(N * 8, device=dev)
src = torch.rand(N * 8, device=dev)
bench("torch copy (read+write, reference point)", 2 * 4 * N * 8, lambda: buf.copy_(src))
# this box's own ceilings: tuned 16-byte-lane copy / read-only / write-only kernels, best of a sweep (tools/ubench/ceiling.py)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench"))
try:
    import ceiling
    for kind, (gbs, cfg) in ceiling.ceilings(4 * N * 8).items():
        rows.append((f"tuned {kind} kernel of this box ({cfg})", 0.0, 0.0, gbs, gbs / PEAK))
        print(f"{'tuned ' + kind + ' (' + cfg + ')':100s} {gbs:8.1f} GB/s {100*gbs/PEAK:5.1f} % of 8 TB/s", flush=True)
    rows.append(("guide's float4 copy (MI355X_MICROARCH.md:35)", 0.0, 0.0, 6290.0, 6290.0 / PEAK))
except Exception as e:      # noqa: BLE001
    print("ceiling ubench unavailable:", e)
bench("weight_fwd_kernel (ray_indices)", 32 * N, lambda: C.render_weight_from_density_fwd(ri, ts, te, sig, None))
bench("weight_bwd_kernel", 36 * N, lambda: C.render_weight_from_density_bwd(ri, ts, te, sig, T, a, gw, None, None))
bench("rendering_fwd_kernel (+ rays without samples)", 44 * N + 20 * R, lambda: C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True))
bench("rendering_bwd_kernel", 56 * N + 20 * R,
      lambda: C.rendering_bwd(ri, ts, te, sig, rgb, w, T, a, opa, dep, R, bk, True, gc, go, gd, None, None, None))
bench("accumulate_kernel<3>", 24 * N + 12 * R, lambda: C.accumulate_along_rays(ri, w, rgb, R))
bench("accumulate_kernel<1> (values=None)", 12 * N + 4 * R, lambda: C.accumulate_along_rays(ri, w, None, R))
bench("scan_keyed_kernel (exclusive sum)", 16 * N, lambda: C.exclusive_sum_cub(ri, x, False))
bench("scan_keyed_kernel (exclusive sum, reverse walk)", 16 * N, lambda: C.exclusive_sum_cub(ri, x, True))
bench("scan_packed_kernel (exclusive sum)", 8 * N + 16 * R, lambda: C.exclusive_sum(pk[:, 0].contiguous(), pk[:, 1].contiguous(), x, False, False))
sig_vis = sig * 0.01      # (rounds 1-5 formed this product inside the timed lambda: a 134 MB elementwise kernel, ~25 us of the row)
# bytes of the filter from what THIS call does: SURVEY 8d's 20 N in (keys 8, t_starts 4, t_ends 4, sigmas 4) + 16 N_out
# (keys, t_starts, t_ends of the survivors); no byte mask is written unless asked for (want_mask = False)
n_vis_out = C.visibility_compact(ri, ts, te, sig_vis, False, 1e-4, 0.0)[0].shape[0]
bench(f"visibility mask+compact ({n_vis_out / N:.3f} of the samples survive)", 20 * N + 16 * n_vis_out,
      lambda: C.visibility_compact(ri, ts, te, sig_vis, False, 1e-4, 0.0))
bench("pack_info_kernel", 16 * R, lambda: C.pack_info(ri, R))

if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        f.write(f"# streaming kernels at N = {N} samples, R = {R} rays (HIP events, median of 10)\n\n")
        f.write("| kernel | algorithmic MB | ms | GB/s | fraction of 8 TB/s |\n|---|---|---|---|---|\n")
        for n_, mb, ms, gbs, fr in rows:
            f.write(f"| {n_} | {mb:.1f} | {ms:.3f} | {gbs:.0f} | {fr:.3f} |\n")
