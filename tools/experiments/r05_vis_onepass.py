"""Visibility filter: the one-pass look-back form against the mask / scan / compaction kernels, per input size and tile.
    python tools/experiments/r05_vis_onepass.py [log2_N ...]
HIP events around the extension call (includes its read-back of the count), median of 10 after 3 warm-ups; 37 algorithmic bytes per sample."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nerfacc_amd
from nerfacc_amd import cuda as C

dev = torch.device("cuda:0")
logs = [int(a) for a in sys.argv[1:]] or [18, 20, 22, 24]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]


for logn in logs:
    g = torch.Generator(device=dev).manual_seed(42)
    R = (1 << logn) // 96
    cnts = torch.randint(0, 193, (R,), device=dev, generator=g)
    ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
    N = ri.shape[0]
    ts = torch.rand(N, device=dev, generator=g) * 4
    te = ts + 5e-3
    for name, sig in (("all kept", torch.rand(N, device=dev, generator=g) * 0.3), ("early stop", torch.rand(N, device=dev, generator=g) * 30)):
        row = []
        ref = None
        for form in ({"vis_onepass": 0},) + tuple({"vis_onepass": 1, "vis_chunks": c} for c in (2, 3, 4, 5, 6)):
            with nerfacc_amd.options(**form):
                out = C.visibility_compact(ri, ts, te, sig, False, 1e-4, 0.0)
                if ref is None:
                    ref = out
                ok = all(torch.equal(a, b) for a, b in zip(ref[:3], out[:3]))
                ms = timeit(lambda: C.visibility_compact(ri, ts, te, sig, False, 1e-4, 0.0))
            k = out[0].shape[0]
            row.append(f"{'3k' if not form['vis_onepass'] else 'c' + str(form['vis_chunks'])} {ms*1e3:7.1f} us {((20 * N + 16 * k) / ms / 1e6):6.0f} GB/s{'' if ok else ' MISMATCH'}")
        print(f"N=2^{logn} ({N}) {name:10s} kept {k/N:.2f} | " + " | ".join(row), flush=True)
