"""GPU timeline of ONE training step of bench.py's loops (torch.profiler, kernel events): start (us from the step's
first kernel), duration, idle gap before, kernel name.  python tools/step_timeline.py [api|overlap] [pretrain]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "api"
pre = sys.argv[2] if len(sys.argv) > 2 else "1500"
sys.argv = ["bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-other-mode", "--no-aux", "--windows", "1", "--mode", mode, "--pretrain", pre]
import bench
from torch.profiler import ProfilerActivity, profile
captured = {}
orig = bench.profile_steps
def hook(step_fn, n_steps):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(6):
            step_fn()
        torch.cuda.synchronize()
    ks = [(float(e.time_range.start), float(e.device_time), e.name) for e in prof.events()
          if "cuda" in str(getattr(e, "device_type", "")).lower() and float(getattr(e, "device_time", 0) or 0) > 0]
    ks.sort()
    # split into steps at the big randint / first kernel pattern: use gaps; simply print steps 3..4 by kernel count
    n = len(ks) // 6
    seg = ks[3 * n: 4 * n + 3]
    t0 = seg[0][0]; prev_end = seg[0][0]
    busy = 0.0
    for s, d, name in seg:
        gap = s - prev_end
        short = name.replace("void ", "").replace("at::native::", "").split("(")[0][:64]
        print(f"{s - t0:8.1f} {d:7.1f}  gap {gap:6.1f}  {short}")
        prev_end = max(prev_end, s + d); busy += d
    print(f"span {prev_end - t0:.1f} us busy {busy:.1f} us kernels {len(seg)}")
    return orig(step_fn, n_steps)
bench.profile_steps = hook
bench.main()
