"""Kernel duration vs N of the tiled streaming kernels (small-N regime), from the rocprofv3 kernel trace.

  driver (on the GPU box):  rocprofv3 --kernel-trace --output-format csv -d D -o lat -- python tools/latency_sweep.py run
  report:                   python tools/latency_sweep.py report D/*kernel_trace.csv
Each N launches every kernel 30 times; the report groups dispatches by (kernel, grid size) = (kernel, N)."""
import csv
import os
import sys

SIZES = [1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21]


def run(cold=False):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nerfacc_amd import cuda as C
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    big = torch.empty(1 << 28, device=dev) if cold else None      # 1 GiB: written between calls, evicts L2 + Infinity Cache
    def thrash():
        if cold:
            big.add_(1.0)
    for n_target in ([1 << 16, 1 << 18, 1 << 20] if cold else SIZES):
        R = n_target // 40
        cnts = torch.randint(10, 71, (R,), device=dev, generator=g)
        ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
        N = ri.shape[0]
        ts = torch.rand(N, device=dev, generator=g) * 4
        te = ts + 5e-3
        sig = torch.rand(N, device=dev, generator=g) * 30
        rgb = torch.rand(N, 3, device=dev, generator=g)
        bk = torch.ones(3, device=dev)
        col, opa, dep, w, T, a = C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
        gc = torch.rand_like(col)
        x = torch.rand(N, device=dev, generator=g)
        print("N", N, flush=True)
        for _ in range(30):
            thrash()
            C.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
            thrash()
            C.rendering_bwd(ri, ts, te, sig, rgb, w, T, a, opa, dep, R, bk, True, gc, None, None, None, None, None)
            thrash()
            C.render_weight_from_density_fwd(ri, ts, te, sig, None)
            thrash()
            C.exclusive_sum_cub(ri, x, False)
            thrash()
            C.visibility_compact(ri, ts, te, sig * 0.02, False, 1e-4, 0.0)
            thrash()
            C.accumulate_along_rays(ri, w, rgb, R)
            thrash()
            y = x + 1.0          # reference point: a plain elementwise kernel of the same N
        torch.cuda.synchronize()


def report(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))          # launch order = ascending N
    groups = {}
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:60]
        key = (name, int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]))
        groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = sorted({k[0] for k in groups if "nfa::" in k[0] or "CUDAFunctorOnSelf_add" in k[0] or "AUnaryFunctor" in k[0]})
    for nm in names:
        line = []
        for (k, grid), v in groups.items():                      # (insertion order: the sizes in the order they ran)
            if k == nm and len(v) >= 20:
                v = sorted(v)
                line.append("%d:%.1f" % (grid, v[len(v) // 2]))
        print("%-62s %s" % (nm, "  ".join(line)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "cold":
        run(cold=True)
    else:
        report(sys.argv[2])
