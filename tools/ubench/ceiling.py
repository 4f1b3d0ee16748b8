"""The box's own streaming ceilings (tools/ubench/ceiling.hip): best of a sweep over workgroups (2..32 waves per SIMD worth),
accesses in flight per lane (1 / 2 / 4) and plain / non-temporal accesses, for copy, read-only and write-only kernels with
16-byte lanes.  `ceilings(n_bytes)` -> {"copy": (GB/s, config), "read": ..., "write": ...}; run as a script for the full table."""
import ctypes, os, subprocess, sys
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libceiling.so")


def _lib():
    src = os.path.join(HERE, "ceiling.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", SO])
    return ctypes.CDLL(SO)


def _time(f, reps=7):
    for _ in range(2):
        f()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]


def ceilings(n_bytes=1 << 30, verbose=False):
    L = _lib()
    dev = "cuda:0"
    n4 = n_bytes // 16
    src = torch.rand(n4 * 4, device=dev)
    dst = torch.empty_like(src)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    best = {}
    for kind, name, traffic in ((0, "copy", 2 * n_bytes), (1, "read", n_bytes), (2, "write", n_bytes)):
        for blocks in (1024, 2048, 4096, 8192, 16384, 65536):
            for unroll in ((1, 2, 4) if kind < 2 else (1,)):
                for nt in (0, 1):
                    f = lambda: L.run(kind, unroll, nt, P(src), P(dst), ctypes.c_int64(n4), blocks,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                    gbs = traffic / _time(f) / 1e6
                    cfg = f"{blocks} workgroups x 256, {unroll} x 16 B in flight per lane, {'non-temporal' if nt else 'plain'}"
                    if verbose:
                        print(f"{name:6s} {cfg:70s} {gbs:7.0f} GB/s")
                    if name not in best or gbs > best[name][0]:
                        best[name] = (gbs, cfg)
    return best


if __name__ == "__main__":
    b = ceilings(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30, verbose=True)
    for k, (g, c) in b.items():
        print(f"best {k:6s} {g:7.0f} GB/s = {g / 8000:.3f} of 8 TB/s   ({c})")
