import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libstream.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "stream.hip"), "-o", so])
L = ctypes.CDLL(so)
dev = "cuda:0"; N = 1 << 24
keys = torch.arange(N, device=dev); a, b, c = (torch.rand(N, device=dev) for _ in range(3)); rgb = torch.rand(N, 3, device=dev)
o = [torch.empty(N, device=dev) for _ in range(3)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
for which, name in ((0, "dword lanes"), (1, "16-byte lanes")):
    for blocks in (2048, 8192, 65536):
        def f(): L.run(which, P(keys), P(a), P(b), P(c), P(rgb), P(o[0]), P(o[1]), P(o[2]), ctypes.c_int64(N), blocks, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(3): f()
        ms = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
        m = sorted(ms)[5]
        print(f"{name:14s} blocks={blocks:6d}  {44*N/m/1e6:8.0f} GB/s  ({m*1e3:.0f} us)")
