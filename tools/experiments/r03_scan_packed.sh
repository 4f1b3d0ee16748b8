cd /root/repo
timeout 600 python -m pytest tests/test_gpu_scan.py tests/test_gpu_tiles.py -x -q 2>&1 | tail -3
cat > /tmp/sp.py <<'PY'
import os, sys, torch
sys.path.insert(0, '/root/repo')
from nerfacc_amd import cuda as C
dev = torch.device('cuda:0')
def run(R, hi, label):
    g = torch.Generator(device=dev).manual_seed(1)
    cnts = torch.randint(0, hi, (R,), device=dev, generator=g)
    starts = torch.cumsum(cnts, 0) - cnts
    N = int(cnts.sum()); x = torch.rand(N, device=dev)
    def t(fn, reps=10):
        for _ in range(3): fn()
        ms = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
        return sorted(ms)[len(ms)//2]
    out = []
    ref = None
    for rw in ("4", "16", "64", ""):
        if rw: os.environ["NFA_SCAN_RW"] = rw
        else: os.environ.pop("NFA_SCAN_RW", None)
        y = C.exclusive_sum(starts, cnts, x, False, False)
        if ref is None: ref = y
        ok = torch.equal(y, ref)
        ms = t(lambda: C.exclusive_sum(starts, cnts, x, False, False))
        out.append(f"rw={rw or 'auto'} {ms*1e3:.1f} us {(8*N+16*R)/ms/1e9:.2f} TB/s{'' if ok else ' MISMATCH'}")
    print(label, N, R, " | ".join(out))
run(174762, 193, "2^24"); run(699050, 193, "2^26"); run(6500, 81, "2^18"); run(2**24//40, 81, "2^24 short rows"); run(2**24//1000, 2001, "2^24 long rows")
PY
timeout 300 python /tmp/sp.py
