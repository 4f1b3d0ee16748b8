cd /root/repo
timeout 300 python -m pytest tests/test_gpu_volrend.py -x -q -k "visibility" 2>&1 | tail -3
cat > /tmp/vis_bench.py <<'PY'
import os, sys, torch
sys.path.insert(0, '/root/repo')
from nerfacc_amd import cuda as C
dev = torch.device('cuda:0')
def run(R, label):
    g = torch.Generator(device=dev).manual_seed(1)
    cnts = torch.randint(0, 193, (R,), device=dev, generator=g)
    ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts); N = ri.shape[0]
    ts = torch.rand(N, device=dev); te = ts + 5e-3; sig = torch.rand(N, device=dev) * 0.01
    def t(fn, reps=10):
        for _ in range(3): fn()
        ms = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
        return sorted(ms)[len(ms)//2]
    out = []
    for env in (("twopass", "", "0"), ("onepass", "", "0"), ("onepass", "128", "0"), ("onepass", "256", "0"), ("onepass", "384", "0"), ("onepass", "640", "0"), ("onepass", "256", "1"), ("onepass", "256", "2")):
        os.environ["NFA_VIS"] = env[0]; os.environ["NFA_VIS_DBG"] = env[2]
        if env[1]: os.environ["NFA_VIS_TILE"] = env[1]
        else: os.environ.pop("NFA_VIS_TILE", None)
        ms = t(lambda: C.visibility_compact(ri, ts, te, sig, False, 1e-4, 0.0))
        out.append(f"{env[0]}{'/'+env[1] if env[1] else ''} dbg{env[2]} {ms*1e3:.1f}")
    print(label, N, " | ".join(out))
run(174762, "2^24"); run(2730, "2^18"); run(699050, "2^26")
PY
timeout 300 python /tmp/vis_bench.py
