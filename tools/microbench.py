"""SURVEY.md 8(d) micro-workloads on one MI355X with the CPU oracle (single-threaded C port of the
reference algorithm) timed beside each:  M1 traversal + weights (128^3, 4096 rays, two grids),
M4 pdf path (4096 rays, 256 -> 96 -> 48), M6 the same traversal on a 256^3 grid.
M2 (roofline-scale streaming) is tools/roofline_sweep.py, M3/M5 is bench.py.

GPU: HIP events on the launch stream, median of 20 after 5 warm-ups (whole C-ABI op, all its
kernels).  CPU: perf_counter around the oracle call, median of 5.  Every GPU result is compared
with the oracle's before it is timed (bit-exact for the traversal and searchsorted, 1e-5 for floats).

    python tools/microbench.py [out.md]
"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from nerfacc_amd import cuda as C

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rows = []


def gpu_ms(fn, reps=20):
    for _ in range(5):
        fn()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]


def cpu_ms(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def row(workload, op, units, unit_name, alg_bytes, g_ms, c_ms):
    rows.append((workload, op, units, unit_name, alg_bytes, g_ms, c_ms))
    print(f"{workload:28s} {op:34s} {units:>10d} {unit_name:8s} GPU {g_ms*1e3:9.1f} us  {alg_bytes/g_ms/1e6:8.1f} GB/s"
          f"   CPU(1 thr) {c_ms:9.2f} ms   x{c_ms/g_ms:8.0f}", flush=True)


def m1_rays(R, seed=42):
    g = np.random.default_rng(seed)
    v = g.normal(size=(R, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = (0.5 + 1.5 * v).astype(np.float32)
    p = g.random((R, 3)).astype(np.float32)
    d = p - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def traversal_case(tag, res, grid_kind, R=4096):
    g = np.random.default_rng(42)
    if grid_kind == "random 50 %":
        occ = g.random((1, res, res, res)) > 0.5
    else:
        c = (np.arange(res) + 0.5) / res
        X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
        occ = (((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2) < 0.3**2)[None]
    aabb = np.array([[0, 0, 0, 1, 1, 1]], np.float32)
    o, d = m1_rays(R)
    step = np.float32(5e-3 / 3)
    near, far = np.zeros(R, np.float32), np.full(R, 1e10, np.float32)
    O, D, OCC, AABB, NEAR, FAR = T(o), T(d), T(occ), T(aabb), T(near), T(far)
    ri, ts, te, pk = C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, float(step), 0.0)
    iv, sm, _ = oracle.traverse_grids(o, d, occ, aabb, near, far, float(step), 0.0)
    assert np.array_equal(ri.cpu().numpy(), sm["ray_indices"]) and np.array_equal(ts.cpu().numpy(), iv["vals"][iv["is_left"]])
    N = ri.shape[0]
    wl = f"{tag} {res}^3 {grid_kind}"
    row(wl, "sampling traversal (count+offsets+emit)", N, "samples", 16 * N + 48 * R + res**3 // 8,
        gpu_ms(lambda: C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, float(step), 0.0)),
        cpu_ms(lambda: oracle.traverse_grids(o, d, occ, aabb, near, far, float(step), 0.0)))
    sig = (g.random(N) * 20).astype(np.float32)
    rgb = g.random((N, 3)).astype(np.float32)
    SIG, RGB = T(sig), T(rgb)
    tsn, ten, rin = ts.cpu().numpy(), te.cpu().numpy(), ri.cpu().numpy()
    w, Tr, al = C.render_weight_from_density_fwd(ri, ts, te, SIG, None)
    w_o, T_o, a_o = oracle.render_weight_from_density(tsn, ten, sig, rin)
    assert np.allclose(w.cpu().numpy(), w_o, atol=1e-5) and np.allclose(Tr.cpu().numpy(), T_o, atol=1e-5)
    row(wl, "render_weight_from_density fwd", N, "samples", 32 * N,
        gpu_ms(lambda: C.render_weight_from_density_fwd(ri, ts, te, SIG, None)),
        cpu_ms(lambda: oracle.render_weight_from_density(tsn, ten, sig, rin)))
    gw = g.random(N).astype(np.float32); GW = T(gw)
    gs = C.render_weight_from_density_bwd(ri, ts, te, SIG, Tr, al, GW, None, None)
    gs_o = oracle.render_weight_from_density_bwd(tsn, ten, sig, rin, g_w=gw)
    assert np.allclose(gs.cpu().numpy(), gs_o, atol=1e-4, rtol=1e-4)
    row(wl, "render_weight_from_density bwd", N, "samples", 28 * N,
        gpu_ms(lambda: C.render_weight_from_density_bwd(ri, ts, te, SIG, Tr, al, GW, None, None)),
        cpu_ms(lambda: oracle.render_weight_from_density_bwd(tsn, ten, sig, rin, g_w=gw)))
    bk = np.ones(3, np.float32); BK = T(bk)
    col = C.rendering_fwd(ri, ts, te, SIG, RGB, R, BK, True)[0]
    col_o = oracle.rendering(tsn, ten, rin, R, sig, rgb, bk)[0]
    assert np.allclose(col.cpu().numpy(), col_o, atol=1e-5)
    row(wl, "fused rendering fwd", N, "samples", 44 * N + 20 * R,
        gpu_ms(lambda: C.rendering_fwd(ri, ts, te, SIG, RGB, R, BK, True)),
        cpu_ms(lambda: oracle.rendering(tsn, ten, rin, R, sig, rgb, bk)))


def pdf_case(R=4096):
    import nerfacc_amd as nerfacc
    from nerfacc_amd.data_specs import RayIntervals
    g = np.random.default_rng(42)
    for n_in, n_out in ((256, 96), (96, 48)):
        edges = np.sort(g.random((R, n_in + 1)).astype(np.float32), -1)
        w = g.random((R, n_in)).astype(np.float32) + 1e-3
        cdf = np.concatenate([np.zeros((R, 1), np.float32), np.cumsum(w / w.sum(-1, keepdims=True), -1, dtype=np.float32)], -1)
        E, CDF = T(edges), T(cdf)
        iv, sm = nerfacc.importance_sampling(RayIntervals(vals=E), CDF, n_out, False)
        e_o, m_o = oracle.importance_sampling(edges, cdf, n_out)
        assert np.allclose(iv.vals.cpu().numpy(), e_o, atol=1e-6) and np.allclose(sm.vals.cpu().numpy(), m_o, atol=1e-6)
        row("M4 pdf 4096 rays", f"importance_sampling {n_in}->{n_out}", R * n_out, "samples", R * (8 * (n_in + 1) + 8 * n_out + 4),
            gpu_ms(lambda: nerfacc.importance_sampling(RayIntervals(vals=E), CDF, n_out, False)),
            cpu_ms(lambda: oracle.importance_sampling(edges, cdf, n_out)))
        q = np.sort(g.random((R, n_out + 1)).astype(np.float32), -1); Q = T(q)
        l, r = nerfacc.searchsorted(RayIntervals(vals=E), RayIntervals(vals=Q))
        l_o, r_o = oracle.searchsorted(edges, q)
        assert np.array_equal(l.cpu().numpy(), l_o) and np.array_equal(r.cpu().numpy(), r_o)
        row("M4 pdf 4096 rays", f"searchsorted {n_out + 1} in {n_in + 1}", R * (n_out + 1), "queries", R * (4 * (n_in + 1) + 20 * (n_out + 1)),
            gpu_ms(lambda: nerfacc.searchsorted(RayIntervals(vals=E), RayIntervals(vals=Q))),
            cpu_ms(lambda: oracle.searchsorted(edges, q)))


only = os.environ.get("NFA_MICROBENCH_ONLY")      # e.g. "M1-random" for a rocprofv3 run of one case
if only is None or only == "M1-random":
    traversal_case("M1", 128, "random 50 %")
if only is None or only == "M1-sphere":
    traversal_case("M1", 128, "sphere 11 %")
if only is None or only == "M6-sphere":
    traversal_case("M6", 256, "sphere 11 %")
if only is None or only == "M6-random":
    traversal_case("M6", 256, "random 50 %")
if only is None or only == "M4":
    pdf_case()
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("| workload | op | units | GPU us | G units/s | algorithmic GB/s | frac of 8 TB/s | CPU oracle ms (1 thread) | GPU/CPU |\n|---|---|---|---|---|---|---|---|---|\n")
        for wl, op, units, un, ab, g_ms, c_ms in rows:
            f.write(f"| {wl} | {op} | {units} {un} | {g_ms*1e3:.1f} | {units/g_ms/1e6:.2f} | {ab/g_ms/1e6:.1f} | {ab/g_ms/1e6/8000:.4f} | {c_ms:.2f} | {c_ms/g_ms:.0f}x |\n")
