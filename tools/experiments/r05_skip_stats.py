"""How much of a training ray's voxel walk is empty space?  (CPU, numpy; approximate float64 DDA — statistics only.)
On bench.py's steady state (profiles/r02_sampling_state.npz: 128^3 grid, 6.1 % occupied, 6 564 rays): voxels visited per ray, the share
that lies in empty 4^3 bricks, and the number of walk iterations when every empty region is left in ONE macro step sized by the brick
distance field (Chebyshev distance to the nearest non-empty brick, capped) — for the whole ray and for the slowest of 16 parts.

    python tools/experiments/r05_skip_stats.py [n_rays]
"""
import os, sys
import numpy as np
from scipy import ndimage
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = np.load(os.path.join(ROOT, "profiles", "r02_sampling_state.npz"))
res = tuple(int(x) for x in st["res"])
occ = np.unpackbits(st["binaries_bits"])[: int(np.prod(res))].astype(bool).reshape(res)[0]
n = res[-1]
b4 = occ.reshape(n // 4, 4, n // 4, 4, n // 4, 4).any(axis=(1, 3, 5))
D = ndimage.distance_transform_cdt(~b4, metric="chessboard")
print(f"grid {n}^3, occupied {occ.mean():.4f}, non-empty bricks {b4.mean():.4f}; bricks by distance 0..8:", np.bincount(np.minimum(D, 8).ravel()).tolist())
lo, hi = st["aabbs"].reshape(-1)[:3], st["aabbs"].reshape(-1)[3:6]
O, Dr = st["rays_o"].astype(np.float64), st["rays_d"].astype(np.float64)
sel = np.random.default_rng(0).choice(O.shape[0], int(sys.argv[1]) if len(sys.argv) > 1 else 500, replace=False)
caps = (1, 2, 3, 4, 15)
its = {c: 0 for c in caps}; part_max = {c: [] for c in caps}; vox = in_empty = rays = 0
for r in sel:
    o, d = O[r], Dr[r]
    inv = 1 / np.where(d == 0, 1e-30, d)
    t0, t1 = (lo - o) * inv, (hi - o) * inv
    tn, tf = np.minimum(t0, t1).max(), np.maximum(t0, t1).min()
    if tn >= tf or tf <= 0:
        continue
    tn = max(tn, 0)
    c = np.clip(((o + d * (tn + 1e-6) - lo) / (hi - lo) * n).astype(int), 0, n - 1)
    s = np.sign(d).astype(int)
    t = (lo + (c + (s > 0)) * (hi - lo) / n - o) * inv
    dt = np.abs((hi - lo) / n * inv)
    cells = []
    while True:
        cells.append(tuple(c))
        a = np.argmin(t); c[a] += s[a]; t[a] += dt[a]
        if c[a] < 0 or c[a] >= n:
            break
    cells = np.array(cells); N = len(cells); rays += 1; vox += N
    in_empty += (~b4[cells[:, 0] // 4, cells[:, 1] // 4, cells[:, 2] // 4]).sum()
    m = np.argmax(np.abs(cells[-1] - cells[0])); mj = np.abs(cells[:, m] - cells[0, m]); nm = mj[-1] + 1

    def iters(i, e, Dc):
        k = 0
        while i < e:
            b = cells[i] // 4; dd = Dc[b[0], b[1], b[2]]
            if dd >= 1:
                j = i
                while j < e and (np.abs(cells[j] // 4 - b) <= dd - 1).all():
                    j += 1
                i = j
            else:
                i += 1
            k += 1
        return k
    for cap in caps:
        Dc = np.minimum(D, cap)
        its[cap] += iters(0, N, Dc)
        mx = 0
        for p in range(16):
            idx = np.where((mj >= p * nm // 16) & (mj < (p + 1) * nm // 16))[0]
            if len(idx):
                mx = max(mx, iters(idx[0], idx[-1] + 1, Dc))
        part_max[cap].append(mx)
print(f"{rays} rays: {vox / rays:.1f} voxels per ray, {in_empty / vox:.3f} of them in empty bricks")
for cap in caps:
    print(f"  distance cap {cap:2d}: {its[cap] / rays:6.1f} iterations per ray; slowest of 16 parts: mean {np.mean(part_max[cap]):.1f}, p90 {np.percentile(part_max[cap], 90):.0f}")
