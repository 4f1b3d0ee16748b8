"""Eight procedural scenes that differ the way nerf_synthetic's eight do (BASELINE.json configs[4]: the reference's
8-scene sweep, docs/source/examples/static/ngp.rst:36-42 — chair, drums, ficus, hotdog, lego, materials, mic, ship) as far
as the OCCUPANCY-GRID path is concerned: occupied fraction, samples per ray, runs per ray, run length.  The datasets do not
exist on this machine; what the sampling kernels see of a scene is its occupancy grid and its rays, and these span the
regimes the data-dependent plan switches of the library react to (lanes per ray of the count pass, grid image in LDS or L2,
emit form).

Every scene is an analytic occupancy function on world points in the aabb [-1.5, 1.5]^3, written against an array module
`xp` (numpy or torch) so that tools/scene_sweep.py (grids + kernels), bench.py's aux leg (teacher fields) and the tests build
the same objects.  `rays(name, n, seed)` draws cameras on the radius-4 sphere aimed into the object, like tests/k2_cases._lego.
"""
import numpy as np

AABB = np.array([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], np.float32)


def _box(xp, x, c, h):
    return (xp.abs(x[..., 0] - c[0]) <= h[0]) & (xp.abs(x[..., 1] - c[1]) <= h[1]) & (xp.abs(x[..., 2] - c[2]) <= h[2])


def _ball(xp, x, c, r):
    return ((x[..., 0] - c[0]) ** 2 + (x[..., 1] - c[1]) ** 2 + (x[..., 2] - c[2]) ** 2) < r * r


def _rod(xp, x, p, q, r):
    """points closer than r to the segment p-q"""
    d = [q[i] - p[i] for i in range(3)]
    dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
    t = ((x[..., 0] - p[0]) * d[0] + (x[..., 1] - p[1]) * d[1] + (x[..., 2] - p[2]) * d[2]) / dd
    t = xp.clip(t, 0.0, 1.0) if xp is np else t.clamp(0.0, 1.0)
    return ((x[..., 0] - p[0] - t * d[0]) ** 2 + (x[..., 1] - p[1] - t * d[1]) ** 2 + (x[..., 2] - p[2] - t * d[2]) ** 2) < r * r


def lego(xp, x):
    """the bench's object (bench.py: lego_like_density): a union of boxes with studs, ~6 % occupied, ~40 samples per ray in long runs"""
    body = _box(xp, x, (0.0, 0.0, -0.25), (0.75, 0.45, 0.2))
    cabin = _box(xp, x, (-0.25, 0.0, 0.2), (0.3, 0.35, 0.25)) & ~_box(xp, x, (-0.25, 0.0, 0.25), (0.22, 0.4, 0.12))
    plate = _box(xp, x, (0.0, 0.0, -0.55), (0.95, 0.7, 0.06))
    arm = _box(xp, x, (0.65, 0.0, 0.1), (0.35, 0.08, 0.08)) | _box(xp, x, (0.95, 0.0, -0.1), (0.06, 0.4, 0.25))
    studs = (xp.sin(x[..., 0] * 24.0) * xp.sin(x[..., 1] * 24.0) > 0.5) & _box(xp, x, (0.0, 0.0, -0.45), (0.9, 0.65, 0.05))
    return body | cabin | plate | arm | studs


def ficus(xp, x):
    """thin structures (ficus / mic): a trunk, twenty thin branches and small leaf blobs — many SHORT runs per ray"""
    g = np.random.default_rng(3)
    occ = _rod(xp, x, (0.0, 0.0, -1.0), (0.0, 0.0, 0.2), 0.05) | _box(xp, x, (0.0, 0.0, -1.05), (0.3, 0.3, 0.06))
    for _ in range(20):
        a = g.uniform(0, 2 * np.pi)
        z0, ln, up = g.uniform(-0.4, 0.2), g.uniform(0.4, 0.9), g.uniform(0.1, 0.7)
        q = (float(ln * np.cos(a)), float(ln * np.sin(a)), float(z0 + up))
        occ = occ | _rod(xp, x, (0.0, 0.0, float(z0)), q, 0.018)
        occ = occ | _ball(xp, x, q, 0.07)
    return occ


def ship(xp, x):
    """a dense slab (ship on water): a quarter of the volume occupied, one very long run per ray"""
    water = _box(xp, x, (0.0, 0.0, -0.55), (1.4, 1.4, 0.4))
    hull = _box(xp, x, (0.0, 0.0, 0.0), (0.7, 0.25, 0.18)) | _rod(xp, x, (0.0, 0.0, 0.0), (0.0, 0.0, 0.9), 0.03)
    return water | hull


def shell(xp, x):
    """a hollow shell (hotdog's plate, a bowl): two runs per ray, far apart"""
    return _ball(xp, x, (0.0, 0.0, 0.0), 1.1) & ~_ball(xp, x, (0.0, 0.0, 0.0), 1.0)


def speck(xp, x):
    """a near-empty grid (< 1 % occupied): most rays have no samples at all"""
    return _ball(xp, x, (0.3, -0.2, 0.1), 0.22) | _ball(xp, x, (-0.6, 0.5, -0.3), 0.12)


def noise(xp, x):
    """the reference's own test grid (tests/test_grid.py: rand > 0.5) as a function of position: a boundary every other voxel at
    ANY grid resolution up to 256 — hundreds of two-sample runs per ray"""
    u = (x + 1.5) * (256.0 / 3.0)
    i = u.astype(np.int64) if xp is np else u.long()
    h = (i[..., 0] * 73856093) ^ (i[..., 1] * 19349663) ^ (i[..., 2] * 83492791)
    return ((h >> 7) & 1) == 1


def drums(xp, x):
    """several separate medium objects (drums, chair legs): 3-6 medium runs per ray"""
    occ = _box(xp, x, (0.0, 0.0, -0.9), (1.2, 1.2, 0.05))
    for cx, cy, cz, r in ((-0.6, -0.5, -0.4, 0.33), (0.55, -0.55, -0.45, 0.3), (0.0, 0.45, -0.35, 0.4), (-0.7, 0.6, 0.2, 0.2),
                          (0.75, 0.5, 0.3, 0.18), (0.1, -0.1, 0.5, 0.16)):
        occ = occ | (_ball(xp, x, (cx, cy, cz), r) & ~_ball(xp, x, (cx, cy, cz), r - 0.06)) | _rod(xp, x, (cx, cy, -0.9), (cx, cy, cz), 0.02)
    return occ


def materials(xp, x):
    """a periodic array of small balls (materials): many medium-short runs, regular spacing"""
    p = 0.42
    fx = x[..., 0] - p * xp.floor(x[..., 0] / p + 0.5)
    fy = x[..., 1] - p * xp.floor(x[..., 1] / p + 0.5)
    fz = x[..., 2] + 0.5
    inside = (xp.abs(x[..., 0]) < 1.2) & (xp.abs(x[..., 1]) < 1.2)
    return ((fx * fx + fy * fy + fz * fz) < 0.15**2) & inside | _box(xp, x, (0.0, 0.0, -0.7), (1.25, 1.25, 0.04))


SCENES = {"lego": lego, "ficus": ficus, "ship": ship, "shell": shell, "speck": speck, "noise": noise, "drums": drums, "materials": materials}


def occupancy_grid(name, res=256):
    """bool [1, res, res, res] (x-major, the estimator's `binaries` layout) of the scene sampled at the voxel centres"""
    g = ((np.arange(res, dtype=np.float32) + 0.5) / res) * 3.0 - 1.5
    out = np.empty((res, res, res), bool)
    Y, Z = np.meshgrid(g, g, indexing="ij")
    for i in range(res):                       # a slab at a time: the full meshgrid of a 256^3 grid is 200 MB per coordinate
        x = np.stack([np.full_like(Y, g[i]), Y, Z], -1)
        out[i] = SCENES[name](np, x)
    return out[None]


def rays(n, seed=0, spread=0.9):
    """cameras on the radius-4 sphere looking at random points of the object's box (tests/k2_cases._lego's generator)"""
    g = np.random.default_rng(seed)
    o = g.standard_normal((n, 3))
    o = (4.0 * o / np.linalg.norm(o, axis=-1, keepdims=True)).astype(np.float32)
    tgt = (g.random((n, 3)) * 3 - 1.5) * spread
    d = tgt - o
    return o, (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
