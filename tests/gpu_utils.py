"""helpers shared by the -m gpu tests"""
import numpy as np
import torch

DEV = "cuda:0"


def t(x, dtype=None):
    x = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(DEV)


def n(x):
    return x.detach().cpu().numpy()


def ragged(rng, n_rays, max_len, p_empty=0.3):
    """random per-ray sample counts -> (ray_indices, packed_info) numpy"""
    cnts = rng.integers(1, max_len + 1, n_rays)
    cnts[rng.random(n_rays) < p_empty] = 0
    starts = np.cumsum(cnts) - cnts
    ri = np.repeat(np.arange(n_rays, dtype=np.int64), cnts)
    return ri, np.stack([starts, cnts], -1).astype(np.int64)


def scene(seed, n_rays=256, levels=4, res=32, occ=0.5):
    rng = np.random.default_rng(seed)
    o = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    base = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    c, e = (base[:3] + base[3:]) / 2, (base[3:] - base[:3]) / 2
    aabbs = np.stack([np.concatenate([c - e * 2**i, c + e * 2**i]) for i in range(levels)]).astype(np.float32)
    binaries = rng.random((levels, res, res, res)) < occ
    return o, d, aabbs, binaries


def lego_like(seed, n_rays, res=128):
    """rays from a sphere of radius 4 aimed into a +-1.5 box holding a blobby object"""
    rng = np.random.default_rng(seed)
    aabb = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32)
    g = (np.arange(res) + 0.5) / res * 3 - 1.5
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    occ = ((X**2 + Y**2 + Z**2) < 0.9**2) & ~((np.abs(X) < 0.3) & (np.abs(Y) < 0.3))
    occ |= (np.abs(X) < 1.2) & (np.abs(Y) < 1.2) & (np.abs(Z + 1.0) < 0.08)
    o = rng.standard_normal((n_rays, 3))
    o = (4.0 * o / np.linalg.norm(o, axis=-1, keepdims=True)).astype(np.float32)
    tgt = (rng.random((n_rays, 3)) * 3 - 1.5) * 0.9
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    return o, d, aabb, occ[None]


def sparse_like(seed, n_rays, res=128):
    """as lego_like, with an object small enough that the grid's sparse image (non-empty bricks) fits LDS beside the crossing-time
    arrays of the 512-thread count kernel — the window of the single-launch sampling call (round 6; the tests assert that it is)"""
    rng = np.random.default_rng(seed)
    aabb = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32)
    g = (np.arange(res) + 0.5) / res * 3 - 1.5
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    occ = ((X**2 + Y**2 + Z**2) < 0.62**2) & ~((np.abs(X) < 0.2) & (np.abs(Y) < 0.2))
    occ |= (np.abs(X) < 0.9) & (np.abs(Y) < 0.9) & (np.abs(Z + 0.8) < 0.05)
    occ |= ((X - 0.9) ** 2 + (Y + 0.8) ** 2 + (Z - 0.7) ** 2) < 0.15**2
    o = rng.standard_normal((n_rays, 3))
    o = (4.0 * o / np.linalg.norm(o, axis=-1, keepdims=True)).astype(np.float32)
    tgt = (rng.random((n_rays, 3)) * 3 - 1.5) * 0.9
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    return o, d, aabb, occ[None]


def sampling_is_fused(o, d, occ, aabb, step):
    """does nfa_traverse_sample take its single-launch form for this call?  (the C ABI's own answer: nfa_traverse_sample_fused)"""
    import ctypes

    from nerfacc_amd.cuda import _backend

    L = _backend.load_library()
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    a = _backend._traverse_args(O, D, None, B, A, None, None, None, None, None, step, 0.0, 0)
    R = O.shape[0]
    cnt, st, tot = (torch.zeros(k, dtype=torch.int64, device=O.device) for k in (R, R, 4))
    a.sm_cnts, a.sm_starts, a.totals = cnt.data_ptr(), st.data_ptr(), tot.data_ptr()
    a.workspace_bytes = L.nfa_traverse_workspace_bytes(R)
    return bool(L.nfa_traverse_sample_fused(ctypes.byref(a)))

