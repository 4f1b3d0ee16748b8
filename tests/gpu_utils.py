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
