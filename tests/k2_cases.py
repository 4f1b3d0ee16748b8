"""Seeded inputs of the K2 (traverse_grids) fixture cases — numpy only, so that the build container
(where the fixtures are generated from the reference's own code, tests/golden/make_k2_golden.py), the CPU
tests (oracle vs fixture) and the GPU tests (HIP vs fixture) regenerate bit-identical inputs.  Large inputs
are never stored: the fixture keeps a SHA-256 of every regenerated input and the tests check it first.

Case list = VERDICT r1 item 1: the reference's tests/test_grid.py:38-68 configuration, SURVEY §8d M1(i) and
M1(ii) at 128^3 / 4096 rays, the lego-like +-1.5 scene at 4 k and 70 k rays, 2 x 256^3, cone-angle,
per-voxel and steps-limit modes, plus near/far planes, the over-allocated test mode with a ray mask, a
non-cubic grid and degenerate rays (zero direction components, origins inside the grid / on voxel faces)."""
import hashlib

import numpy as np


def sha(a) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def _unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def _levels(base, levels):
    base = np.asarray(base, np.float32)
    c, e = (base[:3] + base[3:]) / 2, (base[3:] - base[:3]) / 2
    return np.stack([np.concatenate([c - e * 2**i, c + e * 2**i]) for i in range(levels)]).astype(np.float32)


def _m1_rays(rng, R):
    """origins on the sphere of radius 1.5 about the unit cube's centre, aimed at U[0,1]^3"""
    o = rng.standard_normal((R, 3))
    o = (0.5 + 1.5 * o / np.linalg.norm(o, axis=-1, keepdims=True)).astype(np.float32)
    p = rng.random((R, 3)).astype(np.float32)
    return o, _unit(p - o)


def _sphere_grid(res, lo, hi, centre, radius):
    g = [(np.arange(r) + 0.5) / r * (hi - lo) + lo for r in res]
    X, Y, Z = np.meshgrid(*g, indexing="ij")
    return ((X - centre[0]) ** 2 + (Y - centre[1]) ** 2 + (Z - centre[2]) ** 2) < radius**2


def _lego(rng, R, res=128):
    """same procedural object as tests/gpu_utils.py::lego_like (aabb +-1.5, cameras on radius 4)"""
    g = (np.arange(res) + 0.5) / res * 3 - 1.5
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    occ = ((X**2 + Y**2 + Z**2) < 0.9**2) & ~((np.abs(X) < 0.3) & (np.abs(Y) < 0.3))
    occ |= (np.abs(X) < 1.2) & (np.abs(Y) < 1.2) & (np.abs(Z + 1.0) < 0.08)
    o = rng.standard_normal((R, 3))
    o = (4.0 * o / np.linalg.norm(o, axis=-1, keepdims=True)).astype(np.float32)
    tgt = (rng.random((R, 3)) * 3 - 1.5) * 0.9
    return o, _unit(tgt - o), np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32), occ[None]


def _multi(rng, R, levels, res, occ=0.5):
    o = rng.standard_normal((R, 3)).astype(np.float32)
    d = _unit(rng.standard_normal((R, 3)))
    res3 = (res,) * 3 if isinstance(res, int) else tuple(res)
    return o, d, _levels([-1, -1, -1, 1, 1, 1], levels), rng.random((levels,) + res3) < occ


def _degenerate(rng, R, res):
    """axis-aligned and plane-parallel rays, origins inside the grid, on voxel faces and corners, rays that miss"""
    o = (rng.random((R, 3)) * 2.4 - 1.2).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32)
    k = np.arange(R)
    d[k % 5 == 0, 0] = 0.0                                  # parallel to the yz planes
    d[k % 7 == 0, 1] = 0.0
    d[(k % 11 == 0), 1:] = 0.0                              # along +-x
    d[(k % 11 == 0), 0] = np.where(k[k % 11 == 0] % 2 == 0, 1.0, -1.0)
    d[np.abs(d).sum(-1) == 0, 2] = 1.0
    on_face = k % 3 == 0
    o[on_face] = (np.round((o[on_face] + 1) * res / 2) * 2 / res - 1).astype(np.float32)   # exactly on voxel planes
    d, aabbs = _unit(d), _levels([-1, -1, -1, 1, 1, 1], 2)
    # A ray lying IN a bounding plane of a level (origin on the face, zero direction component) makes the slab
    # test form 0 * inf = NaN and the marcher then converts inf / NaN to int (utils_grid.cuh:73-82): saturating on
    # a GPU, INT_MIN on x86 — the host build of the reference walks out of the grid there.  Such rays are moved
    # off the plane; they stay covered by the GPU-vs-oracle fuzz tests, which use the GPU's conversion rule.
    with np.errstate(all="ignore"):
        slab = (aabbs[None, :, :].reshape(1, -1, 2, 3) - o[:, None, None, :]) * (np.float32(1) / d)[:, None, None, :]
    in_plane = np.isnan(slab).any(axis=(1, 2, 3))
    o[in_plane] = o[in_plane] * np.float32(0.53)
    return o, d, aabbs, rng.random((2, res, res, res)) < 0.35


def _in_plane(rng, R, res, levels=2):
    """rays lying IN a bounding plane of a level (origin on the plane, zero direction component along its axis): the slab
    test forms 0 * inf = NaN and setup_traversal converts inf / NaN to int (utils_grid.cuh:72-82).  Only meaningful with the
    GPU's conversion rule: the fixture for this case comes from the `gpu` build of oracle/ref_shim (prelude.h, REF_GPU_F2I)."""
    aabbs = _levels([-1, -1, -1, 1, 1, 1], levels)
    o = (rng.random((R, 3)) * 3.0 - 1.5).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32)
    k = np.arange(R)
    ax = k % 3
    plane = np.array([-1.0, 1.0, -2.0, 2.0], np.float32)[(k // 3) % 4]
    o[k, ax] = plane
    d[k, ax] = 0.0
    two = k % 5 == 0                       # in two bounding planes at once (an edge line of the box)
    ax2 = (ax + 1) % 3
    o[two, ax2[two]] = np.array([-1.0, 1.0], np.float32)[(k[two] // 5) % 2]
    d[two, ax2[two]] = 0.0
    neg0 = k % 7 == 0                      # -0.0 instead of +0.0: the sign of 1/d flips the infinities
    d[neg0, ax[neg0]] = -0.0
    return o, _unit(d), aabbs, rng.random((levels, res, res, res)) < 0.35


# name -> (builder, traverse kwargs).  builder(rng) -> rays_o, rays_d, aabbs, binaries, extra dict of per-ray arrays
def _case(name):
    rng = np.random.default_rng(int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little"))
    extra, kw = {}, {}
    if name == "m1_noise":            # M1(i): rand > 0.5, the reference test's kind of grid, at 128^3
        o, d = _m1_rays(rng, 4096)
        aabbs, binaries = np.array([[0, 0, 0, 1, 1, 1]], np.float32), (rng.random((1, 128, 128, 128)) > 0.5)
        kw = dict(step_size=5e-3 / 3)
    elif name == "m1_sphere":         # M1(ii): solid sphere, centre 0.5 radius 0.3
        o, d = _m1_rays(rng, 4096)
        aabbs = np.array([[0, 0, 0, 1, 1, 1]], np.float32)
        binaries = _sphere_grid((128,) * 3, 0.0, 1.0, (0.5,) * 3, 0.3)[None]
        kw = dict(step_size=5e-3 / 3)
    elif name in ("lego_4k", "lego_70k", "lego_12k", "lego_160k"):
        # ray counts on either side of every ray-count switch of the one-level count pass: 4 k (16 lanes per ray, grid image in
        # LDS), 12 k (16 lanes, image from L2 — 8 k..16 k rays), 70 k (8 lanes), 160 k (a lane per ray, image from L2)
        o, d, aabbs, binaries = _lego(rng, {"lego_4k": 4096, "lego_12k": 12288, "lego_70k": 70000, "lego_160k": 160000}[name])
        kw = dict(step_size=5e-3)
    elif name == "lego_256":          # configs[4]'s shape: ONE level of 256^3 (the dense brick array is read from L2)
        o, d, aabbs, binaries = _lego(rng, 4096, res=256)
        kw = dict(step_size=5e-3)
    elif name == "two_level_256":     # C5-sized levels: 2 x 256^3
        o, d, aabbs, b0 = _lego(rng, 2048, res=256)
        aabbs = _levels(aabbs[0], 2)
        shell = _sphere_grid((256,) * 3, -3.0, 3.0, (0, 0, 0), 2.6) & ~_sphere_grid((256,) * 3, -3.0, 3.0, (0, 0, 0), 2.3)
        binaries = np.concatenate([b0, shell[None]])
        kw = dict(step_size=5e-3)
    elif name == "cone_angle":
        o, d, aabbs, binaries = _lego(rng, 2048, res=64)
        kw = dict(step_size=1e-2, cone_angle=0.004)
    elif name == "cone_angle_levels":
        o, d, aabbs, binaries = _multi(rng, 512, 4, 32, 0.4)
        kw = dict(step_size=0.02, cone_angle=0.01)
    elif name == "per_voxel":         # step_size <= 0: one interval per occupied voxel (grid.cu:210-211)
        o, d, aabbs, binaries = _multi(rng, 512, 4, 32, 0.5)
        kw = dict(step_size=-1.0)
    elif name == "steps_limit":       # two-pass with traverse_steps_limit (grid.cu:184,208)
        o, d, aabbs, binaries = _multi(rng, 512, 3, 32, 0.5)
        kw = dict(step_size=1e-2, traverse_steps_limit=37)
    elif name == "over_allocate":     # test-time marcher mode: single pass, ray mask honoured (grid.cu:359-404)
        o, d, aabbs, binaries = _lego(rng, 1024, res=64)
        extra["rays_mask"] = rng.random(1024) < 0.7
        extra["near_planes"] = (rng.random(1024) * 3.0).astype(np.float32)
        kw = dict(step_size=1e-2, traverse_steps_limit=8, over_allocate=True)
    elif name == "near_far":
        o, d, aabbs, binaries = _multi(rng, 512, 4, 32, 0.4)
        extra["near_planes"] = (0.3 * rng.random(512)).astype(np.float32)
        extra["far_planes"] = (2.5 * (0.5 + rng.random(512))).astype(np.float32)
        kw = dict(step_size=5e-3)
    elif name == "non_cubic":
        o, d, aabbs, binaries = _multi(rng, 512, 2, (20, 33, 7), 0.4)
        kw = dict(step_size=4e-3)
    elif name == "degenerate":
        o, d, aabbs, binaries = _degenerate(rng, 1024, 32)
        kw = dict(step_size=3e-3)
    elif name == "levels4_inside":    # the unbounded-scene shape: cameras inside the first of four levels, a blob + sparse far clutter
        R, res = 4096, 64
        c = (np.arange(res) + 0.5) / res * 2 - 1
        X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
        binaries = np.stack([(((X * 2.0**l) ** 2 + (Y * 2.0**l) ** 2 + (Z * 2.0**l) ** 2) < 0.25) | (rng.random((res,) * 3) < (0.004 if l else 0.0))
                             for l in range(4)])
        o = (0.6 * _unit(rng.standard_normal((R, 3)).astype(np.float32))).astype(np.float32)
        d = _unit(rng.standard_normal((R, 3)).astype(np.float32))
        aabbs = _levels([-1, -1, -1, 1, 1, 1], 4)
        extra["near_planes"] = np.full(R, 0.2, np.float32)
        kw = dict(step_size=2e-3)
    elif name == "in_plane":
        o, d, aabbs, binaries = _in_plane(rng, 1536, 32)
        kw = dict(step_size=3e-3)
    elif name in ("in_plane_one_level", "in_plane_four_levels"):
        o, d, aabbs, binaries = _in_plane(rng, 1536, 32, levels=1 if name == "in_plane_one_level" else 4)
        kw = dict(step_size=3e-3)
    else:
        raise KeyError(name)
    return dict(rays_o=o, rays_d=d, aabbs=aabbs, binaries=binaries, extra=extra, kw=kw)


# "ref_test_grid" (tests/test_grid.py:38-68 with torch's CPU generator, seed 42) stores its inputs in the fixture
GENERATED = ["m1_noise", "m1_sphere", "lego_4k", "lego_70k", "two_level_256", "cone_angle", "cone_angle_levels",
             "per_voxel", "steps_limit", "over_allocate", "near_far", "non_cubic", "degenerate", "levels4_inside",
             "lego_256", "lego_12k", "lego_160k"]            # (the last three: round 4)
ALL = ["ref_test_grid"] + GENERATED
GPU_RULE = ["in_plane", "in_plane_one_level", "in_plane_four_levels"]     # tests/golden/k2_inplane.npz: the reference built with the GPU's float -> int conversion rule
FULL_LIMIT = 30000      # cases with fewer samples keep every output array in the fixture; the others digests


def build_case(name, fixture=None):
    """inputs of a case; 'ref_test_grid' is read back from the fixture (it was drawn with torch's generator)"""
    if name == "ref_test_grid":
        f = fixture
        return dict(rays_o=f["ref_test_grid/rays_o"], rays_d=f["ref_test_grid/rays_d"], aabbs=f["ref_test_grid/aabbs"],
                    binaries=np.unpackbits(f["ref_test_grid/binaries_bits"]).astype(bool).reshape(4, 32, 32, 32),
                    extra={}, kw={})
    return _case(name)


def input_digest(c) -> str:
    parts = [c["rays_o"], c["rays_d"], c["aabbs"], np.packbits(c["binaries"].ravel())]
    parts += [np.asarray(c["extra"][k]) for k in sorted(c["extra"])]
    return sha(np.concatenate([np.frombuffer(np.ascontiguousarray(p).tobytes(), np.uint8) for p in parts]))


OUTPUT_KEYS = ["iv_vals", "iv_ray_indices", "iv_is_left", "iv_is_right", "iv_chunk_starts", "iv_chunk_cnts",
               "sm_vals", "sm_ray_indices", "sm_is_valid", "sm_chunk_starts", "sm_chunk_cnts", "term_live"]


def pack_outputs(iv, sm, term, live=None):
    """the comparable outputs of one traverse_grids call as a flat dict of numpy arrays.  `iv` / `sm` are
    mappings with the RaySegmentsSpec field names.  terminate_planes is compared on rays that emitted
    something only: the reference never writes it for the others (grid.cu:103-106 skips them in the fill
    pass and the buffer is torch::empty, grid.cu:363)."""
    g = lambda d, k: np.asarray(d[k])
    out = dict(iv_vals=g(iv, "vals").astype(np.float32), iv_ray_indices=g(iv, "ray_indices").astype(np.int64),
               iv_is_left=g(iv, "is_left").astype(bool), iv_is_right=g(iv, "is_right").astype(bool),
               iv_chunk_starts=g(iv, "chunk_starts").astype(np.int64), iv_chunk_cnts=g(iv, "chunk_cnts").astype(np.int64),
               sm_vals=g(sm, "vals").astype(np.float32), sm_ray_indices=g(sm, "ray_indices").astype(np.int64),
               sm_is_valid=g(sm, "is_valid").astype(bool),
               sm_chunk_starts=g(sm, "chunk_starts").astype(np.int64), sm_chunk_cnts=g(sm, "chunk_cnts").astype(np.int64))
    if live is None:     # over-allocated mode: pass the ray mask (every unmasked ray writes its plane, grid.cu:274)
        live = (out["sm_chunk_cnts"] > 0) | (out["iv_chunk_cnts"] > 0)
    out["term_live"] = np.where(live, np.asarray(term, np.float32), np.float32(0))
    return out


def check_against_fixture(name, out, fixture):
    """assert that `out` (pack_outputs) equals what the reference produced; returns the number of samples"""
    for k in OUTPUT_KEYS:
        want = str(fixture[f"{name}/sha/{k}"])
        if f"{name}/full/{k}" in fixture:
            ref = fixture[f"{name}/full/{k}"]
            assert out[k].shape == ref.shape, (name, k, out[k].shape, ref.shape)
            bad = np.flatnonzero(out[k] != ref)
            assert bad.size == 0, (name, k, f"{bad.size} of {ref.size} differ, first at {bad[:5]}")
        elif k in ("sm_chunk_cnts", "iv_chunk_cnts"):
            ref = fixture[f"{name}/cnts/{k}"].astype(np.int64)
            bad = np.flatnonzero(out[k] != ref)
            assert bad.size == 0, (name, k, f"{bad.size} rays differ, first {bad[:5]}")
        assert sha(out[k]) == want, (name, k, "digest differs from the reference's output")
    return int(out["sm_chunk_cnts"].sum())
