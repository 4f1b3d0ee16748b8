"""The single-launch form of the sampling call (round 6: count pass + look-back over its workgroups + emit pass of every wave's own
rays in ONE kernel, csrc/sample_fused.hpp; C ABI nfa_traverse_sample) against the reference-built fixture
(tests/golden/k2_reference.npz: /root/reference/nerfacc/cuda/csrc/grid.cu:68-282, 320-474 compiled for the host) and against the
three-kernel form it replaces — same ray_indices / packed offsets bit for bit, same t_starts / t_ends bit for bit.

What a launch does depends on the caller's GUESS of the output size (the extension sizes it from its previous call's samples per
ray): no guess (count + offsets only, the emit pass follows the read-back), a good guess (everything in the one launch), a guess
that is too small (the waves whose samples end beyond it store nothing; the caller launches the emit pass with exact outputs).
Every one of them is driven here, back to back, so that a sync block left dirty by one launch would break the next."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import k2_cases as K  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu

FUSABLE = ["m1_sphere"]                     # one level, a grid whose sparse image fits LDS beside the crossing-time arrays, 4096 rays: the fused form's window
NOT_FUSABLE = ["lego_4k", "m1_noise"]       # 5 636 / 32 768 non-empty bricks: the image does not fit — nfa_traverse_sample takes the three launches


@pytest.fixture(scope="module")
def k2():
    return dict(np.load(os.path.join(GOLD, "k2_reference.npz")))


def _case(name, k2):
    from test_k2_reference import _inputs

    return _inputs(name, k2)


def _sample(c, rays=None):
    """the extension's fused entry point: (ray_indices, t_starts, t_ends, packed_info)"""
    import torch

    from gpu_utils import t
    from nerfacc_amd import cuda as C

    o, d = c["rays_o"], c["rays_d"]
    if rays is not None:
        o, d = o[rays], d[rays]
    out = C.sample_occgrid(t(o), t(d), t(c["binaries"]), t(c["aabbs"]), None, None, c["kw"]["step_size"], 0.0,
                           near_plane=0.0, far_plane=float("inf"))
    torch.cuda.synchronize()
    return out


def _check(name, k2, out, n_rays):
    from gpu_utils import n

    ri, ts, te, packed = (n(x) for x in out)
    assert K.sha(ri.astype(np.int64)) == str(k2[f"{name}/sha/sm_ray_indices"])
    ref_cnts = k2[f"{name}/cnts/sm_chunk_cnts"] if f"{name}/cnts/sm_chunk_cnts" in k2 else k2[f"{name}/full/sm_chunk_cnts"]
    assert packed.shape == (n_rays, 2)
    assert np.array_equal(packed[:, 1], ref_cnts)                                   # pack offsets: bit-exact (north_star)
    assert np.array_equal(packed[:, 0], np.cumsum(ref_cnts) - ref_cnts)
    assert K.sha(ts.astype(np.float32)) == str(k2[f"{name}/sha/t_starts"])
    assert K.sha(te.astype(np.float32)) == str(k2[f"{name}/sha/t_ends"])


def _is_fused(c):
    """does the library take its single-launch form for this call?  (asked through the C ABI: nfa_traverse_sample_fused)"""
    import ctypes

    import torch

    from gpu_utils import t
    from nerfacc_amd.cuda import _backend

    L = _backend.load_library()
    o, d, ab, B = t(c["rays_o"]), t(c["rays_d"]), t(c["aabbs"]), t(c["binaries"])
    a = _backend._traverse_args(o, d, None, B, ab, None, None, None, None, None, c["kw"]["step_size"], 0.0, 0)
    R = o.shape[0]
    cnt, st, tot = (torch.zeros(k, dtype=torch.int64, device=o.device) for k in (R, R, 4))
    a.sm_cnts, a.sm_starts, a.totals = cnt.data_ptr(), st.data_ptr(), tot.data_ptr()
    a.workspace_bytes = L.nfa_traverse_workspace_bytes(R)
    return bool(L.nfa_traverse_sample_fused(ctypes.byref(a)))


@pytest.mark.parametrize("name", FUSABLE)
def test_fused_form_is_what_serves_the_case(name, k2, force_options):
    c = _case(name, k2)
    assert _is_fused(c)
    force_options(fused_sample=0)
    assert not _is_fused(c)


@pytest.mark.parametrize("name", NOT_FUSABLE)
def test_calls_outside_the_window_take_the_three_launches(name, k2):
    c = _case(name, k2)
    assert not _is_fused(c)
    _check(name, k2, _sample(c), c["rays_o"].shape[0])
    _check(name, k2, _sample(c), c["rays_o"].shape[0])


def test_fused_sampling_equals_the_oracle(force_options):
    """the bench's kind of scene (a grid whose image fits LDS), 6 564 rays with a stratified start: sample lists bit for bit as the C
    restatement of grid.cu:68-282 produces them (oracle/nerfacc_oracle.c)"""
    import torch

    import oracle
    from gpu_utils import n, sampling_is_fused, sparse_like, t
    from nerfacc_amd import cuda as C

    force_options(fused_sample=2)
    o, d, aabb, occ = sparse_like(3, 6564)
    assert sampling_is_fused(o, d, occ, aabb, 5e-3)
    rng = np.random.default_rng(1)
    jit = rng.random(6564, dtype=np.float32)
    args = (t(o), t(d), t(occ), t(aabb), None, None, 5e-3, 0.0)
    kw = dict(near_plane=0.0, far_plane=1e10, jitter=t(jit), jitter_scale=5e-3)
    C.sample_occgrid(*args, **kw)
    out = C.sample_occgrid(*args, **kw)             # (the second call has a guess: everything in the one launch)
    torch.cuda.synchronize()
    r_ri, r_ts, r_te, _ = oracle.sampling(o, d, occ, aabb, near_plane=0.0, far_plane=1e10, render_step_size=5e-3, jitter=jit)
    assert np.array_equal(n(out[0]), r_ri) and np.array_equal(n(out[1]), r_ts) and np.array_equal(n(out[2]), r_te)


@pytest.mark.parametrize("name", FUSABLE)
def test_fused_sampling_reproduces_reference_k2(name, k2, force_options):
    """every guess regime, back to back, against the reference's fixture"""
    c = _case(name, k2)
    R = c["rays_o"].shape[0]
    force_options(fused_sample=2)                # (m1_sphere has 84 samples per ray: beyond what the automatic choice fuses)
    # 1. a call on a handful of rays leaves a samples-per-ray guess that is far too SMALL for the next one only if those rays are
    #    short — take the rays with the fewest samples: the next call's waves mostly end beyond the guess
    ref_cnts = k2[f"{name}/cnts/sm_chunk_cnts"] if f"{name}/cnts/sm_chunk_cnts" in k2 else k2[f"{name}/full/sm_chunk_cnts"]
    few = np.argsort(ref_cnts, kind="stable")[: 3072 + 64]
    few = np.sort(few[ref_cnts[few] > 0][:3072]) if (ref_cnts[few] > 0).sum() >= 3072 else np.sort(few[:3072])
    _sample(c, few)
    _check(name, k2, _sample(c), R)              # guess too small (or, if the short rays are empty, no guess at all)
    _check(name, k2, _sample(c), R)              # good guess: count, offsets and emit in the one launch
    _check(name, k2, _sample(c), R)
    # 2. no speculation: the launch counts and scans, the emit pass follows the read-back
    force_options(speculative_emit=0)
    _check(name, k2, _sample(c), R)
    force_options(speculative_emit=1)
    _check(name, k2, _sample(c), R)
    # 3. the automatic choice (long rays: three launches once there is a guess) and the three-kernel form give the same tensors
    force_options(fused_sample=1)
    _check(name, k2, _sample(c), R)
    _check(name, k2, _sample(c), R)
    force_options(fused_sample=0)
    _check(name, k2, _sample(c), R)


def test_fused_sampling_equals_unfused_on_ragged_batches(force_options):
    """ray counts that do not fill the last workgroup / wave, rays that miss the grid, a stratified near plane: fused == unfused,
    tensor for tensor, 40 times in a row (a stale sync block or a lost hand-off shows up as a difference or a hang)"""
    import torch

    from gpu_utils import sampling_is_fused, sparse_like, t
    from nerfacc_amd import cuda as C

    import nerfacc_amd

    rng = np.random.default_rng(7)
    o, d, aabb, occ = sparse_like(3, 8192)
    assert sampling_is_fused(o, d, occ, aabb, 5e-3) and sampling_is_fused(o[:3072], d[:3072], occ, aabb, 5e-3)
    assert not sampling_is_fused(o[:3071], d[:3071], occ, aabb, 5e-3)            # (below the window: three launches)
    # a third of the rays point away from the box
    flip = rng.random(8192) < 0.33
    d[flip] = -d[flip]
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    for R in (3072, 3073, 3105, 4095, 5000, 6564, 8191, 8192):
        jit = torch.rand(R, device=O.device)
        tmin = torch.rand(R, device=O.device) * 2.0
        outs = {}
        for fused in (1, 0):
            with nerfacc_amd.options(fused_sample=2 if fused else 0):
                for _ in range(3 if fused else 1):
                    outs[fused] = C.sample_occgrid(O[:R].contiguous(), D[:R].contiguous(), B, A, None, None, 5e-3, 0.0, near_plane=0.0,
                                                   far_plane=1e10, t_min=tmin, jitter=jit, jitter_scale=5e-3)
        torch.cuda.synchronize()
        for x, y in zip(outs[1], outs[0]):
            assert x.shape == y.shape and torch.equal(x, y), f"R = {R}"
    force_options(fused_sample=2)
    for _ in range(40):
        out = C.sample_occgrid(O[:6564].contiguous(), D[:6564].contiguous(), B, A, None, None, 5e-3, 0.0, near_plane=0.0, far_plane=1e10)
        with nerfacc_amd.options(fused_sample=0):
            ref = C.sample_occgrid(O[:6564].contiguous(), D[:6564].contiguous(), B, A, None, None, 5e-3, 0.0, near_plane=0.0, far_plane=1e10)
        for x, y in zip(out, ref):
            assert torch.equal(x, y)


def test_fused_sampling_on_two_streams(force_options):
    """two host threads, each on its own stream with its own sync block, sampling concurrently: every result equals the
    single-stream one (the look-back's waits are bounded: a launch that cannot finish it hands over to the separate kernels)"""
    import threading

    import torch

    from gpu_utils import sampling_is_fused, sparse_like, t
    from nerfacc_amd import cuda as C

    force_options(fused_sample=2)
    o, d, aabb, occ = sparse_like(5, 8192)
    assert sampling_is_fused(o, d, occ, aabb, 5e-3)
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    ref = C.sample_occgrid(O, D, B, A, None, None, 5e-3, 0.0, near_plane=0.0, far_plane=1e10)
    torch.cuda.synchronize()
    errors = []

    def work():
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(s):
                for _ in range(25):
                    out = C.sample_occgrid(O, D, B, A, None, None, 5e-3, 0.0, near_plane=0.0, far_plane=1e10)
                    for x, y in zip(out, ref):
                        if not torch.equal(x, y):
                            errors.append("mismatch")
            s.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work) for _ in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
