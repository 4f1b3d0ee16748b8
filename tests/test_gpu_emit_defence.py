"""The emit pass does not store through offsets / counts that cannot be (profiles/r06_oversubscription.md: eight processes on one
GPU were seen to lose part of a count launch's stores; what the emit kernel then read was what the memory held before).  Through
the C ABI: count + offsets of a call, then some rays' offsets / counts overwritten the way the captured event looked (every eighth
group of sixteen rays), then the emit launch into outputs filled with a sentinel — the clean rays come out as in the clean call,
nothing else is touched, and nothing faults."""
import ctypes

import pytest
import torch

import nerfacc_amd
from gpu_utils import sparse_like, t

pytestmark = pytest.mark.gpu
SENTINEL = -7


@pytest.mark.parametrize("what", ["starts", "counts", "total"])
@pytest.mark.parametrize("R", [906, 5000])
def test_emit_skips_what_cannot_be(R, what):
    from nerfacc_amd.cuda import _backend as B
    L = B.load_library()
    o, d, aabb, occ = sparse_like(3, R)
    O, D, OCC, AABB = t(o), t(d), t(occ), t(aabb)
    dev = O.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    with nerfacc_amd.options(fused_sample=0):
        a = B._traverse_args(O, D, None, OCC, AABB, None, None, None, None, None, 5e-3, 0.0, -1, 0.0, 1e10)
        packed = torch.empty((2, R), dtype=torch.int64, device=dev)
        totals = B._host_ints(dev)
        a.workspace_bytes = L.nfa_traverse_workspace_bytes_for(ctypes.byref(a))
        ws = torch.empty(max(a.workspace_bytes, 16), dtype=torch.uint8, device=dev)
        a.sm_starts, a.sm_cnts, a.totals = packed[0].data_ptr(), packed[1].data_ptr(), totals.data_ptr()
        B._check(L.nfa_traverse_count(ctypes.byref(a), ws.data_ptr(), stream))
        B._check(L.nfa_traverse_offsets(ctypes.byref(a), ws.data_ptr(), stream))
        _, n, n_ovf, _ = B._read_ints(totals, dev)
        assert n > 1000 and n_ovf == 0
        starts, cnts = packed[0].clone(), packed[1].clone()
        cap = n + 777

        def emit():
            ri = torch.full((cap,), SENTINEL, dtype=torch.int64, device=dev)
            ts = torch.full((2, (cap + 3) & ~3), float(SENTINEL), device=dev)
            a.sm_ray_indices, a.t_starts, a.t_ends = ri.data_ptr(), ts[0].data_ptr(), ts[1].data_ptr()
            B._check(L.nfa_traverse_emit_speculative(ctypes.byref(a), ws.data_ptr(), cap, stream))
            torch.cuda.synchronize()
            return ri, ts[0, :cap].clone(), ts[1, :cap].clone()

        clean = emit()
        assert bool((clean[0][:n] >= 0).all()) and bool((clean[0][n:] == SENTINEL).all())
        rays = torch.arange(R, device=dev)
        hit = ((rays // 16) % 8 == 3) & (cnts > 0)             # every eighth group of sixteen rays, as in the captured event
        assert int(hit.sum()) > 10
        garbage = torch.randint(1 << 40, 1 << 62, (R,), device=dev)
        if what == "starts":
            packed[0][hit] = torch.where(rays[hit] % 2 == 0, garbage[hit], -garbage[hit])
        elif what == "counts":
            packed[1][hit] = garbage[hit]
        else:                                                  # the device copy of the totals (the last 32 bytes of the base workspace)
            off = L.nfa_traverse_workspace_bytes(R) - 32
            ws[off:off + 32] = torch.tensor([0, -(1 << 60), 0, 0], dtype=torch.int64, device=dev).view(torch.uint8)
        got = emit()
    if what == "total":
        assert all(bool((x == SENTINEL).all()) for x in got)   # a negative total: nothing is stored
        return
    # the samples of the clean rays are where they were; everything else still holds the sentinel
    ok = ~hit & (cnts > 0)
    c_ok, s_ok = cnts[ok], starts[ok]
    within = torch.arange(int(c_ok.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(c_ok, 0) - c_ok, c_ok)
    keep = torch.zeros(cap, dtype=torch.bool, device=dev)
    keep[torch.repeat_interleave(s_ok, c_ok) + within] = True
    for c, x in zip(clean, got):
        assert torch.equal(x[keep], c[keep])
        assert bool((x[~keep] == SENTINEL).all())
