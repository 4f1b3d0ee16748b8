"""The fallback paths behind the single-launch forms (round 6).  Every look-back inside them waits a bounded time; a launch whose wait
runs out says so in its result and the caller goes on with the separate kernels (`nfa_traverse_sample`: totals[1] = -1 ->
nfa_traverse_offsets + emit; `nfa_visibility_compact_sync`: n_out[0] = -1 -> nfa_visibility_compact_resume;
`nfa_grid_occupied_cells`: the workgroup counts the cells before its own itself).  In normal operation that never happens, so option
`sync_spin_us = 0` makes a look-back give up at the first state that is not there yet: with hundreds of workgroups finishing at
different times some launches give up, and everything must come out the same — and the sync block must be clean for the next call."""
import numpy as np
import pytest
import torch

from gpu_utils import DEV, sampling_is_fused, sparse_like, t

pytestmark = pytest.mark.gpu


def test_sampling_gives_the_same_lists_when_look_backs_give_up(force_options):
    from nerfacc_amd import cuda as C

    o, d, aabb, occ = sparse_like(21, 8192)
    assert sampling_is_fused(o, d, occ, aabb, 5e-3)
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    force_options(fused_sample=0)
    want = [C.sample_occgrid(O[:R].contiguous(), D[:R].contiguous(), B, A, None, None, 5e-3, 0.0, near_plane=0.0, far_plane=1e10) for R in (3072, 6564, 8192)]
    force_options(fused_sample=2, sync_spin_us=0)
    gave_up = 0
    for rep in range(12):
        for R, w in zip((3072, 6564, 8192), want):
            got = C.sample_occgrid(O[:R].contiguous(), D[:R].contiguous(), B, A, None, None, 5e-3, 0.0, near_plane=0.0, far_plane=1e10)
            assert all(torch.equal(a, b) for a, b in zip(got, w)), (rep, R)
    # and the usual bound again, on the same sync block
    force_options(sync_spin_us=2000)
    for R, w in zip((3072, 6564, 8192), want):
        got = C.sample_occgrid(O[:R].contiguous(), D[:R].contiguous(), B, A, None, None, 5e-3, 0.0, near_plane=0.0, far_plane=1e10)
        assert all(torch.equal(a, b) for a, b in zip(got, w))


def test_the_c_abi_reports_a_look_back_that_gave_up(force_options):
    """through the ctypes face: with `sync_spin_us = 0` at least one of many launches must come back with totals[1] = -1 (206 workgroups
    never finish counting at the same instant), and nfa_traverse_offsets then produces the totals of the three-launch form"""
    import ctypes

    from nerfacc_amd.cuda import _backend

    L = _backend.load_library()
    o, d, aabb, occ = sparse_like(22, 6564)
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    a = _backend._traverse_args(O, D, None, B, A, None, None, None, None, None, 5e-3, 0.0, 0)
    R = O.shape[0]
    packed = torch.zeros(2, R, dtype=torch.int64, device=DEV)
    a.sm_starts, a.sm_cnts = packed[0].data_ptr(), packed[1].data_ptr()
    totals = torch.zeros(4, dtype=torch.int64, device=DEV)
    a.totals = totals.data_ptr()
    a.workspace_bytes = L.nfa_traverse_workspace_bytes_for(ctypes.byref(a))
    ws = torch.empty(a.workspace_bytes, dtype=torch.uint8, device=DEV)
    sync = torch.zeros(16384, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.nfa_traverse_sample_fused(ctypes.byref(a)) == 1
    force_options(sync_spin_us=2000)
    assert L.nfa_traverse_sample(ctypes.byref(a), ws.data_ptr(), 0, 0, sync.data_ptr(), None, stream) == 0
    torch.cuda.synchronize()
    ref_tot, ref_packed = totals.clone(), packed.clone()
    assert int(ref_tot[1]) > 0 and int(sync.to(torch.int32).abs().sum()) == 0
    force_options(sync_spin_us=0)
    seen = 0
    for _ in range(40):
        totals.zero_(); packed.zero_()
        assert L.nfa_traverse_sample(ctypes.byref(a), ws.data_ptr(), 0, 0, sync.data_ptr(), None, stream) == 0
        torch.cuda.synchronize()
        assert int(sync.to(torch.int32).abs().sum()) == 0                                 # the block is left zero either way
        if int(totals[1]) < 0:
            seen += 1
            assert torch.equal(packed[1], ref_packed[1])                                   # the counts are complete
            assert L.nfa_traverse_offsets(ctypes.byref(a), ws.data_ptr(), stream) == 0
            torch.cuda.synchronize()
        assert torch.equal(totals[:3], ref_tot[:3]) and torch.equal(packed, ref_packed)
    assert seen > 0, "no launch of 40 gave up although sync_spin_us = 0: the fallback was not exercised"


def test_filter_resume_and_occupied_cells_when_look_backs_give_up(force_options):
    from nerfacc_amd import cuda as C

    g = torch.Generator().manual_seed(5)
    cnts = torch.randint(0, 90, (6500,), generator=g)
    ri = torch.repeat_interleave(torch.arange(6500), cnts).to(DEV)
    N = ri.shape[0]
    ts = (torch.rand(N, generator=g) * 4).to(DEV)
    te = ts + 5e-3
    sig = (torch.rand(N, generator=g) * 40).to(DEV)
    want = C.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.0, True)
    binaries = (torch.rand(2, 128, 128, 128, generator=g) < 0.05).to(DEV)
    cells = [torch.nonzero(binaries[l].flatten())[:, 0] for l in range(2)]
    force_options(fused_vis=1, sync_spin_us=0)
    for _ in range(15):
        got = C.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.0, True)
        assert all(torch.equal(a, b) for a, b in zip(got, want))
        for l in range(2):
            assert torch.equal(C.grid_occupied_cells(binaries, l), cells[l])
