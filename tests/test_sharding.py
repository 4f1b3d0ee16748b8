"""Multi-GPU plumbing (nerfacc_amd/sharding.py) on CPU: world_size 2, gloo backend.
The sampling / rendering kernels never communicate (rays are independent); what is tested is
the per-step exchange: one flat gradient all-reduce, the two-scalar count all-reduce, grid
broadcast and the synchronised-RNG grid update."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerfacc_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert sharding.world() == (rank, world)
        # --- a tiny "radiance field"; every rank renders its shard of the SAME global ray batch
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
        gen = torch.Generator().manual_seed(1)
        rays = torch.randn(64, 3, generator=gen)
        target = torch.rand(64, 3, generator=gen)
        b, e = sharding.shard_bounds(64, rank, world)
        loss = torch.nn.functional.smooth_l1_loss(model(rays[b:e]), target[b:e])
        loss.backward()
        sharding.allreduce_gradients(model.parameters())
        flat = torch.cat([p.grad.flatten() for p in model.parameters()])
        # --- a single-tensor field (bench.py's voxel grid) takes the in-place path; a rank without
        # samples (no grad) contributes zeros
        single = torch.nn.Parameter(torch.full((4, 5), float(rank + 1)))
        if rank == 0:
            (single * 3.0).sum().backward()
        sharding.allreduce_gradients([single])
        # --- counts
        s, r = sharding.allreduce_counts(1000 + rank, e - b, "cpu")
        pend = sharding.allreduce_counts_begin(10 + rank, 3, "cpu")      # deferred form: begin now, read next step
        assert sharding.allreduce_counts_end(pend) == (21, 6)
        # --- grid agreement: independent RNG diverges, synchronized_rng / broadcast agree
        from nerfacc_amd import OccGridEstimator

        torch.manual_seed(100 + rank)                       # ranks draw different rays ...
        est = OccGridEstimator([-1.0, -1, -1, 1, 1, 1], resolution=8, levels=1)
        occ_fn = lambda x: (x.norm(dim=-1, keepdim=True) < 0.8).float() * torch.rand(x.shape[0], 1)
        with sharding.synchronized_rng(7):                  # ... but update the grid in lock-step
            est._update(step=0, occ_eval_fn=occ_fn)
        own_stream = torch.rand(1).item()                   # the rank's own stream is restored afterwards
        est_b = OccGridEstimator([-1.0, -1, -1, 1, 1, 1], resolution=8, levels=1)
        est_b._update(step=0, occ_eval_fn=occ_fn)           # diverges between ranks
        sharding.broadcast_grid(est_b, src=0)
        # --- ExchangeAdam: chunked async all-reduce with the Adam update of a chunk as soon as it has arrived
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.Tanh(), torch.nn.Linear(32, 3))
        opt = sharding.ExchangeAdam(net.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6, n_chunks=3)
        for it in range(4):
            if it == 1:
                net.zero_grad(set_to_none=True)              # detaches the flat-buffer views: the hooks / step() re-bind them
            else:
                opt.zero_grad()
            if not (rank == 1 and it == 2):                  # a rank without samples still joins the exchange (zero grads)
                torch.nn.functional.smooth_l1_loss(net(rays[b:e]), target[b:e]).mul(1024.0).backward()
            opt.step()
        chunked = torch.cat([p.detach().flatten() for p in net.parameters()])
        torch.save(dict(grad=flat, single=single.grad.clone(), counts=(s, r), chunked=chunked, occs=est.occs, occs_b=est_b.occs, bin_b=est_b.binaries, own=own_stream),
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_exchange(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{i}.pt") for i in range(world))
    # all-reduced (averaged) shard gradients == the gradient of the whole batch on one process
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    gen = torch.Generator().manual_seed(1)
    rays = torch.randn(64, 3, generator=gen)
    target = torch.rand(64, 3, generator=gen)
    torch.nn.functional.smooth_l1_loss(model(rays), target).backward()
    ref = torch.cat([p.grad.flatten() for p in model.parameters()])
    assert torch.allclose(r0["grad"], ref, atol=1e-6) and torch.equal(r0["grad"], r1["grad"])
    assert torch.equal(r0["single"], torch.full((4, 5), 1.5)) and torch.equal(r1["single"], r0["single"])
    assert r0["counts"] == r1["counts"] == (2001, 64)
    assert torch.equal(r0["occs"], r1["occs"]) and (r0["occs"] > 0).any()
    assert torch.equal(r0["occs_b"], r1["occs_b"]) and torch.equal(r0["bin_b"], r1["bin_b"])
    assert r0["own"] != r1["own"]
    # ExchangeAdam on 2 ranks == torch.optim.Adam on the averaged shard gradients, and the replicas stay identical
    assert torch.equal(r0["chunked"], r1["chunked"])
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.Tanh(), torch.nn.Linear(32, 3))
    ref_opt = torch.optim.Adam(net.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6)
    for it in range(4):
        ref_opt.zero_grad()
        grads = None
        for rk in range(world):
            b, e = sharding.shard_bounds(64, rk, world)
            for p in net.parameters():
                p.grad = None
            if not (rk == 1 and it == 2):
                torch.nn.functional.smooth_l1_loss(net(rays[b:e]), target[b:e]).mul(1024.0).backward()
            g = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in net.parameters()]
            grads = g if grads is None else [a + c for a, c in zip(grads, g)]
        for p, g in zip(net.parameters(), grads):
            p.grad = g / world
        ref_opt.step()
    want = torch.cat([p.detach().flatten() for p in net.parameters()])
    assert torch.allclose(r0["chunked"], want, atol=1e-6, rtol=1e-5)


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 64, 65536, 100003):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_deferred_counts_single_process():
    pend = sharding.allreduce_counts_begin(5, 7, "cpu")
    assert sharding.allreduce_counts_end(pend) == (5, 7)


def test_single_process_is_a_noop():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    sharding.allreduce_gradients([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0))
    assert sharding.allreduce_counts(5, 2, "cpu") == (5, 2)
    assert sharding.world() == (0, 1)


def test_exchange_adam_single_process_matches_torch_adam():
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(300, 3)), torch.nn.Parameter(torch.randn(77))]
    b = [torch.nn.Parameter(x.detach().clone()) for x in a]
    oa = torch.optim.Adam(a, lr=1e-2, eps=1e-15, weight_decay=1e-6)
    ob = sharding.ExchangeAdam(b, lr=1e-2, eps=1e-15, weight_decay=1e-6, n_chunks=5)
    for it in range(6):
        for ps, o in ((a, oa), (b, ob)):
            o.zero_grad()
            (sum(((p * 1.3 - 0.2) ** 2).sum() for p in ps) * (it + 1)).backward()
            o.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-7, rtol=1e-6)


def test_exchange_adam_survives_detached_grads_and_checkpoints():
    """model.zero_grad(set_to_none=True) / p.grad = None detach the flat-buffer views; step() must notice,
    pull the freshly allocated gradients in and re-bind.  state_dict / load_state_dict resume bit-identically."""
    torch.manual_seed(1)
    a = [torch.nn.Parameter(torch.randn(300, 3)), torch.nn.Parameter(torch.randn(77)), torch.nn.Parameter(torch.randn(5, 5))]
    b = [torch.nn.Parameter(x.detach().clone()) for x in a]
    oa = torch.optim.Adam(a, lr=1e-2, eps=1e-15, weight_decay=1e-6)
    ob = sharding.ExchangeAdam(b, lr=1e-2, eps=1e-15, weight_decay=1e-6, n_chunks=4)

    def loss(ps, it):
        # the last parameter gets no gradient at it == 3 (set_to_none leaves it None: counts as zero)
        return sum(((p * 1.3 - 0.2) ** 2).sum() for p in (ps[:2] if it == 3 else ps)) * (it + 1)

    snap = None
    for it in range(8):
        oa.zero_grad(set_to_none=False)
        if it % 2:
            for p in b:                      # what torch's Module.zero_grad() does by default
                p.grad = None
        else:
            ob.zero_grad()
        if it == 5:
            b[1].grad = torch.zeros(77)      # a replaced gradient tensor
        for ps in (a, b):
            loss(ps, it).backward()
        oa.step()
        ob.step()
        assert all(ob._bound(i) for i in range(3))
        if it == 3:
            snap = ob.state_dict()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-7, rtol=1e-6)
    # resume from the snapshot taken after step 4 (it == 3) in a fresh optimizer over fresh parameters
    c = [torch.nn.Parameter(torch.zeros_like(x)) for x in a]
    oc = sharding.ExchangeAdam(c, lr=1.0, n_chunks=2)
    oc.load_state_dict(snap)
    assert oc.t == 4 and oc.lr == 1e-2 and float(oc.step_tensor) == 4.0
    for it in range(4, 8):
        oc.zero_grad()
        loss(c, it).backward()
        oc.step()
    for x, y in zip(b, c):
        assert torch.allclose(x, y, atol=1e-7, rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the step exchange with a MULTI-TENSOR field on up to 8 ranks — a dense feature grid and a
# small MLP, seven parameter tensors from 4 to 13 824 elements, one rank without samples on some steps — in both exchange
# modes (all-reduce / reduce-scatter + sharded Adam + all-gather), launched from step() and from the gradient hooks.
# Every rank must issue the SAME sequence of collectives, replicas must stay bit-identical, and the result must be
# torch.optim.Adam on the averaged shard gradients.
# ---------------------------------------------------------------------------------------------------------------------
class _GridMlpField(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.grid = torch.nn.Parameter(0.1 * torch.randn(1, 8, 12, 12, 12, generator=g))
        self.l1, self.l2, self.l3 = torch.nn.Linear(8, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 4)
        for lin in (self.l1, self.l2, self.l3):
            with torch.no_grad():
                lin.weight.copy_(0.3 * torch.randn(lin.weight.shape, generator=g))
                lin.bias.copy_(0.1 * torch.randn(lin.bias.shape, generator=g))

    def forward(self, x):
        f = torch.nn.functional.grid_sample(self.grid, x.view(1, -1, 1, 1, 3), align_corners=True).view(8, -1).t()
        return self.l3(torch.relu(self.l2(torch.relu(self.l1(f)))))


def _field_batch(step):
    g = torch.Generator().manual_seed(1000 + step)
    return torch.rand(96, 3, generator=g) * 2 - 1, torch.rand(96, 4, generator=g)


def _rank_is_idle(rank, step, world):
    return world > 1 and rank == world - 1 and step % 3 == 1          # the last rank draws no samples on every third step


def _field_worker(rank, world, port, out_dir, mode, overlap, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        field = _GridMlpField()
        opt = sharding.ExchangeAdam(field.parameters(), lr=5e-3, eps=1e-15, weight_decay=1e-6, n_chunks=4, mode=mode,
                                    overlap_backward=overlap, record_collectives=True)
        assert len(opt.params) == 7 and len(opt.bounds) == 4 and all((b - a) % world == 0 for a, b in opt.bounds)
        for step in range(steps):
            x, y = _field_batch(step)
            b, e = sharding.shard_bounds(x.shape[0], rank, world)
            opt.zero_grad()
            # the count exchange of the step (train_ngp_nerf_occ.py:187-194) goes BEFORE backward, as bench.py does
            pend = sharding.allreduce_counts_begin(e - b, e - b, "cpu")
            if not _rank_is_idle(rank, step, world):
                torch.nn.functional.smooth_l1_loss(field(x[b:e]), y[b:e]).mul(64.0).backward()
            opt.step()
            n_s, n_r = sharding.allreduce_counts_end(pend)
            assert n_r == x.shape[0]
        snap = opt.state_dict()                 # (a collective in rs_ag mode: the moments are gathered)
        torch.save(dict(params=torch.cat([p.detach().flatten() for p in field.parameters()]), log=opt.collective_log,
                        m=snap["m"], v=snap["v"]), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _field_reference(world, steps):
    field = _GridMlpField()
    opt = torch.optim.Adam(field.parameters(), lr=5e-3, eps=1e-15, weight_decay=1e-6)
    for step in range(steps):
        x, y = _field_batch(step)
        total = [torch.zeros_like(p) for p in field.parameters()]
        for rk in range(world):
            if _rank_is_idle(rk, step, world):
                continue
            b, e = sharding.shard_bounds(x.shape[0], rk, world)
            field.zero_grad(set_to_none=True)
            torch.nn.functional.smooth_l1_loss(field(x[b:e]), y[b:e]).mul(64.0).backward()
            total = [t + (p.grad if p.grad is not None else 0) for t, p in zip(total, field.parameters())]
        for p, t in zip(field.parameters(), total):
            p.grad = t / world
        opt.step()
    st = opt.state_dict()["state"]
    return (torch.cat([p.detach().flatten() for p in field.parameters()]),
            torch.cat([st[i]["exp_avg"].flatten() for i in range(7)]), torch.cat([st[i]["exp_avg_sq"].flatten() for i in range(7)]))


@pytest.mark.parametrize("world,mode,overlap", [(8, "allreduce", True), (8, "rs_ag", True), (8, "rs_ag", False), (2, "rs_ag", True),
                                                (2, "allreduce", False), (3, "rs_ag", True)])
def test_multi_tensor_field_exchange(tmp_path, world, mode, overlap):
    steps = 20
    mp.spawn(_field_worker, args=(world, _free_port(), str(tmp_path), mode, overlap, steps), nprocs=world, join=True)
    ranks = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    want, want_m, want_v = _field_reference(world, steps)
    ops = {"allreduce": ["all_reduce"] * 4, "rs_ag": ["reduce_scatter"] * 4 + ["all_gather"] * 4}[mode]
    assert [op for op, _, _ in ranks[0]["log"][:len(ops)]] == ops
    assert [k for _, k, _ in ranks[0]["log"][:4]] == [3, 2, 1, 0]                     # the fixed launch order, last chunk first
    for r in ranks[1:]:
        assert r["log"] == ranks[0]["log"], "ranks issued different sequences of collectives"
        assert torch.equal(r["params"], ranks[0]["params"]), "replicas diverged"
        assert torch.equal(r["m"], ranks[0]["m"]) and torch.equal(r["v"], ranks[0]["v"])
    assert len(ranks[0]["log"]) == steps * len(ops)
    assert torch.allclose(ranks[0]["params"], want, atol=2e-6, rtol=1e-5)
    assert torch.allclose(ranks[0]["m"], want_m, atol=1e-6, rtol=1e-4) and torch.allclose(ranks[0]["v"], want_v, atol=1e-8, rtol=1e-4)


def _rules_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        field = _GridMlpField()
        opt = sharding.ExchangeAdam(field.parameters(), lr=5e-3, n_chunks=4, overlap_backward=True, record_collectives=True)
        x, y = _field_batch(0)
        b, e = sharding.shard_bounds(96, rank, world)
        seen = []
        torch.nn.functional.smooth_l1_loss(field(x[b:e]), y[b:e]).backward()
        assert len(opt.collective_log) == 4                  # every chunk went out from inside backward
        for name, call in (("allreduce_counts", lambda: sharding.allreduce_counts(1, 1, "cpu")),
                           ("allreduce_counts_begin", lambda: sharding.allreduce_counts_begin(1, 1, "cpu")),
                           ("allreduce_gradients", lambda: sharding.allreduce_gradients(field.parameters())),
                           ("broadcast_grid", lambda: sharding.broadcast_grid(None))):
            try:
                call()
                seen.append(name + ": no error")
            except RuntimeError as err:
                seen.append(name if "between backward() and ExchangeAdam.step()" in str(err) else f"{name}: {err}")
        try:                                                   # a second backward before step(): gradient accumulation
            torch.nn.functional.smooth_l1_loss(field(x[b:e]), y[b:e]).backward()
            seen.append("second backward: no error")
        except RuntimeError as err:
            seen.append("second backward" if "second backward()" in str(err) else f"second backward: {err}")
        opt.step()
        assert sharding.allreduce_counts(rank + 1, 1, "cpu") == (3, 2)      # after step() collectives are welcome again
        torch.save(seen, os.path.join(out_dir, f"rules{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_overlap_backward_rules_are_enforced(tmp_path):
    """with overlap_backward the chunks leave from inside backward(); this module's other collectives and
    a second backward() must raise until step() has run — and overlap_backward is opt-in"""
    assert sharding.ExchangeAdam([torch.nn.Parameter(torch.zeros(3))]).overlap_backward is False
    mp.spawn(_rules_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rk in range(2):
        assert torch.load(tmp_path / f"rules{rk}.pt") == ["allreduce_counts", "allreduce_counts_begin", "allreduce_gradients",
                                                          "broadcast_grid", "second backward"]
