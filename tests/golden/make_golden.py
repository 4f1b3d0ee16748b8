"""Generate the golden fixtures under tests/golden/ from the Python reference.

Run ONLY in the build container (needs /root/reference; the GPU box has no
such path and never runs this):

    python tests/golden/make_golden.py

What it pins: the pure-torch twins and batched paths of
the reference that execute on CPU — `_ray_aabb_intersect`, `_query`,
batched `render_*` / `rendering` / `accumulate_along_rays` (+ autograd grads),
batched scans (+ grads), `_sample_from_weighted`, torch.searchsorted,
`OccGridEstimator.mark_invisible_cells` / `_update`.  The reference's flattened
CUDA kernels cannot run here (no nvcc, no NVIDIA GPU), so nothing below comes
from them.  `nerfacc.volrend.is_cub_available` is shimmed to True because the
reference consults the CUDA module even for batched CPU inputs
(volrend.py:206,266); that is the only modification.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    import nerfacc  # noqa: F401  (the reference)
    import nerfacc.volrend as V
    import nerfacc.scan as S
    from nerfacc.grid import _enlarge_aabb, _query, _ray_aabb_intersect
    from nerfacc.pdf import _sample_from_weighted
    from nerfacc.estimators.occ_grid import OccGridEstimator

    assert nerfacc.__file__.startswith(REF), nerfacc.__file__
    V.is_cub_available = lambda: True
    g = {}

    # ---- K1 twin (tests/test_grid.py:7-35 generator, smaller) ------------
    torch.manual_seed(42)
    R, G = 256, 24
    rays_o = torch.rand((R, 3))
    rays_d = torch.randn((R, 3))
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    aabb_min = torch.rand((G, 3))
    aabbs = torch.cat([aabb_min, aabb_min + torch.rand_like(aabb_min)], -1)
    tmin, tmax, hits = _ray_aabb_intersect(rays_o, rays_d, aabbs)
    g.update(k1_rays_o=rays_o, k1_rays_d=rays_d, k1_aabbs=aabbs, k1_tmin=tmin, k1_tmax=tmax, k1_hits=hits)

    # ---- _query (grid.py:201-237) ----------------------------------------
    torch.manual_seed(42)
    base = torch.tensor([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0])
    binaries = torch.rand((4, 32, 32, 32)) > 0.5
    pts = (torch.rand((2000, 3)) * 2 - 1) * 9.0
    occ, sel = _query(pts, binaries, base)
    g.update(q_pts=pts, q_binaries=np.packbits(binaries.numpy().ravel()), q_occ=occ, q_sel=sel)
    g["enlarge"] = torch.stack([_enlarge_aabb(base, 2**i) for i in range(4)])

    # ---- batched volrend + grads (volrend.py:219-376, 497-561, 15-164) ----
    torch.manual_seed(7)
    R, S_ = 64, 48
    t0 = torch.sort(torch.rand((R, S_ + 1)) * 4.0, -1)[0]
    ts, te = t0[:, :-1].contiguous(), t0[:, 1:].contiguous()
    sig = (torch.rand((R, S_)) * 20.0 * (torch.rand((R, S_)) > 0.3)).requires_grad_(True)
    rgb = torch.rand((R, S_, 3), requires_grad=True)
    w, T, a = V.render_weight_from_density(ts, te, sig)
    gw, gT, ga = torch.randn_like(w), torch.randn_like(w), torch.randn_like(w)
    (w * gw + T * gT + a * ga).sum().backward()
    g.update(v_ts=ts, v_te=te, v_sig=sig.detach(), v_w=w.detach(), v_T=T.detach(), v_a=a.detach(),
             v_gw=gw, v_gT=gT, v_ga=ga, v_gsig=sig.grad.clone())
    sig.grad = None
    bk = torch.tensor([0.9, 0.5, 0.1])
    col, opa, dep, _ = V.rendering(ts, te, rgb_sigma_fn=lambda a_, b_, c_: (rgb, sig), render_bkgd=bk)
    gc, go, gd = torch.randn_like(col), torch.randn_like(opa), torch.randn_like(dep)
    (col * gc).sum().backward(retain_graph=True)
    g.update(r_rgb=rgb.detach(), r_bk=bk, r_col=col.detach(), r_opa=opa.detach(), r_dep=dep.detach(),
             r_gc=gc, r_gsig_c=sig.grad.clone(), r_grgb_c=rgb.grad.clone())
    sig.grad = None
    rgb.grad = None
    (col * gc).sum().add((opa * go).sum()).add((dep * gd).sum()).backward()
    g.update(r_go=go, r_gd=gd, r_gsig_all=sig.grad.clone(), r_grgb_all=rgb.grad.clone())
    # alpha path (volrend.py:167-216, 281-323, 379-432)
    al = torch.rand((R, S_)) * (torch.rand((R, S_)) > 0.2)
    wa, Ta = V.render_weight_from_alpha(al)
    vis = V.render_visibility_from_alpha(al, early_stop_eps=0.05, alpha_thre=0.35)
    visd = V.render_visibility_from_density(ts, te, sig.detach(), early_stop_eps=1e-2, alpha_thre=0.01)
    g.update(a_al=al, a_w=wa, a_T=Ta, a_vis=vis, a_visd=visd)
    # flattened accumulate on CPU (index_add_ works on CPU tensors)
    torch.manual_seed(3)
    ridx = torch.sort(torch.randint(0, 40, (500,)))[0]
    wv, vv = torch.rand(500), torch.rand(500, 3)
    g.update(acc_idx=ridx, acc_w=wv, acc_v=vv,
             acc_out3=V.accumulate_along_rays(wv, vv, ridx, 40),
             acc_out1=V.accumulate_along_rays(wv, None, ridx, 40))

    # ---- batched scans + grads (scan.py batched branches) -----------------
    torch.manual_seed(42)
    data = torch.rand((5, 1000))
    g["s_in"] = data
    for name, fn in (("isum", S.inclusive_sum), ("esum", S.exclusive_sum),
                     ("iprod", S.inclusive_prod), ("eprod", S.exclusive_prod)):
        x = (data if "sum" in name else data * 0.2 + 0.9).clone().requires_grad_(True)
        y = fn(x)
        y.sum().backward()
        g[f"s_{name}"] = y.detach()
        g[f"s_{name}_grad"] = x.grad.clone()

    # ---- pdf (pdf.py:134-219, tests/test_pdf.py:65-94) --------------------
    torch.manual_seed(42)
    vals = torch.sort(torch.rand((5, 101)), -1)[0]
    cdfs = torch.sort(torch.rand_like(vals), -1)[0]
    edges, mids = [], []
    for i in range(5):
        e, m = _sample_from_weighted(vals[i:i + 1], cdfs[i:i + 1, 1:] - cdfs[i:i + 1, :-1], 100, False,
                                     vals[i].min(), vals[i].max())
        edges.append(e)
        mids.append(m)
    q = torch.sort(torch.rand((5, 77)), -1)[0]
    ss = torch.clamp(torch.searchsorted(vals, q, right=True), 0, vals.shape[-1] - 1)
    g.update(p_vals=vals, p_cdfs=cdfs, p_edges=torch.cat(edges), p_mids=torch.cat(mids), p_q=q, p_ss=ss)

    # ---- estimator (occ_grid.py:262-332 known answer; :366-404 _update) ----
    est = OccGridEstimator(roi_aabb=base, resolution=32, levels=4)
    K = torch.tensor([[[100.0, 0, 50.0], [0, 100.0, 50.0], [0, 0, 1]]])
    pose = torch.tensor([[[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5]]])
    est.mark_invisible_cells(K, pose, 100, 100)
    g["mic_neg"] = torch.tensor(int((est.occs == -1).sum()))
    g["mic_zero"] = torch.tensor(int((est.occs == 0).sum()))
    g["mic_occs_bits"] = np.packbits((est.occs == -1).numpy())

    # _update evolution with the reference's RNG call order (occ_grid.py:345-404)
    from nerfacc.estimators.prop_net import _transform_stot, get_proposal_requires_grad_fn

    torch.manual_seed(123)
    est2 = OccGridEstimator(roi_aabb=base, resolution=16, levels=2)
    occ_fn = lambda x: torch.exp(-2.0 * (x**2).sum(-1, keepdim=True)) * 0.05
    for step in (0, 16, 256, 272):
        est2._update(step=step, occ_eval_fn=occ_fn, occ_thre=0.01)
    g["upd_occs"] = est2.occs.clone()
    g["upd_bin"] = np.packbits(est2.binaries.numpy().ravel())
    fn = get_proposal_requires_grad_fn()
    g["prop_sched"] = np.array([fn(i) for i in range(3000)])
    sv = torch.linspace(0, 1, 17)[None]
    g["stot_uni"] = _transform_stot("uniform", sv, 0.2, 1000.0)
    g["stot_lin"] = _transform_stot("lindisp", sv, 0.2, 1000.0)

    out = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in g.items()}
    np.savez_compressed(os.path.join(OUT, "reference_cpu.npz"), **out)
    print("wrote", os.path.join(OUT, "reference_cpu.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
