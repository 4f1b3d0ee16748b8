import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from gpu_utils import lego_like, t, n
from nerfacc_amd.grid import traverse_grids
o, d, aabb, occ = lego_like(0, 4096)
r_iv, r_sm, r_term = oracle.traverse_grids(o, d, occ, aabb, step_size=5e-3)
iv, sm, term = traverse_grids(t(o), t(d), t(occ), t(aabb), step_size=5e-3)
g = n(sm.packed_info)[:, 1]; w = r_sm["packed_info"][:, 1]
bad = np.nonzero(g != w)[0]
print("mismatching rays", len(bad), bad[:20])
gi = n(iv.packed_info)[:, 1]; wi = r_iv["packed_info"][:, 1]
for b in bad[:8]:
    print(b, "gpu sm", g[b], "ref sm", w[b], "gpu iv", gi[b], "ref iv", wi[b], "o", o[b], "d", d[b])
    s0 = r_sm["packed_info"][b, 0]
    # reference run structure
    e0 = r_iv["packed_info"][b, 0]; ne = wi[b]
    il = r_iv["is_left"][e0:e0+ne]; ir = r_iv["is_right"][e0:e0+ne]; v = r_iv["vals"][e0:e0+ne]
    starts = [float(v[i]) for i in range(ne) if il[i] and not ir[i]]
    ends = [float(v[i]) for i in range(ne) if ir[i] and not il[i]]
    print("   ref runs:", list(zip(starts, ends)))
