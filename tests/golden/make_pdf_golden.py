"""Generate tests/golden/pdf_reference.npz: outputs of the REFERENCE's own importance_sampling / searchsorted kernels
(nerfacc/cuda/csrc/pdf.cu, compiled for the host by oracle/ref_shim/Makefile) driven through the reference's Python layer
(nerfacc/pdf.py with `nerfacc.cuda._backend._C` pointed at oracle/_ref).  stratified = False only (the stratified path draws
cuRAND Philox numbers inside the kernel).  Run ONLY in the build container:

    make -C oracle/ref_shim && python tests/golden/make_pdf_golden.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    sys.path[:0] = [REF, os.path.join(ROOT, "oracle", "_ref"), os.path.join(ROOT, "tests")]
    from pdf_cases import cases
    import nerfacc  # the reference
    import nerfacc.cuda._backend as backend
    from nerfacc.data_specs import RayIntervals
    from nerfacc.pdf import importance_sampling, searchsorted

    assert nerfacc.__file__.startswith(REF)
    backend._C = importlib.import_module("nerfacc_ref_fma")
    other = importlib.import_module("nerfacc_ref_off")
    fx = {}
    tt = torch.from_numpy
    for name, c in cases().items():
        pk = tt(c["packed_info"]) if "packed_info" in c else None
        res = {}
        for tag, mod in (("fma", backend._C), ("off", other)):
            backend._C = mod
            iv, sm = importance_sampling(RayIntervals(vals=tt(c["vals"]), packed_info=pk), tt(c["cdfs"]), c["n"], False)
            # searchsorted of the NEW edges in the old ones (what PropNetEstimator's loss does, prop_net.py:232-256)
            q = RayIntervals(vals=iv.vals, packed_info=iv.packed_info)
            l, r = searchsorted(RayIntervals(vals=tt(c["vals"]), packed_info=pk), q)
            res[tag] = [iv.vals.numpy(), sm.vals.numpy(), l.numpy(), r.numpy()]
        backend._C = importlib.import_module("nerfacc_ref_fma")
        n_diff = sum(int((a != b).sum()) for a, b in zip(res["fma"], res["off"]))
        print(f"{name:16s} edges {res['fma'][0].shape} mids {res['fma'][1].shape}  values differing between the FMA / no-FMA builds: {n_diff}, "
              f"max |diff| {max(float(np.abs(a - b).max()) for a, b in zip(res['fma'][:2], res['off'][:2])):.2e}, ids differing "
              f"{int((res['fma'][2] != res['off'][2]).sum() + (res['fma'][3] != res['off'][3]).sum())}")
        # the floats depend on the FMA choice in their last bits (u_floor + (sid + bias) * u_step, (u - u_lower) * scaling + t_lower),
        # so they are compared with a tolerance and a row subset suffices: every row of the small cases, every 32nd of the C3 ones
        rows = np.arange(res["fma"][0].shape[0]) if res["fma"][0].shape[0] <= 512 else np.arange(0, res["fma"][0].shape[0], 32)
        fx[f"{name}/rows"] = rows.astype(np.int32)
        for k, a in zip(("edges", "mids", "ids_left", "ids_right"), res["fma"]):
            fx[f"{name}/{k}"] = a[rows] if k in ("edges", "mids") else a[rows].astype(np.int32)
        fx[f"{name}/max_abs_diff_between_builds"] = np.array(max(float(np.abs(a - b).max()) for a, b in zip(res["fma"][:2], res["off"][:2])))
        fx[f"{name}/ids_differing_between_builds"] = np.array(int((res["fma"][2] != res["off"][2]).sum() + (res["fma"][3] != res["off"][3]).sum()))
    # the docstring examples (pdf.py:108-120, 40-56)
    backend._C = importlib.import_module("nerfacc_ref_fma")
    iv, sm = importance_sampling(RayIntervals(vals=torch.tensor([[0.0, 1.0], [0.0, 2.0]])), torch.tensor([[0.0, 0.5], [0.0, 0.5]]), 2, False)
    assert iv.vals.tolist() == [[0.0, 0.5, 1.0], [0.0, 1.0, 2.0]] and sm.vals.tolist() == [[0.25, 0.75], [0.5, 1.5]]
    ss = RayIntervals(vals=torch.tensor([0.0, 1.0, 0.0, 1.0, 2.0]), packed_info=torch.tensor([[0, 2], [2, 3]]))
    vv = RayIntervals(vals=torch.tensor([0.5, 1.5, 2.5]), packed_info=torch.tensor([[0, 1], [1, 2]]))
    l, r = searchsorted(ss, vv)
    fx["doc_searchsorted/ids_left"], fx["doc_searchsorted/ids_right"] = l.numpy(), r.numpy()
    print("docstring searchsorted:", l.tolist(), r.tolist())
    np.savez_compressed(os.path.join(HERE, "pdf_reference.npz"), **fx)
    print("wrote", os.path.getsize(os.path.join(HERE, "pdf_reference.npz")), "bytes")


if __name__ == "__main__":
    main()
