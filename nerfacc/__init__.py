"""`import nerfacc` alias for the MI355X implementation (nerfacc_amd).

Lets code written against the reference — e.g. examples/train_ngp_nerf_occ.py and
examples/utils.py, which import `nerfacc.estimators.occ_grid`, `nerfacc.grid`,
`nerfacc.volrend` — run unchanged: every module path of the reference's hot path resolves to
the module of the same name in nerfacc_amd.  Nothing is implemented here.
"""
import importlib
import sys

import nerfacc_amd as _impl
from nerfacc_amd import *  # noqa: F401,F403
from nerfacc_amd import __version__  # noqa: F401

for _name in ("cuda", "data_specs", "grid", "losses", "pack", "pdf", "scan", "volrend", "version", "sharding",
              "estimators", "estimators.base", "estimators.occ_grid", "estimators.prop_net"):
    _mod = importlib.import_module("nerfacc_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        globals()[_name] = _mod

__all__ = list(_impl.__all__)
