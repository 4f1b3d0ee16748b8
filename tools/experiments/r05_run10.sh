#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05i; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_occgrid.py tests/test_gpu_estimator.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
timeout 300 python -m pytest tests/test_k2_reference.py -x -q -m gpu -p no:cacheprovider -k "skip" 2>&1 | tail -2
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu
import torch, time, numpy as np, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import scenes
from nerfacc_amd.cuda import _backend
for res in (128, 256):
    occ = torch.from_numpy(scenes.occupancy_grid("lego", res)).cuda()
    for _ in range(3):
        _backend.packed_bricks(occ.clone())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xs = [occ.clone() for _ in range(20)]
    torch.cuda.synchronize()
    e0.record()
    for x in xs:
        _backend.packed_bricks(x)
    e1.record(); torch.cuda.synchronize()
    print(f"pack_binaries incl. distance field, {res}^3: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per grid")
PY
