#!/bin/bash
# round 6: the packed-grid kernels of a grid update (pack_bricks with 16 lanes per brick, brick distances from an LDS-staged bitmap,
# the occupied-cell list): their tests, then a kernel trace of 40 updates at 128^3 and 256^3
export TMPDIR=/tmp
O=gpurun_out/r06_grid_update; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_occgrid.py tests/test_gpu_grid.py tests/test_gpu_estimator.py -x -q 2>&1 | tail -5 > $O/tests.log
cat > /tmp/upd.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import nerfacc_amd as nerfacc
res = int(sys.argv[1])
dev = "cuda:0"
est = nerfacc.OccGridEstimator(roi_aabb=[-1.5] * 3 + [1.5] * 3, resolution=res, levels=1).to(dev)
est.train()
fn = lambda x: (torch.exp(-6.0 * ((x * x).sum(-1, keepdim=True) - 0.6).abs()) * 0.3)
for step in (0, 16, 32, 48, 256, 272):
    est._update(step=step, occ_eval_fn=fn, occ_thre=0.01)
for step in range(320, 320 + 16 * 40, 16):
    est._update(step=step, occ_eval_fn=fn, occ_thre=0.01)
torch.cuda.synchronize()
print("occupied fraction", est.binaries.float().mean().item())
PY
for r in 128 256; do
  D=$(mktemp -d /tmp/ktXXXX)
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python /tmp/upd.py $r > $O/upd_$r.log 2>&1
  python tools/kernel_summary.py $D > $O/kstats_$r.txt 2>&1
done
cat $O/tests.log; grep "nfa::" $O/kstats_128.txt | cut -c1-70,110-160; echo; grep "nfa::" $O/kstats_256.txt | cut -c1-70,110-160
