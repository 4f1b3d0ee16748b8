#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_volrend.py tests/test_gpu_tiles.py tests/test_gpu_visibility_onepass.py tests/test_gpu_backends.py tests/test_gpu_estimator.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
NFA_VIS_ONEPASS=0 timeout 100 python tools/experiments/r05_vis_onepass.py 18 20 22 24 2>&1 | grep "N=" | cut -c1-95
for lg in 18 24; do
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05v -o kt -- python $GRAFT_REPO_ROOT/tools/roofline_sweep.py $lg > /dev/null 2>&1)
python tools/kernel_summary.py gpurun_out/r05v | grep -i "visib" | cut -c1-60,100-160; rm -rf gpurun_out/r05v
done
