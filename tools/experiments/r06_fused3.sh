#!/bin/bash
# round 6: the three single-launch forms — their test modules, the modules that exercise the same kernels, then a bench line
export TMPDIR=/tmp
O=gpurun_out/r06_fused3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused_sampling.py tests/test_gpu_fused_filter.py tests/test_gpu_fill_fold.py tests/test_gpu_volrend.py tests/test_gpu_estimator.py tests/test_gpu_backends.py -x -q 2>&1 | tail -25 > $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cat $O/tests.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_fused3/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "path_us", d.get("path_us_per_step"), "nfa launches", d["gpu_activity"]["nfa_kernels_per_step"], "path_only", d["path_only_loop"]["ms_per_step"], d["path_only_loop"].get("path_us_per_step"))
PY
