#!/bin/bash
# the driver's round-end checks: the whole GPU suite, smoke(), then the bench line with defaults
export TMPDIR=/tmp
O=gpurun_out/r06_suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/gputest.log; tail -2 $O/smoke.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_suite/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "path_us", d.get("path_us_per_step"), "nfa launches", d["gpu_activity"]["nfa_kernels_per_step"], "path_only", d["path_only_loop"]["ms_per_step"], d["path_only_loop"].get("path_us_per_step"))
print(json.dumps(d["roofline"])[:1200])
PY
