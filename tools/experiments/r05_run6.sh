#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05f; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$(date +%s); timeout -k 5 $secs "$@" > gpurun_out/r05f/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/r05f/summary.txt; }
step bench 600 python bench.py
python - <<'PY'
import json
try:
    l=[x for x in open("gpurun_out/r05f/bench.log") if x.startswith("{")][-1]
    d=json.loads(l)
    print("value", d["value"], "ms/step", d["ms_per_step"])
    print("rank_step", json.dumps(d.get("aux",{}).get("configs3_rank_step"), indent=1)[:5000])
except Exception as e:
    print("bench parse failed", e)
PY
step audit 300 env NFA_TOL_AUDIT=1 python -m pytest tests/test_gpu_volrend.py -q -s -m gpu -p no:cacheprovider
grep "tol-audit\|passed\|failed" gpurun_out/r05f/audit.log | cut -c1-200
step emit_rb 500 python tools/experiments/r04_emit_rb_scenes.py 256 noise,ship,lego
grep -v amdgpu gpurun_out/r05f/emit_rb.log | cut -c1-220
