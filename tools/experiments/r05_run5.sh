#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05e; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$(date +%s); timeout -k 5 $secs "$@" > gpurun_out/r05e/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/r05e/summary.txt; }
step t_est 300 python -m pytest tests/test_gpu_estimator.py -x -q -m gpu -p no:cacheprovider
tail -3 gpurun_out/r05e/t_est.log | cut -c1-300
step ab_new 500 env NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=nerfacc_amd/libnerfacc_hip.so python tools/experiments/r05_count_ab.py new
grep "frame\|1000000\|200000" gpurun_out/r05e/ab_new.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(f\"{d['workload']:24s} {d['rays']:8d} {d.get('form','auto'):16s} count {d['count_us']:8.1f} emit {d['emit_us']:7.1f}\")"
step bench 600 python bench.py
python - <<'PY'
import json
try:
    l=[x for x in open("gpurun_out/r05e/bench.log") if x.startswith("{")][-1]
    d=json.loads(l)
    print("value", d["value"], "ms/step", d["ms_per_step"], d.get("ms_per_step_windows"))
    print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_ms","emit_avg_launch_ms")})
    print("path", d["roofline"].get("path"))
    print("path_only", d.get("path_only_loop"))
    print("rank_step", json.dumps(d.get("aux",{}).get("configs3_rank_step"), indent=1)[:3000])
    print("keys", list(d.keys()), list(d.get("aux",{}).keys()))
except Exception as e:
    print("bench parse failed", e)
PY
tail -5 gpurun_out/r05e/bench.log | cut -c1-400 | grep -v "^{"
step scenes 400 python tools/scene_sweep.py gpurun_out/r05e/scene_sweep.md --quick
grep -v amdgpu gpurun_out/r05e/scenes.log | cut -c1-220
