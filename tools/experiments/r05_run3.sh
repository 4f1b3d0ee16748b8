#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$(date +%s); timeout -k 5 $secs "$@" > gpurun_out/r05c/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/r05c/summary.txt; }
step t_occ_pdf 240 python -m pytest tests/test_gpu_occgrid.py tests/test_gpu_pdf.py -x -q -m gpu
step t_k2_skip 400 python -m pytest tests/test_k2_reference.py -x -q -m gpu -k "skip or lego_160k or lego_70k or levels4_inside or near_far"
for n in 200000 1000000; do for k in 0 1 2; do
  step replay_${n}_skip$k 120 env NFA_SKIP=$k NFA_SPLIT_P=1 python tools/traverse_replay.py profiles/r02_sampling_state.npz 6 --rays=$n
done; done
step ab_new 500 env NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=nerfacc_amd/libnerfacc_hip.so python tools/experiments/r05_count_ab.py new
step phase_new 150 env NFA_PHASE_LIB=tools/_prof/libnerfacc_hip_prof.so python tools/phase_cycles.py --state=profiles/r02_sampling_state.npz 20
step ab_r04 500 env NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=tools/_prof/libnerfacc_hip_r04.so python tools/experiments/r05_count_ab.py r04
grep -h "^rays" gpurun_out/r05c/replay_*.log | cut -c1-120
tail -3 gpurun_out/r05c/t_*.log | cut -c1-300
cat gpurun_out/r05c/phase_new.log
python - <<'PY'
import json
rows={}
for v in ("r04","new"):
    try:
        for l in open(f"gpurun_out/r05c/ab_{v}.log"):
            if not l.startswith("{"): continue
            d=json.loads(l); rows.setdefault((d["workload"],d["rays"],d.get("form","auto")),{})[v]=d
    except Exception as e: print(v, e)
for k,v in rows.items():
    a,b=v.get("r04"),v.get("new")
    f=lambda d,key: f"{d[key]:8.1f}" if d else "       -"
    print(f"{k[0]:24s} {k[1]:8d} {k[2]:14s} count {f(a,'count_us')} -> {f(b,'count_us')}  emit {f(a,'emit_us')} -> {f(b,'emit_us')}  same={(a['digest']==b['digest']) if a and b else '-'}")
PY
