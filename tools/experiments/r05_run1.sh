#!/bin/bash
# gpurun #1 of round 5: GPU suite, then A/B of the count pass (round-4 sources vs HEAD) with phase cycles
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/pytest.log
tail -5 gpurun_out/r05a/pytest.log
for v in r04 new; do
  if [ $v = r04 ]; then lib=tools/_prof/libnerfacc_hip_r04.so; plib=tools/_prof/libnerfacc_hip_r04prof.so; else lib=nerfacc_amd/libnerfacc_hip.so; plib=tools/_prof/libnerfacc_hip_prof.so; fi
  NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=$lib timeout 600 python tools/experiments/r05_count_ab.py $v > gpurun_out/r05a/ab_$v.jsonl 2> gpurun_out/r05a/ab_$v.err
  NFA_PHASE_LIB=$plib timeout 300 python tools/phase_cycles.py --state=profiles/r02_sampling_state.npz 20 > gpurun_out/r05a/phase_$v.txt 2>&1
done
cat gpurun_out/r05a/phase_r04.txt gpurun_out/r05a/phase_new.txt
python - <<'PY'
import json
rows={}
for v in ("r04","new"):
    for l in open(f"gpurun_out/r05a/ab_{v}.jsonl"):
        d=json.loads(l); rows.setdefault((d["workload"],d["rays"],d.get("form","auto")),{})[v]=d
for k,v in rows.items():
    a,b=v.get("r04"),v.get("new")
    if a and b: print(f"{k[0]:24s} {k[1]:8d} {k[2]:7s} count {a['count_us']:8.1f} -> {b['count_us']:8.1f}  emit {a['emit_us']:7.1f} -> {b['emit_us']:7.1f}  same={a['digest']==b['digest']}")
PY
