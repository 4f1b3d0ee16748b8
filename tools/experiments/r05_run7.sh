#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05g; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$(date +%s); timeout -k 5 $secs "$@" > gpurun_out/r05g/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/r05g/summary.txt; }
for i in 1 2 3; do step replay_$i 120 python tools/traverse_replay.py profiles/r02_sampling_state.npz 30 --check; grep "^rays\|oracle" gpurun_out/r05g/replay_$i.log | cut -c1-150; done
step replay_r04 120 env NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=tools/_prof/libnerfacc_hip_r04.so python tools/traverse_replay.py profiles/r02_sampling_state.npz 30
grep "^rays" gpurun_out/r05g/replay_r04.log | cut -c1-150
step replay_ct 120 env NERFACC_AMD_BACKEND=ctypes python tools/traverse_replay.py profiles/r02_sampling_state.npz 30
grep "^rays" gpurun_out/r05g/replay_ct.log | cut -c1-150
step phase_new 150 env NFA_PHASE_LIB=tools/_prof/libnerfacc_hip_prof.so python tools/phase_cycles.py --state=profiles/r02_sampling_state.npz 20
cat gpurun_out/r05g/phase_new.log | grep -v amdgpu
step ab_new 500 env NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=nerfacc_amd/libnerfacc_hip.so python tools/experiments/r05_count_ab.py new --quick
grep "^{" gpurun_out/r05g/ab_new.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d.get('form','auto')=='auto': print(f\"{d['workload']:24s} {d['rays']:8d} count {d['count_us']:8.1f} emit {d['emit_us']:7.1f}\")"
