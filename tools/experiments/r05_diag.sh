#!/bin/bash
# short, individually time-boxed steps: which one hangs?  (every step: timeout -k 5 <s>; log + rc into gpurun_out/r05b/)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05b; export TMPDIR=/tmp
step() { # name seconds cmd...
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 5 $secs "$@" > gpurun_out/r05b/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/r05b/summary.txt
}
step smoke 240 python -c "import __graft_entry__ as g; g.smoke()"
step replay_6k 120 python tools/traverse_replay.py profiles/r02_sampling_state.npz 10 --check
step replay_6k_r04 120 env NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=tools/_prof/libnerfacc_hip_r04.so python tools/traverse_replay.py profiles/r02_sampling_state.npz 10 --check
step replay_200k_skip0 150 env NFA_SKIP=0 NFA_SPLIT_P=1 python tools/traverse_replay.py profiles/r02_sampling_state.npz 5 --rays=200000
step replay_200k_skip1 150 env NFA_SKIP=1 NFA_SPLIT_P=1 python tools/traverse_replay.py profiles/r02_sampling_state.npz 5 --rays=200000 --check
step replay_200k_skip2 150 env NFA_SKIP=2 NFA_SPLIT_P=1 python tools/traverse_replay.py profiles/r02_sampling_state.npz 5 --rays=200000
step bench_short 240 python bench.py --steps 5 --warmup 2 --pretrain 60 --no-aux --no-cpu-baseline --no-other-mode --no-profile --pool 65536
step occgrid_tests 200 python -m pytest tests/test_gpu_occgrid.py tests/test_gpu_pdf.py -x -q -m gpu
tail -3 gpurun_out/r05b/*.log | cut -c1-300
