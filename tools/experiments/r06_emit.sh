#!/bin/bash
# round 6, VERDICT r5 item 4: the emit pass with the next ray block's counts / records requested ahead (NFA_EMIT_AHEAD = 0 none,
# 1 both record rounds, 2 first round only) at 5 / 4 workgroups per CU (NFA_EMIT_MINBLOCKS): HIP-event times of the emit pass from
# tools/traverse_replay.py on the bench's steady state tiled to N rays.  Variant libraries: tools/build_variant.sh emit_<name> -D...
export TMPDIR=/tmp
O=gpurun_out/r06_emit; mkdir -p $O
export NERFACC_AMD_BACKEND=ctypes NFA_FUSED_SAMPLE=0
for n in 6564 160000 1000000; do for v in a0 a1 a2 a0m4 a1m4 a2m4; do
  echo "rays=$n variant=$v $(NERFACC_AMD_LIB=$PWD/tools/_prof/libnerfacc_hip_emit_$v.so python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n 2>/dev/null | grep -o 'count [0-9.]* us  emit [0-9.]* us')"
done; done | tee $O/variants.txt
