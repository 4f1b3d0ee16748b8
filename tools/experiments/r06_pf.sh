#!/bin/bash
# round 6: chunks requested ahead by the tiled walkers (NFA_PF = 0 / 1 (default) / 2 / 3) — at the training size a launch ends with its
# longest ray, whose chunks are walked one after the other: does a deeper queue shorten it?  Variant libraries: tools/build_variant.sh pf<k> -DNFA_PF=<k>
# (at N = 2^24 the depth does not matter: every streaming row within 2 % for 0 ... 3, first version of this script)
export TMPDIR=/tmp
O=gpurun_out/r06_pf; mkdir -p $O
export NERFACC_AMD_BACKEND=ctypes
for rep in 1 2; do for v in default pf0 pf2 pf3; do
  if [ $v = default ]; then unset NERFACC_AMD_LIB; else export NERFACC_AMD_LIB=$PWD/tools/_prof/libnerfacc_hip_$v.so; fi
  D=$(mktemp -d /tmp/ktXXXX)
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/small_n_replay.py same 50 > /dev/null 2>&1
  echo "== $v (N = 2.5e5, same inputs)"; python tools/kernel_summary.py $D | grep "visibility_mask\|visibility_compact\|rendering_fwd\|rendering_bwd" | awk -F'|' '{print substr($2,1,48), $3, $4}'
done; done 2>&1 | tee $O/pf_small.txt
