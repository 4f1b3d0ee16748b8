#!/bin/bash
# round 6: does the single-launch sampling call still win when the rays are long (many samples per ray: the emit half of the launch is
# one wave per four rays)?  SURVEY 8d's M1 / M6 workloads (4096 rays, 84 samples per ray on the sphere grid) fused / in three launches.
export TMPDIR=/tmp
O=gpurun_out/r06_fused_spr; mkdir -p $O
for f in 1 0 1 0; do
  echo "== NFA_FUSED_SAMPLE=$f"; NFA_FUSED_SAMPLE=$f timeout 300 python tools/microbench.py /dev/null 2>&1 | grep "sampling traversal" | cut -c1-110
done | tee $O/m1.txt
