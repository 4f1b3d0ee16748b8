cd /root/repo
for n in 8192 16384 32768 65536 131072; do
  echo "== $n default"; ML_ONLY_LATTICE=1 ML_NO_CHECK=1 python tools/multilevel_bench.py $n 2>&1 | tail -1
  echo "== $n seg cap32 forced"; NFA_SEG_MAX=1000000 ML_ONLY_LATTICE=1 ML_NO_CHECK=1 python tools/multilevel_bench.py $n 2>&1 | tail -1
  echo "== $n seg cap16"; NFA_SEG_MAX=1000000 NFA_SEG_CAP=16 ML_ONLY_LATTICE=1 python tools/multilevel_bench.py $n 2>&1 | tail -1
done
