cd /root/repo
for n in 2048 8192 16384 32768; do
  for f in samples rays; do echo "== $n $f"; NFA_EMIT=$f ML_ONLY_LATTICE=1 ML_NO_CHECK=1 python tools/multilevel_bench.py $n 2>&1 | tail -1; done
done
