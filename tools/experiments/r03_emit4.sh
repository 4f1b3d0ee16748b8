cd /root/repo; export TMPDIR=/tmp
python tools/fuzz_campaign.py 40 905 2>&1 | tail -3
python tools/fuzz_levels.py --cone 2>&1 | tail -1
NFA_EMIT=rays python tools/traverse_replay.py profiles/r02_sampling_state.npz 3 --check | tail -1
for n in 6500 32000 1000000; do
  D=/tmp/e4_$n; mkdir -p $D
  NFA_EMIT=rays rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n > /dev/null 2>&1
  echo "== $n"; python tools/kernel_summary.py $D | grep -E "emit" | cut -c1-150
done
for n in 4096 16384; do
  D=/tmp/e4c_$n; mkdir -p $D
  ML_ONLY_CONE=1 ML_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/multilevel_bench.py $n > /dev/null 2>&1
  echo "== cone $n"; python tools/kernel_summary.py $D | grep -E "emit" | cut -c1-150
done
