cd /root/repo; export TMPDIR=/tmp
python tools/fuzz_campaign.py 40 906 2>&1 | tail -2
python tools/fuzz_levels.py --cone 60 9 2>&1 | tail -1
python tools/microbench.py 2>&1 | grep "^M" | grep "sampling traversal" | cut -c1-140
for n in 6500 32000 1000000; do
  D=/tmp/ed_$n; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n > /dev/null 2>&1
  echo "== $n $(python tools/kernel_summary.py $D | grep -E 'emit' | cut -d'|' -f4)"
done
for n in 4096 16384; do
  D=/tmp/edc_$n; mkdir -p $D
  ML_ONLY_CONE=1 ML_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/multilevel_bench.py $n > /dev/null 2>&1
  echo "== cone $n $(python tools/kernel_summary.py $D | grep -E 'emit' | cut -d'|' -f4)"
done
