cd /root/repo
R="python tools/traverse_replay.py profiles/r02_sampling_state.npz 20"
for n in 2048 4096 6500 8192 10000 13000 16384 20000 32000 50000 65536 98304 130000 160000 300000 1000000; do
  echo "== $n default";           $R --rays=$n 2>&1 | tail -1
  echo "== $n old"; NFA_SPLIT_L2=0 NFA_COUNT_L2=0 $R --rays=$n 2>&1 | tail -1
  if [ $n -le 98304 ]; then echo "== $n bitmap"; NFA_SPLIT_L2=2 $R --rays=$n 2>&1 | tail -1; fi
done
python tools/traverse_replay.py profiles/r02_sampling_state.npz 5 --check | tail -1
python tools/fuzz_campaign.py 24 901 2>&1 | tail -5
