cd /root/repo
R="python tools/traverse_replay.py profiles/r02_sampling_state.npz 30"
for n in 10000 13000 16384 20000 32000 50000 98304; do
  echo "== $n";  $R --rays=$n 2>&1 | tail -1
done
NFA_SPLIT_L2=1 python tools/traverse_replay.py profiles/r02_sampling_state.npz 5 --check | tail -1
NFA_SPLIT_P=8 python tools/traverse_replay.py profiles/r02_sampling_state.npz 5 --check | tail -1
python tools/fuzz_campaign.py 40 904 2>&1 | tail -3
